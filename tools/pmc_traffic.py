#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, csv) into per-launch HBM
traffic of the dominant kernel, with the gfx950 corrections of MI355X_MICROARCH.md (HBM section):
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE counts exactly half of a wide (16 B/lane) coalesced
streaming read, so it is doubled; WRITE_SIZE is taken as is.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <kernel-substring> <total work-items> <points per launch> > profiles/pmc_traffic.json
"""
import csv
import glob
import json
import os
import sys


def collect(d, counter, needle, grid_y):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter or needle not in r["Kernel_Name"]:
                continue
            if grid_y and int(r["Grid_Size"]) != grid_y:      # Grid_Size = total work-items of the dispatch
                continue
            vals.append(float(r["Counter_Value"]))
    return vals


def main():
    fetch_dir, write_dir, needle = sys.argv[1], sys.argv[2], sys.argv[3]
    grid_y = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    points = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    f = collect(fetch_dir, "FETCH_SIZE", needle, grid_y)
    w = collect(write_dir, "WRITE_SIZE", needle, grid_y)
    if not f or not w:
        sys.exit(f"no samples: fetch {len(f)} write {len(w)}")
    fetch_kib, write_kib = sum(f) / len(f), sum(w) / len(w)
    out = {
        "kernel": needle, "launches_sampled": {"FETCH_SIZE": len(f), "WRITE_SIZE": len(w)},
        "FETCH_SIZE_KiB_avg": fetch_kib, "WRITE_SIZE_KiB_avg": write_kib,
        "correction": "read bytes = 2 x FETCH_SIZE x 1024 (gfx950 counts half of a 16 B/lane stream); "
                      "write bytes = WRITE_SIZE x 1024",
        "loss_grad_bytes_per_launch": 2 * fetch_kib * 1024 + write_kib * 1024,
        "points_per_launch": points,
        "loss_grad_bytes_per_point": (2 * fetch_kib * 1024 + write_kib * 1024) / points if points else None,
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
