#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, csv) into per-launch HBM traffic of the
step's kernels, with the gfx950 corrections of MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB;
FETCH_SIZE counts exactly half of a wide (16 B/lane) coalesced streaming read, so it is doubled; WRITE_SIZE is taken as is.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <points per launch> key=kernel-substring [key=substring ...] > profiles/pmc_traffic.json

Only dispatches with a 2-D grid of the batched step (Grid_Size_Y > 1 when the trace has it, else the largest grid seen
for that kernel) are averaged, so warm-up launches of other shapes do not dilute the figure.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def collect(d, counter, needle):
    by_grid = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter or needle not in r["Kernel_Name"]:
                continue
            by_grid[int(r["Grid_Size"])].append(float(r["Counter_Value"]))      # Grid_Size = total work-items
    if not by_grid:
        return [], 0
    grid = max(by_grid)                                                          # the batched launches are the largest
    return by_grid[grid], grid


def main():
    fetch_dir, write_dir, points = sys.argv[1], sys.argv[2], int(sys.argv[3])
    out = {"correction": "read bytes = 2 x FETCH_SIZE x 1024 (gfx950 counts half of a 16 B/lane stream); "
                         "write bytes = WRITE_SIZE x 1024", "points_per_launch": points, "kernels": {}}
    for spec in sys.argv[4:]:
        key, needle = spec.split("=", 1)
        f, gf = collect(fetch_dir, "FETCH_SIZE", needle)
        w, gw = collect(write_dir, "WRITE_SIZE", needle)
        if not f or not w:
            print(f"no samples for {needle}: fetch {len(f)} write {len(w)}", file=sys.stderr)
            continue
        fetch_kib, write_kib = sum(f) / len(f), sum(w) / len(w)
        total = 2 * fetch_kib * 1024 + write_kib * 1024
        out["kernels"][key] = {"kernel": needle, "work_items": gf, "launches_sampled": {"FETCH_SIZE": len(f), "WRITE_SIZE": len(w)},
                               "FETCH_SIZE_KiB_avg": fetch_kib, "WRITE_SIZE_KiB_avg": write_kib, "bytes_per_launch": total,
                               "bytes_per_point": total / points}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
