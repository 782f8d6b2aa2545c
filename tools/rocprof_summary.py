#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats (csv) output directory into a small text summary.

    python tools/rocprof_summary.py gpurun_out/prof1 > profiles/r01_bench_kernel_stats.txt

Lists every kernel whose name contains 'symode' (count, total, average, min, max duration,
grid, VGPR/SGPR/LDS from the trace) followed by the top non-symode kernels by total time.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(d):
    traces = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)
    if not traces:
        sys.exit(f"no *_kernel_trace.csv under {d}")
    agg = defaultdict(lambda: {"n": 0, "tot": 0, "min": 1 << 62, "max": 0, "meta": None})
    for tr in traces:
        for r in csv.DictReader(open(tr)):
            name = r["Kernel_Name"]
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            grid = (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
            key = (name, grid) if "symode" in name else (name, None)
            a = agg[key]
            a["n"] += 1
            a["tot"] += dur
            a["min"] = min(a["min"], dur)
            a["max"] = max(a["max"], dur)
            a["meta"] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"], r["Workgroup_Size_X"])
    total = sum(a["tot"] for a in agg.values())
    print(f"# rocprofv3 kernel-trace summary of {d}  (durations in us; total GPU kernel time {total/1e3:.1f} us)")
    print("# name | grid (blocks x,y,z) | calls | total_us | avg_us | min_us | max_us | vgpr agpr sgpr lds scratch wg")
    sym = sorted([(k, a) for k, a in agg.items() if "symode" in k[0]], key=lambda t: -t[1]["tot"])
    for (name, grid), a in sym:
        short = name.replace("void ", "").split("(")[0]
        print(f"{short} | {grid} | {a['n']} | {a['tot']/1e3:.1f} | {a['tot']/a['n']/1e3:.2f} | {a['min']/1e3:.2f} | {a['max']/1e3:.2f} | {' '.join(a['meta'])}")
    print("# --- other kernels (top 8 by total time) ---")
    oth = sorted([(k, a) for k, a in agg.items() if "symode" not in k[0]], key=lambda t: -t[1]["tot"])[:8]
    for (name, _), a in oth:
        print(f"{name[:100]} | {a['n']} | {a['tot']/1e3:.1f} | {a['tot']/a['n']/1e3:.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
