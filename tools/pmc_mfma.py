#!/usr/bin/env python3
"""MFMA utilisation of the Gram kernel from a rocprofv3 --pmc pass (csv).

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE \
              --output-format csv -d <dir> -- python3 tools/kbench.py --op gram ...
    python tools/pmc_mfma.py <dir> aug_gram_kernel

MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs): GRBM_GUI_ACTIVE is summed over the
8 XCDs by rocprofv3 (MI355X_MICROARCH.md, DVFS section); BUSY_CYCLES counts cycles (64 per f64 16x16x4 MFMA, 16 per 4x4x4).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    d, needle = sys.argv[1], sys.argv[2]
    per = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if needle in r["Kernel_Name"]:
                per[r["Dispatch_Id"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
                per[r["Dispatch_Id"]]["_dur"] = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"])]
    rows = []
    for disp, c in per.items():
        busy = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0]))
        gui = sum(c.get("GRBM_GUI_ACTIVE", [0]))
        n_mfma = sum(c.get("SQ_INSTS_VALU_MFMA_F64", [0]))
        mops = sum(c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", [0]))
        if gui > 0:
            rows.append({"dispatch": disp, "duration_us": c["_dur"][0] / 1e3, "mfma_f64_insts": n_mfma, "mops_f64": mops,
                         "mfma_busy_cycles": busy, "grbm_gui_active_sum_xcd": gui,
                         "mfma_util_pct": 100.0 * busy / (gui / 8.0 * 1024.0)})
    rows.sort(key=lambda r: -r["duration_us"])
    big = rows[: max(1, len(rows) // 2)]
    out = {"kernel": needle, "dispatches": len(rows),
           "avg_over_largest_half": {k: sum(r[k] for r in big) / len(big) for k in big[0] if k != "dispatch"}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
