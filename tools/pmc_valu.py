#!/usr/bin/env python3
"""VALU occupancy of the arithmetic-bound kernels from a rocprofv3 --pmc pass (csv).

    rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d <dir> \
              -- python3 tools/kbench.py --table --only euler_jvp ...
    python tools/pmc_valu.py <dir> euler_jvp_kernel euler_jvp_vjp_kernel odeint_kernel symreg_linear_kernel

Per kernel (largest dispatches): wave-level VALU instructions, instructions per point, and
  valu_issue_frac = SQ_INSTS_VALU x 2 cycles / (kernel cycles x 1024 SIMDs)      (a wave64 fp32 op occupies its SIMD-32 for 2
                    cycles: MI355X_MICROARCH.md cycle table; kernel cycles = GRBM_GUI_ACTIVE / 8, summed over XCDs by rocprofv3)
i.e. the share of the chip's vector-issue capacity the kernel used -- the roofline these K-step kernels sit under; and the
same with 2.85 cycles per instruction, what a three-VGPR-operand v_fmac_f32 -- the kernels' dominant instruction -- reaches in the
chip's own probe (the datasheet's 2 cycles it does not).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    d, needles = sys.argv[1], sys.argv[2:]
    out = {}
    for needle in needles:
        per = defaultdict(lambda: defaultdict(float))
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if needle in r["Kernel_Name"]:
                    per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
                    per[r["Dispatch_Id"]]["_items"] = float(r["Grid_Size"])
        rows = [c for c in per.values() if c.get("GRBM_GUI_ACTIVE", 0) > 0]
        if not rows:
            continue
        big = max(c["_items"] for c in rows)
        rows = [c for c in rows if c["_items"] == big]
        n = len(rows)
        insts = sum(c.get("SQ_INSTS_VALU", 0) for c in rows) / n
        gui = sum(c["GRBM_GUI_ACTIVE"] for c in rows) / n
        cycles = gui / 8.0
        out[needle] = {"dispatches": n, "work_items": big, "SQ_INSTS_VALU": insts, "GRBM_GUI_ACTIVE_sum_xcd": gui,
                       "kernel_cycles": cycles, "valu_issue_frac": insts * 2.0 / (cycles * 1024.0),
                       # against what the kernels' own instruction reaches on this chip: v_fmac_f32 with three VGPR operands
                       # (acc[i] += r * th[i]) retires one wave-instruction per 2.8-2.95 cycles per SIMD at 4-8 resident waves
                       # (tools/micro/valu_rate.hip, profiles/r03_valu_rate.txt); only an FMA that reads ONE VGPR gets to 2.2
                       "valu_issue_frac_at_measured_2p85_cycles": insts * 2.85 / (cycles * 1024.0),
                       "SQ_ACTIVE_INST_VALU": sum(c.get("SQ_ACTIVE_INST_VALU", 0) for c in rows) / n,
                       "SQ_WAVE_CYCLES": sum(c.get("SQ_WAVE_CYCLES", 0) for c in rows) / n,
                       "SQ_BUSY_CYCLES": sum(c.get("SQ_BUSY_CYCLES", 0) for c in rows) / n}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
