#!/usr/bin/env python3
"""GP smoothing (the reference's num_diff_gp) at the README data sizes: T = 10^4 samples, GPU vs CPU.

    python tools/gp_bench.py --T 10000 --n_traj 50
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import symode_amd  # noqa: F401
from symode_amd import data as synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=10000)
    ap.add_argument("--n_traj", type=int, default=50)
    ap.add_argument("--cpu", action="store_true", help="also time the same torch ops on the host")
    a = ap.parse_args()
    x, _ = synth.gen_data("dosc", a.n_traj, dt=0.002, num_steps=a.T, noise=0.2, seed=0, device="cuda:0")
    xt = x.double().transpose(0, 1).contiguous()
    std = xt.std(dim=(0, 1), unbiased=False)
    for dev in (["cuda:0", "cpu"] if a.cpu else ["cuda:0"]):
        xd, sd = xt.to(dev), std.to(dev)
        if dev != "cpu":
            synth.gp_smooth(xd[:256], 0.002, 0.2, sd, 0.1)       # warm-up (solver handles)
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        d, s = synth.gp_smooth(xd, 0.002, 0.2, sd, 0.1)
        if dev != "cpu":
            torch.cuda.synchronize()
        print(f"gp_smooth T={a.T} n_traj={a.n_traj} d=2 on {dev}: {time.perf_counter() - t0:.2f} s", flush=True)


if __name__ == "__main__":
    main()
