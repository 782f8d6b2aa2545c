"""Seed-sweep L-BFGS (sweep.SeedSweepLBFGS) wall time: optimiser side as two fused kernels per inner iteration
(symode_lbfgs_update / _accept) vs the tensor-op form, each eager and replayed from a HIP graph.

    python tools/sweep_bench.py [--shapes 64x50000x3,512x125000x3,1024x125000x5] [--epochs 60]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import symode_amd  # noqa: E402
from symode_amd import data  # noqa: E402
from symode_amd.batched import BatchedClosure  # noqa: E402
from symode_amd.sweep import SeedSweepLBFGS  # noqa: E402


def run(S, n_points, order, epochs):
    n_ics = 50
    steps = n_points // n_ics
    X, DX = data.make_dataset("dosc", n_ics, steps, dt=0.02, noise=0.0, seed=10, device="cuda", n_problems=S)
    p = symode_amd.library.term_count(2, order)
    torch.manual_seed(0)
    inits = torch.randn(S, 2 * p, device="cuda")
    truth = torch.zeros(2, p, dtype=torch.bool)
    truth[:, 1:3] = True
    ref_mask = None
    for fused in ("1", "0"):
        for graph in ("1", "0"):
            os.environ["SYMODE_LBFGS_FUSED"], os.environ["SYMODE_SWEEP_GRAPH"] = fused, graph
            sw = SeedSweepLBFGS(BatchedClosure(X, DX, order), 0.1, 0.05, 50)
            best = 1e9
            for rep in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = sw.fit(inits, epochs)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            ok = int(sum(torch.equal(out["mask"][s].cpu().bool(), truth) for s in range(S)))
            same = "-" if ref_mask is None else str(int((out["mask"] == ref_mask).all(dim=(1, 2)).sum()))
            if ref_mask is None:
                ref_mask = out["mask"].clone()
            print(f"{S:5d} seeds x {n_ics * steps:6d} points order {order}: fused={fused} graph={graph}: {best * 1e3:8.2f} ms "
                  f"({best / S * 1e6:7.1f} us/seed); correct form {ok}/{S}; masks equal to the first row's {same}/{S}; "
                  f"max epochs {int(out['epochs'].max())}", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="64x50000x3,512x125000x3,1024x125000x5")
    ap.add_argument("--epochs", type=int, default=60)
    a = ap.parse_args()
    for shape in a.shapes.split(","):
        S, n, o = (int(v) for v in shape.split("x"))
        run(S, n, o, a.epochs)
