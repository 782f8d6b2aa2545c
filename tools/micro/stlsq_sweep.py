"""BASELINE config[3] shape: selkov 10 x 10000 x 2, 64 seeds x 50 % subsample, order 3: SeedSweepSTLSQ wall time split."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import symode_amd
from symode_amd import data
from symode_amd.sweep import SeedSweepSTLSQ
X, DX = data.make_dataset("selkov", 10, 10000, dt=0.01, noise=0.0, seed=0, device="cuda")
x, dx = X[0].reshape(-1, 2).contiguous(), DX[0].reshape(-1, 2).contiguous()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sw = SeedSweepSTLSQ(x, dx, 3, n_seeds=64, subsample=0.5, seed0=0)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    sw.grams()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    Xi, mask, passes = sw.solve(0.0, 0.05, max_iter=10)
    t3 = time.perf_counter()
    print(f"rep {rep}: index draw {1e3*(t1-t0):.2f} ms | Gram (gather kernel + copy to host) {1e3*(t2-t1):.3f} ms | "
          f"host STLSQ solves {1e3*(t3-t2):.2f} ms ({int(passes.sum())} passes over 64 seeds) | masks per seed identical to seed 0: "
          f"{int((mask == mask[0]).all(dim=(1,2)).sum())}/64", flush=True)
