"""symode_lbfgs_update: kernel time vs the number of stored curvature pairs m (S problems, n parameters, history 100)."""
import os, sys, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import symode_amd
from symode_amd.sweep import BatchedLBFGS
eng = symode_amd.get_engine()
for S, n in ((64, 20), (1, 20), (64, 42), (512, 20)):
    row = []
    for m in (0, 10, 25, 50, 100):
        P = torch.randn(S, n, device="cuda")
        opt = BatchedLBFGS(P, 0.1, engine=eng)
        opt.old_dirs.normal_().mul_(0.1); opt.old_stps.normal_().mul_(0.1); opt.ro.uniform_(0.1, 1.0)
        opt.hist.fill_(m); opt.n_iter.fill_(5); opt.t.fill_(0.1)
        g = torch.randn(S, n, device="cuda") * 0.01
        opt.prev_g.copy_(g)                                   # y = 0 -> no new pair: m stays put
        loss = torch.zeros(S, device="cuda")
        act = torch.ones(S, dtype=torch.bool, device="cuda")
        def f():
            act.fill_(True)
            eng.lbfgs_update(P, g, loss, act, opt, 1e-6, 1e-9)
        for _ in range(3): f()
        gr = torch.cuda.CUDAGraph(); torch.cuda.synchronize()
        with torch.cuda.graph(gr):
            for _ in range(20): f()
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        row.append(f"m={m}: {best:6.1f}")
    print(f"S={S} n={n} (us per [fill + update] pair): " + " | ".join(row), flush=True)
