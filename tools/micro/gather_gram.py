"""Index-table Gram (seed sweeps over sorted subsamples of ONE data set) at BASELINE config[3]'s shape and a larger one:
MFMA form (default) vs the vector-pipe form (SYMODE_GRAM_VALU_GATHER=1; engine.reload_env() after every change)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import symode_amd
eng = symode_amd.get_engine()

def timeit(f, reps=20):
    """eager launches between two events (the gather entry validates its table with a host sync: not capturable)"""
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best

for N, S, frac, order in ((100000, 64, 0.5, 3), (100000, 64, 0.5, 5), (1000000, 64, 0.5, 3), (2000000, 256, 0.1, 3)):
    x = torch.randn(N, 2, device="cuda") * 0.5
    dx = torch.randn(N, 2, device="cuda")
    m = int(N * frac)
    idx = torch.stack([torch.randperm(N, device="cuda")[:m].sort().values for _ in range(S)]).int()
    xs = torch.stack([x[i.long()] for i in idx]); dxs = torch.stack([dx[i.long()] for i in idx])
    row = []
    for env in ("0", "1"):
        os.environ["SYMODE_GRAM_VALU_GATHER"] = env
        symode_amd.engine.reload_env()
        t = timeit(lambda: eng.aug_gram_gather(x, dx, idx, order, 0))
        row.append(f"gather valu={env}: {t:8.1f} us")
    t = timeit(lambda: eng.aug_gram(xs, dxs, order, 0))
    row.append(f"dense copy of the same rows: {t:8.1f} us")
    print(f"N={N} S={S} m={m} order {order}: " + " | ".join(row), flush=True)
    del xs, dxs
