"""symode_seeded_subsamples beside the torch form it replaced (a generator launch per seed, a batched top-k, a sort)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import symode_amd
eng = symode_amd.get_engine()


def torch_form(n, m, seeds, device="cuda"):
    g = torch.Generator(device=device)
    keys = torch.empty(len(seeds), n, dtype=torch.float64, device=device)
    for row, seed in zip(keys, seeds):
        g.manual_seed(int(seed))
        row.uniform_(generator=g)
    idx = torch.topk(keys, m, dim=1, largest=False, sorted=False).indices
    return torch.sort(idx, dim=1).values


def wall(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for S, n in ((64, 100000), (64, 1000000), (256, 2000000), (8, 100000)):
    m = n // 2
    seeds = list(range(S))
    sd = torch.tensor(seeds, dtype=torch.int64, device="cuda")
    out = torch.empty(S, m, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    import ctypes
    k = wall(lambda: eng.lib.symode_seeded_subsamples(n, m, ctypes.c_void_p(sd.data_ptr()), S, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(st)))
    a = wall(lambda: eng.seeded_subsamples(n, m, seeds, "cuda"))
    b = wall(lambda: torch_form(n, m, seeds), reps=5)
    print(f"{S} seeds x {m} of {n} rows: kernel {k:8.1f} us | engine call {a:8.1f} us | torch form {b:9.1f} us")
