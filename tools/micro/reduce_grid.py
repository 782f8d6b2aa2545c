"""Single-problem reduction kernels other than the closures (vjp, jvp_vjp, symreg_linear, euler_jvp_vjp) vs N, for one
value of SYMODE_REDUCE_GRID (workgroup cap, read once per process): us per launch, 20 launches replayed from a graph."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
import symode_amd
from kbench import stream_op
eng = symode_amd.get_engine()

def timeit(f):
    for _ in range(3):
        f()
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(10):
            f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
    return best

ops = ("vjp", "vjp_noxgrad", "jvp_vjp", "symreg_linear", "euler_jvp_vjp")
for N in (20000, 125000, 1000000, 8000000, 64000000):
    row = []
    for op in ops:
        fn, byt = stream_op(eng, op, N, 2, 3, 0, 10)
        t = timeit(fn)
        row.append(f"{op} {t:7.1f}")
        del fn
        torch.cuda.empty_cache()
    print(f"REDUCE_GRID={os.environ.get('SYMODE_REDUCE_GRID', 'default'):>7} N={N:9d}: " + " | ".join(row), flush=True)
