"""Single-problem reduction kernels vs N and the small-grid cap (SYMODE_SMALL_GRID; engine.reload_env() after every change)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import symode_amd
eng = symode_amd.get_engine()
order = int(os.environ.get("ORDER", "3"))
p = symode_amd.library.term_count(2, order)

def timeit(f):
    for _ in range(3):
        f()
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(20):
            f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    return best

for N in (125000, 250000, 500000, 1000000, 2000000, 4000000, 16000000, 64000000):
    x = torch.randn(N, 2, device="cuda") * 0.5
    dx = torch.randn(N, 2, device="cuda")
    gx = torch.randn(1, 1, N, 2, device="cuda") * 0.5
    jg = torch.randn(1, 1, N, 2, 2, device="cuda")
    xi = torch.randn(2, p, device="cuda") * 0.1
    lo, gr = torch.empty(1, device="cuda"), torch.empty(2, p, device="cuda")
    l2 = torch.empty(1, 2, device="cuda")
    ws = eng.new_workspace(x.device, eng.lib.symode_workspace_bytes(2, order, 0, 1, N))
    row = []
    for cap in ("128", "256", "512", "1024", None):
        if cap is None:
            os.environ.pop("SYMODE_SMALL_GRID", None)
        else:
            os.environ["SYMODE_SMALL_GRID"] = cap
        symode_amd.engine.reload_env()
        t1 = timeit(lambda: eng.loss_grad(x, dx, xi, None, order, 0, out=(lo, gr), ws=ws))
        t2 = timeit(lambda: eng.loss_grad_reversed(x[None], dx[None], gx, jg, xi[None], None, order, 0, w_sym=1.0, out=(l2, gr[None]), ws=ws))
        t3 = timeit(lambda: eng.symreg_reversed(x[None], gx, jg, xi[None], None, order, 0, out=(lo, gr[None]), ws=ws, inv_count=1.0 / (2 * N)))
        row.append(f"cap {str(cap):>4}: {t1:6.1f} / {t2:6.1f} / {t3:6.1f}")
    print(f"N={N:9d} order {order} loss_grad / fused closure / symreg_reversed us: " + " | ".join(row), flush=True)
