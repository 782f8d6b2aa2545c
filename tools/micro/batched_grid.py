"""Batched closure kernel time vs the workgroup budget (SYMODE_MAX_GRID, read once per process)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import symode_amd
eng = symode_amd.get_engine()
shapes = [(16, 125000, 3), (64, 50000, 3), (64, 125000, 3), (256, 50000, 3), (512, 125000, 3), (1024, 125000, 5), (2048, 20000, 3)]
out = []
for S, N, order in shapes:
    p = symode_amd.library.term_count(2, order)
    x = torch.randn(S, N, 2, device="cuda") * 0.5
    dx = torch.randn(S, N, 2, device="cuda")
    xi = torch.randn(S, 2, p, device="cuda") * 0.1
    lo, gr = torch.empty(S, device="cuda"), torch.empty(S, 2, p, device="cuda")
    ws = eng.new_workspace(x.device, eng.lib.symode_workspace_bytes(2, order, 0, S, N))
    f = lambda: eng.loss_grad(x, dx, xi, None, order, 0, out=(lo, gr), ws=ws)
    for _ in range(5):
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
    out.append(f"{S}x{N}o{order}: {best:7.1f} us ({S * N * 16 / best / 1e6:5.2f} TB/s)")
print(f"MAX_GRID={os.environ.get('SYMODE_MAX_GRID', 'default')}: " + " | ".join(out), flush=True)
