"""aug_gram (vector-pipe form at order 3, MFMA form at order 5) and the batched fused closure vs their workgroup budgets
(SYMODE_GRAM_VALU_GRID / SYMODE_GRAM_GRID / SYMODE_MAX_GRID: read once per process -> one process per setting)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import symode_amd
eng = symode_amd.get_engine()

def timeit(f, reps=10):
    for _ in range(3):
        f()
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(reps):
            f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best

tag = " ".join(f"{k[7:]}={v}" for k, v in sorted(os.environ.items()) if k in ("SYMODE_GRAM_VALU_GRID", "SYMODE_GRAM_GRID", "SYMODE_MAX_GRID")) or "default"
row = []
for S, N in ((1, 125000), (1, 1000000), (1, 16000000), (16, 125000), (64, 50000), (512, 125000)):
    x = torch.randn(S, N, 2, device="cuda") * 0.5
    dx = torch.randn(S, N, 2, device="cuda")
    for order in (3, 5):
        t = timeit(lambda: eng.aug_gram(x, dx, order, 0))
        row.append(f"gram{order} {S}x{N}: {t:7.1f}")
    if S > 1:
        p = symode_amd.library.term_count(2, 3)
        gx = torch.randn(S, 1, N, 2, device="cuda") * 0.5
        jg = torch.randn(S, 1, N, 2, 2, device="cuda")
        xi = torch.randn(S, 2, p, device="cuda") * 0.1
        l2, gr = torch.empty(S, 2, device="cuda"), torch.empty(S, 2, p, device="cuda")
        ws = eng.new_workspace(x.device, eng.lib.symode_workspace_bytes(2, 3, 0, S, N))
        t = timeit(lambda: eng.loss_grad_reversed(x, dx, gx, jg, xi, None, 3, 0, w_sym=1.0, out=(l2, gr), ws=ws))
        row.append(f"fused3 {S}x{N}: {t:7.1f}")
        del gx, jg
    del x, dx
    torch.cuda.empty_cache()
print(f"[{tag}] " + " | ".join(row), flush=True)
