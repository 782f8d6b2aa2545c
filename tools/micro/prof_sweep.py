import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import symode_amd
from symode_amd import data
from symode_amd.batched import BatchedClosure
from symode_amd.sweep import SeedSweepLBFGS
S, n_ics, steps, order = int(os.environ.get("PS", "64")), 50, int(os.environ.get("PSTEPS", "1000")), 3
X, DX = data.make_dataset("dosc", n_ics, steps, dt=0.02, noise=0.0, seed=10, device="cuda", n_problems=S)
p = symode_amd.library.term_count(2, order)
torch.manual_seed(0)
inits = torch.randn(S, 2 * p, device="cuda")
for graph in ("1", "0"):
    os.environ["SYMODE_SWEEP_GRAPH"] = graph
    sw = SeedSweepLBFGS(BatchedClosure(X, DX, order), 0.1, 0.05, 50)
    for _ in range(2):
        sw.fit(inits, 60)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    t0 = time.perf_counter()
    out = sw.fit(inits, 60)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pr.disable()
    print("graph", graph, "fit", dt * 1e3, "ms; epochs", int(out["epochs"].max()))
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
