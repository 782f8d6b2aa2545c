import sys, time, torch
sys.path.insert(0, '/root/repo')
import symode_amd
from symode_amd import data
from symode_amd.train import _HostShadow
x, dx = data.make_dataset("dosc", 50, 2500, dt=0.02, seed=0, device="cuda")
x, dx = x[0], dx[0]
for order in (3, 5):
    r = symode_amd.SINDyRegression(2, order, False, False, threshold=0.05, device="cuda")
    for use_graph in (False, True):
        sh = _HostShadow(r, x, dx, numpy_vars=True, use_graph=use_graph)
        for _ in range(20): sh.evaluate()
        t0 = time.perf_counter()
        for _ in range(500): sh.evaluate()
        dt = (time.perf_counter() - t0) / 500
        print(f"order {order} graph={use_graph} captured={sh._graph is not None}: {dt*1e6:.1f} us per closure evaluation")
