#!/usr/bin/env python3
"""Single-problem closure latency (the 50x2500x2 shape is launch-latency bound).

    python tools/latency_bench.py [--order 5]

Prints, for the one-launch (ticket) and two-launch (finalize kernel) forms and a few grid caps:
  kernel us  : HIP-event time per call over back-to-back launches on device-resident Xi / outputs
  closure us : host wall time of _HostShadow.evaluate() (Xi in, [loss | grad] out, one sync), zero-copy vs graph + copies
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import symode_amd
from symode_amd.sindy import SINDyRegression
from symode_amd.train import _HostShadow


def ev_time(fn, reps=400, rounds=5):
    best = 1e9
    for _ in range(rounds):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def wall_time(fn, reps=400, rounds=5):
    best = 1e9
    for _ in range(rounds):
        for _ in range(20):
            fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        best = min(best, (time.perf_counter() - t0) * 1e6 / reps)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=125000)
    ap.add_argument("--orders", type=int, nargs="+", default=[3, 5])
    a = ap.parse_args()
    eng = symode_amd.get_engine()
    torch.manual_seed(0)
    x, dx = torch.randn(a.n, 2).cuda() * 0.7, torch.randn(a.n, 2).cuda()
    for order in a.orders:
        p = eng.lib_size(2, order, 0)
        xi = torch.randn(2, p).cuda() * 0.3
        loss, grad = torch.empty(1).cuda(), torch.empty(2, p).cuda()
        for fused in (0, 1, 2):
            for cap in (0, 64, 96, 128, 192):
                os.environ["SYMODE_FUSED_FINALIZE"] = str(fused)
                os.environ["SYMODE_SMALL_GRID"] = str(cap)
                symode_amd.engine.reload_env()
                us = ev_time(lambda: eng.loss_grad(x, dx, xi, None, order, out=(loss, grad)))
                g = torch.cuda.CUDAGraph()
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    eng.loss_grad(x, dx, xi, None, order, out=(loss, grad))
                torch.cuda.current_stream().wait_stream(s)
                ws = eng.new_workspace(x.device, eng.lib.symode_workspace_bytes(2, order, 0, 1, a.n))
                with torch.cuda.graph(g):
                    for _ in range(20):
                        eng.loss_grad(x, dx, xi, None, order, out=(loss, grad), ws=ws)
                gus = ev_time(g.replay, reps=20) / 20
                print(f"order {order} fused={fused} grid_cap={cap:3d}: eager {us:6.2f} us/call, graph of 20 back-to-back {gus:6.2f} us/call", flush=True)
        os.environ["SYMODE_FUSED_FINALIZE"] = "1"
        os.environ.pop("SYMODE_SMALL_GRID", None)
        symode_amd.engine.reload_env()
        reg = SINDyRegression(2, order, False, False, threshold=0.05, device="cuda:0")
        for name, kw in (("zero-copy, one launch", dict(zero_copy=True, use_graph=False)),
                         ("copies + HIP graph", dict(zero_copy=False, use_graph=True)),
                         ("copies, eager", dict(zero_copy=False, use_graph=False))):
            sh = _HostShadow(reg, x, dx, numpy_vars=False, **kw)
            with torch.no_grad():
                us = wall_time(sh.evaluate)
            print(f"order {order} closure ({name}; zero_copy={sh.zero_copy} graph={sh._graph is not None}): {us:6.2f} us wall per evaluate()", flush=True)


if __name__ == "__main__":
    main()
