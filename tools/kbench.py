#!/usr/bin/env python3
"""Micro-benchmark of single HIP entry points (interleaved A/B rounds in one process).

    SYMODE_LOSS_GRAD_VARIANT=1 python tools/kbench.py --op loss_grad --S 1024 --N 125000 --order 5
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import symode_amd


def timeit(fn, reps, rounds=5):
    out = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps)
    return min(out), sorted(out)[len(out) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--op", default="loss_grad")
    ap.add_argument("--S", type=int, default=1024)
    ap.add_argument("--N", type=int, default=125000)
    ap.add_argument("--d", type=int, default=2)
    ap.add_argument("--order", type=int, default=5)
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--graph", action="store_true", help="replay the call from a captured HIP graph (no host launch cost)")
    a = ap.parse_args()
    eng = symode_amd.get_engine()
    p = eng.lib_size(a.d, a.order, a.flags)
    x = torch.randn(a.S, a.N, a.d, device="cuda") * 0.7
    dx = torch.randn(a.S, a.N, a.d, device="cuda")
    xi = torch.randn(a.S, a.d, p, device="cuda") * 0.3
    pts = a.S * a.N
    if a.op == "loss_grad":
        fn = lambda: eng.loss_grad(x, dx, xi, None, a.order, a.flags)  # noqa: E731
        byt = pts * 8 * a.d
    elif a.op == "gram":
        fn = lambda: eng.aug_gram(x, dx, a.order, a.flags)  # noqa: E731
        byt = pts * 8 * a.d
    elif a.op == "forward":
        x2, xi2 = x.reshape(-1, a.d), xi[0]
        fn = lambda: eng.forward(x2, xi2, None, a.order, a.flags)  # noqa: E731
        byt = pts * 8 * a.d
    elif a.op == "theta":
        x2 = x.reshape(-1, a.d)
        fn = lambda: eng.theta(x2, a.order, a.flags)  # noqa: E731
        byt = pts * 4 * (a.d + p)
    else:
        raise SystemExit("unknown op")
    if a.graph:
        if a.op == "loss_grad":
            loss = torch.empty(a.S, device="cuda")
            grad = torch.empty(a.S, a.d, p, device="cuda")
            call = lambda: eng.loss_grad(x, dx, xi, None, a.order, a.flags, out=(loss, grad))  # noqa: E731
        else:
            call = fn
        call()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            call()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            call()
        fn = g.replay
    mn, med = timeit(fn, a.reps)
    print(f"{a.op} variant={os.environ.get('SYMODE_LOSS_GRAD_VARIANT', '0')} S={a.S} N={a.N} d={a.d} order={a.order} p={p}: "
          f"min {mn*1e3:.1f} us  med {med*1e3:.1f} us  {pts/mn/1e6:.1f} Gpts/s  {byt/mn/1e6:.0f} GB/s (alg. bytes)")


if __name__ == "__main__":
    main()
