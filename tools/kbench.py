#!/usr/bin/env python3
"""Micro-benchmark of single HIP entry points (interleaved A/B rounds in one process).

    python tools/kbench.py --op loss_grad --S 1024 --N 125000 --order 5
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


STREAM_OPS = ("symreg_linear", "symreg_reversed", "vjp", "vjp_noxgrad", "forward_jvp", "jvp_vjp", "euler_jvp", "euler_jvp_vjp",
              "odeint", "odeint_rk4", "gram_gather")
EXTRA_OPS = ("weak_gram",)          # not part of --table: one trajectory of n time points against K test functions


def stream_op(eng, op, n, d, order, flags, K=10):
    """(callable, algorithmic bytes per call) for the single-problem streaming entry points at n points
    (bytes per point: SURVEY section 8(d) / DESIGN.md section 4: every operand read once, every output written once)."""
    p = eng.lib_size(d, order, flags)
    f = 4 * d                                            # bytes of one (n, d) fp32 operand per point
    mk = lambda *shape: torch.randn(*shape, device="cuda") * 0.5  # noqa: E731
    x, xi = mk(n, d), mk(d, p) * 0.5
    if op == "symreg_linear":
        L = torch.tensor([[[0.0, 1.0], [-1.0, 0.0]]], device="cuda") if d == 2 else mk(1, d, d)
        return (lambda: eng.symreg_linear(x, xi, None, L, order, flags)), n * f
    if op == "symreg_reversed":
        gx, jgx = mk(1, n, d), mk(1, n, d, d)
        return (lambda: eng.symreg_reversed(x, gx, jgx, xi, None, order, flags)), n * (f + f + 4 * d * d)
    if op == "vjp":
        g = mk(n, d)
        return (lambda: eng.vjp(x, g, xi, None, order, flags)), n * 3 * f
    if op == "vjp_noxgrad":
        g = mk(n, d)
        return (lambda: eng.vjp(x, g, xi, None, order, flags, need_grad_x=False)), n * 2 * f
    if op == "forward_jvp":
        v = mk(n, d)
        return (lambda: eng.forward_jvp(x, v, xi, None, order, flags)), n * 4 * f
    if op == "jvp_vjp":
        v, go, gj = mk(n, d), mk(n, d), mk(n, d)
        return (lambda: eng.jvp_vjp(x, v, go, gj, xi, None, order, flags)), n * 6 * f
    if op == "euler_jvp":
        v = mk(n, d)
        return (lambda: eng.euler_jvp(x, v, xi * 0.1, None, order, flags, K, 0.01)), n * 4 * f
    if op == "euler_jvp_vjp":
        v, go, gj = mk(n, d), mk(n, d), mk(n, d)
        return (lambda: eng.euler_jvp_vjp(x, v, go, gj, xi * 0.1, None, order, flags, K, 0.01)), n * 6 * f
    if op == "odeint":
        return (lambda: eng.odeint(x, xi * 0.1, None, order, flags, K, 0.01, "euler")), n * 2 * f
    if op == "odeint_rk4":
        return (lambda: eng.odeint(x, xi * 0.1, None, order, flags, K, 0.01, "rk4")), n * 2 * f
    if op == "weak_gram":                                # N1, sindy.py:362-381: [V; -V'] (2K, T) x [Theta | x] (T, p + d) on the fp64 MFMA
        V, Vd = mk(K, n), mk(K, n)
        return (lambda: eng.weak_gram(x, V, Vd, order, flags)), n * (f + 8 * K)
    if op == "gram_gather":
        S, m = 64, n // 128
        dx = mk(n, d)
        idx = torch.stack([torch.randperm(n, device="cuda")[:m].sort().values for _ in range(S)]).int()
        return (lambda: eng.aug_gram_gather(x, dx, idx, order, flags)), S * m * (2 * f + 4)
    raise SystemExit(f"unknown op {op}")


def table(a):
    """Markdown table of every streaming entry point at one large N (rooflines: profiles/r02_ops_roofline.md)."""
    eng = symode_amd.get_engine()
    n = a.S * a.N
    print(f"| op | d | order | p | points | us/call (min of 5 rounds) | alg. bytes/pt | GB/s | frac of 8 TB/s |")
    print("|---|---|---|---|---|---|---|---|---|")
    x3 = torch.randn(a.S, a.N, a.d, device="cuda") * 0.7
    dx3 = torch.randn(a.S, a.N, a.d, device="cuda")
    for order, flags in a.libs:
        p = eng.lib_size(a.d, order, flags)
        xi = torch.randn(a.S, a.d, p, device="cuda") * 0.3
        rows = [("loss_grad", (lambda: eng.loss_grad(x3, dx3, xi, None, order, flags)), n * 8 * a.d),
                ("aug_gram", (lambda: eng.aug_gram(x3, dx3, order, flags)), n * 8 * a.d),
                ("forward", (lambda: eng.forward(x3.view(-1, a.d), xi[0], None, order, flags)), n * 8 * a.d),
                ("theta", (lambda: eng.theta(x3.view(-1, a.d), order, flags)), n * 4 * (a.d + p))]
        for op in STREAM_OPS:
            if a.only and op not in a.only:
                continue
            fn, byt = stream_op(eng, op, n, a.d, order, flags, a.K)
            rows.append((op, fn, byt))
        for name, fn, byt in rows:
            if a.only and name not in a.only:
                continue
            mn, _ = timeit(fn, a.reps)
            print(f"| {name} | {a.d} | {order}{'+sin' if flags & 1 else ''}{'+exp' if flags & 2 else ''} | {p} | {n} | {mn*1e3:.1f} | "
                  f"{byt/n:.0f} | {byt/mn/1e6:.0f} | {byt/mn/1e6/8000:.3f} |", flush=True)
            torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--table", action="store_true", help="every entry point at S*N points, markdown")
    ap.add_argument("--libs", type=lambda s: [tuple(int(v) for v in t.split(":")) for t in s.split(",")], default=[(3, 0), (2, 2)],
                    help="order:flags pairs for --table, e.g. 3:0,2:2,5:0")
    ap.add_argument("--only", nargs="*", default=None)
    ap.add_argument("--K", type=int, default=10, help="Euler steps of the odeint / euler_jvp ops")
    ap.add_argument("--op", default="loss_grad")
    ap.add_argument("--S", type=int, default=1024)
    ap.add_argument("--N", type=int, default=125000)
    ap.add_argument("--d", type=int, default=2)
    ap.add_argument("--order", type=int, default=5)
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--graph", action="store_true", help="replay the call from a captured HIP graph (no host launch cost)")
    a = ap.parse_args()
    if a.table:
        return table(a)
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
    elif a.op in STREAM_OPS + EXTRA_OPS:
        fn, byt = stream_op(eng, a.op, pts, a.d, a.order, a.flags, a.K)
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
    print(f"{a.op} S={a.S} N={a.N} d={a.d} order={a.order} p={p}: "
          f"min {mn*1e3:.1f} us  med {med*1e3:.1f} us  {pts/mn/1e6:.1f} Gpts/s  {byt/mn/1e6:.0f} GB/s (alg. bytes)")


if __name__ == "__main__":
    main()
