#!/usr/bin/env python3
"""Per-operation GPU (HIP path) vs CPU (oracle = port of the reference op sequence) timings, following the
protocol of BASELINE.md section 3: 3 warm-ups, median of >= 20 repetitions, host threads stated, same inputs.

    python tests/perf/ops_table.py > profiles/r01_ops_table.md
"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import symode_amd
from symode_amd import data
from oracle import sindy_oracle as O

DEV = "cuda:0"


def med_cpu(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts)


def med_gpu(fn, reps=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts)


def main():
    avail = len(os.sched_getaffinity(0))
    threads = max(1, min(16, avail))
    eng = symode_amd.get_engine()
    x, dx = data.make_dataset("dosc", 50, 2500, dt=0.02, noise=0.2, seed=0, device=DEV)
    x, dx = x[0], dx[0]
    xc, dxc = x.cpu(), dx.cpu()
    N = x.shape[0]
    rows = []

    def add(name, n_pts, gpu_fn, cpu_fn):
        tg = med_gpu(gpu_fn)
        torch.set_num_threads(threads)
        tc = med_cpu(cpu_fn)
        torch.set_num_threads(1)
        t1 = med_cpu(cpu_fn, reps=5, warm=1)
        torch.set_num_threads(threads)
        rows.append((name, n_pts, tg, tc, t1))

    for order in (3, 5):
        p = O.term_count(2, order)
        torch.manual_seed(order)
        Xi = torch.randn(2, p) * 0.3
        mask = torch.ones(2, p)
        Xig, mg = Xi.to(DEV), mask.to(DEV)
        add(f"Theta only (eval_Theta_at), order {order}", N, lambda: eng.theta(x, order), lambda: O.theta(xc, order))
        add(f"closure body: forward + MSE + backward, order {order}", N,
            lambda: eng.loss_grad(x, dx, Xig, mg, order), lambda: O.mse_loss_and_grad(xc, dxc, Xi, mask, order))
    # STLSQ one pass, full and partial mask (order 3)
    reg_g = symode_amd.SINDyRegression(2, 3, False, False, threshold=1e-9, device=DEV)

    def gpu_stlsq(partial):
        reg_g.reset_mask()
        if partial:
            reg_g.mask[:, 3:6] = 0.0
        reg_g._gram_cache = None                       # time the data pass too
        symode_amd.solve_SINDy_one_step(reg_g, x, dx, 0.05, 1e-9)

    def cpu_stlsq(partial):
        r = O.OracleRegressor(2, 3, threshold=1e-9, Xi0=torch.zeros(2, 10))
        if partial:
            r.mask[:, 3:6] = 0.0
        O.stlsq_one_step(r, xc, dxc, 0.05, 1e-9)

    add("solve_SINDy_one_step, full mask, order 3", N, lambda: gpu_stlsq(False), lambda: cpu_stlsq(False))
    add("solve_SINDy_one_step, partial mask (block-diag in the reference), order 3", N, lambda: gpu_stlsq(True), lambda: cpu_stlsq(True))
    # S1 linear-latent symmetry term, N = 20000
    n2 = 20000
    z, zc = x[:n2].contiguous(), xc[:n2].contiguous()
    L = torch.tensor([[[0.0, 1.0], [-1.0, 0.0]]])
    Xi3 = torch.randn(2, 10) * 0.3
    Xi3g, Lg = Xi3.to(DEV), L.to(DEV)

    def cpu_s1():
        r = O.OracleRegressor(2, 3, Xi0=Xi3)
        O.symreg_linear_latent(zc, list(L), r).backward()

    add("S1 linear-latent sym-reg loss + grad, order 3", n2, lambda: eng.symreg_linear(z, Xi3g, None, Lg, 3), cpu_s1)
    # S4 reversed with precomputed (g(x), J_g)
    gx = (zc + 0.01 * torch.randn_like(zc))
    J = torch.eye(2).repeat(n2, 1, 1) + 0.01 * torch.randn(n2, 2, 2)
    gxg, Jg = gx[None].to(DEV).contiguous(), J[None].to(DEV).contiguous()

    def cpu_s4():
        r = O.OracleRegressor(2, 3, Xi0=Xi3)
        O.symreg_reversed_precomputed(zc, [gx], [J], r).backward()

    add("S4 reversed sym-reg (precomputed g, J_g) loss + grad, order 3", n2, lambda: eng.symreg_reversed(z, gxg, Jg, Xi3g, None, 3), cpu_s4)
    # K-step Euler forward
    add("odeint, 10 Euler steps, order 3", N, lambda: eng.odeint(x, Xi3g, None, 3, 0, 10, 0.01),
        lambda: O.odeint(lambda a: O.forward(a, Xi3, torch.ones(2, 10), 3), xc, 0.1 + 1e-9, 0.01))

    print(f"# Per-operation timings, MI355X vs host CPU\n")
    print(f"Inputs: damped oscillator 50x2500x2 fp32 (N = {N} points) unless stated; GPU = HIP path through the C ABI "
          f"(wall time per call incl. Python + launch + sync, median of 50); CPU = oracle port of the reference op "
          f"sequence, torch {torch.__version__}, median of 20 on {threads} threads (box share; {os.cpu_count()} logical CPUs visible) "
          f"and of 5 on 1 thread.\n")
    print("| operation | points | GPU ms | CPU ms (%d thr) | CPU ms (1 thr) | speedup vs %d thr | GPU Mpts/s |" % (threads, threads))
    print("|---|---|---|---|---|---|---|")
    for name, n, tg, tc, t1 in rows:
        print(f"| {name} | {n} | {tg*1e3:.3f} | {tc*1e3:.3f} | {t1*1e3:.3f} | {tc/tg:.1f}x | {n/tg/1e6:.0f} |")


if __name__ == "__main__":
    main()
