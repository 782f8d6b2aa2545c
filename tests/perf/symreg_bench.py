#!/usr/bin/env python3
"""Closure with each symmetry regulariser (BASELINE config 2 shape: lv, order 2 + exp, N = 20 000 points,
autoencoder MLP 512 x 5 with batch norm, n_comps 2, generator (2,1,2), K = 10 Euler steps) -- GPU path
(HIP kernels for everything that touches the library, stock PyTorch-ROCm for the MLP) vs the CPU oracle.

    python tests/perf/symreg_bench.py [--n 20000] [--hidden 512] [--layers 5]
"""
import argparse
import copy
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import symode_amd
from symode_amd import data, model_utils as MU
from symode_amd.autoencoder import AutoEncoder
from symode_amd.lie import LieGenerator
from oracle import sindy_oracle as O


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=20000)
    ap.add_argument("--hidden", type=int, default=512)
    ap.add_argument("--layers", type=int, default=5)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--cpu_reps", type=int, default=2)
    ap.add_argument("--epochs", type=int, default=3, help="epochs (20 L-BFGS iterations each) of the end-to-end trainer timing")
    a = ap.parse_args()
    dev = "cuda:0"
    torch.manual_seed(0)
    ae = AutoEncoder(ae_arch="mlp", input_dim=2, hidden_dim=a.hidden, latent_dim=2, n_layers=a.layers, n_comps=2,
                     activation="ReLU", activation_args=[], batch_norm=True, ortho_ae=False).eval()
    gen = LieGenerator(repr="(2,1,2)", group_idx="0").eval()
    for p in list(ae.parameters()) + list(gen.parameters()):
        p.requires_grad = False
    ae_c, gen_c = copy.deepcopy(ae), copy.deepcopy(gen)
    ae, gen = ae.to(dev), gen.to(dev)
    gen.masks = [m.to(dev) if m is not None else None for m in gen.masks]
    x, dx = data.make_dataset("lv", 200, 10000, dt=0.002, noise=0.0, seed=0, device=dev)
    rows = torch.randperm(x.shape[1], generator=torch.Generator().manual_seed(0))[:a.n].to(dev)
    x, dx = x[0][rows].contiguous(), dx[0][rows].contiguous()
    xc, dxc = x.cpu(), dx.cpu()
    r = symode_amd.SINDyRegression(2, 2, False, True, threshold=0.15, device=dev)
    Xi0 = r.Xi.detach().cpu().clone() * 0.3
    r.Xi.data = Xi0.to(dev)
    K, dt = 10, 0.01
    flow = MU._EulerFlow(r, K * dt + 1e-9, dt)
    s_i, s_f, s_r = MU.make_symmreg_pttrain(ae, gen), MU.make_fsymmreg_pttrain(ae, gen), MU.make_rsymmreg_pttrain(ae, gen)

    def gpu_closure(kind):
        r.Xi.grad = None
        loss = r.mse_loss(x, dx)
        if kind == "i":
            loss = loss + 0.1 * s_i(torch.stack([x, flow(x)], 1), f=flow, x_const=x)     # as train_SIGED_lbfgs calls it
        elif kind == "f":
            loss = loss + 0.1 * s_f(torch.stack([x, flow(x)], 1), f=flow, x_const=x)
        elif kind == "r":
            loss = loss + 0.1 * s_r(x, h=r)
        loss.backward()
        return loss

    basis_c = gen_c.get_full_basis_list()
    gel_c = gen_c.get_deterministic_group_elems()
    gel_r = gen_c.get_deterministic_group_elems(scale=0.01)
    zm = ae_c.encoder[-2].bias

    def cpu_closure(kind):
        reg = O.OracleRegressor(2, 2, False, True, Xi0=Xi0)
        f = lambda a_: O.odeint(reg, a_, K * dt + 1e-9, dt)  # noqa: E731
        loss = torch.nn.functional.mse_loss(reg(xc), dxc)
        if kind == "i":
            loss = loss + 0.1 * O.symreg_infinitesimal(torch.stack([xc, f(xc)], 1), ae_c.encode, ae_c.decode, zm, basis_c, f)
        elif kind == "f":
            loss = loss + 0.1 * O.symreg_finite(torch.stack([xc, f(xc)], 1), ae_c.encode, ae_c.decode, zm, gel_c, f)
        elif kind == "r":
            loss = loss + 0.1 * O.symreg_reversed(xc, ae_c.encode, ae_c.decode, zm, gel_r, reg)
        loss.backward()
        return loss

    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    torch.set_num_threads(threads)
    print(f"closure = MSE + 0.1 * sym-reg, N = {a.n}, AE {a.hidden} x {a.layers}, K = {K}; CPU oracle on {threads} threads")
    print("| regulariser | GPU ms | CPU ms | speedup | GPU loss | CPU loss |")
    print("|---|---|---|---|---|---|")
    for kind in ("none", "i", "f", "r"):
        for _ in range(3):
            lg = gpu_closure(kind)
        torch.cuda.synchronize()
        ts = []
        for _ in range(a.reps):
            t0 = time.perf_counter()
            lg = gpu_closure(kind)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        tg = statistics.median(ts)
        tcs = []
        for _ in range(a.cpu_reps):
            t0 = time.perf_counter()
            lc = cpu_closure(kind)
            tcs.append(time.perf_counter() - t0)
        tc = min(tcs)
        print(f"| {kind} | {tg*1e3:.2f} | {tc*1e3:.1f} | {tc/tg:.0f}x | {lg.item():.6f} | {lc.item():.6f} |")

    # end to end: train_SIGED_lbfgs with each regulariser, L-BFGS variables on the host (default) vs on the device
    import contextlib
    import io
    print()
    print("| regulariser | epochs | trainer, host L-BFGS variables (s) | trainer, device variables (s) |")
    print("|---|---|---|---|")
    for kind in ("i", "f", "r"):
        row = []
        for host in (True, False):
            for rep in range(2):                                  # first run warms up
                rr = symode_amd.SINDyRegression(2, 2, False, True, threshold=0.15, device=dev)
                rr.Xi.data = Xi0.to(dev)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                with contextlib.redirect_stdout(io.StringIO()):
                    symode_amd.train.train_SIGED_lbfgs(
                        train_loader=[(x, dx)], test_loader=[], num_epochs=a.epochs, device=dev, log_interval=10 ** 9, save_interval=10 ** 9,
                        save_dir="symreg_tmp", autoencoder=ae, generator=gen, regressor=rr, regressor_dst=None, use_latent=False,
                        distill_latent=False, lr_sindy=0.1, w_sindy_z=0.0, w_sindy_x=1.0, sindy_reg_type="l1", w_sindy_reg=0.0,
                        sym_reg_type=kind, w_sym_reg=0.1, st_freq=100, threshold=0.15, int_t=K * dt + 1e-9, int_dt=dt, print_eq=False,
                        host_lbfgs=host)
                torch.cuda.synchronize()
                t = time.perf_counter() - t0
            row.append(t)
        print(f"| {kind} | {a.epochs} | {row[0]:.3f} | {row[1]:.3f} |")


if __name__ == "__main__":
    main()
