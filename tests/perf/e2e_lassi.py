#!/usr/bin/env python3
"""BASELINE config 5 end to end: reaction-diffusion latent SINDy (rd/sym_eq.cfg flags) on a synthetic 128x128
rotating-spiral field, `train_lassi` for a few epochs on the GPU; reports seconds per epoch and where a batch's
time goes (autoencoder + GAN on stock PyTorch-ROCm vs the latent least-squares solve on the HIP engine), and the
same solve on the CPU oracle for scale.

    python tests/perf/e2e_lassi.py --epochs 3
"""
import argparse
import contextlib
import io
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import symode_amd
from symode_amd import dataset as D, parser_utils
from symode_amd.autoencoder import AutoEncoder
from symode_amd.lie import Discriminator, LieGenerator


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--n_samples", type=int, default=1000)
    ap.add_argument("--grid", type=int, default=128)
    a = ap.parse_args()
    dev = "cuda:0"
    os.chdir(tempfile.mkdtemp())
    D.RD_SYNTH.update(n=a.grid, n_samples=a.n_samples)
    torch.manual_seed(0)
    np.random.seed(0)
    cfg = os.path.join(os.path.dirname(symode_amd.__file__), "run_configs", "rd", "sym_eq.cfg")
    argv = parser_utils.parse_config(cfg) + ["--num_epochs", str(a.epochs), "--log_interval", "1000", "--save_interval", "1000"]
    args = vars(parser_utils.get_args(argv=argv))
    args["device"] = dev
    with contextlib.redirect_stdout(io.StringIO()):
        tr, va, args = D.get_dataset(args)
    ae, disc, gen = AutoEncoder(**args).to(dev), Discriminator(**args).to(dev), LieGenerator(**args).to(dev)
    args["L_list"] = [L[:2, :2].detach().cpu() for L in gen.get_full_basis_list()]
    with contextlib.redirect_stdout(io.StringIO()):
        reg = symode_amd.SINDyRegression(**args).to(dev)
    tl = D.make_loader(tr, args["batch_size"], True, dev)          # as main.py builds them: device-resident, one gather per batch
    vl = D.make_loader(va, args["batch_size"], False, dev)
    run = lambda n: symode_amd.train.train_lassi(autoencoder=ae, discriminator=disc, generator=gen, regressor=reg,  # noqa: E731
                                                  regressor_dst=None, train_loader=tl, test_loader=vl, **dict(args, num_epochs=n))
    with contextlib.redirect_stdout(io.StringIO()):
        run(1)                                                   # warm-up (allocator, hipBLASLt heuristics)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        rec = run(a.epochs)
    torch.cuda.synchronize()
    per_epoch = (time.perf_counter() - t0) / a.epochs
    n_batches = len(tl)
    print(f"train_lassi: {per_epoch:.3f} s/epoch, {n_batches} batches of {args['batch_size']} x 2 x {args['input_dim']}"
          f" -> {per_epoch / n_batches * 1e3:.2f} ms/batch; final loss_ae {rec['loss_ae']:.4f} loss_sindy_z {rec['loss_sindy_z']:.4f}")

    # the latent solve alone, on one batch of real latents
    from oracle import sindy_oracle as O                         # CPU reference timing only (tools/, not the product path)
    ae.eval()
    xb, dxb = next(iter(tl))
    xb, dxb = xb.to(dev), dxb.to(dev)
    z = ae.encode(xb)[:, 0].detach().contiguous()
    dz = ae.compute_dz(xb, dxb)[:, 0].contiguous()
    reps = 50
    zs = [z.clone() for _ in range(reps)]                        # fresh tensors: the Gram cache keys on the object
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for zi in zs:
        symode_amd.sindy.solve_SINDy(reg, zi, dz, args["w_sindy_reg"], args["threshold"])
    torch.cuda.synchronize()
    t_hip = (time.perf_counter() - t0) / reps
    zc, dzc = z.cpu(), dz.cpu()
    ro = O.OracleRegressor(2, 2, L_list=[L.clone() for L in reg.L_list], threshold=args["threshold"], constrain_constant=True)
    t0 = time.perf_counter()
    for _ in range(reps):
        O.stlsq(ro, zc, dzc, args["w_sindy_reg"], args["threshold"])
    t_cpu = (time.perf_counter() - t0) / reps
    print(f"latent solve_SINDy on (64, 2): HIP path {t_hip * 1e3:.3f} ms, CPU oracle {t_cpu * 1e3:.3f} ms"
          f" ({t_hip / (per_epoch / n_batches) * 100:.1f} % of a batch)")


if __name__ == "__main__":
    main()
