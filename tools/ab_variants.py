#!/usr/bin/env python3
"""A/B of SYMODE_LOSS_GRAD_VARIANT builds inside ONE gpurun call (box-to-box variance is ~10 %).
Variants 2, 5, 6, 8 (measured and rejected in round 1, profiles/r01_ab_variants.txt) are compiled only by
`make -C symmetry-ode-discovery_amd/csrc AB=1`; the default build holds 0, 4 and 7 (the default schedule).

Each variant runs in its own child process (the variant is latched at first launch); every child
times the batched closure and dumps loss/grad so the parent can check the variants agree bit for bit.

    python tools/ab_variants.py --variants 4 6 --orders 3 5 --d 2
"""
import argparse
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, os, torch, numpy as np
sys.path.insert(0, %(root)r)
import symode_amd
S, N, d, order, flags, out = %(S)d, %(N)d, %(d)d, %(order)d, %(flags)d, %(out)r
eng = symode_amd.get_engine()
p = eng.lib_size(d, order, flags)
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn(S, N, d, device='cuda', generator=g) * 0.7
dx = torch.randn(S, N, d, device='cuda', generator=g)
xi = torch.randn(S, d, p, device='cuda', generator=g) * 0.3
fn = lambda: eng.loss_grad(x, dx, xi, None, order, flags)
best = []
for _ in range(5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    best.append(e0.elapsed_time(e1) / 20)
loss, grad = fn()
np.savez(out, loss=loss.cpu().numpy(), grad=grad.cpu().numpy(), ms=np.array([min(best), sorted(best)[2]]))
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", type=int, nargs="+", default=[4, 6])
    ap.add_argument("--orders", type=int, nargs="+", default=[3, 5])
    ap.add_argument("--d", type=int, default=2)
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--S", type=int, default=2048)
    ap.add_argument("--N", type=int, default=125000)
    ap.add_argument("--rounds", type=int, default=2)
    a = ap.parse_args()
    for order in a.orders:
        ref = None
        for rnd in range(a.rounds):
            for v in a.variants:
                out = f"/tmp/ab_{order}_{v}.npz"
                env = dict(os.environ, SYMODE_LOSS_GRAD_VARIANT=str(v))
                code = CHILD % dict(root=ROOT, S=a.S, N=a.N, d=a.d, order=order, flags=a.flags, out=out)
                subprocess.run([sys.executable, "-c", code], env=env, check=True)
                r = np.load(out)
                byt = a.S * a.N * 8 * a.d
                same = ""
                if ref is None:
                    ref = (r["loss"], r["grad"])
                else:
                    same = " bit-identical=%s maxrel=%.2e" % (
                        bool(np.array_equal(ref[0], r["loss"]) and np.array_equal(ref[1], r["grad"])),
                        float(np.max(np.abs(ref[1] - r["grad"]) / (np.abs(ref[1]) + 1e-30))))
                print(f"d={a.d} order={order} variant={v} round={rnd}: min {r['ms'][0]:.4f} ms  med {r['ms'][1]:.4f} ms"
                      f"  {byt / r['ms'][0] / 1e6:.0f} GB/s{same}", flush=True)


if __name__ == "__main__":
    main()
