import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import symode_amd
from symode_amd.batched import BatchedClosure
from symode_amd.sweep import SeedSweepLBFGS, BatchedLBFGS
g = np.load("tests/golden/f4_lbfgs.npz")
DEV = "cuda"
t = lambda a: torch.from_numpy(np.asarray(a)).float()
x, dx = t(g["dosc_sindy_x"]).to(DEV), t(g["dosc_sindy_dx"]).to(DEV)
torch.manual_seed(3)
inits = torch.cat([t(g["dosc_sindy_init_Xi"]).reshape(1, -1), torch.randn(5, 20)]).to(DEV)
rep = lambda v, n: v[None].expand(n, -1, -1).contiguous()
for fused in ("1", "0"):
    os.environ["SYMODE_LBFGS_FUSED"] = fused
    os.environ["SYMODE_SWEEP_GRAPH"] = "0"
    sw = SeedSweepLBFGS(BatchedClosure(rep(x, 6), rep(dx, 6), 3), 0.1, 0.05, 50, w_sindy_x=0.1, sindy_reg_type="l1", w_sindy_reg=0.1)
    # manual: a few optimiser steps
    P = inits.clone()
    sw.mask = torch.ones(6, 2, 10, device=DEV)
    sw._can_alias = True
    opt = BatchedLBFGS(P, 0.1, engine=sw.c.engine)
    if opt.fused:
        opt.data_term = (sw._data_term, sw.w_x, sw.w_reg)
    for ep in range(4):
        l = opt.step(sw._closure)
        print(fused, ep, l.cpu().numpy().round(5), opt.n_iter.cpu().numpy(), opt._act.cpu().numpy(), P[0, :4].cpu().numpy())
