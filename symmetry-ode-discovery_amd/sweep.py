"""Seed sweep of the sequential-threshold least-squares fit, batched on the GPU.

The reference sweeps seeds with a shell loop, one process per seed (run_scripts/*.sh,
``for i in {0..49}``): every seed draws its own random subsample of the same data set
(``--lbfgs_subsample``, main.py:36-38) and fits it.  For the least-squares fit everything a seed
needs is its augmented Gram matrix, so the whole sweep is ONE gather kernel launch over the
(seed, point) index table (symode_aug_gram_gather, fp64 MFMA), an optional all-reduce of the
(S, (p+d)^2) fp64 matrices over the ranks' point shards (RCCL over xGMI), and S tiny host solves
per thresholding pass.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from .engine import get_engine, library_flags
from .sindy import stlsq_solve_from_gram


class SeedSweepSTLSQ:
    def __init__(self, x, dx, poly_order, include_sine=False, include_exp=False, n_seeds=64, subsample=0.5, seed0=0,
                 group=None, engine=None):
        """x, dx: this rank's shard (N_local, d) of the flattened data set (dataset.py:193-194)."""
        assert x.dim() == 2 and x.shape == dx.shape
        self.engine = engine or get_engine()
        self.x, self.dx = x.contiguous(), dx.contiguous()
        self.n_local, self.d = x.shape
        self.order, self.flags = poly_order, library_flags(include_sine, include_exp)
        self.p = self.engine.lib_size(self.d, poly_order, self.flags)
        self.S = n_seeds
        self.group = group
        rank = dist.get_rank(group) if group is not None else 0
        m = max(1, int(self.n_local * subsample))
        rows = []
        for s in range(n_seeds):                      # seeded permutation per (seed, rank): reproducible subsets
            g = torch.Generator().manual_seed(1_000_003 * (seed0 + s) + rank)
            rows.append(torch.randperm(self.n_local, generator=g)[:m])
        self.idx = torch.stack(rows).to(torch.int32).to(x.device)
        self.m_local = m
        self._gram = None

    def grams(self):
        """(S, p+d, p+d) fp64 on the host, summed over the ranks' shards."""
        if self._gram is None:
            G = self.engine.aug_gram_gather(self.x, self.dx, self.idx, self.order, self.flags)
            n = torch.tensor([float(self.m_local)], dtype=torch.float64, device=G.device)
            if self.group is not None:
                dist.all_reduce(G, op=dist.ReduceOp.SUM, group=self.group)
                dist.all_reduce(n, op=dist.ReduceOp.SUM, group=self.group)
            self._gram, self.n_points = G.cpu().numpy(), int(n.item())
        return self._gram

    def solve(self, w_sindy_reg, threshold, max_iter=10, lstsq_driver="gelsy"):
        """STLSQ to convergence for every seed (train.py:872-887 per seed).
        Returns (Xi (S, d, p) float32, mask (S, d, p) float32, passes (S,))."""
        G = self.grams()
        S, d, p = self.S, self.d, self.p
        Xi = np.zeros((S, d, p), dtype=np.float32)
        mask = np.ones((S, d, p), dtype=bool)
        passes = np.zeros(S, dtype=np.int64)
        for s in range(S):
            for it in range(max_iter):
                xi, _ = stlsq_solve_from_gram(G[s], self.n_points, mask[s], float(w_sindy_reg), d, lstsq_driver)
                xi32 = xi.astype(np.float32)
                new_mask = np.logical_and(np.abs(xi32) > np.float32(threshold), mask[s])   # strict >, monotone (sindy.py:194)
                converged = np.array_equal(new_mask, mask[s])
                Xi[s], mask[s], passes[s] = xi32, new_mask, it + 1
                if converged:
                    break
        return torch.from_numpy(Xi), torch.from_numpy(mask.astype(np.float32)), passes
