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


# =================================================================================================
# Seed sweep of the L-BFGS fit (train_SIGED_lbfgs for S seeds at once)
# =================================================================================================
class BatchedLBFGS:
    """torch.optim.LBFGS (no line search), restated for S independent problems stepped in lockstep.

    Every problem keeps its own state (iteration count, direction, step, curvature history, scale of the
    initial Hessian) in padded (S, ...) tensors; one ``step`` evaluates the closure for all problems with
    one fused kernel launch per inner iteration and applies, per problem, exactly the update rules and
    stopping tests of torch/optim/lbfgs.py (defaults: max_iter 20, tolerance_grad 1e-7, tolerance_change
    1e-9, history 100).  Problems that stop early simply stop changing.
    """

    def __init__(self, params, lr, max_iter=20, tolerance_grad=1e-7, tolerance_change=1e-9, history_size=100):
        self.P = params                                   # (S, n), updated in place
        S, n = params.shape
        self.lr, self.max_iter, self.tol_g, self.tol_c, self.H = lr, max_iter, tolerance_grad, tolerance_change, history_size
        dev, dt = params.device, params.dtype
        self.n_iter = torch.zeros(S, dtype=torch.long, device=dev)
        self.d = torch.zeros(S, n, device=dev, dtype=dt)
        self.t = torch.zeros(S, device=dev, dtype=dt)
        self.old_dirs = torch.zeros(S, history_size, n, device=dev, dtype=dt)
        self.old_stps = torch.zeros(S, history_size, n, device=dev, dtype=dt)
        self.ro = torch.zeros(S, history_size, device=dev, dtype=dt)
        self.hist = torch.zeros(S, dtype=torch.long, device=dev)
        self.H_diag = torch.ones(S, device=dev, dtype=dt)
        self.prev_g = torch.zeros(S, n, device=dev, dtype=dt)
        self.prev_loss = torch.zeros(S, device=dev, dtype=dt)

    def reset(self, which):
        """Fresh optimiser for the selected problems (the reference re-creates LBFGS after thresholding)."""
        self.n_iter[which] = 0
        self.hist[which] = 0
        self.H_diag[which] = 1.0

    @torch.no_grad()
    def step(self, closure, frozen=None):
        """closure(P) -> (loss (S,), grad (S, n)).  ``frozen`` (S,) bool: problems that must not move."""
        P, S = self.P, self.P.shape[0]
        loss, g = closure(P)
        loss, g = loss.clone(), g.clone()
        act = g.abs().amax(dim=1) > self.tol_g                              # optimality test
        if frozen is not None:
            act &= ~frozen
        idx = torch.arange(S, device=P.device)
        for it in range(1, self.max_iter + 1):
            if not bool(act.any()):
                break
            self.n_iter[act] += 1
            first = act & (self.n_iter == 1)
            upd = act & ~first
            # ---- direction ------------------------------------------------------------------
            if bool(first.any()):
                self.d[first] = -g[first]
                self.hist[first] = 0
                self.H_diag[first] = 1.0
            if bool(upd.any()):
                y = g - self.prev_g
                s = self.d * self.t[:, None]
                ys = (y * s).sum(1)
                mem = upd & (ys > 1e-10)
                if bool(mem.any()):
                    full = mem & (self.hist == self.H)
                    if bool(full.any()):                                   # limited memory: drop the oldest pair
                        self.old_dirs[full] = torch.roll(self.old_dirs[full], -1, dims=1)
                        self.old_stps[full] = torch.roll(self.old_stps[full], -1, dims=1)
                        self.ro[full] = torch.roll(self.ro[full], -1, dims=1)
                        self.hist[full] -= 1
                    rows, pos = idx[mem], self.hist[mem]
                    self.old_dirs[rows, pos] = y[mem]
                    self.old_stps[rows, pos] = s[mem]
                    self.ro[rows, pos] = 1.0 / ys[mem]
                    self.hist[mem] += 1
                    self.H_diag[mem] = ys[mem] / (y[mem] * y[mem]).sum(1)
                # two-loop recursion over the padded history (slots beyond hist[s] are skipped per problem)
                num_old = int(self.hist[upd].max().item())
                q = -g
                al = torch.zeros(S, max(num_old, 1), device=P.device, dtype=P.dtype)
                for i in range(num_old - 1, -1, -1):
                    live = (i < self.hist).to(P.dtype)
                    a = (self.old_stps[:, i] * q).sum(1) * self.ro[:, i] * live
                    al[:, i] = a
                    q = q - a[:, None] * self.old_dirs[:, i]
                r = q * self.H_diag[:, None]
                for i in range(num_old):
                    live = (i < self.hist).to(P.dtype)
                    be = (self.old_dirs[:, i] * r).sum(1) * self.ro[:, i] * live
                    r = r + self.old_stps[:, i] * ((al[:, i] - be) * live)[:, None]
                self.d[upd] = r[upd]
            self.prev_g[act] = g[act]
            self.prev_loss[act] = loss[act]
            # ---- step length -----------------------------------------------------------------
            t_first = torch.clamp(1.0 / g.abs().sum(1), max=1.0) * self.lr
            self.t[act] = torch.where(self.n_iter[act] == 1, t_first[act], torch.full_like(t_first[act], self.lr))
            gtd = (g * self.d).sum(1)
            act = act & ~(gtd > -self.tol_c)                               # directional derivative below tolerance
            if not bool(act.any()):
                break
            P[act] += self.t[act, None] * self.d[act]
            if it != self.max_iter:                                        # no re-evaluation on the last iteration
                nl, ng = closure(P)
                loss[act], g[act] = nl[act], ng[act]
                stop = (g.abs().amax(1) <= self.tol_g) | ((self.d * self.t[:, None]).abs().amax(1) <= self.tol_c) \
                    | ((loss - self.prev_loss).abs() < self.tol_c)
                act = act & ~stop
        return loss


class SeedSweepLBFGS:
    """``train_SIGED_lbfgs`` (non-latent, MSE [+ L1]) for S seeds in lockstep: the per-epoch logic of
    train.py:692-725 -- NaN guard, update-norm convergence test, threshold + optimiser reset, final
    convergence -- evaluated per problem on (S, ...) tensors; the closure of all seeds is ONE launch of the
    fused Theta + residual + gradient kernel (BatchedClosure), all-reduced over point shards if sharded."""

    def __init__(self, closure, lr_sindy, threshold, st_freq, w_sindy_x=1.0, sindy_reg_type="l1", w_sindy_reg=0.0, tol=1e-3):
        self.c = closure
        self.lr, self.threshold, self.st_freq, self.tol = lr_sindy, threshold, st_freq, tol
        self.w_x, self.reg_type, self.w_reg = w_sindy_x, sindy_reg_type, w_sindy_reg
        if sindy_reg_type not in ("l1", "none"):
            raise ValueError(f"Unknown regularization type: {sindy_reg_type}")

    def _split(self, P):
        c = self.c
        if c.Q is None:
            return P.view(c.S, c.d, c.p), None
        r = c.Q.shape[1]
        return P[:, :r], P[:, r:].reshape(c.S, c.d, 1)

    def _closure(self, P):
        c = self.c
        a, b = self._split(P)
        loss, ga, gb = c.evaluate(a.contiguous(), b, mask=self.mask)
        g = ga.reshape(c.S, -1) if gb is None else torch.cat([ga, gb.reshape(c.S, -1)], dim=1)
        loss, g = self.w_x * loss, self.w_x * g
        if self.reg_type == "l1":                                           # over the raw parameters (train.py:681)
            loss = loss + self.w_reg * P.abs().sum(1)
            g = g + self.w_reg * torch.sign(P)
        return loss, g

    def _xi(self, P):
        a, b = self._split(P)
        return self.c.xi_from(a.contiguous(), b)

    def _norms(self, A, B):
        """sum over parameter tensors of ||A - B|| per problem (train.py:702-704)."""
        c = self.c
        if c.Q is None:
            return (A - B).norm(dim=1)
        r = c.Q.shape[1]
        return (A[:, :r] - B[:, :r]).norm(dim=1) + (A[:, r:] - B[:, r:]).norm(dim=1)

    @torch.no_grad()
    def fit(self, P0, num_epochs, mask0=None):
        """P0 (S, n): initial flat parameters per seed ([Xi] or [beta | const]).
        Returns dict(Xi, mask, params, epochs (S,), finished (S,), nan (S,))."""
        c = self.c
        P = P0.clone().contiguous()
        S = P.shape[0]
        self.mask = torch.ones(S, c.d, c.p, device=P.device) if mask0 is None else mask0.clone()
        opt = BatchedLBFGS(P, self.lr)
        prev, pprev = P.clone(), P.clone()
        n_iters = torch.zeros(S, dtype=torch.long, device=P.device)
        done = torch.zeros(S, dtype=torch.bool, device=P.device)
        nan = torch.zeros(S, dtype=torch.bool, device=P.device)
        epochs = torch.zeros(S, dtype=torch.long, device=P.device)
        for epoch in range(num_epochs):
            live = ~done
            if not bool(live.any()):
                break
            n_iters[live] += 1
            epochs[live] = epoch + 1
            opt.step(self._closure, frozen=done)
            bad = live & torch.isnan(P).any(dim=1)                         # train.py:697-699
            nan |= bad
            done |= bad
            live = ~done
            upd = self._norms(P, prev)
            conv = live & (upd < self.tol)
            final = conv & (self._norms(P, pprev) < self.tol)             # train.py:709-714
            done |= final
            thr_conv = conv & ~final
            thr_freq = live & ~conv & (n_iters % self.st_freq == 0) if self.st_freq > 0 else torch.zeros_like(done)
            ev = thr_conv | thr_freq
            if bool(ev.any()):
                Xi = self._xi(P)
                new_mask = torch.logical_and(Xi.abs() > self.threshold, self.mask > 0).float()
                self.mask[ev] = new_mask[ev]                               # strict >, monotone (sindy.py:194)
                opt.reset(ev)
                n_iters[ev] = 0
                pprev[thr_conv] = P[thr_conv]                              # only on convergence-triggered events (:718)
            upd_prev = live & ~final
            prev[upd_prev] = P[upd_prev]
        return {"Xi": self._xi(P), "mask": self.mask, "params": P, "epochs": epochs, "finished": done & ~nan, "nan": nan}
