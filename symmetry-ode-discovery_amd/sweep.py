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

import os

import numpy as np
import torch
import torch.distributed as dist

from .engine import get_engine, library_flags
from . import lstsq as _lstsq
from .sindy import NEAR_THRESHOLD_BAND, near_threshold_cases, stlsq_solve_from_gram


def seeded_subsamples(n, m, seeds, device, dtype=torch.int64):
    """(len(seeds), m) index table, rows ascending: for every seed an m-subset of range(n) that depends on THAT seed alone -- a
    seed's subsample (main.py:36-38: the first batch of a shuffled loader) does not depend on which other seeds run beside
    it, nor on the world size (every rank draws the same rows and takes its slice).
    On the GPU: ONE launch for all seeds (symode_seeded_subsamples: the m smallest of n counter-based keys per seed, radix
    select + ordered compaction, a workgroup per seed) -- the torch form below (a generator launch per seed, one batched
    top-k, one sort) took 1.1-1.2 ms for 64 seeds of 10^5 rows, three quarters of the STLSQ sweep.  On the CPU (tests of the
    host logic): uniform keys from a torch generator seeded by the seed, the m smallest kept."""
    device = torch.device(device)
    if m >= n:
        return torch.arange(n, device=device, dtype=dtype).expand(len(seeds), n).contiguous()
    if device.type == "cuda":
        return get_engine().seeded_subsamples(n, m, seeds, device).to(dtype)
    g = torch.Generator(device=device)
    keys = torch.empty(len(seeds), n, dtype=torch.float64, device=device)          # fp64 keys: ties are not a concern
    for row, seed in zip(keys, seeds):
        g.manual_seed(int(seed))
        row.uniform_(generator=g)
    idx = torch.topk(keys, m, dim=1, largest=False, sorted=False).indices
    return torch.sort(idx, dim=1).values.to(dtype)


class SeedSweepSTLSQ:
    def __init__(self, x, dx, poly_order, include_sine=False, include_exp=False, n_seeds=64, subsample=0.5, seed0=0,
                 group=None, engine=None, idx=None, idx_sorted=False):
        """x, dx: (N_local, d) rows of the flattened data set (dataset.py:193-194) this rank gathers from.
        ``idx`` (S, m_local) int32: the rows of x this rank contributes to each seed's subsample; by default every
        (seed, rank) draws its own seeded permutation of the rank's shard."""
        assert x.dim() == 2 and x.shape == dx.shape
        self.engine = engine or get_engine()
        self.x, self.dx = x.contiguous(), dx.contiguous()
        self.n_local, self.d = x.shape
        self.order, self.flags = poly_order, library_flags(include_sine, include_exp)
        self.p = self.engine.lib_size(self.d, poly_order, self.flags)
        self.S = n_seeds
        self.group = group
        if idx is not None:
            assert idx.dim() == 2 and idx.shape[0] == n_seeds
            self.idx = idx.to(torch.int32).to(x.device).contiguous()
            self.m_local = idx.shape[1]
        # a seed's Gram is a sum over its rows: visit them in ascending order (near-sequential reads of x, dx)
        if idx is not None:
            if not idx_sorted:                               # (seeded_subsamples' tables already are)
                self.idx = torch.sort(self.idx, dim=1).values.contiguous()
        else:
            rank = dist.get_rank(group) if group is not None else 0
            m = max(1, int(self.n_local * subsample))
            # seeded subset per (seed, rank): reproducible
            self.idx = seeded_subsamples(self.n_local, m, [1_000_003 * (seed0 + s) + rank for s in range(n_seeds)],
                                         x.device, dtype=torch.int32)
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

    def solve(self, w_sindy_reg, threshold, max_iter=10, lstsq_driver=None):
        """STLSQ to convergence for every seed (train.py:872-887 per seed).
        Returns (Xi (S, d, p) float32, mask (S, d, p) float32, passes (S,))."""
        G = self.grams()
        if lstsq_driver is None:                   # torch.linalg.lstsq's default on the device the data lives on (sindy.py)
            lstsq_driver = "gels" if self.x.is_cuda else "gelsy"
        S, d, p = self.S, self.d, self.p
        Xi = np.zeros((S, d, p), dtype=np.float32)
        mask = np.ones((S, d, p), dtype=bool)
        passes = np.zeros(S, dtype=np.int64)
        self.near_threshold = []              # (seed index, pass, row, col, |coef|): BASELINE.md section 3
        todo = range(S)
        lib = _lstsq._native() if os.environ.get('SYMODE_STLSQ_NATIVE', '1') != '0' else None
        if lib and lstsq_driver in ('gels', 'gelsy'):
            # the whole loop below for all seeds in ONE native call (host_lstsq.cpp::symode_host_stlsq_sweep: same solver, same
            # inputs, same arithmetic); seeds that met a near-threshold coefficient are replayed here for the detailed record
            Gc = np.ascontiguousarray(G, dtype=np.float64)
            m8 = np.ones((S, d, p), dtype=np.uint8)
            p32, near, fell = np.zeros(S, dtype=np.int32), np.zeros(S, dtype=np.int32), np.zeros(S, dtype=np.int32)
            rc = lib.symode_host_stlsq_sweep(Gc.ctypes.data, S, d, p, int(self.n_points), float(w_sindy_reg), float(threshold),
                                             int(max_iter), 0 if lstsq_driver == 'gelsy' else 1, float(NEAR_THRESHOLD_BAND),
                                             Xi.ctypes.data, m8.ctypes.data, p32.ctypes.data, near.ctypes.data, fell.ctypes.data)
            if rc != 0:
                raise RuntimeError(f'symode_host_stlsq_sweep failed with code {rc}')
            if fell.any():
                import warnings
                warnings.warn('singular normal equations under the full-rank (gels) driver: minimum-norm solution returned instead',
                              RuntimeWarning)
            mask, passes = m8.astype(bool), p32.astype(np.int64)
            todo = [int(s) for s in np.nonzero(near)[0]]
            for s in todo:
                mask[s] = True
        for s in todo:
            for it in range(max_iter):
                xi, _ = stlsq_solve_from_gram(G[s], self.n_points, mask[s], float(w_sindy_reg), d, lstsq_driver)
                xi32 = xi.astype(np.float32)
                self.near_threshold += [(s, it, i, k, v) for i, k, v in near_threshold_cases(xi32, mask[s], threshold)]
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
    initial Hessian) in (S, ...) tensors; one ``step`` evaluates the closure of all problems with one
    fused kernel launch per inner iteration and applies, per problem, exactly the update rules and
    stopping tests of torch/optim/lbfgs.py (defaults: max_iter 20, tolerance_grad 1e-7, tolerance_change
    1e-9, history 100).  Problems that stop early simply stop changing.

    Everything is mask arithmetic -- no host synchronisation inside ``step``.  With the variables on the GPU the
    optimiser side of an inner iteration is TWO kernels, a wavefront per problem: symode_lbfgs_update (curvature
    memory in ring buffers with per-problem head / count, two-loop recursion, step length, move) and
    symode_lbfgs_accept (take the re-evaluated loss / gradient, stopping tests); the tensor-op form below is the
    same arithmetic statement by statement (CPU / gloo runs, ``SYMODE_LBFGS_FUSED=0``) with only the two-loop
    recursion as a kernel (symode_lbfgs_direction).
    """

    def __init__(self, params, lr, max_iter=20, tolerance_grad=1e-7, tolerance_change=1e-9, history_size=100, engine=None,
                 use_graph=False):
        self.P = params                                   # (S, n), updated in place
        S, n = params.shape
        self.lr, self.max_iter, self.tol_g, self.tol_c, self.H = lr, max_iter, tolerance_grad, tolerance_change, history_size
        dev, dt = params.device, params.dtype
        self.engine = engine if (engine is not None and params.is_cuda and n <= 256 and history_size <= 128) else None
        self.n_iter = torch.zeros(S, dtype=torch.long, device=dev)
        self.d = torch.zeros(S, n, device=dev, dtype=dt)
        self.t = torch.zeros(S, device=dev, dtype=dt)
        self.old_dirs = torch.zeros(S, history_size, n, device=dev, dtype=dt)
        self.old_stps = torch.zeros(S, history_size, n, device=dev, dtype=dt)
        self.ro = torch.zeros(S, history_size, device=dev, dtype=dt)
        self.head = torch.zeros(S, dtype=torch.long, device=dev)       # oldest stored pair
        self.hist = torch.zeros(S, dtype=torch.long, device=dev)       # number of stored pairs
        self.H_diag = torch.ones(S, device=dev, dtype=dt)
        self.prev_g = torch.zeros(S, n, device=dev, dtype=dt)
        self.prev_loss = torch.zeros(S, device=dev, dtype=dt)
        self._rows = torch.arange(S, device=dev)
        self._loss = torch.zeros(S, device=dev, dtype=dt)              # closure values / gradients / live flags of the running step
        self._g = torch.zeros(S, n, device=dev, dtype=dt)
        self._act = torch.zeros(S, dtype=torch.bool, device=dev)
        # HIP-graph replay of the inner iteration: GPU variables + the direction kernel (no host sync inside) only
        self.use_graph = bool(use_graph and params.is_cuda and self.engine is not None)
        # the whole optimiser side of an iteration as two kernels (symode_lbfgs_update / _accept), a wave per problem
        self.data_term = None          # (closure of the bare data term, w_x, w_reg): objective = w_x * data + w_reg * |P|_1
        self.merged = False            # accept + next update as one launch (set below)
        self.fused = bool(self.engine is not None and hasattr(self.engine, 'lbfgs_update') and dt == torch.float32
                          and params.is_contiguous() and os.environ.get('SYMODE_LBFGS_FUSED', '1') != '0')
        self.merged = bool(self.fused and hasattr(self.engine, 'lbfgs_accept_update') and os.environ.get('SYMODE_LBFGS_MERGED', '1') != '0')
        self._graph, self._graph_closure, self._warm = None, None, 0

    def reset(self, which):
        """Fresh optimiser for the selected problems (the reference re-creates LBFGS after thresholding).  In place:
        a captured iteration graph holds these buffers."""
        self.n_iter.masked_fill_(which, 0)
        self.hist.masked_fill_(which, 0)
        self.head.masked_fill_(which, 0)
        self.H_diag.masked_fill_(which, 1.0)

    def _direction(self, g):
        if self.engine is not None:
            return self.engine.lbfgs_direction(g, self.old_dirs, self.old_stps, self.ro, self.head, self.hist, self.H_diag)
        S = g.shape[0]
        num_old = int(self.hist.max().item())
        q = -g
        al = torch.zeros(S, max(num_old, 1), device=g.device, dtype=g.dtype)
        for k in range(num_old - 1, -1, -1):                               # logical slot k of every problem
            live = (k < self.hist).to(g.dtype)
            slot = (self.head + k) % self.H
            a = (self.old_stps[self._rows, slot] * q).sum(1) * self.ro[self._rows, slot] * live
            al[:, k] = a
            q = q - a[:, None] * self.old_dirs[self._rows, slot]
        r = q * self.H_diag[:, None]
        for k in range(num_old):
            live = (k < self.hist).to(g.dtype)
            slot = (self.head + k) % self.H
            be = (self.old_dirs[self._rows, slot] * r).sum(1) * self.ro[self._rows, slot] * live
            r = r + self.old_stps[self._rows, slot] * ((al[:, k] - be) * live)[:, None]
        return r

    def _iteration(self, closure, evaluate):
        """One inner iteration of every problem on the persistent buffers (self.P, _loss, _g, _act and the optimiser
        state): everything is updated in place, so the sequence of launches can be captured once and replayed."""
        P, g, loss, act = self.P, self._g, self._loss, self._act
        if self.fused:                                         # two launches around the closure instead of ~60
            self.engine.lbfgs_update(P, g, loss, act, self, self.lr, self.tol_c)
            if evaluate and self.data_term is not None:        # bare data term; scale + L1 term added by the accept kernel
                raw, w_x, w_reg = self.data_term
                nl, ng = raw(P)
                self.engine.lbfgs_accept(nl, ng, loss, g, act, self, self.tol_g, self.tol_c, l1=(P, w_x, w_reg))
            elif evaluate:
                nl, ng = closure(P)
                self.engine.lbfgs_accept(nl, ng, loss, g, act, self, self.tol_g, self.tol_c)
            return
        self.n_iter.add_(act.long())
        first = act & (self.n_iter == 1)
        upd = act & ~first
        # ---- curvature memory (torch: "do lbfgs update (update memory)") -------------------
        self.hist.masked_fill_(first, 0)
        self.head.masked_fill_(first, 0)
        self.H_diag.masked_fill_(first, 1.0)
        y = g - self.prev_g
        s = self.d * self.t[:, None]
        ys = (y * s).sum(1)
        mem = upd & (ys > 1e-10)
        full = mem & (self.hist == self.H)
        pos = torch.where(full, self.head, (self.head + self.hist) % self.H)      # overwrite the oldest when full
        m3 = mem[:, None]
        self.old_dirs[self._rows, pos] = torch.where(m3, y, self.old_dirs[self._rows, pos])
        self.old_stps[self._rows, pos] = torch.where(m3, s, self.old_stps[self._rows, pos])
        safe_ys = torch.where(mem, ys, torch.ones_like(ys))
        self.ro[self._rows, pos] = torch.where(mem, 1.0 / safe_ys, self.ro[self._rows, pos])
        self.head.copy_(torch.where(full, (self.head + 1) % self.H, self.head))
        self.hist.copy_(torch.where(mem & ~full, self.hist + 1, self.hist))
        yy = (y * y).sum(1)
        self.H_diag.copy_(torch.where(mem, ys / torch.where(mem, yy, torch.ones_like(yy)), self.H_diag))
        # ---- direction: empty history gives d = -g, as torch's first iteration -----------------
        d_new = self._direction(g)
        self.d.copy_(torch.where(act[:, None], d_new, self.d))
        self.prev_g.copy_(torch.where(act[:, None], g, self.prev_g))
        self.prev_loss.copy_(torch.where(act, loss, self.prev_loss))
        # ---- step length ---------------------------------------------------------------------
        t_first = torch.clamp(1.0 / g.abs().sum(1), max=1.0) * self.lr
        t_new = torch.where(self.n_iter == 1, t_first, torch.full_like(t_first, self.lr))
        self.t.copy_(torch.where(act, t_new, self.t))
        gtd = (g * self.d).sum(1)
        live = act & ~(gtd > -self.tol_c)                                  # directional derivative below tolerance
        P.add_(torch.where(live, self.t, torch.zeros_like(self.t))[:, None] * self.d)
        if evaluate:                                                       # (no re-evaluation on the last iteration)
            nl, ng = closure(P)
            loss.copy_(torch.where(live, nl, loss))
            g.copy_(torch.where(live[:, None], ng, g))
            stop = (g.abs().amax(1) <= self.tol_g) | ((self.d * self.t[:, None]).abs().amax(1) <= self.tol_c) \
                | ((loss - self.prev_loss).abs() < self.tol_c)
            live = live & ~stop
        act.copy_(live)

    def _evaluate_accept_update(self, closure):
        """closure at the moved parameters, then ONE optimiser launch: finish the running iteration (accept + stopping
        tests) and start the next one (update + move) for the problems still active."""
        if self.data_term is not None:
            raw, w_x, w_reg = self.data_term
            nl, ng = raw(self.P)
            l1 = (w_x, w_reg)
        else:
            nl, ng = closure(self.P)
            l1 = None
        self.engine.lbfgs_accept_update(nl, ng, self.P, self._g, self._loss, self._act, self, self.lr, self.tol_g, self.tol_c, l1=l1)

    def _replayed(self, body, closure):
        """``body(closure)`` is a handful of launches on persistent buffers: after a few eager runs (lazy initialisations
        done) it is captured ONCE in a HIP graph and replayed -- same kernels, same buffers."""
        if self._graph is None or self._graph_closure is not closure:
            if self._warm < 3 or (self._graph is not None and self._graph_closure is not closure):
                self._warm += 1
                return body(closure)
            try:
                torch.cuda.synchronize(self.P.device)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    body(closure)
                self._graph, self._graph_closure = graph, closure
            except Exception:                                              # pragma: no cover - depends on the runtime
                self.use_graph = False
                return body(closure)
        self._graph.replay()

    def _iteration_replayed(self, closure):
        """The iteration is ~60 small launches around one fused closure kernel: launch-bound.  After a few eager runs
        (lazy initialisations done) it is captured ONCE in a HIP graph and replayed -- same kernels, same buffers."""
        if self._graph is None or self._graph_closure is not closure:
            if self._warm < 3 or (self._graph is not None and self._graph_closure is not closure):
                self._warm += 1
                return self._iteration(closure, True)
            try:
                torch.cuda.synchronize(self.P.device)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    self._iteration(closure, True)
                self._graph, self._graph_closure = graph, closure
            except Exception:                                              # pragma: no cover - depends on the runtime
                self.use_graph = False
                return self._iteration(closure, True)
        self._graph.replay()

    @torch.no_grad()
    def step(self, closure, frozen=None):
        """closure(P) -> (loss (S,), grad (S, n)).  ``frozen`` (S,) bool: problems that must not move."""
        loss, g = closure(self.P)
        self._loss.copy_(loss)
        self._g.copy_(g)
        # optimality test in torch's sense (`if flat_grad.abs().max() <= tolerance_grad: return`): a NaN gradient does NOT
        # stop the step -- the parameters go to NaN and the trainer's NaN guard ends the run (train.py:697)
        act = ~(g.abs().amax(dim=1) <= self.tol_g)
        if frozen is not None:
            act = act & ~frozen
        self._act.copy_(act)
        if self.merged:
            # iteration 1's update alone, then (closure, accept + next update) max_iter - 1 times: the last update is the
            # one torch performs without re-evaluating (n_iter == max_iter)
            self.engine.lbfgs_update(self.P, self._g, self._loss, self._act, self, self.lr, self.tol_c)
            for it in range(1, self.max_iter):
                if self.use_graph:
                    self._replayed(self._evaluate_accept_update, closure)
                else:
                    self._evaluate_accept_update(closure)
                if it % 5 == 0 and not bool(self._act.any()):               # the only host sync, every 5th iteration
                    break
            return self._loss.clone()
        for it in range(1, self.max_iter + 1):
            if it == self.max_iter:
                self._iteration(closure, False)
                break
            if self.use_graph:
                self._iteration_replayed(closure)
            else:
                self._iteration(closure, True)
            if it % 5 == 0 and not bool(self._act.any()):                   # the only host sync, every 5th iteration
                break
        return self._loss.clone()


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

    def _data_term(self, P, alias=True):
        """loss, gradient of the bare MSE (+ regulariser the closure carries) w.r.t. the flat parameters.  ``alias``:
        the results may be views of the closure's output buffer (they are consumed before the next evaluation)."""
        c = self.c
        a, b = self._split(P)
        loss, ga, gb = c.evaluate(a.contiguous(), b, mask=self.mask, **({'alias': True} if alias and self._can_alias else {}))
        if gb is None and c.Q is not None:                                  # constrain_constant: const is a parameter the
            gb = torch.zeros(c.S, c.d, 1, device=P.device, dtype=P.dtype)   # model does not read (sindy.py:60, 173-175)
        g = ga.reshape(c.S, -1) if gb is None else torch.cat([ga, gb.reshape(c.S, -1)], dim=1)
        return loss, g

    def _closure(self, P):
        loss, g = self._data_term(P, alias=False)
        if self.w_x != 1.0:
            loss, g = self.w_x * loss, self.w_x * g
        if self.reg_type == "l1" and self.w_reg != 0.0:                     # over the raw parameters (train.py:681)
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

    def _native_ok(self, P0):
        """The device-resident trainer (device_lbfgs.DeviceTrainer: optimiser AND epoch logic as kernels of the library, no
        stock torch op between the epochs) serves every GPU fit; the tensor-op form below is its restatement for CPU /
        gloo test doubles and ``SYMODE_LBFGS_FUSED=0``."""
        c = self.c
        eng = getattr(c, 'engine', None)
        return bool(P0.is_cuda and eng is not None and hasattr(getattr(eng, 'lib', None), 'symode_trainer_run')
                    and os.environ.get('SYMODE_LBFGS_FUSED', '1') == '1' and os.environ.get('SYMODE_LBFGS_MERGED', '1') == '1'
                    and P0.shape[1] <= 256 and c.d * c.p <= 256 and getattr(c, 'n_chunks', 1) == 1)

    def _fit_native(self, P0, num_epochs, mask0, on_epoch):
        from .device_lbfgs import DeviceTrainer
        c = self.c
        tr = DeviceTrainer(c.x, c.dx, c.order, c.flags, Q=c.Q, use_kron_product=c.use_kron, allow_constant=c.allow_constant,
                           reversed_sym=c.sym, lr=self.lr, threshold=self.threshold, st_freq=self.st_freq, w_x=self.w_x,
                           w_reg=self.w_reg if self.reg_type == 'l1' else 0.0, l1=True, tol=self.tol, inv_count=c.inv_count,
                           engine=c.engine, group=(c.group or dist.group.WORLD) if c.distributed else None,
                           detail=on_epoch is not None and c.S <= 64)
        self.trainer = tr
        cb = None
        if on_epoch is not None:
            dev = P0.device

            def cb(epoch, rec):                              # the tensor-op form's callback signature
                done = torch.from_numpy((rec['code'] == 3) | (rec['code'] == 4) | (rec['code'] == -1))
                return on_epoch(epoch, tr.field('params'), tr.field('mask').view(c.S, c.d, c.p), done.to(dev))
        out = tr.fit(P0, num_epochs, mask0=mask0, on_epoch=cb)
        self.mask = out['mask'].to(P0.device)
        return {k: v.to(P0.device) for k, v in out.items()}

    @torch.no_grad()
    def fit(self, P0, num_epochs, mask0=None, on_epoch=None):
        """P0 (S, n): initial flat parameters per seed ([Xi] or [beta | const]).
        ``on_epoch(epoch, P, mask, done)`` (optional) is called after the epoch's logic with the live device tensors (a
        caller that reads them synchronises; returning True ends the fit) -- per-epoch logs / interval checkpoints.
        Returns dict(Xi, mask, params, epochs (S,), finished (S,), nan (S,))."""
        c = self.c
        if self._native_ok(P0):
            return self._fit_native(P0, num_epochs, mask0, on_epoch)
        P = P0.clone().contiguous()
        S = P.shape[0]
        self.mask = torch.ones(S, c.d, c.p, device=P.device) if mask0 is None else mask0.clone()
        graph_ok = P.is_cuda and not getattr(c, 'distributed', False) and os.environ.get('SYMODE_SWEEP_GRAPH', '1') != '0'
        opt = BatchedLBFGS(P, self.lr, engine=getattr(c, 'engine', None) if P.is_cuda else None, use_graph=graph_ok)
        closure = self._closure                               # ONE bound-method object: the captured graph is tied to it
        self._can_alias = 'alias' in getattr(getattr(c.evaluate, '__code__', None), 'co_varnames', ())
        if opt.fused:
            opt.data_term = (self._data_term, self.w_x, self.w_reg if self.reg_type == "l1" else 0.0)
        prev, pprev = P.clone(), P.clone()
        n_iters = torch.zeros(S, dtype=torch.long, device=P.device)
        done = torch.zeros(S, dtype=torch.bool, device=P.device)
        nan = torch.zeros(S, dtype=torch.bool, device=P.device)
        epochs = torch.zeros(S, dtype=torch.long, device=P.device)
        near = torch.zeros(S, dtype=torch.long, device=P.device)         # near-threshold coefficients met per seed
        for epoch in range(num_epochs):
            if epoch % 4 == 0 and bool(done.all()):                        # host sync every 4th epoch only
                break
            live = ~done
            n_iters = n_iters + live.long()
            epochs = torch.where(live, torch.full_like(epochs, epoch + 1), epochs)
            opt.step(closure, frozen=done)
            bad = live & torch.isnan(P).any(dim=1)                         # train.py:697-699
            nan |= bad
            done |= bad
            live = ~done
            upd = self._norms(P, prev)
            conv = live & (upd < self.tol)
            final = conv & (self._norms(P, pprev) < self.tol)             # train.py:709-714
            done |= final
            thr_conv = conv & ~final
            thr_freq = (live & ~conv & (n_iters % self.st_freq == 0)) if self.st_freq > 0 else torch.zeros_like(done)
            ev = thr_conv | thr_freq
            Xi = self._xi(P)
            new_mask = torch.logical_and(Xi.abs() > self.threshold, self.mask > 0).float()
            close = ((Xi.abs() - self.threshold).abs() < NEAR_THRESHOLD_BAND) & (self.mask > 0)
            near = near + torch.where(ev, close.sum(dim=(1, 2)), torch.zeros_like(near))
            self.mask.copy_(torch.where(ev[:, None, None], new_mask, self.mask))   # strict >, monotone (sindy.py:194); in place:
            #                                                                       the captured iteration reads this buffer
            opt.reset(ev)
            n_iters = torch.where(ev, torch.zeros_like(n_iters), n_iters)
            pprev = torch.where(thr_conv[:, None], P, pprev)               # only on convergence-triggered events (:718)
            prev = torch.where((live & ~final)[:, None], P, prev)
            if on_epoch is not None and on_epoch(epoch, P, self.mask, done):
                break
        return {"Xi": self._xi(P), "mask": self.mask, "params": P, "epochs": epochs, "finished": done & ~nan, "nan": nan,
                "near_threshold": near}
