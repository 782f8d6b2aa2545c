"""Batched / sharded evaluation of the SINDy closure over independent (trajectory, seed) problems.

One process per GPU.  Every rank keeps, resident in HBM, its shard of the points of each of S
problems: x, dx (S, N_local, d).  One call evaluates loss and gradient of all S problems with
the fused Theta + residual + gradient kernel and, when a process group is given, sums the
per-rank partials with ONE all-reduce per chunk of problems (RCCL over xGMI when the backend
is "nccl"); chunks are pipelined so the collective of chunk c overlaps the kernel of chunk
c+1.  The message is S*(1 + d*p) floats -- latency-bound, hence few, fused collectives.

The reference runs one process per seed and has no distributed code (run_scripts/*.sh loop
over seeds); this is the new design SURVEY section 8(e) calls point-sharding.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .engine import get_engine, library_flags


def allreduce_sums(sums, params, group):
    """Point-sharded scalar sums and their parameter gradients through ONE collective.

    ``sums``: scalar tensors, each a sum over THIS rank's points (with or without an autograd graph); ``params``: the
    leaf tensors they are differentiated by.  Every sum is differentiated locally, the values and gradients of all of
    them travel in one packed fp64 buffer (n * (1 + |params|) numbers -- a few hundred bytes) through one all-reduce,
    and what comes back is, per sum, a scalar tensor whose VALUE is the global sum and whose first derivative with
    respect to ``params`` is the global gradient (value + g . (p - p.detach())).  Whatever the caller builds from them
    -- a mean, a ratio of two batch means (the relative regularisers, model_utils.py:62, 118-121: numerator and
    denominator must be summed over the ranks SEPARATELY before the division) -- autograd differentiates on that
    handful of scalars: the quotient rule is applied after the collective, on global numbers, on every rank alike."""
    params = list(params)
    sizes = [p.numel() for p in params]
    P = sum(sizes)
    dev = sums[0].device
    buf = torch.zeros(len(sums), 1 + P, dtype=torch.float64, device=dev)
    for i, s in enumerate(sums):
        buf[i, 0] = s.detach()
        if s.requires_grad:
            gs = torch.autograd.grad(s, params, retain_graph=True, allow_unused=True)
            o = 1
            for g, n in zip(gs, sizes):
                if g is not None:
                    buf[i, o:o + n] = g.reshape(-1)
                o += n
    if group is not None:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    out = []
    for i in range(len(sums)):
        v = buf[i, 0].to(sums[i].dtype)
        o = 1
        for p, n in zip(params, sizes):
            g = buf[i, o:o + n].to(p.dtype).view_as(p)
            v = v + (g * (p - p.detach())).sum()
            o += n
        out.append(v)
    return out


class BatchedClosure:
    def __init__(self, x, dx, poly_order, include_sine=False, include_exp=False, Q=None, use_kron_product=True,
                 allow_constant=True, group=None, world_size=None, n_chunks=1, engine=None, reversed_sym=None, fuse_sym=True):
        """``reversed_sym = (gx (S, n_g, N_local, d), jgx (S, n_g, N_local, d, d), weight)`` adds  weight * the reversed
        symmetry regulariser (model_utils.py:126-170 on precomputed (g(x), J_g(x))) to every problem's loss and gradient:
        by default residual and regulariser run as ONE launch per chunk (symode_loss_grad_reversed: Theta(x) shared, x read
        once); ``fuse_sym=False`` keeps them as two launches summed into the same packed buffer before the collective."""
        assert x.dim() == 3 and x.shape == dx.shape, "x, dx must be (S, N_local, d)"
        self.engine = engine or get_engine()
        self.x, self.dx = x.contiguous(), dx.contiguous()
        self.S, self.n_local, self.d = x.shape
        self.order = poly_order
        self.flags = library_flags(include_sine, include_exp)
        self.p = self.engine.lib_size(self.d, poly_order, self.flags)
        self.Q = Q
        self.use_kron, self.allow_constant = use_kron_product, allow_constant
        self.group = group
        self.distributed = group is not None or (world_size or 1) > 1
        self.world = world_size if world_size is not None else (dist.get_world_size(group) if self.distributed else 1)
        self.n_global = self.n_local * self.world
        if group is not None:                                # shards need not be equal: the count is summed once, at set-up
            cnt = torch.tensor([float(self.n_local)], dtype=torch.float64, device=x.device)
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
            self.n_global = int(cnt.item())
        self.inv_count = 1.0 / (self.n_global * self.d)
        self.n_chunks = max(1, min(n_chunks, self.S))
        nacc = 1 + self.d * self.p
        # packed [loss | grad] per chunk so that one collective carries both
        bounds = torch.linspace(0, self.S, self.n_chunks + 1).long().tolist()
        self.chunks = [(a, b) for a, b in zip(bounds[:-1], bounds[1:]) if b > a]
        self.buffers = [torch.empty((b - a) * nacc, dtype=torch.float32, device=x.device) for a, b in self.chunks]
        # private scratch: a captured HIP graph (sweep.BatchedLBFGS) replays these launches from its own stream
        self._ws_kw = {}
        if x.is_cuda and hasattr(self.engine, 'new_workspace'):
            self._ws_kw = {'ws': self.engine.new_workspace(x.device, self.engine.lib.symode_workspace_bytes(
                self.d, poly_order, self.flags, max(b - a for a, b in self.chunks), self.n_local))}
        self.sym = None
        if reversed_sym is not None:
            gx, jgx, weight = reversed_sym
            assert gx.dim() == 4 and gx.shape[0] == self.S and gx.shape[2:] == x.shape[1:] and jgx.shape == gx.shape + (self.d,)
            self.sym = (gx.contiguous(), jgx.contiguous(), float(weight))
            self.sym_buffers = [torch.empty_like(b) for b in self.buffers]
        self.fuse_sym = fuse_sym

    # -- coefficient plumbing (batched get_Xi, sindy.py:169-176) ----------------------------
    def xi_from(self, beta, const=None):
        if self.Q is None:
            return beta                                         # beta IS Xi (S, d, p)
        flat = beta @ self.Q.T                                  # (S, d*p)
        Xi = flat.view(self.S, self.d, self.p) if self.use_kron else flat.view(self.S, self.p, self.d).transpose(1, 2)
        if self.allow_constant and const is not None:
            Xi = Xi.clone()
            Xi[:, :, 0:1] += const
        return Xi.contiguous()

    def grads_to(self, grad_xi):
        if self.Q is None:
            return grad_xi, None
        g = grad_xi if self.use_kron else grad_xi.transpose(1, 2)
        g_beta = g.reshape(self.S, -1) @ self.Q
        g_const = grad_xi[:, :, 0:1].clone() if self.allow_constant else None
        return g_beta, g_const

    # -- the hot path -------------------------------------------------------------------------
    def loss_grad_xi(self, Xi, mask=None, alias=False):
        """loss (S,), dloss/dXi (S, d, p) summed over all ranks' shards.  ``alias=True``: with a single chunk the results
        are returned as views of the packed buffer the kernel wrote (no copies; overwritten by the next evaluation)."""
        works = []
        fused = self.sym is not None and self.fuse_sym and hasattr(self.engine, 'loss_grad_reversed')
        for ci, ((a, b), buf) in enumerate(zip(self.chunks, self.buffers)):
            n = b - a
            loss = buf[:n]
            grad = buf[n:].view(n, self.d, self.p)
            if fused:
                # ONE launch per chunk: residual and regulariser share Theta(x), x is read once; the kernel leaves
                # (mse, regulariser) per problem and the gradient of mse + weight * regulariser
                gx, jgx, weight = self.sym
                sb = self.sym_buffers[ci]
                l2 = sb[:2 * n].view(n, 2)
                self.engine.loss_grad_reversed(self.x[a:b], self.dx[a:b], gx[a:b], jgx[a:b], Xi[a:b],
                                               None if mask is None else mask[a:b], self.order, self.flags, w_sym=weight,
                                               inv_count=self.inv_count, out=(l2, grad), **self._ws_kw)
                torch.add(l2[:, 0], l2[:, 1], alpha=weight, out=loss)
                if self.distributed:
                    works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                continue
            l, g = self.engine.loss_grad(self.x[a:b], self.dx[a:b], Xi[a:b], None if mask is None else mask[a:b],
                                         self.order, self.flags, inv_count=self.inv_count, out=(loss, grad), **self._ws_kw)
            if l.data_ptr() != loss.data_ptr():          # an engine that does not write in place
                loss.copy_(l)
                grad.copy_(g)
            if self.sym is not None:
                gx, jgx, weight = self.sym
                sb = self.sym_buffers[ci]
                sl, sg = sb[:n], sb[n:].view(n, self.d, self.p)
                l2, g2 = self.engine.symreg_reversed(self.x[a:b], gx[a:b], jgx[a:b], Xi[a:b], None if mask is None else mask[a:b],
                                                     self.order, self.flags, out=(sl, sg), inv_count=self.inv_count, **self._ws_kw)
                if l2.data_ptr() != sl.data_ptr():
                    sl.copy_(l2)
                    sg.copy_(g2)
                buf.add_(sb, alpha=weight)
            if self.distributed:
                works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in works:
            w.wait()
        if alias and len(self.chunks) == 1:
            n = self.S
            return self.buffers[0][:n], self.buffers[0][n:].view(n, self.d, self.p)
        loss = torch.cat([buf[:b - a] for (a, b), buf in zip(self.chunks, self.buffers)])
        grad = torch.cat([buf[b - a:].view(b - a, self.d, self.p) for (a, b), buf in zip(self.chunks, self.buffers)])
        return loss, grad

    def evaluate(self, beta, const=None, mask=None, alias=False):
        """Closure of all S problems: (loss (S,), d/dbeta [or d/dXi when unconstrained], d/dconst)."""
        Xi = self.xi_from(beta, const)
        loss, grad = self.loss_grad_xi(Xi, mask, alias=alias)
        g_beta, g_const = self.grads_to(grad)
        return loss, g_beta, g_const

    def aug_gram(self):
        """fp64 (S, p+d, p+d) augmented Gram matrices, all-reduced over the point shards."""
        G = self.engine.aug_gram(self.x, self.dx, self.order, self.flags)
        if self.distributed:
            dist.all_reduce(G, op=dist.ReduceOp.SUM, group=self.group)
        return G
