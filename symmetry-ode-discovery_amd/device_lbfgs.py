"""``train_SIGED_lbfgs`` (non-latent branch, reference train.py:617-766) for S problems with the optimiser AND the
per-epoch logic resident on the GPU: the host enqueues whole epochs (``symode_trainer_run``: closure, optimiser launch,
... , epoch-end launch -- 2 * max_iter + 1 launches, include/symode.h) and reads one small record per epoch from
pinned memory.  No stock torch GPU op runs between the first and the last epoch, so a one-seed process pays no
code-object loading beyond this library's own kernels.

The arithmetic is torch.optim.LBFGS's (no line search) statement by statement, as in ``sweep.BatchedLBFGS`` (the
tensor-op form of the same iteration, kept for CPU / gloo runs and as the test double of these kernels).
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch
import torch.distributed as dist

from .engine import TRAINER_FIELDS, SymodeError, TrainerDesc, get_engine

NEAR_THRESHOLD_BAND = 1e-4            # sindy.NEAR_THRESHOLD_BAND; repeated here to keep imports light

EVENT_NONE, EVENT_THRESHOLD_CONVERGED, EVENT_THRESHOLD_PERIOD, EVENT_FINAL, EVENT_NAN, EVENT_IDLE = 0, 1, 2, 3, 4, -1
_FIELD_DTYPES = {"act": torch.uint8, "n_iter": torch.int64, "head": torch.int64, "count": torch.int64, "n_iters": torch.int32,
                 "done": torch.uint8, "nan": torch.uint8, "finished": torch.uint8, "epochs": torch.int32, "near": torch.int32}


def effective_Q(Q, d, p, use_kron_product):
    """Q (d p, r) with its rows permuted into Xi's (d, p) row-major order: sindy.py:171-173 reads ``Q @ beta`` as
    ``view(d, -1)`` on the Kronecker branch and as ``view(-1, d).T`` otherwise."""
    Q = np.ascontiguousarray(Q.detach().cpu().numpy() if torch.is_tensor(Q) else Q, dtype=np.float32)
    if use_kron_product:
        return Q
    rows = (np.arange(p)[None, :] * d + np.arange(d)[:, None]).reshape(-1)      # Xi[i, t] = flat[t * d + i]
    return np.ascontiguousarray(Q[rows])


class DeviceTrainer:
    LOG_RING = 8                      # epochs of records kept; the host runs at most two epochs ahead of its reading

    def __init__(self, x, dx, poly_order, flags=0, Q=None, use_kron_product=True, allow_constant=True, reversed_sym=None,
                 lr=1.0, threshold=0.1, st_freq=0, w_x=1.0, w_reg=0.0, l1=True, tol=1e-3, max_iter=20, history=100,
                 tol_grad=1e-7, tol_change=1e-9, inv_count=None, engine=None, detail=None, group=None):
        """x, dx (S, N_local, d) device tensors; ``reversed_sym = (gx (S, n_g, N, d), jgx (S, n_g, N, d, d), weight)`` as
        batched.BatchedClosure; ``group``: point shards, [loss | grad] summed over the ranks between closure and update;
        ``detail``: keep coefficients and mask of every epoch in the record (default: for S <= 64)."""
        self.engine = engine or get_engine()
        if not (x.is_cuda and x.dim() == 3 and x.shape == dx.shape and x.dtype == torch.float32):
            raise SymodeError("DeviceTrainer expects x, dx as (S, N, d) fp32 device tensors; there is no CPU fallback")
        lib = self.engine.lib
        self.x, self.dx = x.contiguous(), dx.contiguous()
        self.S, self.n_points, self.d = x.shape
        self.order, self.flags = int(poly_order), int(flags)
        self.p = self.engine.lib_size(self.d, self.order, self.flags)
        self.dp = self.d * self.p
        self.group = group
        world = dist.get_world_size(group) if group is not None else 1
        dev = x.device
        self.q_eff = None
        if Q is not None:
            q = effective_Q(Q, self.d, self.p, use_kron_product)
            if q.shape[0] != self.dp:
                raise SymodeError(f"Q has {q.shape[0]} rows, expected d*p = {self.dp}")
            self.r = q.shape[1]
            self.q_eff = torch.from_numpy(q).to(dev)
            self.n = self.r + self.d
        else:
            self.r, self.n = 0, self.dp
        if self.n > 256 or self.dp > 256 or history > 128:
            raise SymodeError("DeviceTrainer handles at most 256 parameters / coefficients and 128 curvature pairs")
        self.sym = None
        if reversed_sym is not None:
            gx, jgx, weight = reversed_sym
            if gx.dim() != 4 or gx.shape[0] != self.S or tuple(gx.shape[2:]) != tuple(x.shape[1:]) or tuple(jgx.shape) != tuple(gx.shape) + (self.d,):
                raise SymodeError("reversed_sym operands do not match x")
            self.sym = (gx.contiguous(), jgx.contiguous(), float(weight))
        self.detail = (self.S <= 64) if detail is None else bool(detail)
        self.distributed = group is not None
        # --- state block: ONE allocation, laid out by the library
        offs = (ctypes.c_size_t * len(TRAINER_FIELDS))()
        nbytes = lib.symode_trainer_layout(self.S, self.n, self.dp, history, 1 if Q is not None else 0, offs)
        if nbytes == 0:
            raise SymodeError("symode_trainer_layout refused the problem sizes")
        self.state = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self._off = dict(zip(TRAINER_FIELDS, [int(o) for o in offs]))
        self._shape = {"params": (self.S, self.n), "xi": (self.S, self.d, self.p), "mask": (self.S, self.d, self.p),
                       "cl_loss": (self.S, 2), "cl_grad": (self.S, self.dp), "g": (self.S, self.n), "loss": (self.S,)}
        ws_bytes = lib.symode_workspace_bytes(self.d, self.order, self.flags, self.S, self.n_points)
        self.ws = self.engine.new_workspace(dev, ws_bytes)
        # --- per-epoch records: pinned host memory the kernels write directly (sharded runs: device memory, copied)
        R = self.LOG_RING
        mk = (lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)) if self.distributed else \
             (lambda *s: torch.zeros(*s, dtype=torch.float32).pin_memory())
        self.log = mk(R, self.S, 8)
        self.log_test = mk(R, self.S, 2)
        self.log_xi = mk(R, self.S, self.dp) if self.detail else None
        self.log_mask = mk(R, self.S, self.dp) if self.detail else None
        self.log_params = mk(R, self.S, self.n) if self.detail else None
        n_global = self.n_points * world
        if group is not None and inv_count is None:          # shards need not be equal: the count is summed once, at set-up
            cnt = torch.tensor([float(self.n_points)], dtype=torch.float64, device=dev)
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
            n_global = int(cnt.item())
        T = TrainerDesc()
        T.x, T.dx = self.x.data_ptr(), self.dx.data_ptr()
        if self.sym is not None:
            T.gx, T.jgx, T.n_g, T.w_sym = self.sym[0].data_ptr(), self.sym[1].data_ptr(), self.sym[0].shape[1], self.sym[2]
        else:
            T.gx, T.jgx, T.n_g, T.w_sym = None, None, 0, 0.0
        T.n_problems, T.n_points, T.d, T.order, T.flags = self.S, self.n_points, self.d, self.order, self.flags
        T.inv_count = float(inv_count) if inv_count is not None else 1.0 / (n_global * self.d)
        T.workspace, T.workspace_bytes = self.ws.data_ptr(), self.ws.numel() * 8
        T.q_eff = self.q_eff.data_ptr() if self.q_eff is not None else None
        T.r, T.allow_constant, T.n_params = self.r, int(bool(allow_constant)), self.n
        T.w_x, T.w_reg, T.l1 = float(w_x), float(w_reg), int(bool(l1))
        T.lr, T.tol_grad, T.tol_change, T.max_iter, T.history = float(lr), float(tol_grad), float(tol_change), int(max_iter), int(history)
        T.threshold, T.tol_update, T.near_band, T.st_freq = float(threshold), float(tol), NEAR_THRESHOLD_BAND, int(st_freq)
        T.state, T.state_bytes = self.state.data_ptr(), nbytes
        T.log, T.log_test = self.log.data_ptr(), self.log_test.data_ptr()
        T.log_xi = self.log_xi.data_ptr() if self.detail else None
        T.log_mask = self.log_mask.data_ptr() if self.detail else None
        T.log_params = self.log_params.data_ptr() if self.detail else None
        T.log_epochs = R
        self.T = T
        self._Tp = ctypes.byref(T)
        self.max_iter = int(max_iter)
        self.threshold = float(threshold)

    # -- plumbing -----------------------------------------------------------------------------------------------------
    def _check(self, rc, what):
        if rc != 0:
            # a launch that failed half-way may have left tickets of the one-launch reductions behind: start them afresh
            self.engine.lib.symode_workspace_init(self.ws.data_ptr(), self.ws.numel() * 8, self._st())
            self.engine._check(rc, what)

    def _st(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.x.device).cuda_stream)

    def field(self, name):
        """A device view of one array of the state block."""
        dt = _FIELD_DTYPES.get(name, torch.float32)
        shape = self._shape.get(name, (self.S,))
        n = int(np.prod(shape)) * torch.empty((), dtype=dt).element_size()
        off = self._off[name]
        return self.state[off:off + n].view(dt).view(*shape)

    def _record(self, slot):
        log = self.log[slot].cpu().numpy() if self.distributed else self.log[slot].numpy().copy()
        test = self.log_test[slot].cpu().numpy() if self.distributed else self.log_test[slot].numpy().copy()
        rec = {"code": log[:, 0].astype(np.int64), "mse": log[:, 1], "sym": log[:, 2], "l1": log[:, 3], "update_norm": log[:, 4],
               "update_norm_2": log[:, 5], "near": log[:, 6].astype(np.int64), "epoch": log[:, 7].astype(np.int64),
               "test": test[:, 0] if self.sym is not None else test.reshape(-1)[:self.S], "xi": None, "mask": None, "params": None}
        if self.detail:
            get = (lambda a: a[slot].cpu().numpy()) if self.distributed else (lambda a: a[slot].numpy().copy())
            rec["xi"] = get(self.log_xi).reshape(self.S, self.d, self.p)
            rec["mask"] = get(self.log_mask).reshape(self.S, self.d, self.p)
            rec["params"] = get(self.log_params)
        return rec

    def _epoch_sharded(self, epoch, test_eval):
        """One epoch with the ranks' partial [loss | grad] summed between closure and update (RCCL / gloo)."""
        lib, st = self.engine.lib, self._st()
        width = (2 if self.sym is not None else 1) * self.S
        cl = self.state[self._off["cl_loss"]:self._off["cl_grad"] + self.S * self.dp * 4].view(torch.float32)
        # (cl_loss is (S, 2) floats with cl_grad right behind it; the plain closure fills the first S floats only)
        for it in range(self.max_iter):
            self._check(lib.symode_trainer_closure(self._Tp, None, None, st), "symode_trainer_closure")
            if width < 2 * self.S:
                dist.all_reduce(cl[:width], group=self.group)
                dist.all_reduce(cl[2 * self.S:], group=self.group)
            else:
                dist.all_reduce(cl, group=self.group)
            self._check(lib.symode_trainer_update(self._Tp, 2 if it == 0 else 1, st), "symode_trainer_update")
        self._check(lib.symode_trainer_epoch_end(self._Tp, epoch, st), "symode_trainer_epoch_end")
        if test_eval:
            slot = epoch % self.LOG_RING
            tg = self.field("test_grad")
            self._check(lib.symode_trainer_closure(self._Tp, ctypes.c_void_p(self.log_test[slot].data_ptr()),
                                                   ctypes.c_void_p(tg.data_ptr()), st), "symode_trainer_closure")
            dist.all_reduce(self.log_test[slot], group=self.group)

    # -- the fit --------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def fit(self, P0, num_epochs, mask0=None, on_epoch=None, test_eval=False):
        """P0 (S, n) start parameters ([Xi] or [beta | const]), host or device.  ``on_epoch(epoch, record)`` is called once
        per epoch, in order, with the epoch's record (numpy arrays over the problems: code, mse, sym, l1, update_norm,
        near, test, and with ``detail`` xi / mask after the epoch's events); returning True ends the fit.
        Returns dict(Xi, mask, params, epochs, finished, nan, near_threshold) of host tensors."""
        lib, dev = self.engine.lib, self.x.device
        P0 = P0.detach().to(torch.float32).contiguous()
        if tuple(P0.shape) != (self.S, self.n):
            raise SymodeError(f"P0 must be ({self.S}, {self.n}), got {tuple(P0.shape)}")
        m0 = None
        if mask0 is not None:
            m0 = mask0.detach().to(torch.float32).contiguous()
            if m0.numel() != self.S * self.dp:
                raise SymodeError("mask0 does not match the coefficient shape")
        stream = torch.cuda.current_stream(dev)
        self._check(lib.symode_trainer_init(self._Tp, ctypes.c_void_p(P0.data_ptr()),
                                            None if m0 is None else ctypes.c_void_p(m0.data_ptr()), self._st()),
                    "symode_trainer_init")
        if not P0.is_cuda or (m0 is not None and not m0.is_cuda):
            stream.synchronize()                            # pageable host sources: the copies must finish before they go away
        done = np.zeros(self.S, dtype=bool)

        def enqueue(e):
            if self.distributed:
                self._epoch_sharded(e, test_eval)
            else:
                self._check(lib.symode_trainer_run(self._Tp, e, 1, 1 if test_eval else 0, self._st()), "symode_trainer_run")
            ev = torch.cuda.Event()
            ev.record(stream)
            return ev

        pending = [enqueue(0)] if num_epochs > 0 else []
        for e in range(num_epochs):
            if e + 1 < num_epochs:
                pending.append(enqueue(e + 1))              # the next epoch is in flight while this one's record is read
            pending.pop(0).synchronize()
            rec = self._record(e % self.LOG_RING)
            stop = bool(on_epoch(e, rec)) if on_epoch is not None else False
            done |= (rec["code"] == EVENT_FINAL) | (rec["code"] == EVENT_NAN) | (rec["code"] == EVENT_IDLE)
            if stop or done.all():
                break
        stream.synchronize()
        out = {k: self.field(k).cpu() for k in ("params", "xi", "mask", "epochs", "finished", "nan", "near")}
        return {"Xi": out["xi"].reshape(self.S, self.d, self.p), "mask": out["mask"].reshape(self.S, self.d, self.p),
                "params": out["params"], "epochs": out["epochs"].long(), "finished": out["finished"].bool(),
                "nan": out["nan"].bool(), "near_threshold": out["near"].long()}
