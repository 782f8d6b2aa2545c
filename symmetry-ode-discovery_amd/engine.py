"""ctypes binding of libsymode_hip.so (C ABI: include/symode.h) on PyTorch-ROCm tensors.

PyTorch is used here for device memory, streams and nothing else: every method takes CUDA
(= HIP) fp32 tensors, hands their ``data_ptr()`` to the C ABI on torch's current stream and
returns tensors allocated by torch.  There is NO CPU fallback: a missing library, a missing
GPU or a CPU tensor raises.
"""
from __future__ import annotations

import ctypes
import os
import weakref
from ctypes import c_char_p, c_float, c_int, c_long, c_size_t, c_void_p

import torch

FLAG_SINE = 1
FLAG_EXP = 2

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SYMODE_LIB") or os.path.join(_HERE, "libsymode_hip.so")

# name -> (restype, argtypes); kept in step with include/symode.h (tests check every symbol)
_SIGNATURES = {
    "symode_abi_version": (c_int, []),
    "symode_reload_env": (None, []),
    "symode_error_string": (c_char_p, [c_int]),
    "symode_lib_size": (c_int, [c_int, c_int, c_int]),
    "symode_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_long, c_long]),
    "symode_workspace_init": (c_int, [c_void_p, c_size_t, c_void_p]),
    "symode_theta": (c_int, [c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_void_p]),
    "symode_forward": (c_int, [c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "symode_odeint": (c_int, [c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_float, c_int,
                              c_void_p, c_void_p]),
    "symode_odeint_traj": (c_int, [c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_float, c_int,
                              c_void_p, c_void_p]),
    "symode_loss_grad": (c_int, [c_void_p, c_void_p, c_long, c_long, c_int, c_int, c_int, c_void_p, c_void_p, c_float,
                                 c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "symode_aug_gram": (c_int, [c_void_p, c_void_p, c_long, c_long, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t,
                                c_void_p]),
    "symode_aug_gram_gather": (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_long, c_long, c_int, c_int, c_int, c_void_p,
                                       c_void_p, c_size_t, c_void_p]),
    "symode_symreg_linear": (c_int, [c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                     c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "symode_symreg_reversed": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_long, c_int, c_int, c_int, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "symode_symreg_reversed_batched": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_long, c_long, c_int, c_int, c_int, c_void_p,
                                               c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "symode_weak_gram": (c_int, [c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t,
                                 c_void_p]),
    "symode_loss_grad_reversed": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_long, c_long, c_int, c_int, c_int, c_void_p,
                                          c_void_p, c_float, c_float, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "symode_vjp": (c_int, [c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                           c_void_p, c_size_t, c_void_p]),
    "symode_forward_jvp": (c_int, [c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p]),
    "symode_jvp_vjp": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "symode_rk4_traj": (c_int, [c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_int, ctypes.c_double, c_int, c_void_p, c_void_p,
                                c_void_p]),
    "symode_seeded_subsamples": (c_int, [c_long, c_long, c_void_p, c_int, c_void_p, c_void_p]),
    "symode_euler_jvp": (c_int, [c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_float, c_void_p,
                                 c_void_p, c_void_p]),
    "symode_euler_jvp_vjp": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_void_p,
                                     c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "symode_lbfgs_direction": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int,
                                       c_int, c_void_p, c_void_p]),
    "symode_selftest_wave_sum": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_void_p]),
    "symode_lbfgs_update": (c_int, [c_void_p] * 15 + [c_long, c_int, c_int, c_float, c_float, c_void_p]),
    "symode_lbfgs_accept_update": (c_int, [c_void_p, c_void_p, c_float, c_int, c_float, c_float] + [c_void_p] * 15
                                   + [c_long, c_int, c_int, c_float, c_float, c_void_p]),
    "symode_lbfgs_accept": (c_int, [c_void_p] * 8 + [c_long, c_int, c_float, c_float, c_void_p, c_float, c_float, c_void_p]),
    "symode_trainer_layout": (c_size_t, [c_long, c_int, c_int, c_int, c_int, c_void_p]),
    "symode_trainer_init": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "symode_trainer_closure": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "symode_trainer_update": (c_int, [c_void_p, c_int, c_void_p]),
    "symode_trainer_epoch_end": (c_int, [c_void_p, c_int, c_void_p]),
    "symode_trainer_run": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "symode_host_stlsq_sweep": (c_int, [c_void_p, c_int, c_int, c_int, c_long, ctypes.c_double, ctypes.c_double, c_int, c_int,
                                        ctypes.c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "symode_host_lstsq_normal": (c_int, [c_void_p, c_void_p, c_int, c_int, c_long, c_int, ctypes.c_double, c_void_p, c_void_p]),
}

ABI_VERSION = 5


class TrainerDesc(ctypes.Structure):
    """``symode_trainer`` of include/symode.h, field for field."""
    _fields_ = [("x", c_void_p), ("dx", c_void_p), ("gx", c_void_p), ("jgx", c_void_p), ("n_g", c_int), ("w_sym", c_float),
                ("n_problems", c_long), ("n_points", c_long), ("d", c_int), ("order", c_int), ("flags", c_int),
                ("inv_count", c_float), ("workspace", c_void_p), ("workspace_bytes", c_size_t),
                ("q_eff", c_void_p), ("r", c_int), ("allow_constant", c_int), ("n_params", c_int),
                ("w_x", c_float), ("w_reg", c_float), ("l1", c_int), ("lr", c_float), ("tol_grad", c_float), ("tol_change", c_float),
                ("max_iter", c_int), ("history", c_int),
                ("threshold", c_float), ("tol_update", c_float), ("near_band", c_float), ("st_freq", c_int),
                ("state", c_void_p), ("state_bytes", c_size_t),
                ("log", c_void_p), ("log_test", c_void_p), ("log_xi", c_void_p), ("log_mask", c_void_p), ("log_params", c_void_p), ("log_epochs", c_int)]


TRAINER_FIELDS = ("params", "xi", "mask", "cl_loss", "cl_grad", "g", "loss", "act", "n_iter", "d", "t", "old_dirs", "old_stps", "ro",
                  "head", "count", "h_diag", "prev_g", "prev_loss", "prev", "pprev", "n_iters", "done", "nan", "finished", "epochs",
                  "near", "l1_last", "test_grad")


class SymodeError(RuntimeError):
    pass


def load_library(path: str = LIB_PATH) -> ctypes.CDLL:
    """dlopen the HIP library and declare every entry point; raises if it is not built."""
    if not os.path.exists(path):
        raise SymodeError(
            f"{path} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype, fn.argtypes = res, args
    if lib.symode_abi_version() != ABI_VERSION:
        raise SymodeError(f"libsymode_hip ABI {lib.symode_abi_version()} != binding ABI {ABI_VERSION}")
    return lib


def reload_env() -> None:
    """Have the loaded library read its optional SYMODE_* variables again (it reads them once, at first use)."""
    if _ENGINE is not None:
        _ENGINE.lib.symode_reload_env()


def library_flags(include_sine: bool, include_exp: bool) -> int:
    return (FLAG_SINE if include_sine else 0) | (FLAG_EXP if include_exp else 0)


class HipEngine:
    """The product's only compute backend for the hot path."""

    def __init__(self, path: str = LIB_PATH):
        self.lib = load_library(path)
        self._ws = {}     # (device index, stream) -> scratch tensor (grown on demand, reused)

    # -- plumbing ----------------------------------------------------------------------
    def _check(self, code: int, what: str):
        if code > 0:
            # a HIP error from a launch: whatever state the scratch buffers' tickets are in, the next call gets fresh ones
            self._ws.clear()
        if code != 0:
            raise SymodeError(f"{what} failed: {self.lib.symode_error_string(code).decode()} (code {code})")

    @staticmethod
    def _dev(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise SymodeError(f"{name} must be a CUDA/HIP tensor; the hot path has no CPU fallback")
        if t.dtype != dtype:
            raise SymodeError(f"{name} must be {dtype}, got {t.dtype}")
        return t.contiguous()

    @staticmethod
    def _dev_or_pinned(t: torch.Tensor, name: str) -> torch.Tensor:
        """fp32 device tensor, or a pinned host tensor (hipHostMalloc memory is mapped into the device's address space
        at the same address): lets a latency-bound caller hand coefficients in and take [loss | grad] out of ONE launch
        without copy nodes on either side."""
        if isinstance(t, torch.Tensor) and not t.is_cuda and t.is_pinned() and t.dtype == torch.float32 and t.is_contiguous():
            return t
        return HipEngine._dev(t, name)

    @staticmethod
    def _ptr(t):
        return None if t is None else c_void_p(t.data_ptr())

    @staticmethod
    def _stream(t: torch.Tensor):
        return c_void_p(torch.cuda.current_stream(t.device).cuda_stream)

    def lib_size(self, d: int, order: int, flags: int) -> int:
        p = self.lib.symode_lib_size(d, order, flags)
        if p < 0:
            raise SymodeError(f"library d={d} order={order} flags={flags} is not compiled into libsymode_hip")
        return p

    def workspace(self, device, d, order, flags, n_problems, n, min_bytes=0) -> torch.Tensor:
        need = max(self.lib.symode_workspace_bytes(d, order, flags, n_problems, n), min_bytes)
        dev = torch.device(device)
        key = (dev.index or 0, torch.cuda.current_stream(dev).cuda_stream)   # one scratch per (device, stream)
        ws = self._ws.get(key)
        if ws is None or ws.numel() * 8 < need:
            ws = self.new_workspace(dev, need)
            self._ws[key] = ws
        return ws

    def new_workspace(self, device, nbytes) -> torch.Tensor:
        """A private scratch buffer, header initialised on the current stream (symode_workspace_init): for callers
        that replay launches from a HIP graph or from several streams in turn and must not share the engine's."""
        ws = torch.empty(max(nbytes // 8 + 1, 1024), dtype=torch.float64, device=device)
        self._check(self.lib.symode_workspace_init(self._ptr(ws), ws.numel() * 8, self._stream(ws)), "symode_workspace_init")
        return ws

    # -- entry points ------------------------------------------------------------------
    def theta(self, x, order, flags=0):
        x = self._dev(x, "x")
        lead, d = x.shape[:-1], x.shape[-1]
        n = x.numel() // d if d else 0
        p = self.lib_size(d, order, flags)
        out = torch.empty(*lead, p, dtype=torch.float32, device=x.device)
        self._check(self.lib.symode_theta(self._ptr(x), n, d, order, flags, self._ptr(out), self._stream(x)), "symode_theta")
        return out

    def forward(self, x, xi, mask, order, flags=0):
        x = self._dev(x, "x")
        d = x.shape[-1]
        n = x.numel() // d
        xi = self._dev(xi, "xi")
        mask = None if mask is None else self._dev(mask, "mask")
        self._check_coef(xi, mask, d, order, flags)
        out = torch.empty_like(x)
        self._check(self.lib.symode_forward(self._ptr(x), n, d, order, flags, self._ptr(xi), self._ptr(mask),
                                            self._ptr(out), self._stream(x)), "symode_forward")
        return out

    def odeint(self, x, xi, mask, order, flags, n_steps, dt, method="euler"):
        x = self._dev(x, "x")
        d = x.shape[-1]
        n = x.numel() // d
        xi = self._dev(xi, "xi")
        mask = None if mask is None else self._dev(mask, "mask")
        self._check_coef(xi, mask, d, order, flags)
        m = {"euler": 0, "rk4": 1}.get(method)
        if m is None:
            raise ValueError("Unrecognized ODEInt method.")
        out = torch.empty_like(x)
        self._check(self.lib.symode_odeint(self._ptr(x), n, d, order, flags, self._ptr(xi), self._ptr(mask), int(n_steps),
                                           float(dt), m, self._ptr(out), self._stream(x)), "symode_odeint")
        return out

    def odeint_traj(self, x, xi, mask, order, flags, n_steps, dt, method="euler"):
        """(n_steps, n, d): the state after every step (odeint(..., full_traj=True))."""
        x = self._dev(x, "x")
        d = x.shape[-1]
        n = x.numel() // d
        xi = self._dev(xi, "xi")
        mask = None if mask is None else self._dev(mask, "mask")
        self._check_coef(xi, mask, d, order, flags)
        m = {"euler": 0, "rk4": 1}.get(method)
        if m is None:
            raise ValueError("Unrecognized ODEInt method.")
        traj = torch.empty(int(n_steps), n, d, dtype=torch.float32, device=x.device)
        self._check(self.lib.symode_odeint_traj(self._ptr(x), n, d, order, flags, self._ptr(xi), self._ptr(mask), int(n_steps),
                                                float(dt), m, self._ptr(traj), self._stream(x)), "symode_odeint_traj")
        return traj

    def _check_coef(self, xi, mask, d, order, flags, n_problems=1):
        p = self.lib_size(d, order, flags)
        want = n_problems * d * p
        if xi.numel() != want:
            raise SymodeError(f"xi has {xi.numel()} elements, expected {n_problems}x{d}x{p}")
        if mask is not None and mask.numel() != want:
            raise SymodeError(f"mask has {mask.numel()} elements, expected {n_problems}x{d}x{p}")
        return p

    def loss_grad(self, x, dx, xi, mask, order, flags=0, inv_count=None, out=None, ws=None):
        """x, dx: (S, N, d) or (N, d); xi, mask: (S, d, p) or (d, p).  Returns (loss (S,), grad (S, d, p)).
        ``xi`` and ``out`` may be pinned host tensors (zero-copy); ``ws`` a private workspace from new_workspace()."""
        x, dx = self._dev(x, "x"), self._dev(dx, "dx")
        if x.shape != dx.shape:
            raise SymodeError(f"x {tuple(x.shape)} and dx {tuple(dx.shape)} differ")
        batched = x.dim() == 3
        S = x.shape[0] if batched else 1
        n, d = x.shape[-2], x.shape[-1]
        xi = self._dev_or_pinned(xi, "xi")
        mask = None if mask is None else self._dev(mask, "mask")
        p = self._check_coef(xi, mask, d, order, flags, S)
        if out is None:
            loss = torch.empty(S, dtype=torch.float32, device=x.device)
            grad = torch.empty(S, d, p, dtype=torch.float32, device=x.device)
        else:
            loss, grad = (self._dev_or_pinned(o, "out") for o in out)
            if loss.numel() != S or grad.numel() != S * d * p:
                raise SymodeError(f"out buffers hold {loss.numel()} / {grad.numel()} elements, expected {S} / {S * d * p}")
        if ws is None:
            ws = self.workspace(x.device, d, order, flags, S, n)
        elif ws.numel() * 8 < self.lib.symode_workspace_bytes(d, order, flags, S, n):
            raise SymodeError("private workspace too small for this call")
        inv = 1.0 / (n * d) if inv_count is None else float(inv_count)
        self._check(self.lib.symode_loss_grad(self._ptr(x), self._ptr(dx), S, n, d, order, flags, self._ptr(xi),
                                              self._ptr(mask), inv, self._ptr(loss), self._ptr(grad), self._ptr(ws),
                                              ws.numel() * 8, self._stream(x)), "symode_loss_grad")
        if not batched:
            return loss[0], grad[0]
        return loss, grad

    def aug_gram(self, x, dx, order, flags=0):
        """fp64 augmented Gram [Theta | dx]^T [Theta | dx]: (S, p+d, p+d) or (p+d, p+d)."""
        x, dx = self._dev(x, "x"), self._dev(dx, "dx")
        if x.shape != dx.shape:
            raise SymodeError(f"x {tuple(x.shape)} and dx {tuple(dx.shape)} differ")
        batched = x.dim() == 3
        S = x.shape[0] if batched else 1
        n, d = x.shape[-2], x.shape[-1]
        p = self.lib_size(d, order, flags)
        gram = torch.empty(S, p + d, p + d, dtype=torch.float64, device=x.device)
        ws = self.workspace(x.device, d, order, flags, S, n)
        self._check(self.lib.symode_aug_gram(self._ptr(x), self._ptr(dx), S, n, d, order, flags, self._ptr(gram),
                                             self._ptr(ws), ws.numel() * 8, self._stream(x)), "symode_aug_gram")
        return gram if batched else gram[0]

    def aug_gram_gather(self, x, dx, idx, order, flags=0):
        """Gram matrices of S index subsets of one shared data set: x, dx (N, d); idx (S, M) int32 rows."""
        x, dx = self._dev(x, "x"), self._dev(dx, "dx")
        idx = self._dev(idx, "idx", torch.int32)
        if x.dim() != 2 or x.shape != dx.shape or idx.dim() != 2:
            raise SymodeError("aug_gram_gather expects x, dx (N, d) and idx (S, M)")
        n_src, d = x.shape
        S, m = idx.shape
        # the kernel trusts the table: checked once per live table tensor (one reduction, one sync -- ~40 us at config[3]'s
        # 64 x 50 000 rows, more than the 25 us kernel), remembered by object, in-place version and N
        seen = getattr(self, "_idx_checked", None)
        if seen is None or seen[0]() is not idx or seen[1:] != (idx._version, n_src):
            lo, hi = torch.stack(torch.aminmax(idx)).tolist() if idx.numel() else (0, 0)
            if lo < 0 or hi >= max(n_src, 1):
                raise SymodeError("idx holds row indices outside [0, N)")
            self._idx_checked = (weakref.ref(idx), idx._version, n_src)
        p = self.lib_size(d, order, flags)
        gram = torch.empty(S, p + d, p + d, dtype=torch.float64, device=x.device)
        ws = self.workspace(x.device, d, order, flags, S, m)
        self._check(self.lib.symode_aug_gram_gather(self._ptr(x), self._ptr(dx), n_src, self._ptr(idx), S, m, d, order, flags,
                                                    self._ptr(gram), self._ptr(ws), ws.numel() * 8, self._stream(x)),
                    "symode_aug_gram_gather")
        return gram

    def symreg_linear(self, z, xi, mask, L, order, flags=0):
        z = self._dev(z, "z")
        n, d = z.shape[-2], z.shape[-1]
        xi = self._dev(xi, "xi")
        mask = None if mask is None else self._dev(mask, "mask")
        p = self._check_coef(xi, mask, d, order, flags)
        L = self._dev(L, "L").reshape(-1, d, d)
        loss = torch.empty(1, dtype=torch.float32, device=z.device)
        grad = torch.empty(d, p, dtype=torch.float32, device=z.device)
        ws = self.workspace(z.device, d, order, flags, 1, n)
        self._check(self.lib.symode_symreg_linear(self._ptr(z), n, d, order, flags, self._ptr(xi), self._ptr(mask),
                                                  self._ptr(L), L.shape[0], self._ptr(loss), self._ptr(grad),
                                                  self._ptr(ws), ws.numel() * 8, self._stream(z)), "symode_symreg_linear")
        return loss[0], grad

    def symreg_reversed(self, x, gx, jgx, xi, mask, order, flags=0, out=None, ws=None, inv_count=None):
        """Reversed symmetry regulariser on precomputed (g(x), J_g(x)).
        One problem: x (N, d), gx (n_g, N, d), jgx (n_g, N, d, d), xi / mask (d, p) -> (loss scalar, grad (d, p)).
        S problems in one launch: x (S, N, d), gx (S, n_g, N, d), jgx (S, n_g, N, d, d), xi / mask (S, d, p) -> ((S,), (S, d, p)).
        ``xi`` / ``out`` may be pinned host tensors for the one-problem form; ``inv_count`` as in loss_grad."""
        x, gx, jgx = self._dev(x, "x"), self._dev(gx, "gx"), self._dev(jgx, "jgx")
        batched = x.dim() == 3
        S = x.shape[0] if batched else 1
        n, d = x.shape[-2], x.shape[-1]
        n_g = gx.shape[1] if batched else gx.shape[0]
        want_g = (S, n_g, n, d) if batched else (n_g, n, d)
        if tuple(gx.shape) != want_g or tuple(jgx.shape) != want_g + (d,):
            raise SymodeError(f"gx {tuple(gx.shape)} / jgx {tuple(jgx.shape)} do not match x {tuple(x.shape)}")
        xi = self._dev_or_pinned(xi, "xi")
        mask = None if mask is None else self._dev(mask, "mask")
        p = self._check_coef(xi, mask, d, order, flags, S)
        if out is None:
            loss = torch.empty(S, dtype=torch.float32, device=x.device)
            grad = torch.empty(S, d, p, dtype=torch.float32, device=x.device)
        else:
            loss, grad = (self._dev_or_pinned(o, "out") for o in out)
            if loss.numel() != S or grad.numel() != S * d * p:
                raise SymodeError(f"out buffers hold {loss.numel()} / {grad.numel()} elements, expected {S} / {S * d * p}")
        if ws is None:
            ws = self.workspace(x.device, d, order, flags, S, n)
        elif ws.numel() * 8 < self.lib.symode_workspace_bytes(d, order, flags, S, n):
            raise SymodeError("private workspace too small for this call")
        inv = 1.0 / (n * d) if inv_count is None else float(inv_count)
        self._check(self.lib.symode_symreg_reversed_batched(self._ptr(x), self._ptr(gx), self._ptr(jgx), n_g, S, n, d, order,
                                                            flags, self._ptr(xi), self._ptr(mask), inv, self._ptr(loss),
                                                            self._ptr(grad), self._ptr(ws), ws.numel() * 8, self._stream(x)),
                    "symode_symreg_reversed_batched")
        if not batched:
            return loss.reshape(-1)[0], grad.reshape(d, p)
        return loss, grad


    def loss_grad_reversed(self, x, dx, gx, jgx, xi, mask, order, flags=0, w_sym=1.0, inv_count=None, out=None, ws=None):
        """The closure MSE + w_sym * reversed regulariser in ONE pass (x read once, Theta(x) shared by both terms).
        Shapes as loss_grad / symreg_reversed.  Returns (loss2, grad): loss2 (S, 2) [or (2,)] = (mse, regulariser), grad =
        d(mse + w_sym * regulariser)/dXi (S, d, p) [or (d, p)]."""
        x, dx, gx, jgx = self._dev(x, "x"), self._dev(dx, "dx"), self._dev(gx, "gx"), self._dev(jgx, "jgx")
        if x.shape != dx.shape:
            raise SymodeError(f"x {tuple(x.shape)} and dx {tuple(dx.shape)} differ")
        batched = x.dim() == 3
        S = x.shape[0] if batched else 1
        n, d = x.shape[-2], x.shape[-1]
        n_g = gx.shape[1] if batched else gx.shape[0]
        want_g = (S, n_g, n, d) if batched else (n_g, n, d)
        if n_g < 1 or tuple(gx.shape) != want_g or tuple(jgx.shape) != want_g + (d,):
            raise SymodeError(f"gx {tuple(gx.shape)} / jgx {tuple(jgx.shape)} do not match x {tuple(x.shape)}")
        xi = self._dev_or_pinned(xi, "xi")
        mask = None if mask is None else self._dev(mask, "mask")
        p = self._check_coef(xi, mask, d, order, flags, S)
        if out is None:
            loss2 = torch.empty(S, 2, dtype=torch.float32, device=x.device)
            grad = torch.empty(S, d, p, dtype=torch.float32, device=x.device)
        else:
            loss2, grad = (self._dev_or_pinned(o, "out") for o in out)
            if loss2.numel() != 2 * S or grad.numel() != S * d * p:
                raise SymodeError(f"out buffers hold {loss2.numel()} / {grad.numel()} elements, expected {2 * S} / {S * d * p}")
        if ws is None:
            ws = self.workspace(x.device, d, order, flags, S, n)
        elif ws.numel() * 8 < self.lib.symode_workspace_bytes(d, order, flags, S, n):
            raise SymodeError("private workspace too small for this call")
        inv = 1.0 / (n * d) if inv_count is None else float(inv_count)
        self._check(self.lib.symode_loss_grad_reversed(self._ptr(x), self._ptr(dx), self._ptr(gx), self._ptr(jgx), n_g, S, n, d, order,
                                                       flags, self._ptr(xi), self._ptr(mask), inv, float(w_sym), self._ptr(loss2),
                                                       self._ptr(grad), self._ptr(ws), ws.numel() * 8, self._stream(x)),
                    "symode_loss_grad_reversed")
        if not batched:
            return loss2.reshape(2), grad.reshape(d, p)
        return loss2.reshape(S, 2), grad.reshape(S, d, p)

    def bind_closure(self, x, dx, xi, mask, order, flags, out, ws, stream, reversed_sym=None, w_sym=1.0):
        """A zero-argument callable that launches the single-problem closure on fixed buffers: every check and every
        ctypes conversion is done once, here (a latency-bound caller -- one L-BFGS closure is 8 us of GPU time -- pays
        ~15 us of Python per call otherwise).  ``out`` = (loss, grad) [(loss2, grad) with ``reversed_sym = (gx, jgx)``];
        ``xi`` / ``out`` may be pinned host tensors; ``stream`` a torch stream.  The callable keeps the tensors alive."""
        x, dx = self._dev(x, "x"), self._dev(dx, "dx")
        n, d = x.shape[-2], x.shape[-1]
        xi = self._dev_or_pinned(xi, "xi")
        mask = None if mask is None else self._dev(mask, "mask")
        p = self._check_coef(xi, mask, d, order, flags, 1)
        loss, grad = (self._dev_or_pinned(o, "out") for o in out)
        if grad.numel() != d * p or loss.numel() != (2 if reversed_sym is not None else 1):
            raise SymodeError("out buffers do not match the closure's outputs")
        if ws.numel() * 8 < self.lib.symode_workspace_bytes(d, order, flags, 1, n):
            raise SymodeError("private workspace too small for this call")
        keep = (x, dx, xi, mask, loss, grad, ws, stream)
        st = c_void_p(stream.cuda_stream)
        inv = c_float(1.0 / (n * d))
        if reversed_sym is None:
            fn = self.lib.symode_loss_grad
            args = (self._ptr(x), self._ptr(dx), c_long(1), c_long(n), c_int(d), c_int(order), c_int(flags), self._ptr(xi),
                    self._ptr(mask), inv, self._ptr(loss), self._ptr(grad), self._ptr(ws), c_size_t(ws.numel() * 8), st)
        else:
            gx, jgx = self._dev(reversed_sym[0], "gx"), self._dev(reversed_sym[1], "jgx")
            n_g = gx.shape[0]
            if tuple(gx.shape) != (n_g, n, d) or tuple(jgx.shape) != (n_g, n, d, d) or n_g < 1:
                raise SymodeError("gx / jgx do not match x")
            keep += (gx, jgx)
            fn = self.lib.symode_loss_grad_reversed
            args = (self._ptr(x), self._ptr(dx), self._ptr(gx), self._ptr(jgx), c_int(n_g), c_long(1), c_long(n), c_int(d),
                    c_int(order), c_int(flags), self._ptr(xi), self._ptr(mask), inv, c_float(float(w_sym)), self._ptr(loss),
                    self._ptr(grad), self._ptr(ws), c_size_t(ws.numel() * 8), st)

        def launch(_fn=fn, _args=args, _keep=keep):
            rc = _fn(*_args)
            if rc != 0:
                self._check(rc, "bound closure")
        return launch

    def weak_gram(self, x, V, V_drv, order, flags=0):
        """Weak-SINDy contraction: x (T, d) one trajectory, V / V_drv (K, T) -> (G = V Theta(x) (K, p), b = -V_drv x (K, d)), fp64."""
        x, V, V_drv = self._dev(x, "x"), self._dev(V, "V"), self._dev(V_drv, "V_drv")
        if x.dim() != 2 or V.dim() != 2 or V.shape != V_drv.shape or V.shape[1] != x.shape[0]:
            raise SymodeError(f"weak_gram expects x (T, d) and V, V_drv (K, T); got {tuple(x.shape)}, {tuple(V.shape)}, {tuple(V_drv.shape)}")
        T, d = x.shape
        K = V.shape[0]
        p = self.lib_size(d, order, flags)
        R, C = 16 * ((2 * K + 15) // 16), 16 * ((p + d + 15) // 16)
        out = torch.empty(R, C, dtype=torch.float64, device=x.device)
        # room for the widest launch (64 time slabs x row tiles x column tiles of 256 fp64 partials behind the header):
        # a scratch sized for the closures alone made the library halve the grid to 5 workgroups at T = 10^4
        ws = self.workspace(x.device, d, order, flags, 1, T, min_bytes=8 * (8 + 32768 + 64 * (R // 16) * (C // 16) * 256) + 4096)
        self._check(self.lib.symode_weak_gram(self._ptr(x), T, d, order, flags, self._ptr(V), self._ptr(V_drv), K, self._ptr(out),
                                              self._ptr(ws), ws.numel() * 8, self._stream(x)), "symode_weak_gram")
        return out[:K, :p], out[K:2 * K, p:p + d]

    def vjp(self, x, g, xi, mask, order, flags=0, need_grad_x=True):
        """Reverse mode of forward: returns (grad_x (N, d) or None, grad_xi (d, p))."""
        x, g = self._dev(x, "x"), self._dev(g, "g")
        n, d = x.shape[-2], x.shape[-1]
        xi = self._dev(xi, "xi")
        mask = None if mask is None else self._dev(mask, "mask")
        p = self._check_coef(xi, mask, d, order, flags)
        gx = torch.empty_like(x) if need_grad_x else None
        gxi = torch.empty(d, p, dtype=torch.float32, device=x.device)
        ws = self.workspace(x.device, d, order, flags, 1, n)
        self._check(self.lib.symode_vjp(self._ptr(x), self._ptr(g), n, d, order, flags, self._ptr(xi), self._ptr(mask),
                                        self._ptr(gx), self._ptr(gxi), self._ptr(ws), ws.numel() * 8, self._stream(x)),
                    "symode_vjp")
        return gx, gxi

    def forward_jvp(self, x, v, xi, mask, order, flags=0, need_out=True):
        """Forward mode: (out, J.v) for tangents v of x's shape."""
        x, v = self._dev(x, "x"), self._dev(v, "v")
        d = x.shape[-1]
        n = x.numel() // d
        xi = self._dev(xi, "xi")
        mask = None if mask is None else self._dev(mask, "mask")
        self._check_coef(xi, mask, d, order, flags)
        out = torch.empty_like(x) if need_out else None
        jv = torch.empty_like(x)
        self._check(self.lib.symode_forward_jvp(self._ptr(x), self._ptr(v), n, d, order, flags, self._ptr(xi),
                                                self._ptr(mask), self._ptr(out), self._ptr(jv), self._stream(x)),
                    "symode_forward_jvp")
        return out, jv


    def jvp_vjp(self, x, v, g_out, g_jv, xi, mask, order, flags=0):
        """Reverse mode of forward_jvp: returns (grad_x, grad_v, grad_xi); g_out may be None."""
        x, v, g_jv = self._dev(x, "x"), self._dev(v, "v"), self._dev(g_jv, "g_jv")
        g_out = None if g_out is None else self._dev(g_out, "g_out")
        d = x.shape[-1]
        n = x.numel() // d
        xi = self._dev(xi, "xi")
        mask = None if mask is None else self._dev(mask, "mask")
        p = self._check_coef(xi, mask, d, order, flags)
        gx, gv = torch.empty_like(x), torch.empty_like(x)
        gxi = torch.empty(d, p, dtype=torch.float32, device=x.device)
        ws = self.workspace(x.device, d, order, flags, 1, n)
        self._check(self.lib.symode_jvp_vjp(self._ptr(x), self._ptr(v), self._ptr(g_out), self._ptr(g_jv), n, d, order,
                                            flags, self._ptr(xi), self._ptr(mask), self._ptr(gx), self._ptr(gv),
                                            self._ptr(gxi), self._ptr(ws), ws.numel() * 8, self._stream(x)),
                    "symode_jvp_vjp")
        return gx, gv, gxi


    def rk4_traj(self, x0, xi, order, flags, n_steps, dt, subsample=1):
        """fp32 (x, dx) of shape (n_traj, ceil(n_steps/subsample), d): fp64 RK4 orbits of dx/dt = Theta(x) xi^T."""
        x0 = self._dev(x0, "x0", torch.float64)
        xi = self._dev(xi, "xi", torch.float64)
        n_traj, d = x0.shape
        p = self.lib_size(d, order, flags)
        if xi.shape != (d, p):
            raise SymodeError(f"xi must be ({d}, {p}), got {tuple(xi.shape)}")
        n_out = (n_steps + subsample - 1) // subsample
        x = torch.empty(n_traj, n_out, d, dtype=torch.float32, device=x0.device)
        dx = torch.empty_like(x)
        self._check(self.lib.symode_rk4_traj(self._ptr(x0), n_traj, d, order, flags, self._ptr(xi), int(n_steps), float(dt),
                                             int(subsample), self._ptr(x), self._ptr(dx), self._stream(x0)), "symode_rk4_traj")
        return x, dx

    def seeded_subsamples(self, n, m, seeds, device):
        """(len(seeds), m) int32 index table, rows ascending: per seed the m rows of range(n) with the smallest counter-based
        keys (symode_seeded_subsamples); depends on each seed alone."""
        s = torch.as_tensor([int(v) for v in seeds], dtype=torch.int64).to(device)
        out = torch.empty(len(s), int(m), dtype=torch.int32, device=s.device)
        self._check(self.lib.symode_seeded_subsamples(int(n), int(m), self._ptr(s), len(s), self._ptr(out), self._stream(out)),
                    "symode_seeded_subsamples")
        return out

    def euler_jvp(self, x, v, xi, mask, order, flags, n_steps, dt):
        """(f(x), J_f(x) v) for f = n_steps Euler steps of the regressor ODE; one launch."""
        x, v = self._dev(x, "x"), self._dev(v, "v")
        d = x.shape[-1]
        n = x.numel() // d
        xi = self._dev(xi, "xi")
        mask = None if mask is None else self._dev(mask, "mask")
        self._check_coef(xi, mask, d, order, flags)
        xo, to = torch.empty_like(x), torch.empty_like(x)
        self._check(self.lib.symode_euler_jvp(self._ptr(x), self._ptr(v), n, d, order, flags, self._ptr(xi), self._ptr(mask),
                                              int(n_steps), float(dt), self._ptr(xo), self._ptr(to), self._stream(x)),
                    "symode_euler_jvp")
        return xo, to

    def euler_jvp_vjp(self, x, v, g_x, g_t, xi, mask, order, flags, n_steps, dt):
        """Reverse mode of euler_jvp: (grad_x, grad_v, grad_xi)."""
        x, v, g_x, g_t = (self._dev(a, nm) for a, nm in ((x, "x"), (v, "v"), (g_x, "g_x"), (g_t, "g_t")))
        d = x.shape[-1]
        n = x.numel() // d
        xi = self._dev(xi, "xi")
        mask = None if mask is None else self._dev(mask, "mask")
        p = self._check_coef(xi, mask, d, order, flags)
        gx, gv = torch.empty_like(x), torch.empty_like(x)
        gxi = torch.empty(d, p, dtype=torch.float32, device=x.device)
        ws = self.workspace(x.device, d, order, flags, 1, n)
        self._check(self.lib.symode_euler_jvp_vjp(self._ptr(x), self._ptr(v), self._ptr(g_x), self._ptr(g_t), n, d, order,
                                                  flags, self._ptr(xi), self._ptr(mask), int(n_steps), float(dt),
                                                  self._ptr(gx), self._ptr(gv), self._ptr(gxi), self._ptr(ws),
                                                  ws.numel() * 8, self._stream(x)), "symode_euler_jvp_vjp")
        return gx, gv, gxi

    def lbfgs_direction(self, g, old_dirs, old_stps, ro, head, count, h_diag):
        """d = -H g by the two-loop recursion for S problems at once (ring-buffered curvature pairs)."""
        g = self._dev(g, "g")
        S, n = g.shape
        H = old_dirs.shape[1]
        out = torch.empty_like(g)
        self._check(self.lib.symode_lbfgs_direction(self._ptr(g), self._ptr(self._dev(old_dirs, "old_dirs")),
                                                    self._ptr(self._dev(old_stps, "old_stps")), self._ptr(self._dev(ro, "ro")),
                                                    self._ptr(self._dev(head, "head", torch.int64)),
                                                    self._ptr(self._dev(count, "count", torch.int64)),
                                                    self._ptr(self._dev(h_diag, "h_diag")), S, n, H, self._ptr(out),
                                                    self._stream(g)), "symode_lbfgs_direction")
        return out

    def lbfgs_update(self, params, g, loss, act, st, lr, tol_change):
        """In place: one L-BFGS inner iteration up to the move x += t d for every active problem (``st``: the
        optimiser's state tensors, see sweep.BatchedLBFGS); ``act`` (S,) bool: active in, moved out."""
        S, n = params.shape
        H = st.old_dirs.shape[1]
        for name, ten, dt in (("params", params, torch.float32), ("g", g, torch.float32), ("loss", loss, torch.float32),
                              ("act", act, torch.bool), ("n_iter", st.n_iter, torch.int64), ("d", st.d, torch.float32),
                              ("t", st.t, torch.float32), ("old_dirs", st.old_dirs, torch.float32),
                              ("old_stps", st.old_stps, torch.float32), ("ro", st.ro, torch.float32),
                              ("head", st.head, torch.int64), ("hist", st.hist, torch.int64),
                              ("H_diag", st.H_diag, torch.float32), ("prev_g", st.prev_g, torch.float32),
                              ("prev_loss", st.prev_loss, torch.float32)):
            if not (ten.is_cuda and ten.dtype == dt and ten.is_contiguous()):
                raise SymodeError(f"lbfgs_update: {name} must be a contiguous {dt} GPU tensor (updated in place)")
        self._check(self.lib.symode_lbfgs_update(self._ptr(params), self._ptr(g), self._ptr(loss), self._ptr(act),
                                                 self._ptr(st.n_iter), self._ptr(st.d), self._ptr(st.t), self._ptr(st.old_dirs),
                                                 self._ptr(st.old_stps), self._ptr(st.ro), self._ptr(st.head), self._ptr(st.hist),
                                                 self._ptr(st.H_diag), self._ptr(st.prev_g), self._ptr(st.prev_loss), S, n, H,
                                                 float(lr), float(tol_change), self._stream(params)), "symode_lbfgs_update")

    def lbfgs_accept_update(self, new_loss, new_g, params, g, loss, act, st, lr, tol_grad, tol_change, l1=None):
        """lbfgs_accept followed by lbfgs_update in ONE launch (state tensors checked by the preceding lbfgs_update call of
        the same optimiser step).  ``l1 = (w_x, w_reg)``: new_loss / new_g are the bare data term."""
        S, n = params.shape
        new_loss, new_g = self._dev(new_loss, "new_loss"), self._dev(new_g, "new_g")
        w_x, w_reg = (1.0, 0.0) if l1 is None else l1
        self._check(self.lib.symode_lbfgs_accept_update(
            self._ptr(new_loss), self._ptr(new_g), float(tol_grad), 0 if l1 is None else 1, float(w_x), float(w_reg),
            self._ptr(params), self._ptr(g), self._ptr(loss), self._ptr(act), self._ptr(st.n_iter), self._ptr(st.d), self._ptr(st.t),
            self._ptr(st.old_dirs), self._ptr(st.old_stps), self._ptr(st.ro), self._ptr(st.head), self._ptr(st.hist),
            self._ptr(st.H_diag), self._ptr(st.prev_g), self._ptr(st.prev_loss), S, n, st.old_dirs.shape[1], float(lr),
            float(tol_change), self._stream(params)), "symode_lbfgs_accept_update")

    def lbfgs_accept(self, new_loss, new_g, loss, g, act, st, tol_grad, tol_change, l1=None):
        """In place: moved problems take the re-evaluated loss / gradient and run the stopping tests; ``act``: moved in,
        still active out.  ``l1 = (params, w_x, w_reg)``: new_loss / new_g are the bare data term, the kernel forms
        w_x * loss + w_reg * |params|_1 and its gradient."""
        S, n = g.shape
        params, w_x, w_reg = (None, 1.0, 0.0) if l1 is None else l1
        if params is not None and not (params.is_cuda and params.dtype == torch.float32 and params.is_contiguous()
                                       and params.shape == g.shape):
            raise SymodeError("lbfgs_accept: params must be a contiguous fp32 GPU tensor shaped like g")
        new_loss, new_g = self._dev(new_loss, "new_loss"), self._dev(new_g, "new_g")
        for name, ten, dt in (("loss", loss, torch.float32), ("g", g, torch.float32), ("act", act, torch.bool)):
            if not (ten.is_cuda and ten.dtype == dt and ten.is_contiguous()):
                raise SymodeError(f"lbfgs_accept: {name} must be a contiguous {dt} GPU tensor (updated in place)")
        self._check(self.lib.symode_lbfgs_accept(self._ptr(new_loss), self._ptr(new_g), self._ptr(loss), self._ptr(g),
                                                 self._ptr(act), self._ptr(st.d), self._ptr(st.t), self._ptr(st.prev_loss), S, n,
                                                 float(tol_grad), float(tol_change), self._ptr(params), float(w_x),
                                                 float(w_reg), self._stream(g)), "symode_lbfgs_accept")


_ENGINE = None


def get_engine() -> HipEngine:
    """Process-wide engine; raises SymodeError if libsymode_hip.so is not built."""
    global _ENGINE
    if _ENGINE is None:
        _ENGINE = HipEngine()
    return _ENGINE
