"""SINDy operator layer on the HIP engine -- same surface as the reference's ``sindy.py``.

``SINDyRegression`` keeps the reference's constructor, parameter names (``Xi`` or
``beta`` + ``const`` in the state_dict), plain-tensor attributes (``mask``, ``Q``) and methods
(sindy.py:33-247); ``solve_SINDy_one_step`` / ``solve_SINDy`` keep their signatures and
in-place update semantics (sindy.py:250-324).  What changes is how the numbers are made:

  * Theta is never materialised on the training path: ``forward`` is one fused kernel,
    ``mse_loss`` is the fused Theta + residual + loss + gradient kernel (symode_loss_grad);
  * the sequential-threshold least-squares step reads the data ONCE into the fp64 augmented
    Gram matrix (symode_aug_gram, MFMA) and solves every masked / constrained variant of the
    reference's block-diagonal system on the host from that (p+d)^2 matrix (lstsq.py), instead
    of building a dense (d(N+p)) x (dp) matrix per pass (sindy.py:270-273).

There is no CPU fallback: tensors must live on the GPU.
"""
from __future__ import annotations

import weakref

import numpy as np
import torch
import torch.nn as nn

from . import library
from .constraint import constraint_M, constraint_Q
from .engine import get_engine, library_flags
from .lstsq import lstsq_normal


# BASELINE.md section 3 / SURVEY H5: masks come from the strict test |coef| > threshold after an iterative fit in fp32, so
# a coefficient that lands within this band of the threshold could fall on the other side under a different (equally
# valid) rounding.  Every thresholding event records such cases instead of hiding them; parity runs assert there are none.
NEAR_THRESHOLD_BAND = 1e-4
# Coefficients are RECORDED within this wider band: an iterative fit ends inside its optimiser's stopping ball (L-BFGS
# trainer: parameter update < 1e-3, train.py:643, 705), so two runs that differ in the last bit of a closure can differ
# by that much in a coefficient at a thresholding event -- ``near_threshold_within(1e-3)`` is the list that explains a
# mask flip there; ``near_threshold`` is the BASELINE.md section 3 list (1e-4).
NEAR_THRESHOLD_RECORD_BAND = 1e-3


def near_threshold_cases(xi, mask, threshold, band=NEAR_THRESHOLD_BAND):
    """[(row, col, |coef|)] for the still-active coefficients with | |coef| - threshold | < band (numpy in, list out)."""
    a = np.abs(np.asarray(xi, dtype=np.float64))
    hit = np.logical_and(np.abs(a - float(threshold)) < band, np.asarray(mask) > 0)
    return [(int(i), int(k), float(a[i, k])) for i, k in zip(*np.nonzero(hit))]


class _Forward(torch.autograd.Function):
    """dx_hat = Theta(x) (Xi*mask)^T, differentiable once w.r.t. x and Xi (HIP forward + vjp)."""

    @staticmethod
    def forward(ctx, x, xi, mask, reg):
        eng = reg.engine
        ctx.reg = reg
        ctx.save_for_backward(x, xi, mask)
        return eng.forward(x.detach(), xi.detach(), mask, reg.poly_order, reg.flags)

    @staticmethod
    def backward(ctx, g):
        x, xi, mask = ctx.saved_tensors
        reg = ctx.reg
        gx, gxi = reg.engine.vjp(x.detach(), g.contiguous(), xi.detach(), mask, reg.poly_order, reg.flags,
                                 need_grad_x=ctx.needs_input_grad[0])
        return gx, (gxi if ctx.needs_input_grad[1] else None), None, None


class _ForwardJvp(torch.autograd.Function):
    """(out, jv) = (Theta(x), J_Theta(x) v) (Xi*mask)^T, differentiable once w.r.t. x, v and Xi."""

    @staticmethod
    def forward(ctx, x, v, xi, mask, reg):
        ctx.reg = reg
        ctx.save_for_backward(x, v, xi, mask)
        out, jv = reg.engine.forward_jvp(x.detach(), v.detach(), xi.detach(), mask, reg.poly_order, reg.flags)
        return out, jv

    @staticmethod
    def backward(ctx, g_out, g_jv):
        x, v, xi, mask = ctx.saved_tensors
        reg = ctx.reg
        g_jv = torch.zeros_like(x) if g_jv is None else g_jv.contiguous()
        gx, gv, gxi = reg.engine.jvp_vjp(x.detach(), v.detach(), None if g_out is None else g_out.contiguous(), g_jv,
                                         xi.detach(), mask, reg.poly_order, reg.flags)
        return gx, gv, gxi, None, None


class _FusedMSE(torch.autograd.Function):
    """mean((Theta(x)(Xi*mask)^T - dx)^2) with its Xi-gradient from the same single pass."""

    @staticmethod
    def forward(ctx, xi, mask, x, dx, reg, inv_count):
        loss, grad = reg.engine.loss_grad(x, dx, xi.detach(), mask, reg.poly_order, reg.flags, inv_count=inv_count)
        ctx.save_for_backward(grad)
        return loss.clone()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return g * grad, None, None, None, None, None


class _LstsqResidual(torch.autograd.Function):
    """The value ``lm.residuals.mean() / N`` of the ridge least-squares solve as a differentiable function of
    the data (x, y) -- what ``train_lassi`` back-propagates into the encoder (train.py:166-174).

    By the envelope theorem the optimal coefficients need no derivative: with r = Theta(x) Xi*^T - y,
    d res / dy = -2 c r  and  d res / dx = 2 c J_Theta(x)^T (Xi* r)  (c = the reference's averaging factor);
    both come from the HIP forward and vjp kernels.  Forward returns the residual computed from the Gram.
    """

    @staticmethod
    def forward(ctx, x, y, value, xi_sol, scale, reg):
        ctx.reg, ctx.scale = reg, scale
        ctx.save_for_backward(x, y, xi_sol)
        return value.clone()

    @staticmethod
    def backward(ctx, g):
        x, y, xi = ctx.saved_tensors
        reg = ctx.reg
        d = reg.latent_dim
        x2, y2 = x.detach().reshape(-1, d).contiguous(), y.detach().reshape(-1, d).contiguous()
        r = reg.engine.forward(x2, xi, None, reg.poly_order, reg.flags) - y2
        gy = (-2.0 * ctx.scale) * g * r
        gx = None
        if ctx.needs_input_grad[0]:
            gx, _ = reg.engine.vjp(x2, ((2.0 * ctx.scale) * g * r).contiguous(), xi, None, reg.poly_order, reg.flags)
            gx = gx.reshape(x.shape)
        return gx, (gy.reshape(y.shape) if ctx.needs_input_grad[1] else None), None, None, None, None


class SINDyRegression(nn.Module):
    """
    Arguments (reference sindy.py:33-42):
        latent_dim: dimension of the state
        poly_order: highest polynomial order (reference: max 3; here up to 5)
        include_sine / include_exp: append sin / exp columns (forced off under the constraint)
        L_list: list of Lie-algebra generators (d, d) -> equivariance constraint
        kwargs: threshold, device, constrain_constant (required with L_list), lstsq_driver ('gels' | 'gelsy';
                default: what torch.linalg.lstsq uses on that device -- gels on the GPU, gelsy on the CPU)
    """

    def __init__(self, latent_dim, poly_order, include_sine, include_exp, L_list=[], **kwargs):
        super().__init__()
        self.latent_dim = latent_dim
        self.poly_order = poly_order
        self.constraint = (len(L_list) != 0)
        self.include_sine = include_sine and not self.constraint        # sindy.py:47
        self.include_exp = include_exp and not self.constraint          # sindy.py:48
        self.L_list = L_list
        self.threshold = kwargs["threshold"]
        device = kwargs["device"]
        # torch.linalg.lstsq's default driver depends on where the data lives: ?gels (full rank) on a GPU -- what the
        # reference computes with its default --gpu 0 -- and the rank-truncating ?gelsy on the CPU (rcond = eps_fp32 *
        # max(m, n) = 1.5e-2 at m = 125 010 rows: any library with cond > ~67 is solved rank-deficient there).  This
        # engine is GPU-only, so 'gels' is the default on a HIP device; 'gelsy' reproduces the CPU-made golden vectors.
        self.lstsq_driver = kwargs.get("lstsq_driver") or ("gels" if torch.device(device).type == "cuda" else "gelsy")
        self.engine = kwargs.get("engine") or get_engine()
        self.flags = library_flags(self.include_sine, self.include_exp)
        n_terms = self.engine.lib_size(latent_dim, poly_order, self.flags)   # raises if not compiled in
        assert n_terms == library.term_count(latent_dim, poly_order, self.include_sine, self.include_exp)

        if self.constraint:
            print('Computing equivariance constraint...')
            self.Q = self.get_Q().to(device)
            self.beta = nn.Parameter(torch.randn((self.Q.shape[1]), device=device))
            self.const = nn.Parameter(torch.randn((latent_dim, 1), device=device))
            self.allow_constant = not kwargs['constrain_constant']
            self.Xi = self.get_Xi()
        else:
            self.Xi = nn.Parameter(torch.randn(self.latent_dim, n_terms, device=device))
        self.mask = torch.ones_like(self.Xi, device=device)
        self._gram_cache = None
        self._near = []                   # one dict per near-threshold coefficient met at a thresholding event
        self._near_wide = []              # ... recorded within NEAR_THRESHOLD_RECORD_BAND (see near_threshold_within)
        self._near_pending = []           # device-side events not yet looked at (no host sync inside the solve loops)

    # ------------------------------------------------------------------ evaluation
    def _coef(self):
        self.Xi = self.get_Xi() if self.constraint else self.Xi
        return self.Xi

    def forward(self, x):
        """Theta(x) @ (Xi * mask)^T                                      (sindy.py:79-82)"""
        xi = self._coef()
        lead = x.shape[:-1]
        out = _Forward.apply(x.reshape(-1, self.latent_dim), xi, self.mask, self)
        return out.reshape(*lead, self.latent_dim)

    def mse_loss(self, x, dx, inv_count=None):
        """Fused replacement for ``MSELoss()(regressor(x), dx)`` (train.py:663-664)."""
        xi = self._coef()
        return _FusedMSE.apply(xi, self.mask, x.reshape(-1, self.latent_dim), dx.reshape(-1, self.latent_dim), self,
                               inv_count)

    def eval_Theta_at(self, x):
        """Function library without coefficients                          (sindy.py:201-203)"""
        return self.engine.theta(x, self.poly_order, self.flags)

    def jvp(self, x, v):
        """J_regressor(x) . v  =  (J_Theta(x) v) (Xi*mask)^T -- analytic, no double backward."""
        xi = self._coef()
        return self.engine.forward_jvp(x, v, xi.detach(), self.mask, self.poly_order, self.flags)[1]

    def forward_and_jvp(self, x, v):
        """(regressor(x), J_regressor(x) v), differentiable once w.r.t. x, v and the parameters
        (HIP forward-mode kernel + its hand-written reverse, second-order term included)."""
        xi = self._coef()
        lead = x.shape[:-1]
        out, jv = _ForwardJvp.apply(x.reshape(-1, self.latent_dim), v.reshape(-1, self.latent_dim), xi, self.mask, self)
        return out.reshape(*lead, self.latent_dim), jv.reshape(*lead, self.latent_dim)

    # ------------------------------------------------------------------ constraint
    def get_M_list(self):
        return [constraint_M(L, self.latent_dim, self.poly_order) for L in self.L_list]

    def get_Q(self):
        Q, self.use_kron_product = constraint_Q(self.L_list, self.latent_dim, self.poly_order)
        return Q

    def update_Q(self, new_Li):                                           # sindy.py:117-120
        self.L_list = new_Li
        self.Q = self.get_Q().to(self.Xi.device)
        self.beta = nn.Parameter(torch.randn((self.Q.shape[1]), device=self.Xi.device))
        self._gram_cache = None

    def get_Xi(self):                                                     # sindy.py:169-176
        if not self.constraint:
            return self.Xi
        if self.use_kron_product:
            Xi = (self.Q @ self.beta).view(self.latent_dim, -1)
        else:
            Xi = (self.Q @ self.beta).view(-1, self.latent_dim).transpose(0, 1)
        if self.allow_constant:
            Xi = Xi + torch.cat([self.const, torch.zeros((Xi.shape[0], Xi.shape[1] - 1), device=Xi.device)], dim=1)
        return Xi

    def get_term_num(self):                                               # sindy.py:179-189
        return library.term_count(self.latent_dim, self.poly_order, self.include_sine, self.include_exp)

    # ------------------------------------------------------------------ sparsity
    def note_near_threshold(self, xi, mask, threshold, where=''):
        """Record the coefficients within NEAR_THRESHOLD_BAND of ``threshold``.  Device tensors are only snapshotted here
        (two tiny copies, no synchronisation: the latent solve runs this five times per batch) and examined when
        ``near_threshold`` is read."""
        if torch.is_tensor(xi) and xi.is_cuda:
            self._near_pending.append((xi.detach().clone(), mask.detach().clone(), float(threshold), where))
            if len(self._near_pending) > 256:
                self._flush_near()
            return
        xi = xi.detach().cpu().numpy() if torch.is_tensor(xi) else xi
        mask = mask.detach().cpu().numpy() if torch.is_tensor(mask) else mask
        for i, k, v in near_threshold_cases(xi, mask, threshold, band=NEAR_THRESHOLD_RECORD_BAND):
            self._near_wide.append({'where': where, 'threshold': float(threshold), 'index': (i, k), 'abs_coef': v})

    def _flush_near(self):
        pending, self._near_pending = self._near_pending, []
        for xi, mask, threshold, where in pending:
            self.note_near_threshold(xi.cpu(), mask.cpu(), threshold, where)

    @property
    def near_threshold(self):
        """[{'where', 'threshold', 'index': (row, col), 'abs_coef'}] for every coefficient that sat within 1e-4 of the
        threshold at a thresholding event."""
        return self.near_threshold_within(NEAR_THRESHOLD_BAND)

    def near_threshold_within(self, band):
        """The same record for a wider band (at most NEAR_THRESHOLD_RECORD_BAND = 1e-3, the L-BFGS trainer's stopping ball)."""
        self._flush_near()
        return [e for e in self._near + self._near_wide if abs(e['abs_coef'] - e['threshold']) < band]

    def set_threshold(self, threshold):                                   # sindy.py:192-194 (strict >)
        self.Xi = self.get_Xi() if self.constraint else self.Xi
        self.note_near_threshold(self.Xi, self.mask, threshold, 'set_threshold')
        self.mask.data = torch.logical_and(torch.abs(self.Xi) > threshold, self.mask).float()

    def reset_mask(self):                                                 # sindy.py:197-198
        self.mask.data = torch.ones_like(self.Xi, device=self.Xi.device)

    # ------------------------------------------------------------------ reporting
    def print(self):                                                      # sindy.py:206-247
        Xi = (self.get_Xi() if self.constraint else self.Xi).detach().cpu()
        mask = self.mask.cpu()
        names = library.term_names(self.latent_dim, self.poly_order, self.include_sine, self.include_exp)
        for i in range(self.latent_dim):
            equation = f'dz{i} ='
            for k, name in enumerate(names):
                if mask[i, k]:
                    equation += f' {Xi[i, k]:.3f}' + (f'*{name}' if name else '') + ' +'
            print(equation)

    # ------------------------------------------------------------------ Gram access
    def aug_gram(self, x, y):
        """fp64 [Theta | y]^T [Theta | y] on the host, cached while the SAME tensor objects (x, y) are unchanged
        (identity through weak references + version counters: an address can be recycled by the allocator, e.g. the
        encoder output of the next batch, a live object cannot)."""
        c = self._gram_cache
        hit = (c is not None and c[0]() is x and c[1]() is y and c[2] == (x._version, y._version, self.poly_order, self.flags))
        if not hit:
            G = self.engine.aug_gram(x.detach().reshape(-1, self.latent_dim), y.detach().reshape(-1, self.latent_dim),
                                     self.poly_order, self.flags)
            c = (weakref.ref(x), weakref.ref(y), (x._version, y._version, self.poly_order, self.flags), G.cpu().numpy())
            self._gram_cache = c
        return c[3]


def _normal_system(regressor, G, gamma):
    """Normal equations of the reference's (possibly block-diagonal, column-selected,
    Q-projected) system, from the augmented Gram.  Returns (Gn, Cn, yy, m_rows, info)."""
    d, p = regressor.latent_dim, G.shape[0] - regressor.latent_dim
    Gtt = G[:p, :p] + (gamma * gamma) * np.eye(p)          # [Theta; gamma I]^T [Theta; gamma I]  (sindy.py:262-263)
    Gty = G[:p, p:]                                         # Theta^T y
    yy = np.trace(G[p:, p:])
    return Gtt, Gty, yy, d, p


def stlsq_solve_from_gram(G, N, mask, gamma, d, driver="gelsy"):
    """Unconstrained ridge least squares on the support ``mask`` (d, p) from the augmented Gram G.

    Returns (Xi (d, p) float64, residual).  Full mask: one (p, p) system with d right-hand sides
    (sindy.py:288, 300); otherwise the reference's block-diagonal, column-selected system in
    equation-major order (sindy.py:270-274, 296-298) -- solved as ONE system because gelsy's
    rank decision is taken over all equations jointly.
    """
    p = G.shape[0] - d
    Gtt = G[:p, :p] + (gamma * gamma) * np.eye(p)
    Gty = G[:p, p:]
    yy = np.trace(G[p:, p:])
    if mask.all():
        W, _ = lstsq_normal(Gtt, Gty, N + p, driver)
        res = np.mean([G[p + j, p + j] - 2 * W[:, j] @ Gty[:, j] + W[:, j] @ Gtt @ W[:, j] for j in range(d)])
        return W.T.copy(), res
    flat = mask.reshape(-1)
    Gb = np.zeros((d * p, d * p))
    for j in range(d):
        Gb[j * p:(j + 1) * p, j * p:(j + 1) * p] = Gtt
    cb = Gty.T.reshape(-1)
    Gm, cm = Gb[flat][:, flat], cb[flat]
    w, _ = lstsq_normal(Gm, cm, d * (N + p), driver)
    Xi = np.zeros((d, p))
    Xi[mask] = w
    return Xi, yy - 2 * w @ cm + w @ Gm @ w


def _mask_on_host(regressor):
    """Boolean host copy of the mask without a device round trip when the mask on the device is still the tensor this
    module uploaded last (the mirror keeps that tensor alive, so an equal address means the same storage; in-place
    edits bump the version counter)."""
    m = regressor.mask
    c = getattr(regressor, '_mask_mirror', None)
    if c is not None and c[0].data_ptr() == m.data_ptr() and c[1] == m._version and c[0].shape == m.shape:
        return c[2]
    host = (m > 0.0).cpu().numpy()
    regressor._mask_mirror = None
    return host


def _upload_mask(regressor, new_mask, t=None):
    if t is None:
        t = torch.from_numpy(new_mask.astype(np.float32)).to(regressor.mask.device)
    regressor.mask.data = t
    regressor._mask_mirror = (t, regressor.mask._version, new_mask.copy())


def _upload_packed(dev, *arrays):
    """Several small host arrays to the device in ONE copy; returns fp32 device views of their shapes."""
    flat = np.concatenate([np.asarray(a, dtype=np.float32).reshape(-1) for a in arrays])
    buf = torch.from_numpy(flat).to(dev)
    out, off = [], 0
    for a in arrays:
        n = int(np.asarray(a).size)
        out.append(buf[off:off + n].view(np.asarray(a).shape))
        off += n
    return out


def _constraint_on_host(regressor):
    """fp64 host copy of Q (d*p, r), refreshed when the regressor's Q tensor changes (update_Q)."""
    c = getattr(regressor, '_q_mirror', None)
    if c is None or c[0] is not regressor.Q or c[1] != regressor.Q._version:
        c = (regressor.Q, regressor.Q._version, regressor.Q.detach().cpu().double().numpy())
        regressor._q_mirror = c
    return c[2]


def solve_SINDy_one_step(regressor, x, y, w_sindy_reg, st_threshold, **kwargs):
    '''
    Solve  argmin_w ||y - w Theta(x)||^2 + w_sindy_reg^2 ||w||^2  on the current support, then
    threshold (reference sindy.py:250-315; note the ridge enters as gamma*I rows => gamma^2).

    Same signature, same in-place updates of Xi / beta / const / mask, same return
    ``(residual, converged)``.  ``residual`` is the squared residual of the solved system
    averaged as the reference does (``lm.residuals.mean() / N``) -- the reference's CPU path
    returns NaN there (empty ``residuals`` with the gelsy driver), its GPU path this number.
    '''
    N = x.reshape(-1, regressor.latent_dim).shape[0]
    G = regressor.aug_gram(x, y)
    Gtt, Gty, yy, d, p = _normal_system(regressor, G, float(w_sindy_reg))
    driver = kwargs.get("lstsq_driver", regressor.lstsq_driver)
    mask = _mask_on_host(regressor)
    dev = regressor.mask.device
    needs_grad = torch.is_grad_enabled() and (x.requires_grad or y.requires_grad)
    prev_mask = regressor.mask.clone() if needs_grad else None

    uploads = {}                                                            # name -> host array: one packed copy at the end
    if mask.all() and not regressor.constraint:
        W, _ = lstsq_normal(Gtt, Gty, N + p, driver)                       # (p, d); sindy.py:288, 300
        res = np.array([G[p + j, p + j] - 2 * W[:, j] @ Gty[:, j] + W[:, j] @ Gtt @ W[:, j] for j in range(d)])
        xi_host = W.T.astype(np.float32)                                    # what .float() makes of it on the device
        uploads['Xi'] = xi_host
        residual = res.mean()
    else:
        # block-diagonal system over all equations, equation-major flattening (sindy.py:270-274)
        flat = mask.reshape(-1)
        Gb = np.zeros((d * p, d * p))
        for j in range(d):
            Gb[j * p:(j + 1) * p, j * p:(j + 1) * p] = Gtt
        cb = Gty.T.reshape(-1)                                              # index j*p + k
        Gm, cm = Gb[flat][:, flat], cb[flat]
        m_rows = d * (N + p)
        if not regressor.constraint:
            w, _ = lstsq_normal(Gm, cm, m_rows, driver)
            new_coef = np.zeros((d, p))
            new_coef[mask] = w                                              # sindy.py:296-298
            xi_host = new_coef.astype(np.float32)
            uploads['Xi'] = xi_host
            residual = yy - 2 * w @ cm + w @ Gm @ w
        else:
            Q = _constraint_on_host(regressor)
            if regressor.allow_constant:                                    # sindy.py:277-280
                Q = np.concatenate([Q, np.zeros((Q.shape[0], d))], axis=1)
                for i in range(d):
                    Q[i * Q.shape[0] // d, Q.shape[1] - d + i] = 1.0
            Qm = Q[flat]                                                    # sindy.py:282 (equation-major rows)
            # the reference drops columns of A @ Q[mask] that are exactly zero (sindy.py:284);
            # with a full-column-rank masked library that is a zero column of Q[mask]
            effective = np.any(Qm != 0.0, axis=0)
            Qe = Qm[:, effective]
            Gq, cq = Qe.T @ Gm @ Qe, Qe.T @ cm
            b, _ = lstsq_normal(Gq, cq, m_rows, driver)
            full = np.zeros(Q.shape[1])
            full[effective] = b
            if not regressor.allow_constant:                                # sindy.py:302-305
                uploads['beta'] = full.astype(np.float32)
            else:                                                           # sindy.py:307-311
                uploads['beta'] = full[:-d].astype(np.float32)
                uploads['const'] = full[-d:].astype(np.float32).reshape(-1, 1)
            residual = yy - 2 * b @ cq + b @ Gq @ b
            # get_Xi of THIS solution on the host, in fp32 like the device product (sindy.py:169-176)
            beta_h = (full[:-d] if regressor.allow_constant else full).astype(np.float32)
            flat_xi = _constraint_on_host(regressor).astype(np.float32) @ beta_h
            xi_host = flat_xi.reshape(d, p).copy() if regressor.use_kron_product else flat_xi.reshape(p, d).T.copy()
            if regressor.allow_constant:
                xi_host[:, 0] += full[-d:].astype(np.float32)
    # The solution was made on the host: threshold it there too (same fp32 values, same strict >, sindy.py:192-194, 312),
    # then ONE packed copy carries coefficients, mask and residual up -- no device launches, no synchronising allclose
    # (sindy.py:313), no mask download on the next pass.
    regressor.note_near_threshold(xi_host, mask, st_threshold, 'solve_SINDy_one_step')
    new_mask = np.logical_and(np.abs(xi_host) > np.float32(st_threshold), mask)
    converged = bool(np.array_equal(new_mask, mask))
    names = list(uploads)
    parts = _upload_packed(dev, *[uploads[k] for k in names], new_mask.astype(np.float32), np.float32(residual / N))
    for k, t_dev in zip(names, parts):
        getattr(regressor, k).data = t_dev
    xi_sol = None
    if needs_grad:       # coefficients of THIS solve on the support it was solved on, before the new threshold applies
        xi_sol = ((regressor.get_Xi() if regressor.constraint else regressor.Xi).detach() * prev_mask).contiguous()
    _upload_mask(regressor, new_mask, parts[len(names)])
    if regressor.constraint:
        regressor.Xi = regressor.get_Xi()                                   # as set_threshold leaves it (sindy.py:193)
    value = parts[-1].reshape(())
    if needs_grad:
        # lm.residuals is per right-hand side: d columns for the full-mask solve, one for the flattened system
        per_col = mask.all() and not regressor.constraint
        scale = 1.0 / (N * (d if per_col else 1))
        value = _LstsqResidual.apply(x, y, value, xi_sol, scale, regressor)
    return value, converged


def solve_SINDy(regressor, x, y, w_sindy_reg, st_threshold, max_iter=5, **kwargs):
    """reference sindy.py:318-324"""
    regressor.reset_mask()
    residual = None
    for _ in range(max_iter):
        residual, converged = solve_SINDy_one_step(regressor, x, y, w_sindy_reg, st_threshold, **kwargs)
        if converged:
            break
    return residual


class WSINDyWrapper():
    """
    Weak SINDy as a regularised least-squares problem (reference sindy.py:327-395): trigonometric
    test functions g_k(t) = sqrt(2/T) sin(k pi t / T), V = dt g, V' = dt g'; G = V Theta(x), b = -V' x;
    solve  min || [V^T G; sqrt(gamma) I] w - [V^T b; 0] ||  on the current support, then threshold.

    The data pass is ONE fused launch (symode_weak_gram: Theta built in registers, G = V Theta(x) and b = -V' x
    accumulated by the fp64 matrix cores, Theta never written).  The reference's system  [V^T G; sqrt(gamma) I] w =
    [V^T b; 0]  has the normal equations  G^T (V V^T) G + gamma I  and  G^T (V V^T) b  -- the (K, K) matrix V V^T does not
    depend on the data and is formed once -- which the host solves with the same LAPACK semantics as solve_SINDy_one_step.
    """

    def __init__(self, regressor, t, t_max, num_test_funcs=50, test_func_family='trig', device='cuda', **kwargs):
        self.t = t.to(device)
        self.dt = self.t[1] - self.t[0]
        self.regressor = regressor
        if test_func_family != 'trig':
            raise NotImplementedError(f'test_func_family={test_func_family} not implemented')
        k = torch.arange(1, num_test_funcs + 1, dtype=torch.float32, device=device).view(-1, 1)
        amp = (2 / t_max) ** 0.5
        g = amp * torch.sin(k * torch.pi * self.t / t_max)
        g_drv = amp * k * np.pi / t_max * torch.cos(k * np.pi * self.t / t_max)
        self.V = (self.dt * g).contiguous()                                       # sindy.py:346-347
        self.V_drv = (self.dt * g_drv).contiguous()
        Vh = self.V.double().cpu().numpy()
        self._VVt = Vh @ Vh.T                                                     # (K, K) fp64, data independent

    def solve(self, x, w_sindy_reg, st_threshold, **kwargs):
        reg = self.regressor
        d = reg.latent_dim
        with torch.no_grad():
            if hasattr(reg.engine, 'weak_gram'):
                G, b = reg.engine.weak_gram(x.reshape(-1, d), self.V, self.V_drv, reg.poly_order, reg.flags)   # ONE fused launch
                G, b = G.cpu().numpy(), b.cpu().numpy()                           # (K, p), (K, d) fp64
            else:                                                                 # engines without the fused entry (test double)
                G = (self.V.double() @ reg.eval_Theta_at(x).double()).cpu().numpy()
                b = (-self.V_drv.double() @ x.double()).cpu().numpy()
            p = G.shape[1]
            MG = self._VVt @ G                                                    # A^T A = G^T (V V^T) G with A = V^T G
            Gn = G.T @ MG + float(w_sindy_reg) * np.eye(p)
            Cn = MG.T @ b                                                         # (p, d) = A^T (V^T b)
            bb = float(np.sum(b * (self._VVt @ b)))
            m_rows = self.V.shape[1] + p
            mask = (reg.mask > 0.0).cpu().numpy()
            driver = kwargs.get('lstsq_driver', reg.lstsq_driver)
            prev_mask = reg.mask.clone()
            if mask.all():
                W, _ = lstsq_normal(Gn, Cn, m_rows, driver)
                reg.Xi.data = torch.from_numpy(W.T.copy()).float().to(reg.mask.device)
                residual = np.mean([Cn[:, j] @ W[:, j] * -2 + W[:, j] @ Gn @ W[:, j] for j in range(d)]) + bb / d
            else:
                flat = mask.reshape(-1)
                Gb = np.zeros((d * p, d * p))
                for j in range(d):
                    Gb[j * p:(j + 1) * p, j * p:(j + 1) * p] = Gn
                cb = Cn.T.reshape(-1)
                Gm, cm = Gb[flat][:, flat], cb[flat]
                w, _ = lstsq_normal(Gm, cm, d * m_rows, driver)
                new_coef = np.zeros((d, p))
                new_coef[mask] = w
                reg.Xi.data = torch.from_numpy(new_coef).float().to(reg.mask.device)
                residual = bb - 2 * w @ cm + w @ Gm @ w
            reg.set_threshold(st_threshold)
            converged = torch.allclose(prev_mask, reg.mask)
        return float(residual), converged
