"""CPU oracle: a plain torch-CPU / numpy restatement of the reference's SINDy hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the reference
file:line it follows (paths relative to the reference checkout).  The restatement is
op-for-op faithful where rounding matters (cat-of-products library, ``Theta @ (Xi*mask).T``,
MSE + autograd backward, ``torch.linalg.lstsq`` on the ridge-augmented / block-diagonal
system, strict ``>`` thresholding), so that it can serve both as the parity checker and as
the timed CPU baseline ("port") in bench.py.

Pinning: tests/test_oracle_golden.py checks every function here against
tests/golden/*.npz, which tools/gen_golden.py produced in the build container by importing
the reference itself (read-only, with a stub ``wandb``).  Library orders 4-5 are an
extension of the reference's ordering (it stops at cubic, sindy.py:37) and are therefore
"parity unpinned" beyond the recurrence they share with orders 2-3.
"""
from __future__ import annotations

import itertools
import math
from functools import partial

import numpy as np
import torch
from torch.autograd.functional import jvp as _torch_jvp

# --------------------------------------------------------------------------------------
# Library Theta(x)                                              ref: sindy.py:7-30, 68-82
# --------------------------------------------------------------------------------------


def poly_tuples(d: int, n: int):
    """Index tuples i1<=i2<=...<=in in the reference's nested-loop (lexicographic) order.

    ref: sindy.py:14-17 (i<=j), sindy.py:20-24 (i<=j<=k); orders 4-5 continue the nesting.
    """
    return list(itertools.combinations_with_replacement(range(d), n))


def term_count(d: int, order: int, include_sine: bool = False, include_exp: bool = False) -> int:
    """ref: sindy.py:179-189 (closed forms up to cubic; C(d+n-1, n) generalises them)."""
    p = 1 + d
    for n in range(2, order + 1):
        p += math.comb(d + n - 1, n)
    if include_sine:
        p += d
    if include_exp:
        p += d
    return p


def term_exponents(d: int, order: int):
    """Exponent vector of every polynomial column, in library order (constant first)."""
    exps = [tuple([0] * d)]
    for n in range(1, order + 1):
        for tup in poly_tuples(d, n):
            e = [0] * d
            for i in tup:
                e[i] += 1
            exps.append(tuple(e))
    return exps


def theta(x: torch.Tensor, order: int, include_sine: bool = False, include_exp: bool = False) -> torch.Tensor:
    """Theta(x): columns [1 | x | x_i x_j (i<=j) | (x_i x_j) x_k (i<=j<=k) | ... | sin x | exp x].

    ref: sindy.py:7-30 (term functions), sindy.py:201-203 (eval_Theta_at).  Products are
    evaluated left to right, ``(x_i * x_j) * x_k`` (sindy.py:20), one fp32 rounding per
    multiply, which is what makes the polynomial columns bit-reproducible.
    """
    d = x.shape[-1]
    blocks = [torch.ones(*x.shape[:-1], 1, dtype=x.dtype, device=x.device), x]     # sindy.py:7-11 (ones on x's device)
    prev = {(i,): x[..., i] for i in range(d)}
    for n in range(2, order + 1):
        cur, cols = {}, []
        for tup in poly_tuples(d, n):
            v = prev[tup[:-1]] * x[..., tup[-1]]                      # sindy.py:14, 20
            cur[tup] = v
            cols.append(v.reshape(*x.shape[:-1], 1))
        blocks.append(torch.cat(cols, dim=-1))
        prev = cur
    if include_sine:
        blocks.append(torch.sin(x))                                   # sindy.py:26-27
    if include_exp:
        blocks.append(torch.exp(x))                                   # sindy.py:29-30
    return torch.cat(blocks, dim=-1)                                  # sindy.py:81, 202


def forward(x, Xi, mask, order, include_sine=False, include_exp=False):
    """dx_hat = Theta(x) @ (Xi * mask)^T                              ref: sindy.py:79-82"""
    return theta(x, order, include_sine, include_exp) @ (Xi * mask).T


def mse_loss_and_grad(x, dx, Xi, mask, order, include_sine=False, include_exp=False):
    """MSE closure body: loss = mean((Theta Xi_m^T - dx)^2), grad = dloss/dXi via autograd.

    ref: train.py:663-664 (forward + MSELoss), train.py:689 (backward).
    """
    Xi = Xi.detach().clone().requires_grad_(True)
    pred = forward(x, Xi, mask, order, include_sine, include_exp)
    loss = torch.nn.functional.mse_loss(pred, dx)
    loss.backward()
    return loss.detach(), Xi.grad.detach()


# --------------------------------------------------------------------------------------
# Equivariance constraint                                        ref: sindy.py:85-176
# --------------------------------------------------------------------------------------


def constraint_M(L: torch.Tensor, d: int, order: int) -> torch.Tensor:
    """M with  J_Theta(z) . (L z) = M . Theta(z)  for the polynomial library.

    ref: sindy.py:123-144 builds M symbolically (sympy Jacobian, expand, coeff).  The same
    matrix follows in closed form from the product rule: for Theta_t = prod z_i^{a_i},
        sum_m dTheta_t/dz_m (L z)_m = sum_{m,n} a_m L[m,n] z^{a - e_m + e_n},
    so M[t, idx(a - e_m + e_n)] += a_m L[m,n].  Stored as float32 like the reference's
    ``torch.zeros(p, p)`` (sindy.py:134).
    """
    exps = term_exponents(d, order)
    index = {e: t for t, e in enumerate(exps)}
    p = len(exps)
    M = np.zeros((p, p), dtype=np.float64)
    Ld = L.detach().cpu().double().numpy()
    for t, a in enumerate(exps):
        for m in range(d):
            if a[m] == 0:
                continue
            for n in range(d):
                b = list(a)
                b[m] -= 1
                b[n] += 1
                M[t, index[tuple(b)]] += a[m] * Ld[m, n]
    return torch.from_numpy(M).float()


def constraint_Q(L_list, d: int, order: int):
    """Null-space basis Q of the stacked constraint matrix, plus the branch flag.

    ref: sindy.py:85-115.  ``det(L) < 1e-5`` has no abs (sindy.py:90) and the flag is the
    one left by the *last* generator in the list (sindy.py:91, 95); both kept.
    Returns (Q, use_kron_product).
    """
    C_list, use_kron = [], None
    for L in L_list:
        M = constraint_M(L, d, order)
        if torch.det(L) < 1e-5:                                                # sindy.py:90
            use_kron = False
            MT = M.transpose(0, 1)
            C = torch.kron(-MT.contiguous(), torch.eye(L.shape[0])) + torch.kron(torch.eye(MT.shape[0]), L)
        else:
            use_kron = True
            C = torch.kron(L.inverse(), M.T)                                    # sindy.py:96
            C = C - torch.eye(C.shape[0])
        C_list.append(C)
    C_total = torch.cat(C_list, dim=0)
    U, Sigma, V = torch.svd(C_total)                                            # sindy.py:100
    r = 0
    for r in range(len(Sigma)):                                                 # sindy.py:102-104
        if abs(Sigma[-1 - r]) > 5e-3:
            break
    Q = V[:, -r:]                                                               # sindy.py:106
    return Q, use_kron


def xi_from_beta(Q, beta, const, d: int, use_kron: bool, allow_constant: bool):
    """Xi = reshape(Q beta) (+ const in column 0).                 ref: sindy.py:169-176"""
    if use_kron:
        Xi = (Q @ beta).view(d, -1)
    else:
        Xi = (Q @ beta).view(-1, d).transpose(0, 1)
    if allow_constant:
        Xi = Xi + torch.cat([const, torch.zeros((Xi.shape[0], Xi.shape[1] - 1))], dim=1)
    return Xi


class OracleRegressor:
    """State holder mirroring ``SINDyRegression`` (sindy.py:33-77) without nn.Module.

    Parameters are plain leaf tensors: ``Xi`` (d,p) when unconstrained, ``beta`` (r,) and
    ``const`` (d,1) under the equivariance constraint.  ``mask`` is a 0/1 float tensor.
    Initial values are injected (the reference draws them from the global RNG,
    sindy.py:58-59, 64; parity runs capture them from the oracle run instead).
    """

    def __init__(self, latent_dim, poly_order, include_sine=False, include_exp=False, L_list=(),
                 threshold=0.1, constrain_constant=False, Xi0=None, beta0=None, const0=None):
        self.latent_dim, self.poly_order = latent_dim, poly_order
        self.constraint = len(L_list) != 0
        self.include_sine = include_sine and not self.constraint            # sindy.py:47
        self.include_exp = include_exp and not self.constraint              # sindy.py:48
        self.L_list = list(L_list)
        self.threshold = threshold
        p = term_count(latent_dim, poly_order, self.include_sine, self.include_exp)
        if self.constraint:
            self.Q, self.use_kron_product = constraint_Q(self.L_list, latent_dim, poly_order)
            r = self.Q.shape[1]
            self.beta = (torch.randn(r) if beta0 is None else beta0.clone().float()).requires_grad_(True)
            self.const = (torch.randn(latent_dim, 1) if const0 is None else const0.clone().float()).requires_grad_(True)
            self.allow_constant = not constrain_constant                     # sindy.py:60
            self.Xi = None
        else:
            self.Xi = (torch.randn(latent_dim, p) if Xi0 is None else Xi0.clone().float()).requires_grad_(True)
        self.mask = torch.ones(latent_dim, p)                                # sindy.py:66
        self.near_threshold = []                                             # see set_threshold

    # -- parameter plumbing --------------------------------------------------------
    def parameters(self):
        return [self.beta, self.const] if self.constraint else [self.Xi]

    def get_Xi(self):
        if not self.constraint:
            return self.Xi
        return xi_from_beta(self.Q, self.beta, self.const, self.latent_dim,
                            self.use_kron_product, self.allow_constant)

    def theta(self, x):
        return theta(x, self.poly_order, self.include_sine, self.include_exp)

    def __call__(self, x):                                                   # sindy.py:79-82
        return self.theta(x) @ (self.get_Xi() * self.mask).T

    def set_threshold(self, thr):                                            # sindy.py:192-194
        with torch.no_grad():
            a = torch.abs(self.get_Xi())
            # live coefficients within 1e-4 of the threshold at a thresholding event, as (row, column)
            self.near_threshold += [(int(i), int(k)) for i, k in
                                    torch.nonzero(((a - thr).abs() < 1e-4) & (self.mask > 0)).tolist()]
            self.mask = torch.logical_and(a > thr, self.mask).float()

    def reset_mask(self):                                                    # sindy.py:197-198
        self.mask = torch.ones_like(self.mask)


# --------------------------------------------------------------------------------------
# Sequential-threshold least squares                              ref: sindy.py:250-324
# --------------------------------------------------------------------------------------


class _LstsqResult:
    def __init__(self, solution, residuals, rank):
        self.solution, self.residuals, self.rank = solution, residuals, rank


def lstsq_cpu(A: torch.Tensor, B: torch.Tensor) -> _LstsqResult:
    """``torch.linalg.lstsq(A, B)`` as the reference's CPU path defines it (sindy.py:288):
    LAPACK ?gelsy with torch's default ``rcond = eps(dtype) * max(m, n)``, empty ``residuals``.

    The LAPACK routine is reached through scipy instead of torch because torch 2.10's CPU wrapper
    is not reproducible on ill-conditioned systems (identical calls return different ranks --
    measured in the build container); scipy initialises the pivot array, so
    this is what the reference computes whenever its own call behaves.
    """
    import scipy.linalg as sl
    a = A.detach().numpy()
    b = B.detach().numpy()
    rcond = float(np.finfo(a.dtype).eps) * max(a.shape)
    if a.shape[1] == 0:
        return _LstsqResult(torch.zeros((0,) + tuple(b.shape[1:]), dtype=A.dtype), torch.empty(0, dtype=A.dtype), 0)
    sol, _, rank, _ = sl.lstsq(a, b, cond=rcond, lapack_driver="gelsy", check_finite=False)
    return _LstsqResult(torch.from_numpy(np.ascontiguousarray(sol)).to(A.dtype), torch.empty(0, dtype=A.dtype), rank)


def stlsq_one_step(reg: OracleRegressor, x, y, w_sindy_reg, st_threshold, lstsq=None):
    """One ridge-augmented least-squares solve + hard threshold.   ref: sindy.py:250-315

    Returns (residual, converged, solution).  ``residual`` follows the reference literally
    (``lm.residuals.mean() / N``), which is NaN on CPU where lstsq returns no residuals.
    ``lstsq``: the solver called at sindy.py:288 -- default ``lstsq_cpu`` (see there); bench.py's timed CPU baseline
    passes ``torch.linalg.lstsq``, the reference's own call.
    """
    lstsq = lstsq or lstsq_cpu
    theta_x = reg.theta(x)                                                               # :261
    p = theta_x.shape[1]
    A = torch.cat([theta_x, w_sindy_reg * torch.eye(p)], dim=0)                          # :262-263
    B = torch.cat([y, torch.zeros(p, y.shape[1])], dim=0)                                # :264
    mask = reg.mask > 0.0                                                                # :267-268
    effective = None
    if (not torch.all(mask)) or reg.constraint:                                          # :269
        A = torch.block_diag(*([A] * y.shape[-1]))                                       # :270-272
        A = A[:, mask.flatten()]                                                         # :273
        B = B.transpose(0, 1).reshape(-1)                                                # :274
        if reg.constraint:
            Q = reg.Q
            if reg.allow_constant:                                                       # :277-280
                Q = torch.cat([Q, torch.zeros((Q.shape[0], reg.latent_dim))], dim=1)
                for i in range(reg.latent_dim):
                    Q[i * Q.shape[0] // reg.latent_dim, Q.shape[1] - reg.latent_dim + i] = 1.0
            A = A @ Q[mask.flatten()]                                                    # :282
            effective = torch.any(A != 0.0, dim=0)                                       # :284
            A = A[:, effective]
    lm = lstsq(A, B)                                                                     # :288
    sol = lm.solution
    prev_mask = reg.mask.clone()
    with torch.no_grad():
        if not reg.constraint:
            if not torch.all(mask):                                                      # :295-298
                new = torch.zeros(reg.latent_dim, p)
                new[mask] = sol
                reg.Xi = new.requires_grad_(True)
            else:
                reg.Xi = sol.T.clone().requires_grad_(True)                              # :300
        else:
            if not reg.allow_constant:                                                   # :302-305
                nb = torch.zeros_like(reg.beta)
                nb[effective] = sol
                reg.beta = nb.requires_grad_(True)
            else:                                                                        # :307-311
                ns = torch.zeros(reg.beta.shape[0] + reg.latent_dim)
                ns[effective] = sol
                reg.beta = ns[:-reg.latent_dim].clone().requires_grad_(True)
                reg.const = ns[-reg.latent_dim:].view(-1, 1).clone().requires_grad_(True)
    reg.set_threshold(st_threshold)                                                      # :312
    converged = torch.allclose(prev_mask, reg.mask)                                      # :313
    residual = lm.residuals.mean() / x.shape[0]                                          # :315
    return residual, converged, sol


def stlsq(reg: OracleRegressor, x, y, w_sindy_reg, st_threshold, max_iter=5):
    """ref: sindy.py:318-324"""
    reg.reset_mask()
    residual = None
    for _ in range(max_iter):
        residual, converged, _ = stlsq_one_step(reg, x, y, w_sindy_reg, st_threshold)
        if converged:
            break
    return residual


def stlsq_until_converged(reg: OracleRegressor, x, y, num_epochs, w_sindy_reg, threshold):
    """Loop of one-step solves until the mask stops changing.     ref: train.py:872-887

    Returns the list of (mask, Xi) after every pass.
    """
    history = []
    for _ in range(num_epochs):
        _, done, _ = stlsq_one_step(reg, x, y, w_sindy_reg, threshold)
        history.append((reg.mask.clone(), reg.get_Xi().detach().clone()))
        if done:
            break
    return history


# --------------------------------------------------------------------------------------
# ODE integrator                                                  ref: model_utils.py:223-255
# --------------------------------------------------------------------------------------


def odeint(f, x0, t, dt, method="euler", full_traj=False):
    n_steps = int(t / dt)                                                    # :233
    traj = []
    if method == "euler":
        for _ in range(n_steps):
            x0 = x0 + dt * f(x0)                                             # :238
            traj.append(x0)
    elif method == "rk4":
        for _ in range(n_steps):                                             # :242-247
            k1 = f(x0)
            k2 = f(x0 + dt / 2 * k1)
            k3 = f(x0 + dt / 2 * k2)
            k4 = f(x0 + dt * k3)
            x0 = x0 + dt / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
            traj.append(x0)
    else:
        raise ValueError("Unrecognized ODEInt method.")
    return torch.stack(traj, dim=0) if full_traj else x0


# --------------------------------------------------------------------------------------
# Symmetry regularisers                                           ref: model_utils.py:8-211
# --------------------------------------------------------------------------------------
# The autoencoder halves are passed in as callables (stock PyTorch modules in eval mode);
# ``z_mean`` is the latent offset the reference reads from ``encoder[-2].bias``
# (model_utils.py:46, 101, 151).  ``basis`` / ``group_elems`` are the matrices the
# reference obtains from LieGenerator.get_full_basis_list / get_deterministic_group_elems.


def block_basis(L: torch.Tensor, n_comps: int):
    """Channels of ``L`` (ch,k,k) replicated block-diagonally over ``n_comps`` components.

    ref: gan.py:306-330 for a single representation tuple (split_channel=True).
    """
    ch, k, _ = L.shape
    out = []
    for c in range(ch):
        V = torch.zeros(k * n_comps, k * n_comps)
        for j in range(n_comps):
            V[j * k:(j + 1) * k, j * k:(j + 1) * k] = L[c]
        out.append(V)
    return out


def group_elems(basis_stack: torch.Tensor, sigma: torch.Tensor, scale: float = 1.0):
    """exp(sigma * L * scale) per (sigma, basis) pair.            ref: gan.py:332-348

    With split_channel=False the reference zips ``self.sigma`` (one (ch,ch) matrix per
    representation tuple) with the un-split basis (ch,n,n); for ch == 1 this is
    ``matrix_exp(sigma[0,0] * L[0] * scale)`` broadcast over the leading channel axis.
    """
    out = []
    for Li in basis_stack:
        out.append(torch.matrix_exp(sigma * Li * scale))
    return out


def symreg_linear_latent(z, basis, regressor, dz_pred=None):
    """S1: sum_v || J_h(z) (v z) - v h(z) ||_F^2,  h = regressor.   ref: train.py:502-507

    The shipped line subtracts from the (output, jvp) tuple (TypeError); the intended
    quantity indexes ``[1]`` as model_utils.py:53,56,166 do.
    """
    if dz_pred is None:
        dz_pred = regressor(z)
    loss = 0.0
    for v in basis:
        vz = torch.einsum("ij,bj->bi", v, z)
        jv = _torch_jvp(regressor, z, vz, create_graph=True)[1]
        loss = loss + torch.norm(jv - torch.einsum("ij,bj->bi", v, dz_pred)) ** 2
    return loss


def symreg_infinitesimal(x_fx, encode, decode, z_mean, basis, f, relative=True, require_grad=True):
    """S2.                                                          ref: model_utils.py:8-67"""
    jvp_fn = partial(_torch_jvp, create_graph=True, strict=True) if require_grad else _torch_jvp
    with torch.set_grad_enabled(require_grad):
        loss = 0.0
        z = encode(x_fx) - z_mean                                                # :38, :47
        x = x_fx[:, 0]
        zs = z.shape
        for v in basis:                                                          # :50
            v_z = torch.einsum("jk,...k->...j", v, z.reshape(zs[0], -1)).reshape(zs)
            v_x_fx = jvp_fn(decode, z, v=v_z)[1]                                 # :53 (decodes z - mean)
            v_x, v_fx = v_x_fx[:, 0], v_x_fx[:, 1]
            var = jvp_fn(f, x, v_x)[1]                                           # :56
            if relative:
                loss = loss + torch.mean((var - v_fx) ** 2) / torch.mean(var ** 2)   # :62
            else:
                loss = loss + torch.mean((var - v_fx) ** 2)
    return loss


def symreg_finite(x_fx, encode, decode, z_mean, gelems, f, relative=True, require_grad=True):
    """S3.                                                          ref: model_utils.py:69-124"""
    with torch.set_grad_enabled(require_grad):
        loss = 0.0
        z = encode(x_fx) - z_mean
        fx = x_fx[:, 1]
        zs = z.shape
        for g in gelems:                                                         # :105
            g_z = torch.einsum("jk,...k->...j", g, z.reshape(zs[0], -1)).reshape(zs) + z_mean
            g_x_fx = decode(g_z)                                                 # :109
            g_x, g_fx = g_x_fx[:, 0], g_x_fx[:, 1]
            f_g_x = f(g_x)
            if relative:
                loss = loss + torch.mean((f_g_x - g_fx) ** 2) / torch.mean((f_g_x - fx) ** 2)  # :119
            else:
                loss = loss + torch.mean((f_g_x - g_fx) ** 2)
    return loss


def _group_transform(x, g, encode, decode, z_mean):
    """x -> dec(g (enc([x,x]) - mu) + mu)[:, 0]                    ref: model_utils.py:144-158"""
    xx = torch.stack([x, x], dim=1)
    z = encode(xx) - z_mean
    zs = z.shape
    g_z = torch.einsum("jk,...k->...j", g, z.reshape(zs[0], -1)).reshape(zs) + z_mean
    return decode(g_z)[:, 0]


def symreg_reversed(x, encode, decode, z_mean, gelems, h, require_grad=True):
    """S4: sum_g mean((J_g(x) h(x) - h(g(x)))^2).                   ref: model_utils.py:126-170"""
    jvp_fn = partial(_torch_jvp, create_graph=True, strict=True) if require_grad else _torch_jvp
    with torch.set_grad_enabled(require_grad):
        loss = 0.0
        for g in gelems:
            tr = partial(_group_transform, g=g, encode=encode, decode=decode, z_mean=z_mean)
            gx = tr(x)
            hx = h(x)
            var1 = jvp_fn(tr, x, v=hx)[1]                                        # :166
            var2 = h(gx)
            loss = loss + torch.mean((var1 - var2) ** 2)                         # :168
    return loss


def symreg_reversed_precomputed(x, gx_list, Jgx_list, h):
    """S4 with (g(x), J_g(x)) precomputed once (they do not depend on Xi).

    ref: model_utils.py:172-211 produces the pair for the PySR objective; the loss it
    feeds is model_utils.py:166-168 with the jvp replaced by the explicit matvec.
    """
    loss = 0.0
    hx = h(x)
    for gx, Jg in zip(gx_list, Jgx_list):
        var1 = torch.einsum("bij,bj->bi", Jg, hx)
        loss = loss + torch.mean((var1 - h(gx)) ** 2)
    return loss


def precompute_group_jacobians(x, encode, decode, z_mean, gelems):
    """(g(x), J_g(x)) per group element, J_g (B,d,d).              ref: model_utils.py:172-211

    The reference's vmap(jacfwd) result carries an extra singleton axis (B,1,d,d) from a
    ``dim=1`` stack on 1-D per-sample input (model_utils.py:186); the oracle returns the
    squeezed (B,d,d) Jacobian, computed column by column with jvp.
    """
    gx_list, J_list = [], []
    d = x.shape[-1]
    with torch.no_grad():
        for g in gelems:
            tr = partial(_group_transform, g=g, encode=encode, decode=decode, z_mean=z_mean)
            gx_list.append(tr(x))
            cols = []
            for j in range(d):
                e = torch.zeros_like(x)
                e[:, j] = 1.0
                cols.append(_torch_jvp(tr, x, v=e)[1])
            J_list.append(torch.stack(cols, dim=-1))
    return gx_list, J_list


# --------------------------------------------------------------------------------------
# L-BFGS training loop                                            ref: train.py:617-766
# --------------------------------------------------------------------------------------


def lbfgs_fit(reg: OracleRegressor, x, dx, num_epochs, lr_sindy, w_sindy_x=1.0, sindy_reg_type="l1",
              w_sindy_reg=0.0, w_sym_reg=0.0, sym_loss=None, st_freq=100, threshold=0.1, tol=1e-3):
    """Full-batch L-BFGS fit with convergence-triggered / periodic thresholding.

    ref: train.py:617-766, non-latent branch.  ``sym_loss`` (optional) is a callable
    ``sym_loss(reg, x) -> scalar tensor`` standing for train.py:667-676.
    Returns a history dict: per-epoch loss, events ('conv', 'final', 'freq', 'nan'),
    masks and coefficient snapshots.
    """
    def new_opt():
        return torch.optim.LBFGS(reg.parameters(), lr=lr_sindy)                        # :630, :717

    opt = new_opt()
    losses = {}
    prev = [p.detach().clone() for p in reg.parameters()]                              # :641
    pprev = [p.detach().clone() for p in reg.parameters()]                             # :642
    hist = {"loss": [], "events": [], "mask": [], "Xi": [], "n_closure": 0}

    def closure():                                                                     # :645-690
        opt.zero_grad()
        pred = reg(x)
        loss_x = torch.nn.functional.mse_loss(pred, dx)                                # :664
        losses["loss_sindy_x"] = loss_x.item()
        loss = w_sindy_x * loss_x
        if w_sym_reg > 0.0:
            ls = sym_loss(reg, x)
            losses["loss_sym_reg"] = ls.item()
            loss = loss + w_sym_reg * ls                                               # :679
        if sindy_reg_type == "l1":                                                     # :680-683
            lr_ = sum(torch.norm(p, 1) for p in reg.parameters())
            losses["loss_sindy_reg"] = lr_.item()
            loss = loss + w_sindy_reg * lr_
        elif sindy_reg_type != "none":
            raise ValueError(f"Unknown regularization type: {sindy_reg_type}")
        loss.backward()
        hist["n_closure"] += 1
        return loss

    n_iters = 0
    for epoch in range(num_epochs):                                                    # :693
        n_iters += 1
        opt.step(closure)
        if any(torch.isnan(p).any() for p in reg.parameters()):                        # :697
            hist["events"].append((epoch, "nan"))
            break
        with torch.no_grad():
            upd = sum(torch.norm(p - q) for p, q in zip(reg.parameters(), prev))       # :702
        event = None
        if upd < tol:                                                                  # :705
            upd2 = sum(torch.norm(p - q) for p, q in zip(reg.parameters(), pprev))
            if upd2 < tol:                                                             # :709
                hist["events"].append((epoch, "final"))                                # :710-714 (break before logging)
                break
            n_iters = 0
            reg.set_threshold(threshold)                                               # :716
            opt = new_opt()
            pprev = [p.detach().clone() for p in reg.parameters()]                     # :718
            event = "conv"
        elif st_freq > 0 and n_iters % st_freq == 0:                                   # :720
            n_iters = 0
            reg.set_threshold(threshold)
            opt = new_opt()
            event = "freq"
        prev = [p.detach().clone() for p in reg.parameters()]                          # :725
        if event:
            hist["events"].append((epoch, event))
        hist["loss"].append(dict(losses))
        hist["mask"].append(reg.mask.clone())
        hist["Xi"].append(reg.get_Xi().detach().clone())
    return hist


# --------------------------------------------------------------------------------------
# Evaluation                                                      ref: evaluation/eval_eq.py
# --------------------------------------------------------------------------------------

# Ground-truth coefficient tables (data).                          ref: eval_eq.py:88-105
SINDY_TRUTH = {
    "lv": np.array([[2 / 3, 0, 0, 0, 0, 0, 0, -4 / 3], [-1.0, 0, 0, 0, 0, 0, 1.0, 0]]),
    "selkov": np.array([[0.75, -0.1, 0, 0, 0, 0, 0, 0, -1.0, 0], [0, 0.1, -1.0, 0, 0, 0, 0, 0, 1.0, 0]]),
    "dosc": np.array([[0, -0.1, -1, 0, 0, 0], [0, 1, -0.1, 0, 0, 0]], dtype=float),
    "growth": np.array([[0, -0.3, 0, 0, 0, 0.1], [0, 0, 1.0, 0, 0, 0]]),
}


def eval_coefficients(coef: np.ndarray, mask: np.ndarray, truth: np.ndarray):
    """(coef_masked, correct_form, mse, correct_form_all, mse_all)  ref: eval_eq.py:7-34"""
    mask = mask.astype(bool)
    coef = np.where(mask, coef, 0.0)
    tmask = truth != 0
    n_eqs = coef.shape[0]
    cf = np.zeros(n_eqs)
    mse = np.ones(n_eqs) * -1.0
    for i in range(n_eqs):
        cf[i] = np.all(mask[i] == tmask[i])                                      # :25
        mse[i] = np.mean((coef[i, tmask[i]] - truth[i, tmask[i]]) ** 2)          # :28
    return coef, cf, mse, np.all(cf), np.mean(mse)


# --------------------------------------------------------------------------------------
# Synthetic trajectories                              ref: data_utils/ode.py:7-28 + RHS files
# --------------------------------------------------------------------------------------


def rhs_dosc(x, a=0.1):                                  # ref: data_utils/damped_oscillator.py:20-24
    return np.stack([-a * x[..., 0] - x[..., 1], x[..., 0] - a * x[..., 1]], axis=-1)


def rhs_selkov(x, a=0.75, b=0.1, c=0.1):                 # ref: data_utils/selkov.py:18-22
    return np.stack([a - b * x[..., 0] - x[..., 0] * x[..., 1] ** 2,
                     -x[..., 1] + c * x[..., 0] + x[..., 0] * x[..., 1] ** 2], axis=-1)


def rhs_lv(x, a=2 / 3, b=4 / 3, c=1.0, d=1.0):           # ref: data_utils/lotka.py:33-41 (canonical)
    return np.stack([a - b * np.exp(x[..., 1]), c * np.exp(x[..., 0]) - d], axis=-1)


def rhs_growth(x, a=0.1, b=0.3):                         # ref: data_utils/growth.py:18-22
    return np.stack([a * x[..., 1] ** 2 - b * x[..., 0], x[..., 1]], axis=-1)


def ics_dosc(n, rng):                                    # ref: damped_oscillator.py:10-17
    r = rng.uniform(0.5, 2, n)
    th = rng.uniform(0, 2 * np.pi, n)
    return np.stack([r * np.cos(th), r * np.sin(th)], axis=-1)


def ics_selkov(n, rng):                                  # ref: selkov.py:10-15
    return rng.uniform(0.5, 1, (n, 2))


def ics_growth(n, rng):                                  # ref: growth.py:10-15
    return rng.uniform(0.2, 1, (n, 2))


def ics_lv(n, rng, h_min=3.0, h_max=4.5):                # ref: lotka.py:10-31 (rejection on H)
    out = []
    while len(out) < n:
        x0 = np.log(rng.uniform(0, 1, 2))
        h = np.exp(x0[0]) - x0[0] + 4 / 3 * np.exp(x0[1]) - 2 / 3 * x0[1]
        if h_min <= h <= h_max:
            out.append(x0)
    return np.array(out)


def rk4_trajectories(rhs, x0, dt, num_steps):
    """Batch RK4; returns (x, dx) of shape (n_ics, num_steps, d).  ref: data_utils/ode.py:7-28, 46-47"""
    x = np.zeros((num_steps, *x0.shape))
    dx = np.zeros_like(x)
    x[0] = x0
    for i in range(num_steps):
        d1 = rhs(x[i])
        dx[i] = d1
        if i == num_steps - 1:
            break
        k1 = dt * d1
        k2 = dt * rhs(x[i] + 0.5 * k1)
        k3 = dt * rhs(x[i] + 0.5 * k2)
        k4 = dt * rhs(x[i] + k3)
        x[i + 1] = x[i] + (k1 + 2 * k2 + 2 * k3 + k4) / 6
    return np.transpose(x, (1, 0, 2)), np.transpose(dx, (1, 0, 2))
