"""Symmetry regularisers and the fixed-step integrator -- surface of the reference's model_utils.py.

``make_*symmreg*`` return callables with the reference's signatures (model_utils.py:214-221);
``odeint`` keeps its signature (model_utils.py:223-255).  Division of labour:

  * the autoencoder / generator halves are stock PyTorch-ROCm modules (out of scope by
    north_star) and are only *called* here;
  * everything that touches the SINDy library runs in HIP kernels: the regressor's forward and
    vjp (autograd route), its analytic JVP ``(J_Theta(x) v) Xi^T`` (no double-backward trick:
    SURVEY H3), the fused K-step integrator, and the fully fused reversed regulariser once
    ``(g(x), J_g(x))`` have been precomputed (they do not depend on Xi: model_utils.py:172-211).
"""
from __future__ import annotations

import weakref
from functools import partial

import torch
from torch.autograd.functional import jvp

from .sindy import SINDyRegression


# --------------------------------------------------------------------------------------------
# odeint                                                            ref: model_utils.py:223-255
# --------------------------------------------------------------------------------------------
def odeint(f, x0, t, dt, method='euler', full_traj=False):
    '''
    Integrate an ODE f over a time interval differentiably (fixed step).
    f: nn.Module / callable RHS;  x0: initial state;  t: time;  dt: step;  method: 'euler' | 'rk4'.
    A SINDyRegression RHS with no gradient requested runs as ONE fused kernel (all K steps in registers;
    ``full_traj`` writes the state after every step); otherwise the steps are chained through autograd.
    '''
    n_steps = int(t / dt)
    if method not in ('euler', 'rk4'):
        raise ValueError('Unrecognized ODEInt method.')
    fused = (isinstance(f, SINDyRegression) and x0.is_cuda
             and not (torch.is_grad_enabled() and (x0.requires_grad or any(p.requires_grad for p in f.parameters()))))
    if fused:
        xi = f.get_Xi().detach()
        lead = x0.shape[:-1]
        flat = x0.reshape(-1, f.latent_dim)
        if full_traj:                                       # (K, ..., d): every step of the roll-out from ONE launch
            out = f.engine.odeint_traj(flat, xi, f.mask, f.poly_order, f.flags, n_steps, dt, method)
            return out.reshape(n_steps, *lead, f.latent_dim)
        out = f.engine.odeint(flat, xi, f.mask, f.poly_order, f.flags, n_steps, dt, method)
        return out.reshape(*lead, f.latent_dim)
    traj = []
    for _ in range(n_steps):
        if method == 'euler':
            x0 = x0 + dt * f(x0)
        else:
            k1 = f(x0)
            k2 = f(x0 + dt / 2 * k1)
            k3 = f(x0 + dt / 2 * k2)
            k4 = f(x0 + dt * k3)
            x0 = x0 + dt / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
        if full_traj:
            traj.append(x0)
    return torch.stack(traj, dim=0) if full_traj else x0


class _EulerFlowFn(torch.autograd.Function):
    """(f(x), J_f(x) v) for f = K Euler steps of the regressor ODE: ONE fused launch forward, one backward
    (symode_euler_jvp / symode_euler_jvp_vjp), differentiable once w.r.t. x, v and Xi."""

    @staticmethod
    def forward(ctx, x, v, xi, mask, reg, n_steps, dt):
        ctx.reg, ctx.n_steps, ctx.dt = reg, n_steps, dt
        ctx.save_for_backward(x, v, xi, mask)
        return reg.engine.euler_jvp(x.detach(), v.detach(), xi.detach(), mask, reg.poly_order, reg.flags, n_steps, dt)

    @staticmethod
    def backward(ctx, g_x, g_t):
        x, v, xi, mask = ctx.saved_tensors
        reg = ctx.reg
        gx, gv, gxi = reg.engine.euler_jvp_vjp(x.detach(), v.detach(), g_x.contiguous(), g_t.contiguous(), xi.detach(), mask,
                                               reg.poly_order, reg.flags, ctx.n_steps, ctx.dt)
        return gx, gv, gxi, None, None, None, None


class _EulerFlow:
    """f = odeint(regressor, ., int_t, int_dt) together with its tangent map.

    ``tangent(x, v)`` returns ``(f(x), J_f(x) v)`` by propagating the tangent through the same
    steps with the regressor's analytic JVP -- the quantity the reference obtains from
    ``jvp(f, x, v, create_graph=True)`` (model_utils.py:56) -- differentiable w.r.t. Xi.
    """

    def __init__(self, regressor, int_t, int_dt, method='euler'):
        self.regressor, self.int_t, self.int_dt, self.method = regressor, int_t, int_dt, method

    def _fused(self, x):
        reg = self.regressor
        return (self.method == 'euler' and isinstance(reg, SINDyRegression) and x.is_cuda
                and hasattr(reg.engine, 'euler_jvp'))

    def __call__(self, x):
        if self._fused(x) and torch.is_grad_enabled():
            reg = self.regressor
            lead = x.shape[:-1]
            x2 = x.reshape(-1, reg.latent_dim)
            out, _ = _EulerFlowFn.apply(x2, torch.zeros_like(x2), reg.get_Xi(), reg.mask, reg,
                                        int(self.int_t / self.int_dt), self.int_dt)
            return out.reshape(*lead, reg.latent_dim)
        return odeint(self.regressor, x, self.int_t, self.int_dt, self.method)

    def tangent(self, x, v):
        n_steps = int(self.int_t / self.int_dt)
        reg, dt = self.regressor, self.int_dt
        if self._fused(x):
            lead = x.shape[:-1]
            xo, to = _EulerFlowFn.apply(x.reshape(-1, reg.latent_dim), v.reshape(-1, reg.latent_dim), reg.get_Xi(), reg.mask,
                                        reg, n_steps, dt)
            return xo.reshape(*lead, reg.latent_dim), to.reshape(*lead, reg.latent_dim)
        for _ in range(n_steps):
            if self.method != 'euler':
                raise NotImplementedError('tangent flow is implemented for the Euler integrator')
            h, jv = reg.forward_and_jvp(x, v)
            x = x + dt * h
            v = v + dt * jv
        return x, v


def _z_mean(autoencoder, z_mean):
    return autoencoder.encoder[-2].bias if z_mean is None else z_mean          # model_utils.py:46


def _jvp_fn(require_grad):
    return partial(jvp, create_graph=True, strict=True) if require_grad else jvp


def mlp_jvp(module, h, t):
    """(module(h), J_module(h) t) by forward-mode propagation through the stock layers of the autoencoder: Linear
    (parametrised weights included), eval-mode BatchNorm1d, Reshape / Identity, nested Sequential and the elementwise
    activations.  One pass of twice the GEMMs instead of the double-backward trick of ``torch.autograd.functional.jvp``
    (forward + a backward w.r.t. a dummy cotangent + its differentiation); ordinary autograd differentiates the result.
    Returns None when a layer is not covered (the caller falls back to the functional jvp)."""
    import torch.nn as nn
    from .autoencoder import Reshape
    if isinstance(module, nn.Sequential):
        for layer in module:
            out = mlp_jvp(layer, h, t)
            if out is None:
                return None
            h, t = out
        return h, t
    if isinstance(module, nn.Identity):
        return h, t
    if isinstance(module, Reshape):
        return h.reshape(module.shape), t.reshape(module.shape)
    if isinstance(module, nn.Linear):
        W = module.weight
        return torch.nn.functional.linear(h, W, module.bias), torch.nn.functional.linear(t, W)
    if isinstance(module, nn.BatchNorm1d):
        if module.training or module.running_var is None:
            return None
        scale = torch.rsqrt(module.running_var + module.eps)
        if module.weight is not None:
            scale = scale * module.weight
        shift = -module.running_mean * scale + (module.bias if module.bias is not None else 0.0)
        return h * scale + shift, t * scale
    if isinstance(module, nn.ReLU):
        return torch.relu(h), t * (h > 0).to(t.dtype)
    if isinstance(module, nn.LeakyReLU):
        slope = torch.where(h > 0, torch.ones_like(h), torch.full_like(h, module.negative_slope))
        return torch.nn.functional.leaky_relu(h, module.negative_slope), t * slope
    if isinstance(module, nn.Tanh):
        y = torch.tanh(h)
        return y, t * (1.0 - y * y)
    if isinstance(module, nn.Sigmoid):
        y = torch.sigmoid(h)
        return y, t * (y * (1.0 - y))
    if isinstance(module, nn.SiLU):
        sg = torch.sigmoid(h)
        return h * sg, t * (sg * (1.0 + h * (1.0 - sg)))
    if isinstance(module, nn.ELU):
        e = module.alpha * torch.exp(h)
        return torch.where(h > 0, h, e - module.alpha), t * torch.where(h > 0, torch.ones_like(h), e)
    if isinstance(module, nn.Softplus) and module.beta == 1:
        return torch.nn.functional.softplus(h, threshold=module.threshold), t * torch.sigmoid(h)
    return None


def _module_jvp(module, h, t, require_grad):
    """J_module(h) t: analytic forward mode where the layers allow it, the reference's functional jvp otherwise."""
    out = mlp_jvp(module, h, t)
    if out is not None:
        return out[1]
    return _jvp_fn(require_grad)(module, h, v=t)[1]


# --------------------------------------------------------------------------------------------
# The Xi-independent half of S2 / S3.  x_fx = [x | f(x)] goes through the autoencoder component by component (Linear
# layers act on the last axis, eval-mode BatchNorm is a per-feature affine map), and the generators / group elements are
# block diagonal over the components: everything computed from the x component -- enc(x), the decoder tangent
# J_dec(z_x)(L z_x), the decoded g z_x -- does not depend on the ODE.  The L-BFGS trainer evaluates hundreds of closures
# on ONE fixed batch (train.py:626), so with a frozen autoencoder that half is computed once per batch and only the
# f(x) component runs through the MLP per closure.  Opt-in: the caller passes the fixed batch as ``x_const``.
# --------------------------------------------------------------------------------------------
_CONST_HALF = {}


def _fingerprint(*modules):
    """Cheap identity of the frozen halves a cached quantity was computed from: storage address and in-place version
    of every parameter, buffer and plain-tensor attribute the stock modules read (generator masks).  load_state_dict,
    an optimiser step or ``mask *= ...`` all bump a version; a re-created module changes the addresses."""
    out = []
    for m in modules:
        tensors = list(m.parameters()) + list(m.buffers())
        tensors += [t for t in getattr(m, 'masks', []) if isinstance(t, torch.Tensor)]
        out.append(tuple((t.data_ptr(), t._version) for t in tensors))
    return tuple(out)


def _zmean_key(z_mean):
    return None if z_mean is None else (id(z_mean), z_mean._version)


def _diag_block(mat, n_comps):
    """the common diagonal block of a block-diagonal (n_comps k, n_comps k) matrix, or None"""
    k = mat.shape[-1] // n_comps
    if k * n_comps != mat.shape[-1]:
        return None
    L = mat[:k, :k]
    return L if torch.equal(mat, torch.block_diag(*([L] * n_comps))) else None


def _split_ok(x_const, x_fx, autoencoder, generator, normalize, numpy):
    from .autoencoder import AutoEncoder
    if x_const is None or numpy or normalize != 'global' or not isinstance(autoencoder, AutoEncoder):
        return False
    if not isinstance(autoencoder.encoder, torch.nn.Sequential) or autoencoder.training:
        return False
    if x_fx.dim() != 3 or x_fx.shape[0] % x_fx.shape[1] != 0 or x_const.shape != x_fx[:, 0].shape:
        return False
    return not any(p.requires_grad for m in (autoencoder, generator) for p in m.parameters())


def _by_rows(module_call, a, n_comps):
    """an (B, k) batch through a module that expects (., n_comps, k): consecutive samples stand in for the components"""
    return module_call(a.reshape(-1, n_comps, a.shape[-1])).reshape(a.shape[0], -1)


def _const_half(kind, x_const, autoencoder, generator, blocks, zm, n_comps, x_fx=None, z_mean=None):
    """cached per live batch tensor, latent offset and state of the frozen modules: (z_x, [decoder tangent | decoded g z_x per block])"""
    key = (kind, x_const._version, _fingerprint(autoencoder, generator), _zmean_key(z_mean), len(blocks))
    hit = _CONST_HALF.get(kind)
    if hit is None or hit[0]() is not x_const or hit[1] != key:
        if x_fx is not None and not torch.equal(x_const, x_fx[:, 0]):       # checked when the half is (re)computed
            raise ValueError('x_const must be the x component of x_fx (x_fx[:, 0])')
        with torch.no_grad():
            z0 = _by_rows(autoencoder.encode, x_const, n_comps) - zm
            outs = []
            for L in blocks:
                if kind == 'i':
                    t = _module_jvp(autoencoder.decoder, z0.reshape(-1, n_comps, z0.shape[-1]),
                                    (z0 @ L.T).reshape(-1, n_comps, z0.shape[-1]), False)
                    outs.append(t.reshape(x_const.shape[0], -1))
                else:
                    outs.append(_by_rows(autoencoder.decode, z0 @ L.T + zm, n_comps))
        hit = (weakref.ref(x_const), key, z0, outs)
        _CONST_HALF[kind] = hit
    return hit[2], hit[3]


def _sharded_params(f, params):
    if params is not None:
        return list(params)
    reg = getattr(f, 'regressor', None)
    if reg is None:
        raise ValueError('group=...: pass params= (the tensors f depends on) unless f is the Euler flow of a regressor.')
    return list(reg.parameters())


def _finish_sharded(nums, dens, count, relative, params, group, also):
    """``nums`` / ``dens``: per-generator sums of squares over THIS rank's points.  The relative loss is a ratio of two
    batch means (model_utils.py:62, 118-121), so numerators, denominators and their gradients cross the ranks
    separately -- in ONE packed buffer (batched.allreduce_sums) together with whatever else the closure needs summed
    (``also``: e.g. the residual's sum of squares) -- and the division happens after the collective."""
    from .batched import allreduce_sums
    cnt = torch.tensor(float(count), device=nums[0].device)
    n = len(nums)
    red = allreduce_sums(list(nums) + (list(dens) if relative else []) + [cnt] + list(also or []), params, group)
    loss = 0.0
    for k in range(n):
        loss = loss + (red[k] / red[n + k] if relative else red[k] / red[n].detach())
    extra = red[(2 * n if relative else n) + 1:]
    return loss if also is None else (loss, extra)


# --------------------------------------------------------------------------------------------
# S2: infinitesimal                                                   ref: model_utils.py:8-67
# --------------------------------------------------------------------------------------------
def symmreg_i(x_fx, autoencoder, generator, f=None, dfdx=None, normalize='global', z_mean=None, relative=True,
              require_grad=False, numpy=False, x_const=None, group=None, also=None, params=None):
    """``group`` (a torch.distributed process group): ``x_fx`` is this rank's POINT SHARD of the batch.
    Returns the loss of the whole batch -- per generator, numerator and denominator of the relative loss and their
    parameter gradients are summed locally, all-reduced in one packed buffer, and the ratio is formed after the
    collective; the returned scalar carries the global first derivative with respect to ``params`` (default: the
    parameters of the regressor behind ``f``), identical on every rank.  ``also``: further local sums to ride in the same
    buffer; then ``(loss, [global sums])`` is returned."""
    sharded = group is not None or also is not None
    if sharded and (numpy or not require_grad or normalize == 'in_batch'):
        raise ValueError('group=... needs require_grad=True, torch inputs and a batch-independent normalisation.')
    nums, dens = [], []
    if numpy:
        x_fx = torch.from_numpy(x_fx).float().to(autoencoder.device)
        if z_mean is not None:
            z_mean = torch.from_numpy(z_mean).float().to(autoencoder.device)
        if require_grad:
            raise ValueError('Cannot require grad when numpy=True.')
    if f is None and dfdx is None:
        raise ValueError('Either f or dfdx must be specified.')
    if f is not None and dfdx is not None:
        raise ValueError('Only one of f and dfdx can be specified.')
    jvp_fn = _jvp_fn(require_grad)
    autoencoder.eval()
    generator.eval()
    if f is not None and _split_ok(x_const, x_fx, autoencoder, generator, normalize, numpy):
        nc = x_fx.shape[1]
        blocks = [_diag_block(v, nc) for v in generator.get_full_basis_list()]
        if all(b is not None for b in blocks):
            with torch.set_grad_enabled(require_grad):
                zm = _z_mean(autoencoder, z_mean)
                _, v_xs = _const_half('i', x_const, autoencoder, generator, blocks, zm, nc, x_fx=x_fx, z_mean=z_mean)
                x, fx = x_fx[:, 0], x_fx[:, 1]
                z1 = _by_rows(autoencoder.encode, fx, nc) - zm
                loss = 0.0
                for L, v_x in zip(blocks, v_xs):
                    v_fx = _module_jvp(autoencoder.decoder, z1.reshape(-1, nc, z1.shape[-1]),
                                       (z1 @ L.T).reshape(-1, nc, z1.shape[-1]), require_grad).reshape(fx.shape[0], -1)
                    input_variation = f.tangent(x, v_x)[1] if isinstance(f, _EulerFlow) else jvp_fn(f, x, v_x)[1]
                    if sharded:
                        nums.append(torch.sum((input_variation - v_fx) ** 2))
                        dens.append(torch.sum(input_variation ** 2))
                        continue
                    err = torch.mean((input_variation - v_fx) ** 2)
                    loss += err / torch.mean(input_variation ** 2) if relative else err
                if sharded:
                    return _finish_sharded(nums, dens, fx.numel(), relative, _sharded_params(f, params), group, also)
            return loss
    with torch.set_grad_enabled(require_grad):
        loss = 0.0
        z = autoencoder.encode(x_fx)
        x = x_fx[:, 0]
        if normalize == 'in_batch':
            z = z - z.mean(dim=0, keepdim=True)
        elif normalize == 'global':
            z = z - _z_mean(autoencoder, z_mean)
        z_shape = z.shape
        for v in generator.get_full_basis_list():
            v_z = torch.einsum('jk,...k->...j', v, z.reshape(z_shape[0], -1)).reshape(z_shape)
            v_x_fx = _module_jvp(autoencoder.decoder, z, v_z, require_grad)      # stock PyTorch MLP, forward-mode tangent
            v_x, v_fx = v_x_fx[:, 0], v_x_fx[:, 1]
            if f is not None:
                if isinstance(f, _EulerFlow):
                    input_variation = f.tangent(x, v_x)[1]                       # HIP analytic tangent flow
                else:
                    input_variation = jvp_fn(f, x, v_x)[1]                       # arbitrary f: autograd
            else:
                input_variation = torch.einsum('bjk,bk->bj', dfdx, v_x)
            if sharded:
                nums.append(torch.sum((input_variation - v_fx) ** 2))
                dens.append(torch.sum(input_variation ** 2))
            elif not relative:
                loss += torch.mean((input_variation - v_fx) ** 2)
            else:
                loss += torch.mean((input_variation - v_fx) ** 2) / torch.mean(input_variation ** 2)
        if sharded:
            return _finish_sharded(nums, dens, x.numel(), relative, _sharded_params(f, params), group, also)
    if numpy:
        loss = loss.cpu().numpy()
    return loss


# --------------------------------------------------------------------------------------------
# S3: finite                                                         ref: model_utils.py:69-124
# --------------------------------------------------------------------------------------------
def symmreg_f(x_fx, autoencoder, generator, f, normalize='global', z_mean=None, relative=True, require_grad=False,
              numpy=False, x_const=None, group=None, also=None, params=None):
    """``group`` / ``also`` / ``params``: point-sharded evaluation, as in symmreg_i."""
    sharded = group is not None or also is not None
    if sharded and (numpy or not require_grad or normalize == 'in_batch'):
        raise ValueError('group=... needs require_grad=True, torch inputs and a batch-independent normalisation.')
    nums, dens = [], []
    autoencoder.eval()
    generator.eval()
    if numpy:
        x_fx = torch.from_numpy(x_fx).float().to(generator.Li[0].device)
        if z_mean is not None:
            z_mean = torch.from_numpy(z_mean).float().to(generator.Li[0].device)
        if require_grad:
            raise ValueError('Cannot require grad when numpy=True.')
    if _split_ok(x_const, x_fx, autoencoder, generator, normalize, numpy):
        nc = x_fx.shape[1]
        blocks = [_diag_block(g, nc) for g in generator.get_deterministic_group_elems()]
        if all(b is not None for b in blocks):
            with torch.set_grad_enabled(require_grad):
                zm = _z_mean(autoencoder, z_mean)
                _, g_xs = _const_half('f', x_const, autoencoder, generator, blocks, zm, nc, x_fx=x_fx, z_mean=z_mean)
                fx = x_fx[:, 1]
                z1 = _by_rows(autoencoder.encode, fx, nc) - zm
                loss = 0.0
                for G, g_x in zip(blocks, g_xs):
                    g_fx = _by_rows(autoencoder.decode, z1 @ G.T + zm, nc)
                    f_g_x = f(g_x)
                    if sharded:
                        nums.append(torch.sum((f_g_x - g_fx) ** 2))
                        dens.append(torch.sum((f_g_x - fx) ** 2))
                        continue
                    err = torch.mean((f_g_x - g_fx) ** 2)
                    loss += err / torch.mean((f_g_x - fx) ** 2) if relative else err
                if sharded:
                    return _finish_sharded(nums, dens, fx.numel(), relative, _sharded_params(f, params), group, also)
            return loss
    with torch.set_grad_enabled(require_grad):
        loss = 0.0
        z = autoencoder.encode(x_fx)
        fx = x_fx[:, 1]
        if normalize == 'in_batch':
            zm = z.mean(dim=0, keepdim=True)
        else:
            zm = _z_mean(autoencoder, z_mean)
        z = z - zm
        z_shape = z.shape
        for g in generator.get_deterministic_group_elems():
            g_z = torch.einsum('jk,...k->...j', g, z.reshape(z_shape[0], -1)).reshape(z_shape) + zm
            g_x_fx = autoencoder.decode(g_z)
            g_x, g_fx = g_x_fx[:, 0], g_x_fx[:, 1]
            if numpy:
                g_x = g_x.cpu().numpy()
            f_g_x = f(g_x)
            if numpy:
                f_g_x = torch.from_numpy(f_g_x).float().to(generator.Li[0].device)
            if sharded:
                nums.append(torch.sum((f_g_x - g_fx) ** 2))
                dens.append(torch.sum((f_g_x - fx) ** 2))
            elif not relative:
                loss += torch.mean((f_g_x - g_fx) ** 2)
            else:
                loss += torch.mean((f_g_x - g_fx) ** 2) / torch.mean((f_g_x - fx) ** 2)
        if sharded:
            return _finish_sharded(nums, dens, fx.numel(), relative, _sharded_params(f, params), group, also)
    if numpy:
        loss = loss.cpu().numpy()
    return loss


# --------------------------------------------------------------------------------------------
# S4: reversed                                                      ref: model_utils.py:126-211
# --------------------------------------------------------------------------------------------
def _group_transform(x, autoencoder, g, normalize='global', z_mean=None):
    xx = torch.stack([x, x], dim=1)
    z = autoencoder.encode(xx)
    zm = z.mean(dim=0, keepdim=True) if normalize == 'in_batch' else _z_mean(autoencoder, z_mean)
    z = z - zm
    z_shape = z.shape
    g_z = torch.einsum('jk,...k->...j', g, z.reshape(z_shape[0], -1)).reshape(z_shape) + zm
    return autoencoder.decode(g_z)[:, 0]


def _group_transform_jvp(x, v, autoencoder, g, normalize='global', z_mean=None):
    """(g(x), J_g(x) v) by forward mode through encoder -> group element -> decoder; None if a layer is not covered."""
    xx, vv = torch.stack([x, x], dim=1), torch.stack([v, v], dim=1)
    enc = mlp_jvp(autoencoder.encoder, xx, vv)
    if enc is None:
        return None
    z, tz = enc
    if normalize == 'in_batch':
        zm, tzm = z.mean(dim=0, keepdim=True), tz.mean(dim=0, keepdim=True)
    else:
        zm, tzm = _z_mean(autoencoder, z_mean), 0.0
    shape = z.shape
    act = lambda a: torch.einsum('jk,...k->...j', g, a.reshape(shape[0], -1)).reshape(shape)  # noqa: E731
    g_z, t_gz = act(z - zm) + zm, act(tz - tzm) + tzm
    dec = mlp_jvp(autoencoder.decoder, g_z, t_gz)
    if dec is None:
        return None
    return dec[0][:, 0], dec[1][:, 0]


def precompute_symmreg_r(x, autoencoder, generator, z_mean=None, scale=0.01):
    '''
    g(x) and J_g(x) for every deterministic group element -- they do not depend on the ODE, so
    a frozen autoencoder lets them be computed once per dataset (model_utils.py:172-211).
    Returns (gx_list, Jgx_list) with Jgx of shape (B, d, d); the reference's vmap version
    carries a stray singleton axis and transposed per-sample stacking (PySR-only path).
    '''
    autoencoder.eval()
    generator.eval()
    gx_list, Jgx_list = [], []
    d = x.shape[-1]
    with torch.no_grad():
        for g in generator.get_deterministic_group_elems(scale=scale):
            tr = partial(_group_transform, autoencoder=autoencoder, g=g, normalize='global', z_mean=z_mean)
            gx_list.append(tr(x))
            cols = []
            for j in range(d):
                e = torch.zeros_like(x)
                e[:, j] = 1.0
                fast = _group_transform_jvp(x, e, autoencoder, g, normalize='global', z_mean=z_mean)
                cols.append(fast[1] if fast is not None else jvp(tr, x, v=e)[1])
            Jgx_list.append(torch.stack(cols, dim=-1))
    return gx_list, Jgx_list


class _ReversedFused(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xi, mask, x, gx, jgx, reg):
        loss, grad = reg.engine.symreg_reversed(x, gx, jgx, xi.detach(), mask, reg.poly_order, reg.flags)
        ctx.save_for_backward(grad)
        return loss.clone()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return g * grad, None, None, None, None, None


_R_CACHE = {}


def symmreg_r(x, autoencoder, generator, h, normalize='global', z_mean=None, require_grad=False, scale=0.01):
    '''
    Reversed symmetry loss  sum_g mean((J_g(x) h(x) - h(g(x)))^2).
    With a SINDyRegression ``h`` and a frozen autoencoder, (g(x), J_g(x)) are computed once per
    (x, autoencoder, generator) and every later call is one fused HIP kernel; any other ``h``
    takes the reference's autograd route.
    '''
    autoencoder.eval()
    generator.eval()
    frozen = not any(p.requires_grad for p in list(autoencoder.parameters()) + list(generator.parameters()))
    if isinstance(h, SINDyRegression) and normalize == 'global' and frozen and x.is_cuda:
        # identity of the LIVE tensor object + its version, not its address: the allocator hands the address of a freed
        # batch to the next one (every epoch draws a new subsample, main.py:36-38), a weak reference to it dies instead
        key = (x._version, tuple(x.shape), _fingerprint(autoencoder, generator), _zmean_key(z_mean), float(scale))
        hit = _R_CACHE.get('entry')
        if hit is None or hit[0]() is not x or hit[1] != key:
            gx, jgx = precompute_symmreg_r(x, autoencoder, generator, z_mean=z_mean, scale=scale)
            hit = (weakref.ref(x), key, torch.stack(gx).contiguous(), torch.stack(jgx).contiguous())
            _R_CACHE['entry'] = hit
        gx, jgx = hit[2], hit[3]
        with torch.set_grad_enabled(require_grad):
            return _ReversedFused.apply(h.get_Xi(), h.mask, x, gx, jgx, h)
    jvp_fn = _jvp_fn(require_grad)
    with torch.set_grad_enabled(require_grad):
        loss = 0.0
        for g in generator.get_deterministic_group_elems(scale=scale):
            tr = partial(_group_transform, autoencoder=autoencoder, g=g, normalize=normalize, z_mean=z_mean)
            gx = tr(x)
            hx = h(x)
            variation1 = jvp_fn(tr, x, v=hx)[1]
            variation2 = h(gx)
            loss += torch.mean((variation1 - variation2) ** 2)
    return loss


# --------------------------------------------------------------------------------------------
# S1: linear latent regulariser                                          ref: train.py:502-507
# --------------------------------------------------------------------------------------------
class _LinearFused(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xi, mask, z, L, reg):
        loss, grad = reg.engine.symreg_linear(z, xi.detach(), mask, L, reg.poly_order, reg.flags)
        ctx.save_for_backward(grad)
        return loss.clone()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return g * grad, None, None, None, None


def symmreg_linear(z, regressor, basis_list):
    """sum_v || J_h(z)(v z) - v h(z) ||_F^2 for h = regressor: one fused kernel over all generators.
    (z is treated as data: the gradient reaches Xi only, as in the latent L-BFGS use.)"""
    d = regressor.latent_dim
    L = torch.stack([v[:d, :d] for v in basis_list]).to(z.device).float().contiguous()
    return _LinearFused.apply(regressor.get_Xi(), regressor.mask, z.detach().reshape(-1, d), L, regressor)


make_symmreg = lambda autoencoder, generator: partial(symmreg_i, autoencoder=autoencoder, generator=generator)  # noqa: E731
make_symmreg_pttrain = lambda autoencoder, generator: partial(symmreg_i, autoencoder=autoencoder, generator=generator, require_grad=True)  # noqa: E731
make_symmreg_np = lambda autoencoder, generator: partial(symmreg_i, autoencoder=autoencoder, generator=generator, numpy=True)  # noqa: E731
make_fsymmreg = lambda autoencoder, generator: partial(symmreg_f, autoencoder=autoencoder, generator=generator)  # noqa: E731
make_fsymmreg_pttrain = lambda autoencoder, generator: partial(symmreg_f, autoencoder=autoencoder, generator=generator, require_grad=True)  # noqa: E731
make_fsymmreg_np = lambda autoencoder, generator: partial(symmreg_f, autoencoder=autoencoder, generator=generator, numpy=True)  # noqa: E731
make_rsymmreg = lambda autoencoder, generator: partial(symmreg_r, autoencoder=autoencoder, generator=generator)  # noqa: E731
make_rsymmreg_pttrain = lambda autoencoder, generator: partial(symmreg_r, autoencoder=autoencoder, generator=generator, require_grad=True)  # noqa: E731
