"""Synthetic trajectories of the four benchmark systems, generated with torch on any device.

Restates the reference's offline data recipe -- fixed-step RK4 over a batch of initial
conditions (data_utils/ode.py:7-28), the four right-hand sides and their initial-condition
samplers (data_utils/{damped_oscillator,selkov,lotka,growth}.py), additive noise scaled by the
per-dimension std (data_utils/ode.py:33-37) -- so that benchmark inputs of any size can be made
directly in HBM.  fp64 arithmetic like the numpy original, cast to fp32 at the end
(dataset.py:188-189).  Not part of the hot path.
"""
from __future__ import annotations

import math

import torch


def rhs_dosc(x, a=0.1):
    return torch.stack([-a * x[..., 0] - x[..., 1], x[..., 0] - a * x[..., 1]], dim=-1)


def rhs_selkov(x, a=0.75, b=0.1, c=0.1):
    xy2 = x[..., 0] * x[..., 1] ** 2
    return torch.stack([a - b * x[..., 0] - xy2, -x[..., 1] + c * x[..., 0] + xy2], dim=-1)


def rhs_lv(x, a=2 / 3, b=4 / 3, c=1.0, d=1.0):
    return torch.stack([a - b * torch.exp(x[..., 1]), c * torch.exp(x[..., 0]) - d], dim=-1)


def rhs_growth(x, a=0.1, b=0.3):
    return torch.stack([a * x[..., 1] ** 2 - b * x[..., 0], x[..., 1]], dim=-1)


def ics_dosc(n, gen, device):
    r = 0.5 + 1.5 * torch.rand(n, generator=gen, device=device, dtype=torch.float64)
    th = 2 * math.pi * torch.rand(n, generator=gen, device=device, dtype=torch.float64)
    return torch.stack([r * torch.cos(th), r * torch.sin(th)], dim=-1)


def ics_selkov(n, gen, device):
    return 0.5 + 0.5 * torch.rand(n, 2, generator=gen, device=device, dtype=torch.float64)


def ics_growth(n, gen, device):
    return 0.2 + 0.8 * torch.rand(n, 2, generator=gen, device=device, dtype=torch.float64)


def ics_lv(n, gen, device, h_min=3.0, h_max=4.5):
    out = torch.empty(0, 2, dtype=torch.float64, device=device)
    while out.shape[0] < n:                       # rejection on the Hamiltonian, lotka.py:10-31
        x0 = torch.log(torch.rand(4 * n, 2, generator=gen, device=device, dtype=torch.float64))
        h = torch.exp(x0[:, 0]) - x0[:, 0] + 4 / 3 * torch.exp(x0[:, 1]) - 2 / 3 * x0[:, 1]
        out = torch.cat([out, x0[(h >= h_min) & (h <= h_max)]])
    return out[:n]


SYSTEMS = {
    # name: (rhs, initial-condition sampler, default dt)
    "dosc": (rhs_dosc, ics_dosc, 0.02),
    "selkov": (rhs_selkov, ics_selkov, 0.002),
    "lv": (rhs_lv, ics_lv, 0.002),
    "growth": (rhs_growth, ics_growth, 0.002),
}


def rk4_trajectories(rhs, x0, dt, num_steps):
    """(x, dx) of shape (n_ics, num_steps, d): states and exact derivatives along RK4 orbits."""
    x = torch.empty(num_steps, *x0.shape, dtype=x0.dtype, device=x0.device)
    dx = torch.empty_like(x)
    cur = x0
    for i in range(num_steps):
        d1 = rhs(cur)
        x[i], dx[i] = cur, d1
        if i == num_steps - 1:
            break
        k1 = dt * d1
        k2 = dt * rhs(cur + 0.5 * k1)
        k3 = dt * rhs(cur + 0.5 * k2)
        k4 = dt * rhs(cur + k3)
        cur = cur + (k1 + 2 * k2 + 2 * k3 + k4) / 6
    return x.transpose(0, 1).contiguous(), dx.transpose(0, 1).contiguous()


# The four systems as library coefficients (evaluation/eval_eq.py:88-105): (poly_order, include_exp, Xi)
_LIBRARY_FORM = {
    "dosc": (2, False, [[0.0, -0.1, -1, 0.0, 0.0, 0.0], [0.0, 1, -0.1, 0.0, 0.0, 0.0]]),
    "growth": (2, False, [[0.0, -0.3, 0.0, 0.0, 0.0, 0.1], [0.0, 0.0, 1.0, 0.0, 0.0, 0.0]]),
    "selkov": (3, False, [[0.75, -0.1, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, -1.0, 0.0], [0.0, 0.1, -1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0]]),
    "lv": (2, True, [[2 / 3, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, -4 / 3], [-1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0]]),
}


def rk4_trajectories_fused(name, x0, dt, num_steps, subsample=1):
    """Same orbits from ONE HIP launch (symode_rk4_traj: a trajectory per thread, fp64), fp32 output."""
    from .engine import get_engine
    order, use_exp, xi = _LIBRARY_FORM[name]
    xi = torch.tensor(xi, dtype=torch.float64, device=x0.device)
    return get_engine().rk4_traj(x0, xi, order, 2 if use_exp else 0, num_steps, dt, subsample)


def make_dataset(name, n_ics, num_steps, dt=None, noise=0.0, seed=0, device="cpu", n_problems=1, fused=None):
    """fp32 (n_problems, n_ics*num_steps, d) trajectories and derivatives, flattened like ODEDataset
    (dataset.py:193-194).  Every problem gets its own initial conditions and noise draw.
    ``fused`` (default: on for CUDA devices) integrates with the HIP RK4 kernel instead of torch ops."""
    rhs, ics, dt0 = SYSTEMS[name]
    dt = dt0 if dt is None else dt
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    x0 = ics(n_problems * n_ics, gen, device)
    if fused is None:
        fused = torch.device(device).type == "cuda"
    if fused:
        x, dx = rk4_trajectories_fused(name, x0, dt, num_steps)
        x, dx = x.double(), dx
    else:
        x, dx = rk4_trajectories(rhs, x0, dt, num_steps)
    d = x.shape[-1]
    x = x.reshape(n_problems, n_ics * num_steps, d)
    dx = dx.reshape(n_problems, n_ics * num_steps, d)
    if noise > 0:
        std = x.std(dim=1, keepdim=True)
        x = x + noise * std * torch.randn(x.shape, generator=gen, device=device, dtype=x.dtype)
    return x.float().contiguous(), dx.float().contiguous()


# -------------------------------------------------------------------------------------------------
# Noisy-data recipe of the reference (data_utils/ode.py:30-49) with its two derivative estimators
# -------------------------------------------------------------------------------------------------
def gp_smooth(x, dt, noise_level, std_base, sigma_in, delta=1e-3):
    """Gaussian-process smoothing and numerical differentiation of noisy trajectories
    (reference data_utils/smoothing.py:155-196, ``num_diff_gp``).

    x (T, n_traj, d) float64 on any device.  Per state dimension the reference builds a GP-PCA model with as many
    factors as trajectories -- its loading matrix is then square and orthonormal, A A^T = I -- and takes the
    predictive mean, which is ordinary GP regression with an RBF kernel shared by all trajectories:

        X_hat(t*) = K(t*, t) (K(t, t) + sigma^2 I)^-1 Y,     K = sigma_out^2 exp(-(t - t')^2 / (2 sigma_in^2)),
        sigma = noise_level * std_base[d],  sigma_out = std_base[d];

    the derivative is the reference's forward difference of the predictive mean, (X_hat(t + delta) - X_hat(t)) / delta.
    One Cholesky factorisation of the (T, T) system per dimension (fp64; rocSOLVER when x lives on the GPU: T = 10^4
    takes about a second there, minutes with the reference's chain of dense numpy inverses).  Returns (dX, X_hat).
    """
    T = x.shape[0]
    t = torch.arange(T, device=x.device, dtype=torch.float64) * dt
    gap = t[:, None] - t[None, :]
    eye = torch.eye(T, device=x.device, dtype=torch.float64)
    xs, dxs = [], []
    for k in range(x.shape[-1]):
        so, sn = float(std_base[k]), float(noise_level * std_base[k])
        K = (so * so) * torch.exp(gap.square() * (-1.0 / (2.0 * sigma_in ** 2)))
        alpha = torch.cholesky_solve(x[:, :, k].to(torch.float64), torch.linalg.cholesky(K + (sn * sn) * eye))
        mean = K @ alpha
        ahead = ((so * so) * torch.exp((gap + delta).square() * (-1.0 / (2.0 * sigma_in ** 2)))) @ alpha
        xs.append(mean)
        dxs.append((ahead - mean) / delta)
    return torch.stack(dxs, -1), torch.stack(xs, -1)


def gen_data(name, n_ics, dt=0.002, num_steps=2000, subsample_rate=1, noise=0.0, multiplicative_noise=False, smoothing=None,
             gp_sigma_in=0.1, seed=0, device="cpu", fused=None):
    """One data set the way the reference's generators make it (data_utils/ode.py:30-49 behind
    ``get_{dosc,lv,selkov,growth}_data``): RK4 orbits; noise relative to each dimension's standard deviation over
    (time, trajectory) -- multiplicative for ``growth`` -- and then the derivative of the NOISY series: forward
    differences (the last sample keeps the exact RHS), or ``smoothing='gp'``: ``gp_smooth`` replaces both x and dx.
    Returns fp32 (n_ics, num_steps / subsample_rate, d) tensors x, dx on ``device``."""
    rhs, ics, _ = SYSTEMS[name]
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    x0 = ics(n_ics, gen, device)
    if fused is None:
        fused = torch.device(device).type == "cuda"
    if fused:
        x, dx = rk4_trajectories_fused(name, x0, dt, num_steps)
    else:
        x, dx = rk4_trajectories(rhs, x0, dt, num_steps)
    x, dx = x.double().transpose(0, 1).contiguous(), dx.double().transpose(0, 1).contiguous()      # (T, n_ics, d)
    if noise > 0:
        std = x.std(dim=(0, 1), unbiased=False)
        eps = torch.randn(x.shape, generator=gen, device=device, dtype=torch.float64)
        x = x * (1 + eps * noise) if multiplicative_noise else x + eps * noise * std
        if smoothing is None:
            dx[:-1] = (x[1:] - x[:-1]) / dt
        elif smoothing == "gp":
            print("Smoothing with Gaussian process...")
            dx, x = gp_smooth(x, dt, noise, std, gp_sigma_in)
        else:
            raise NotImplementedError(f"smoothing={smoothing!r}")
    x, dx = x[::subsample_rate].transpose(0, 1), dx[::subsample_rate].transpose(0, 1)
    return x.float().contiguous(), dx.float().contiguous()
