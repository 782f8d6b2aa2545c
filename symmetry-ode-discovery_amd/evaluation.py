"""Success criterion and cross-seed aggregation (reference evaluation/eval_eq.py).

"Identical recovered sparsity pattern" is defined here: an equation has the correct form when
its mask equals the support of the ground-truth row (eval_eq.py:25); coefficient error is the
MSE over the truth support (eval_eq.py:28).
"""
from __future__ import annotations

import os

import numpy as np
import torch

# Ground-truth coefficient tables for the shipped library choices (eval_eq.py:88-105):
# lv: order 2 + exp (p = 8); selkov: order 3 (p = 10); dosc / growth: order 2 (p = 6).
sindy_truth = {
    'lv': np.array([[2 / 3, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, -4 / 3], [-1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0]]),
    'selkov': np.array([[0.75, -0.1, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, -1.0, 0.0],
                        [0.0, 0.1, -1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0]]),
    'dosc': np.array([[0.0, -0.1, -1, 0.0, 0.0, 0.0], [0.0, 1, -0.1, 0.0, 0.0, 0.0]]),
    'growth': np.array([[0.0, -0.3, 0.0, 0.0, 0.0, 0.1], [0.0, 0.0, 1.0, 0.0, 0.0, 0.0]]),
}


def eval_sindy_regressor(regressor, truth, threshold=0.05):
    """Returns (coef, correct_form, mse, correct_form_all, mse_all) -- eval_eq.py:7-34."""
    with torch.no_grad():
        coef = (regressor.get_Xi() if regressor.constraint else regressor.Xi).cpu().numpy()
        mask = regressor.mask.bool().cpu().numpy()
    coef = np.where(mask, coef, 0.0)
    truth_mask = truth != 0
    n_eqs = coef.shape[0]
    correct_form = np.zeros(n_eqs)
    mse = np.ones(n_eqs) * -1.0
    for i in range(n_eqs):
        correct_form[i] = np.all(mask[i, :] == truth_mask[i, :])
        mse[i] = np.mean((coef[i, truth_mask[i, :]] - truth[i, truth_mask[i, :]]) ** 2)
    return coef, correct_form, mse, np.all(correct_form), np.mean(mse)


result_dir = 'eval_results'


def aggregate_results(run_name, min_seed=0, max_seed=100, mse_multiplier=1.0):
    """Success rates and RMSE over eval_results/<run_name>/seed*.npz (eval_eq.py:38-85)."""
    directory = os.path.join(result_dir, run_name)
    cf, mse, cf_all, mse_all = [], [], [], []
    for filename in sorted(os.listdir(directory)):
        if not filename.endswith('.npz'):
            continue
        seed = int(filename.split('.')[0][4:])
        if seed >= max_seed or seed < min_seed:
            continue
        res = np.load(os.path.join(directory, filename))
        cf.append(res['correct_form'])
        mse.append(res['mse'])
        cf_all.append(res['correct_form_all'])
        mse_all.append(res['mse_all'])
    print(f'Loaded results from {len(cf)} runs.')
    cf, cf_all = np.stack(cf), np.array(cf_all)
    out = {'n_runs': len(cf), 'success': np.sum(cf, axis=0).astype(int), 'joint_success': int(np.sum(cf_all))}
    for i, each in enumerate(out['success']):
        print(f'Equation {i} success rate = {each}/{cf.shape[0]}')
    print(f"Joint success rate = {out['joint_success']}/{cf.shape[0]}")
    rmse = np.sqrt(np.stack(mse))
    out['rmse'], out['rmse_any'] = [], []
    for i in range(rmse.shape[1]):
        ok = rmse[np.where(cf[:, i]), i]
        v, s = np.mean(ok) * mse_multiplier, np.std(ok) * mse_multiplier
        va, sa = np.mean(rmse[:, i]) * mse_multiplier, np.std(rmse[:, i]) * mse_multiplier
        out['rmse'].append((v, s))
        out['rmse_any'].append((va, sa))
        print(f'Equation {i} RMSE = {v:.4f} ({s:.4f})')
        print(f'Equation {i} RMSE (any) = {va:.4f} ({sa:.4f})')
    ra = np.sqrt(np.stack(mse_all))
    ok = ra[np.where(cf_all)]
    out['rmse_all'] = (np.mean(ok) * mse_multiplier, np.std(ok) * mse_multiplier)
    out['rmse_all_any'] = (np.mean(ra) * mse_multiplier, np.std(ra) * mse_multiplier)
    print(f"All equations RMSE = {out['rmse_all'][0]:.4f} ({out['rmse_all'][1]:.4f})")
    print(f"All equations RMSE (any) = {out['rmse_all_any'][0]:.4f} ({out['rmse_all_any'][1]:.4f})")
    return out


@torch.no_grad()
def eval_ltp_accuracy(regressor, autoencoder, x, dt=None, **kwargs):
    """Long-term prediction with the learned dynamics (reference evaluation/eval_ltp.py:9-45): roll the regressor's
    ODE out from x[:, 0] with RK4 over the length of x (through the autoencoder's latent space when one is given)
    and report the squared error per step.  x: (n_ics, n_steps, d) or (n_ics, n_steps, n_comps, d).
    The roll-out is ONE launch of the fused full-trajectory integrator (symode_odeint_traj)."""
    from .dataset import ode_dt_dict
    from .model_utils import odeint
    x0 = x[:, 0]
    if x.dim() == 3:
        n_ics, n_steps, n_dim = x.shape
    else:
        n_ics, n_steps, _, n_dim = x.shape
    n_steps -= 1
    if dt is None:
        dt = ode_dt_dict[kwargs['task'].split('_')[-1]]
    t_max = n_steps * dt
    if autoencoder is not None:
        z0 = autoencoder.encode(x0)
        if z0.dim() == 3:
            z0 = z0.flatten(0, 1)
        z_pred = odeint(regressor, z0, t_max, dt, method='rk4', full_traj=True).transpose(0, 1)
        x_pred = autoencoder.decode(z_pred.flatten(0, 1)).reshape(n_ics, n_steps, n_dim)
    else:
        x_pred = odeint(regressor, x0, t_max, dt, method='rk4', full_traj=True).transpose(0, 1)
    error = torch.mean((x[:, 1:] - x_pred) ** 2, dim=-1)
    res = {'x_pred': x_pred, 't': torch.arange(1, n_steps + 1) * dt, 'error': error}
    return {k: v.cpu().numpy() for k, v in res.items()}
