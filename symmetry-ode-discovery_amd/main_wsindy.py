"""Weak-SINDy driver (reference main_wsindy.py:18-80): a random 80 % sub-trajectory of one initial
condition, trigonometric test functions, regularised least squares + thresholding."""
from __future__ import annotations

import os

import numpy as np
import torch

from . import train as T
from .dataset import get_dataset, ode_dt_dict
from .evaluation import eval_sindy_regressor, sindy_truth
from .parser_utils import get_args
from .sindy import SINDyRegression, WSINDyWrapper


def main(argv=None):
    args = get_args(argv=argv)
    seed = args.seed
    torch.manual_seed(seed)
    np.random.seed(seed)
    args = vars(args)
    train_dataset, _, args = get_dataset(args)
    n_ics, n_steps = train_dataset.n_ics, train_dataset.n_steps
    train_x = train_dataset.x.reshape(n_ics, n_steps, -1)
    start = np.random.randint(0, n_steps - int(0.8 * n_steps))
    traj = np.random.randint(0, n_ics)
    train_x = train_x[traj, start:start + int(0.8 * n_steps)]
    n_steps = int(0.8 * n_steps)
    dt = ode_dt_dict[args['task']]
    t = torch.arange(n_steps) * dt
    regressor = SINDyRegression(**args).to(args['device'])
    wrapper = WSINDyWrapper(regressor, t, n_steps * dt, **args)
    T.train_WSINDy(wrapper=wrapper, train_x=train_x, **args)
    out = f'saved_models/{args["save_dir"]}'
    os.makedirs(out, exist_ok=True)
    torch.save(regressor.state_dict(), f'{out}/regressor.pt')
    coef, cf, mse, cf_all, mse_all = eval_sindy_regressor(regressor, sindy_truth[args['task']])
    print(f'Correct form: {cf}')
    os.makedirs(f'eval_results/{args["save_dir"]}', exist_ok=True)
    np.savez(f'eval_results/{args["save_dir"]}/seed{seed}.npz', coefficients=coef, correct_form=cf, mse=mse,
             correct_form_all=cf_all, mse_all=mse_all)
    return regressor


if __name__ == '__main__':
    main()
