"""Input contract of the ODE tasks (reference dataset.py:161-200, 16-57).

Files: ``./data/{ode}-{mode}-noise{NN}[-{smoothing}]-{x,dx}.pt`` holding float32 tensors of
shape (n_ics, n_steps, d); the dataset flattens them to (n_ics*n_steps, d) points.  When the
files are absent (the authors distribute them out of band) trajectories are synthesised with the
RK4 restatement in data.py using the README recipe sizes, and written under ./data like the
reference's fallback (dataset.py:178-186).  GP smoothing is not reproduced: synthesised noisy data
keep the exact derivative of the clean orbit.
"""
from __future__ import annotations

import os

import torch
from torch.utils.data import Dataset

from . import data as synth

data_path = './data'

ode_dt_dict = {'lv': 0.002, 'selkov': 0.002, 'dosc': 0.2, 'growth': 0.02, 'rd': 0.05}      # dataset.py:161-167

# README data recipes: (n_ics train, n_ics val, num_steps, subsample_rate, dt)
_RECIPES = {'dosc': (50, 10, 10000, 100, 0.002), 'growth': (100, 20, 1000, 10, 0.002),
            'lv': (200, 20, 10000, 1, 0.002), 'selkov': (10, 5, 10000, 1, 0.002)}


class ODEDataset(Dataset):
    def __init__(self, path=data_path, ode_name='lv', mode='train', noise=0.0, smoothing=None):
        super().__init__()
        sm = f'-{smoothing}' if smoothing is not None else ''
        stem = f'{path}/{ode_name}-{mode}-noise{int(100 * noise):02d}{sm}'
        try:
            print(f'Loading existing {ode_name} {mode} data...')
            x = torch.load(f'{stem}-x.pt', weights_only=True)
            dx = torch.load(f'{stem}-dx.pt', weights_only=True)
        except FileNotFoundError:
            print(f'Load data failed. Generating {ode_name} {mode} data...')
            n_tr, n_va, steps, sub, dt = _RECIPES[ode_name]
            n_ics = n_tr if 'train' in mode else n_va
            xs, dxs = synth.make_dataset(ode_name, n_ics, steps, dt=dt, noise=noise, seed=0 if 'train' in mode else 1)
            d = xs.shape[-1]
            x = xs.reshape(n_ics, steps, d)[:, ::sub].contiguous()
            dx = dxs.reshape(n_ics, steps, d)[:, ::sub].contiguous()
            os.makedirs(path, exist_ok=True)
            torch.save(x, f'{stem}-x.pt')
            torch.save(dx, f'{stem}-dx.pt')
        x, dx = x.to(torch.float32), dx.to(torch.float32)
        self.n_ics, self.n_steps, self.input_dim = x.shape
        self.x = x.reshape(self.n_ics * self.n_steps, self.input_dim)           # dataset.py:193-194
        self.dx = dx.reshape(self.n_ics * self.n_steps, self.input_dim)

    def __len__(self):
        return len(self.x)

    def __getitem__(self, idx):
        return self.x[idx], self.dx[idx]


def get_dataset(args):
    task = args['task']
    if task in ('lv', 'selkov', 'dosc', 'growth'):
        tr = ODEDataset(ode_name=task, mode='train', noise=args['noise'], smoothing=args['smoothing'])
        va = ODEDataset(ode_name=task, mode='val', noise=args['noise'], smoothing=args['smoothing'])
        args['input_dim'] = tr[0][0].shape[-1]
        return tr, va, args
    raise NotImplementedError(f"task {task!r}: only the ODE tasks (lv, selkov, dosc, growth) are on the MI355X path; "
                              "reaction-diffusion / multi-timestep discovery stay with the reference on stock PyTorch")
