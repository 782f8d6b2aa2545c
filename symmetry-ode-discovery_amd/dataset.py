"""Input contract of the ODE tasks (reference dataset.py:161-200, 16-57).

Files: ``./data/{ode}-{mode}-noise{NN}[-{smoothing}]-{x,dx}.pt`` holding float32 tensors of
shape (n_ics, n_steps, d); the dataset flattens them to (n_ics*n_steps, d) points.  When the
files are absent (the authors distribute them out of band) trajectories are synthesised with the
RK4 restatement in data.py using the README recipe sizes, and written under ./data like the
reference's fallback (dataset.py:178-186), following its generators' recipe (data.gen_data): noise relative to each
dimension's spread, then forward differences of the noisy series or, with ``--smoothing gp``, GP regression
(data.gp_smooth = the reference's num_diff_gp, one Cholesky solve per dimension).

Reaction-diffusion tasks (``rd``, ``mt_rd``; dataset.py:59-159) read ``./data/reaction_diffusion.mat`` (fields t, x, y,
uf, duf -- the SINDy-autoencoder example file, distributed out of band) when it exists; otherwise a rigidly rotating
spiral wave of the same layout is synthesised (``synthetic_spiral``), whose snapshots live on a two-dimensional
manifold with harmonic latent dynamics.  Multi-timestep ODE tasks (``mt_lv``, ``mt_selkov``) window the ODE files.
"""
from __future__ import annotations

import os

import numpy as np
import torch
from torch.utils.data import Dataset

from . import data as synth

data_path = './data'

ode_dt_dict = {'lv': 0.002, 'selkov': 0.002, 'dosc': 0.2, 'growth': 0.02, 'rd': 0.05}      # dataset.py:161-167

# README data recipes: (n_ics train, n_ics val, num_steps, subsample_rate, dt)
_RECIPES = {'dosc': (50, 10, 10000, 100, 0.002), 'growth': (100, 20, 1000, 10, 0.002),
            'lv': (200, 20, 10000, 1, 0.002), 'selkov': (10, 5, 10000, 1, 0.002)}


class DeviceBatches:
    """Drop-in for ``DataLoader(dataset, batch_size, shuffle)`` over an in-memory data set: the arrays are moved to
    ``device`` once and every batch is ONE index gather there (a fresh permutation per epoch when shuffling, last
    batch short), instead of batch_size Python ``__getitem__`` calls + collation + an H2D copy per batch -- which is
    what bounds the reference's loaders at batch sizes like 8192 (70 ms per batch for the multi-timestep LV set).
    ``window`` > 0: item i is rows i .. i+window-1 (the multi-timestep reaction-diffusion layout)."""

    def __init__(self, arrays, n_items, batch_size, shuffle, device, window=0, limit_bytes=8 << 30):
        total = sum(a.numel() * a.element_size() for a in arrays)
        self.arrays = [a.to(device) if total <= limit_bytes else a.pin_memory() for a in arrays]
        self.device, self.n, self.bs, self.shuffle, self.window = device, n_items, batch_size, shuffle, window

    def __len__(self):
        return (self.n + self.bs - 1) // self.bs

    def __iter__(self):
        dev = self.arrays[0].device
        order = torch.randperm(self.n, device=dev) if self.shuffle else torch.arange(self.n, device=dev)
        for lo in range(0, self.n, self.bs):
            idx = order[lo:lo + self.bs]
            if self.window:
                idx = idx[:, None] + torch.arange(self.window, device=dev)[None, :]
            yield tuple(a[idx].to(self.device, non_blocking=True) for a in self.arrays)


def make_loader(dataset, batch_size, shuffle, device):
    """``DeviceBatches`` for the in-memory data sets of this module, ``DataLoader`` for anything else."""
    if str(device) != 'cpu' and hasattr(dataset, 'device_arrays'):
        arrays, n_items, window = dataset.device_arrays()
        return DeviceBatches(arrays, n_items, batch_size, shuffle, device, window)
    from torch.utils.data import DataLoader
    return DataLoader(dataset, batch_size=batch_size, shuffle=shuffle)


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
            gen_dev = 'cuda' if torch.cuda.is_available() else 'cpu'      # HIP RK4 kernel + rocSOLVER for the GP solve
            x, dx = synth.gen_data(ode_name, n_ics, dt=dt, num_steps=steps, subsample_rate=sub, noise=noise,
                                   multiplicative_noise=(ode_name == 'growth'), smoothing=smoothing, gp_sigma_in=0.1,
                                   seed=0 if 'train' in mode else 1, device=gen_dev)
            x, dx = x.cpu(), dx.cpu()
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

    def device_arrays(self):
        return [self.x, self.dx], len(self.x), 0


class MTODEDataset(Dataset):
    """Windows of ``n_timesteps`` states, ``interval`` samples apart, along every trajectory
    (reference dataset.py:203-241): item -> (x, dx) of shape (n_timesteps, d)."""

    def __init__(self, path=data_path, ode_name='lv', mode='train', n_timesteps=2, interval=10, noise=0.0, smoothing=None):
        super().__init__()
        if n_timesteps < 2:
            raise ValueError('n_timesteps must be greater than 1 for multi-timestep dataset')
        base = ODEDataset(path=path, ode_name=ode_name, mode=mode, noise=noise, smoothing=smoothing)
        self.n_ics, self.n_steps, self.input_dim, self.n_timesteps = base.n_ics, base.n_steps, base.input_dim, n_timesteps
        x = base.x.reshape(base.n_ics, base.n_steps, base.input_dim)
        dx = base.dx.reshape(base.n_ics, base.n_steps, base.input_dim)
        starts = torch.arange(base.n_steps - n_timesteps * interval)
        t_idx = starts[:, None] + interval * torch.arange(n_timesteps)[None, :]          # (windows, n_timesteps)
        self.x = x[:, t_idx].reshape(-1, n_timesteps, base.input_dim)
        self.dx = dx[:, t_idx].reshape(-1, n_timesteps, base.input_dim)

    def __len__(self):
        return len(self.x)

    def __getitem__(self, idx):
        return self.x[idx], self.dx[idx]

    def device_arrays(self):
        return [self.x, self.dx], len(self.x), 0


# synthetic stand-in for reaction_diffusion.mat: grid points per side, snapshots, time step (ode_dt_dict['rd']),
# angular velocity of the spiral, half-width of the square domain
RD_SYNTH = {'n': 128, 'n_samples': 1000, 'dt': 0.05, 'omega': 1.0, 'half_width': 10.0}


def synthetic_spiral(n, n_samples, dt, omega, half_width):
    """Rigidly rotating one-armed spiral u(x, y, t) = tanh(r) cos(theta - r + omega t) and its time derivative,
    in the .mat layout: dict(t (T,1), x (n,1), y (n,1), uf (n,n,T), duf (n,n,T)).  The initial frame is the
    initial condition of the lambda-omega example the reference's file was made from."""
    axis = np.linspace(-half_width, half_width, n)
    X, Y = np.meshgrid(axis, axis)
    r, theta = np.sqrt(X * X + Y * Y), np.arctan2(Y, X)
    t = np.arange(n_samples) * dt
    phase = (theta - r)[:, :, None] + omega * t[None, None, :]
    amp = np.tanh(r)[:, :, None]
    return {'t': t[:, None], 'x': axis[:, None], 'y': axis[:, None], 'uf': amp * np.cos(phase), 'duf': -omega * amp * np.sin(phase)}


def _load_reaction_diffusion(path):
    if os.path.exists(path):
        import scipy.io as sio
        data = sio.loadmat(path)
    else:
        print(f'{path} not found. Synthesising a rotating-spiral field {RD_SYNTH}...')
        data = synthetic_spiral(**RD_SYNTH)
    uf = data['uf'] + 1e-6 * np.random.randn(*data['uf'].shape)                        # dataset.py:66-67, 124-125
    duf = data['duf'] + 1e-6 * np.random.randn(*data['duf'].shape)
    T = data['t'].size
    split = {'train': np.arange(int(.8 * T)), 'val': np.arange(int(.8 * T), int(.9 * T)), 'test': np.arange(int(.9 * T), T)}
    return data, uf, duf, split


class ReactionDiffusionDataset(Dataset):
    """Single snapshots (N = n*n,) of the field and its time derivative (reference dataset.py:59-116)."""

    def __init__(self, path=f'{data_path}/reaction_diffusion.mat', mode='train'):
        data, uf, duf, split = _load_reaction_diffusion(path)
        idx = split[mode]
        self.x = torch.from_numpy(uf[:, :, idx].reshape(-1, len(idx)).T.copy()).float()
        self.dx = torch.from_numpy(duf[:, :, idx].reshape(-1, len(idx)).T.copy()).float()

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, idx):
        return self.x[idx], self.dx[idx], self.dx[idx]


class MultiTimestepReactionDiffusionDataset(Dataset):
    """Item i -> (x, dx) of shape (n_timesteps, N): the n_timesteps consecutive snapshots that end before
    sample i + n_timesteps of the split (reference dataset.py:119-159).  The snapshots are stored once; windows
    are views."""

    def __init__(self, path=f'{data_path}/reaction_diffusion.mat', n_timesteps=2, mode='train'):
        data, uf, duf, split = _load_reaction_diffusion(path)
        idx = split[mode]
        self.n_timesteps = n_timesteps
        self.x = torch.from_numpy(np.transpose(uf[:, :, idx], (2, 0, 1)).reshape(len(idx), -1).copy()).float()
        self.dx = torch.from_numpy(np.transpose(duf[:, :, idx], (2, 0, 1)).reshape(len(idx), -1).copy()).float()

    def __len__(self):
        return max(self.x.shape[0] - self.n_timesteps, 0)

    def __getitem__(self, idx):
        return self.x[idx:idx + self.n_timesteps], self.dx[idx:idx + self.n_timesteps]

    def device_arrays(self):
        return [self.x, self.dx], len(self), self.n_timesteps


def get_dataset(args):
    """reference dataset.py:16-57"""
    task = args['task']
    if task in ('lv', 'selkov', 'dosc', 'growth'):
        tr = ODEDataset(ode_name=task, mode='train', noise=args['noise'], smoothing=args['smoothing'])
        va = ODEDataset(ode_name=task, mode='val', noise=args['noise'], smoothing=args['smoothing'])
        args['input_dim'] = tr[0][0].shape[-1]
    elif task in ('mt_lv', 'mt_selkov'):
        win = {} if task == 'mt_lv' else {'n_timesteps': 2, 'interval': 50}
        tr = MTODEDataset(ode_name=task[3:], mode='train', noise=args['noise'], smoothing=args['smoothing'], **win)
        va = MTODEDataset(ode_name=task[3:], mode='val', noise=args['noise'], smoothing=args['smoothing'], **win)
        args['input_dim'] = tr[0][0].shape[-1]
        args['mt_data'] = True
    elif task == 'rd':
        tr, va = ReactionDiffusionDataset(mode='train'), ReactionDiffusionDataset(mode='val')
        args['input_dim'] = tr[0][0].shape[0]
        args['flatten'] = False
    elif task == 'mt_rd':
        tr, va = MultiTimestepReactionDiffusionDataset(mode='train'), MultiTimestepReactionDiffusionDataset(mode='val')
        args['input_dim'] = tr[0][0].shape[1]
        args['mt_data'] = True
    else:
        raise NotImplementedError
    return tr, va, args
