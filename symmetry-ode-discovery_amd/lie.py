"""Lie-algebra generator container -- stock PyTorch, the part of the reference's gan.LieGenerator
that the SINDy path consumes: representation-string parsing, the fixed bases, block-diagonal
replication over components (``get_full_basis_list``) and deterministic group elements
(``get_deterministic_group_elems``).  GAN sampling / regularisers / the discriminator belong to
symmetry *discovery* and are out of scope.  Parameter names (``Li.k``, ``sigma.k``,
``struct_const.k``) follow gan.py:63-70 so that reference ``generator.pt`` files load.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


def so(n):
    """Basis of so(n): E_ij - E_ji for j < i                                    (utils.py:16-24)"""
    L = torch.zeros(n * (n - 1) // 2, n, n)
    k = 0
    for i in range(n):
        for j in range(i):
            L[k, i, j], L[k, j, i] = 1.0, -1.0
            k += 1
    return L


def _so3p1():
    L = torch.zeros(3, 4, 4)
    L[:, :3, :3] = so(3)
    return L


# name -> (basis (channels, k, k))                                              gan.py:110-168
FIXED_GROUPS = {
    'so2': lambda: torch.tensor([[[0.0, 1.0], [-1.0, 0.0]]]),
    'sim2': lambda: torch.tensor([[[-0.2, 1.0], [-1.0, 0.0]]]),
    'scaling2': lambda: torch.tensor([[[2.0, 0.0], [0.0, 1.0]]]),
    'so2*r': lambda: torch.tensor([[[0.0, 1.0], [-1.0, 0.0]], [[0.1, 0.0], [0.0, 0.1]]]),
    'so3': lambda: so(3),
    'so3+1': _so3p1,
    'so4': lambda: so(4),
}


def parse_repr(repr_str):
    """'(2,sim2)+(1)' -> [('2','sim2'), ('1',)]                               (gan.py:41-49)"""
    out = []
    for t in repr_str.split('+'):
        t = t.strip()
        if t.startswith('(') and t.endswith(')'):
            out.append(tuple(e.strip() for e in t[1:-1].split(',')))
    return out


class LieGenerator(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        self.repr = kwargs['repr']
        self.sigma_init = kwargs.get('sigma_init', 1)
        self.threshold = kwargs.get('gan_st_thres', 0.3)
        self.keep_center = kwargs.get('keep_center', False)
        self.group_idx = kwargs.get('group_idx', '0').split(',')
        tuples = parse_repr(self.repr)
        if len(self.group_idx) != len(tuples):
            raise ValueError('Number of group indices does not match number of components in representation string.')
        self.Li, self.sigma, self.struct_const = nn.ParameterList(), nn.ParameterList(), nn.ParameterList()
        self.masks, self.n_comps, self.n_channels, self.learnable, self.f_Li = [], [], [], [], []
        self.n_dims = 0
        for i, r in enumerate(tuples):
            if len(r) >= 3:                                  # (n_comps, n_channels, n_dims[, 'o']): learnable
                nc, ch, nd = int(r[0]), int(r[1]), int(r[2])
                if len(r) == 4 and r[3] != 'o':
                    raise ValueError(f'Group {r[3]} not implemented yet.')
                self.f_Li.append((lambda L: L - L.transpose(-1, -2)) if len(r) == 4 else (lambda L: L))
                self._add(torch.randn(ch, nd, nd), True, nc, torch.ones(ch, nd, nd), torch.eye(ch) * self.sigma_init)
                self.n_dims += nd * nc
            elif len(r) == 1:                                # (n,): n untouched scalars
                n = int(r[0])
                self.f_Li.append(lambda L: L)
                self._add(torch.zeros(1, n, n), False, 1, None, torch.eye(1))
                self.n_dims += n
            elif len(r) == 2:                                # (n_comps, name): fixed basis
                nc, name = int(r[0]), r[1]
                if name not in FIXED_GROUPS:
                    raise ValueError(f'Group {name} not implemented yet.')
                basis = FIXED_GROUPS[name]()
                self.f_Li.append(lambda L: L)
                self._add(basis, False, nc, None, torch.eye(basis.shape[0]) * self.sigma_init)
                self.n_dims += nc * basis.shape[-1]
            else:
                raise ValueError(f'Invalid representation string at position {i}: {r}')
        dev = kwargs.get('device', 'cpu')
        self.masks = [m.to(dev) if m is not None else None for m in self.masks]

    def _add(self, basis, learnable, n_comps, mask, sigma):
        ch = basis.shape[0]
        self.Li.append(nn.Parameter(basis, requires_grad=learnable))
        self.struct_const.append(nn.Parameter(torch.zeros(ch, ch, ch), requires_grad=learnable))
        self.sigma.append(nn.Parameter(sigma, requires_grad=False))
        self.masks.append(mask)
        self.n_comps.append(n_comps)
        self.n_channels.append(ch)
        self.learnable.append(learnable)

    def get_full_basis_list(self, split_channel=True):
        """Generators replicated block-diagonally over components; tuples that share a group index
        are summed into one (n_dims, n_dims) generator per channel.                (gan.py:306-330)"""
        start, groups = 0, {idx: [] for idx in self.group_idx}
        for Li, f, gidx, mask, nc, learnable in zip(self.Li, self.f_Li, self.group_idx, self.masks, self.n_comps, self.learnable):
            if learnable and mask is not None:
                Li = f(Li) * mask
            comp = 0
            for _ in range(nc):
                end = start + Li.shape[1]
                comp = comp + F.pad(Li, (start, self.n_dims - end, start, self.n_dims - end))
                start = end
            groups[gidx].append(comp)
        v = []
        for idx in groups:
            total = sum(groups[idx])
            v += [ch for ch in total] if split_channel else [total]
        return v

    def get_deterministic_group_elems(self, split_channel=False, scale=1.0):
        """exp(sigma * L * scale) per (sigma, un-split basis) pair                  (gan.py:332-348)"""
        g_list = []
        for sigma, L in zip(self.sigma, self.get_full_basis_list(split_channel=split_channel)):
            if len(L.shape) == 3:
                g_list += [torch.matrix_exp(sigma * Li * scale) for Li in L]
            else:
                g_list.append(torch.matrix_exp(sigma * L * scale))
        return g_list

    def set_threshold(self, threshold):                                           # gan.py:269-276
        for Li, f, mask in zip(self.Li, self.f_Li, self.masks):
            if mask is None:
                continue
            mx = torch.amax(torch.abs(f(Li)), dim=(1, 2), keepdim=True)
            mask.data = torch.logical_and(torch.abs(f(Li)) > threshold * mx, mask).float()
