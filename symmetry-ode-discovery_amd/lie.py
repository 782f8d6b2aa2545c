"""Lie-algebra generator and discriminator of the LaLiGAN loop -- stock PyTorch, no HIP code
(north_star keeps the autoencoder / GAN side on stock PyTorch-ROCm).

``LieGenerator``: representation-string parsing, the fixed bases, block-diagonal replication over
components (``get_full_basis_list``), deterministic group elements
(``get_deterministic_group_elems``) -- what the SINDy path consumes -- plus what ``train_lassi``
needs: random group elements acting on the latent batch (``forward``), the norm / orthogonality /
closure regularisers and sequential thresholding of the learned basis.  ``Discriminator``: the MLP
critic.  Parameter names (``Li.k``, ``sigma.k``, ``struct_const.k``, ``model.*``) follow gan.py:63-70, 395-404
so that reference ``generator.pt`` / ``discriminator.pt`` files load.
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


class IntParameter(nn.Module):
    """Noisy rounding of a learned basis to the integer grid {-k..k}              (gan.py:9-17)"""

    def __init__(self, k=2, noise=0.1):
        super().__init__()
        self.k, self.noise = k, noise

    def forward(self, data):
        jitter = torch.randn_like(data) * self.noise
        return torch.round(torch.clamp(self.k * (data + jitter), -self.k - 0.49, self.k + 0.49))


def _unit_frobenius(L):
    """every channel of L (k, n, n) scaled to unit Frobenius norm (+1e-6 guard)   (gan.py:224-225, 237-238)"""
    return L / (torch.sqrt(torch.einsum('kdf,kdf->k', L, L))[:, None, None] + 1e-6)


class LieGenerator(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        self.repr = kwargs['repr']
        self.sigma_init = kwargs.get('sigma_init', 1)
        self.threshold = kwargs.get('gan_st_thres', 0.3)
        self.keep_center = kwargs.get('keep_center', False)
        self.group_idx = kwargs.get('group_idx', '0').split(',')
        self.uniform_max = kwargs.get('uniform_max', 1)
        self.coef_dist = kwargs.get('coef_dist', 'normal')
        self.g_init = kwargs.get('g_init', 'random')
        self.task = kwargs.get('task')
        self.int_param = kwargs.get('int_param', False)
        self.int_param_approx = IntParameter(k=kwargs.get('int_param_max', 2), noise=kwargs.get('int_param_noise', 0.1))
        self.activated_channel = None                        # None = every channel
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
        by_group = {}
        for i, idx in enumerate(self.group_idx):             # tuples sharing a group index share coefficients
            by_group.setdefault(idx, []).append(i)
        for idx, members in by_group.items():
            if any(self.n_channels[i] != self.n_channels[members[0]] for i in members):
                raise ValueError(f'Group index {idx} contains channels of different dimensions.')
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

    # ---- LaLiGAN side (train_lassi) ------------------------------------------------------------
    def set_activated_channel(self, ch):
        self.activated_channel = ch

    def activate_all_channels(self):
        self.activated_channel = None

    def _learned(self):
        """(f(Li) * mask, struct_const) of every learnable tuple."""
        return [(f(Li) * mask, c) for Li, f, c, mask, learnable
                in zip(self.Li, self.f_Li, self.struct_const, self.masks, self.learnable) if learnable]

    def reg_norm(self):
        """sum_k max(0, 1/2 - ||L_k||_F^2): keeps learned channels from collapsing     (gan.py:212-217)"""
        s = 0.0
        for L, _ in self._learned():
            s = s + torch.clamp(0.5 - torch.einsum('kdf,kdf->k', L, L), min=0.0).sum()
        return s

    def reg_ortho(self):
        """squared cosines between distinct unit-norm channels                         (gan.py:219-227)"""
        s = 0.0
        for L, _ in self._learned():
            U = _unit_frobenius(L)
            s = s + torch.triu(torch.einsum('bij,cij->bc', U, U), diagonal=1).square().sum()
        return s

    def reg_closure(self):
        """|| [L_i, L_j] - sum_k c_ijk L_k ||^2 over i < j, unit-norm channels          (gan.py:229-242)"""
        s = 0.0
        for L, c in self._learned():
            U = _unit_frobenius(L)
            for i in range(U.shape[0]):
                for j in range(i + 1, U.shape[0]):
                    bracket = U[i] @ U[j] - U[j] @ U[i]
                    s = s + (bracket - torch.einsum('k,kij->ij', c[i, j], U)).square().sum()
        return s

    def sample_coefficient(self, batch_size, n_channels, params, device):           # gan.py:350-366
        if self.coef_dist == 'normal':
            z = torch.randn(batch_size, n_channels, device=device) @ params
        elif self.coef_dist == 'uniform':
            z = torch.rand(batch_size, n_channels, device=device) * 2 * params - params
        elif self.coef_dist == 'uniform_int_grid':
            z = torch.randint(-int(params), int(params), (batch_size, n_channels), device=device, dtype=torch.float32)
        else:
            raise ValueError(f'Unknown coefficient distribution: {self.coef_dist}')
        if self.activated_channel is not None:
            keep = torch.zeros_like(z)
            keep[:, self.activated_channel] = 1
            z = z * keep
        return z

    def sample_group_element(self, batch_size, device):
        """(B, n_dims, n_dims): exp(sum_k z_k L_k) per tuple, block-diagonal over components; one coefficient
        draw per group index, in order of first appearance                           (gan.py:278-304)"""
        coef = {}
        for i, idx in enumerate(self.group_idx):
            if idx not in coef:
                coef[idx] = self.sample_coefficient(batch_size, self.n_channels[i], self.sigma[i], device)
        g, start = 0, 0
        for Li, f, idx, mask, nc, learnable in zip(self.Li, self.f_Li, self.group_idx, self.masks, self.n_comps, self.learnable):
            if learnable and self.int_param:
                Li = self.int_param_approx(f(Li))
            if learnable and mask is not None:
                Li = f(Li) * mask
            block = torch.matrix_exp(torch.einsum('bj,jkl->bkl', coef[idx], Li))
            for _ in range(nc):
                end = start + block.shape[1]
                g = g + F.pad(block, (start, self.n_dims - end, start, self.n_dims - end))
                start = end
        return g

    def forward(self, x):
        """A random group element per sample acting on x (B, *, n_dims) (components flattened).  (gan.py:244-262)"""
        if not self.keep_center:
            centre = x.mean(dim=list(range(x.dim() - 1)), keepdim=True)
            x = x - centre
        shape = x.shape
        flat = x.reshape(shape[0], -1) if x.dim() == 3 else x
        out = torch.einsum('bij,bj->bi', self.sample_group_element(shape[0], x.device), flat).reshape(shape)
        return out if self.keep_center else out + centre

    def infinitesimal_transform(self, x, L_idx):                                    # gan.py:264-281
        if not self.keep_center:
            x = x - x.mean(dim=list(range(x.dim() - 1)), keepdim=True)
        shape = x.shape
        flat = x.reshape(shape[0], -1) if x.dim() == 3 else x
        return torch.einsum('ij,bj->bi', self.get_full_basis_list()[L_idx], flat).reshape(shape)

    def getLi(self):
        return self.get_full_basis_list(split_channel=False)

    def getStructureConst(self):
        return [c.reshape(-1, c.shape[-1]) for c, learnable in zip(self.struct_const, self.learnable) if learnable]

    def set_threshold(self, threshold):                                           # gan.py:269-276
        for Li, f, mask in zip(self.Li, self.f_Li, self.masks):
            if mask is None:
                continue
            mx = torch.amax(torch.abs(f(Li)), dim=(1, 2), keepdim=True)
            mask.data = torch.logical_and(torch.abs(f(Li)) > threshold * mx, mask).float()


class Discriminator(nn.Module):
    """MLP critic on the flattened latent (optionally with the decoded input and an invariant label);
    slot layout of ``model`` as gan.py:395-404 so that ``discriminator.pt`` files load."""

    def __init__(self, latent_dim, n_comps, hidden_dim, n_layers, activation='ReLU', **kwargs):
        super().__init__()
        width = latent_dim * n_comps
        if kwargs.get('use_original_x'):
            width += kwargs['input_dim'] * n_comps
        self.embed_y = False
        if kwargs.get('use_invariant_y'):
            self.embed_y = bool(kwargs.get('embed_y'))
            if self.embed_y:
                self.y_embedding = nn.Embedding(kwargs['y_classes'], kwargs['y_embed_dim'])
                width += kwargs['y_embed_dim']
            else:
                width += kwargs['y_dim']
        self.input_dim = width
        act = getattr(nn, activation)
        self.model = nn.Sequential(
            nn.Linear(width, hidden_dim), act(),
            *[nn.Sequential(nn.Linear(hidden_dim, hidden_dim), act()) for _ in range(n_layers - 1)],
            nn.Linear(hidden_dim, 1), nn.Sigmoid())

    def forward(self, z, y=None, x=None):
        parts = [z.reshape(z.shape[0], -1)]
        if y is not None:
            parts.append(self.y_embedding(y) if self.embed_y else y)
        if x is not None:
            parts.append(x.reshape(x.shape[0], -1))
        return self.model(torch.cat(parts, dim=-1) if len(parts) > 1 else parts[0])
