"""Autoencoder used by the symmetry regularisers -- stock PyTorch-ROCm, NOT part of the HIP path.

north_star keeps the autoencoder / latent path on stock PyTorch; this module exists so that the
drivers run and that checkpoints written by the reference (``saved_models/*/autoencoder.pt``)
load: the ``nn.Sequential`` slot layout of encoder / decoder (Linear, optional flatten +
BatchNorm1d + unflatten, activation, ...) reproduces autoencoder.py:36-66 of the reference so
the state_dict keys agree.  Architectures: 'mlp', 'mlp_split' (two independent MLPs on the two halves of the
last axis, reference model.py:17-70) and 'none' (the CNN names in the reference point at classes that do not exist there).
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.autograd.functional import jvp
from torch.nn.utils.parametrizations import orthogonal


class Reshape(nn.Module):
    def __init__(self, *shape):
        super().__init__()
        self.shape = shape

    def forward(self, x):
        return x.reshape(self.shape)


def _bn_slots(width, batch_norm, n_comps):
    """[flatten, BatchNorm1d, unflatten] slots (Identity when unused) -- three slots either way."""
    wrap = batch_norm and n_comps > 1
    return [Reshape(-1, width) if wrap else nn.Identity(),
            nn.BatchNorm1d(width) if batch_norm else nn.Identity(),
            Reshape(-1, n_comps, width) if wrap else nn.Identity()]


def _encoder_layers(kwargs):
    din, dh, dz = kwargs['input_dim'], kwargs['hidden_dim'], kwargs['latent_dim']
    nl, nc, bn = kwargs['n_layers'], kwargs['n_comps'], kwargs['batch_norm']
    act = lambda: getattr(nn, kwargs['activation'])(*kwargs.get('activation_args', []))  # noqa: E731
    last = nn.Linear(dh, dz)
    return nn.Sequential(
        nn.Linear(din, dh), *_bn_slots(dh, bn, nc), act(),
        *[nn.Sequential(nn.Linear(dh, dh), *_bn_slots(dh, bn, nc), act()) for _ in range(nl - 1)],
        orthogonal(last) if kwargs.get('ortho_ae') else last, *_bn_slots(dz, bn, nc))


def _decoder_layers(kwargs):
    din, dh, dz, nl = kwargs['input_dim'], kwargs['hidden_dim'], kwargs['latent_dim'], kwargs['n_layers']
    act = lambda: getattr(nn, kwargs['activation'])(*kwargs.get('activation_args', []))  # noqa: E731
    return nn.Sequential(
        nn.Linear(dz, dh), act(),
        *[nn.Sequential(nn.Linear(dh, dh), act()) for _ in range(nl - 1)],
        nn.Linear(dh, din))


class _Layers(nn.Module):
    """``layers`` attribute around a Sequential: the state_dict prefix of the reference's EncoderMLP / DecoderMLP."""

    def __init__(self, layers):
        super().__init__()
        self.layers = layers

    def forward(self, x):
        return self.layers(x)


class SplitModel(nn.Module):
    """Two independent copies of a model on the two halves of the last axis (reference model.py:62-70)."""

    def __init__(self, make, **kwargs):
        super().__init__()
        self.model1, self.model2 = _Layers(make(kwargs)), _Layers(make(kwargs))

    def forward(self, x):
        x1, x2 = torch.split(x, x.shape[-1] // 2, dim=-1)
        return torch.cat([self.model1(x1), self.model2(x2)], dim=-1)


class AutoEncoder(nn.Module):
    """x (B, n_comps, input_dim) -> z (B, n_comps, latent_dim) -> xhat; kwargs as in the reference."""

    def __init__(self, **kwargs):
        super().__init__()
        arch = kwargs['ae_arch']
        if arch == 'none':
            self.encoder, self.decoder = nn.Identity(), nn.Identity()
        elif arch == 'mlp':
            self.encoder, self.decoder = _encoder_layers(kwargs), _decoder_layers(kwargs)
        elif arch == 'mlp_split':
            self.encoder, self.decoder = SplitModel(_encoder_layers, **kwargs), SplitModel(_decoder_layers, **kwargs)
        else:
            raise NotImplementedError(f"ae_arch={arch!r}: 'mlp', 'mlp_split' and 'none' are provided")

    def forward(self, x):
        z = self.encode(x)
        return z, self.decode(z)

    def encode(self, x):
        return self.encoder(x)

    def decode(self, z):
        return self.decoder(z)

    def compute_dz(self, x, dx):                       # autoencoder.py:102-104
        return jvp(self.encode, x, v=dx)[1]

    def compute_dx(self, z, dz):                       # autoencoder.py:106-108
        return jvp(self.decode, z, v=dz)[1]
