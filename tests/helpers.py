"""Shared test helpers (test infrastructure, not product code)."""
import numpy as np
import torch


def t(a, dtype=torch.float32):
    return torch.from_numpy(np.asarray(a)).to(dtype)


class TinyAE:
    """Rebuilds the fixture autoencoder (stock MLP + eval-mode BatchNorm) from golden arrays.

    The reference AE is ``Linear -> BatchNorm1d(eval) -> act`` per hidden layer with a final
    ``Linear -> BatchNorm1d(eval)`` in the encoder and ``Linear -> act`` / final ``Linear`` in
    the decoder, applied per component (last axis).  Eval-mode BatchNorm is the affine map
    ``(h - mean) / sqrt(var + eps) * w + b``.
    """

    def __init__(self, g, prefix, activation):
        self.act = {"ReLU": torch.relu, "Tanh": torch.tanh}[activation]
        self.enc, self.dec = self._load(g, prefix, "enc"), self._load(g, prefix, "dec")
        self.z_mean = t(g[f"{prefix}_zmean"])

    @staticmethod
    def _load(g, prefix, tag):
        layers = []
        for k in range(int(g[f"{prefix}_{tag}_n"])):
            L = {"W": t(g[f"{prefix}_{tag}_W{k}"]), "b": t(g[f"{prefix}_{tag}_b{k}"])}
            if f"{prefix}_{tag}_bnw{k}" in g.files:
                L.update(bnw=t(g[f"{prefix}_{tag}_bnw{k}"]), bnb=t(g[f"{prefix}_{tag}_bnb{k}"]),
                         bnm=t(g[f"{prefix}_{tag}_bnm{k}"]), bnv=t(g[f"{prefix}_{tag}_bnv{k}"]),
                         eps=float(g[f"{prefix}_{tag}_bneps{k}"]))
            layers.append(L)
        return layers

    def to(self, device):
        for layers in (self.enc, self.dec):
            for L in layers:
                for k, v in L.items():
                    if isinstance(v, torch.Tensor):
                        L[k] = v.to(device)
        self.z_mean = self.z_mean.to(device)
        return self

    def _run(self, layers, h):
        n = len(layers)
        for i, L in enumerate(layers):
            h = torch.nn.functional.linear(h, L["W"], L["b"])
            if "bnw" in L:
                h = (h - L["bnm"]) / torch.sqrt(L["bnv"] + L["eps"]) * L["bnw"] + L["bnb"]
            if i < n - 1:
                h = self.act(h)
        return h

    def encode(self, x):
        return self._run(self.enc, x)

    def decode(self, z):
        return self._run(self.dec, z)


def projector(Q):
    Q = torch.as_tensor(Q, dtype=torch.float64)
    return Q @ torch.linalg.pinv(Q)
