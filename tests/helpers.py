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


def load_fixture_autoencoder(g, prefix, activation, device="cpu"):
    """Instantiate the product's stock AutoEncoder and fill it with the golden fixture's weights
    (same module walk order as tools/gen_golden.py::_ae_arrays)."""
    from symode_amd.autoencoder import AutoEncoder
    hidden = g[f"{prefix}_enc_W0"].shape[0]
    n_layers = int(g[f"{prefix}_enc_n"]) - 1
    ae = AutoEncoder(ae_arch="mlp", input_dim=2, hidden_dim=hidden, latent_dim=2, n_layers=n_layers, n_comps=2,
                     activation=activation, activation_args=[], batch_norm=True, ortho_ae=False)
    with torch.no_grad():
        for tag, seq in (("enc", ae.encoder), ("dec", ae.decoder)):
            k = 0
            for m in seq.modules():
                if isinstance(m, torch.nn.Linear):
                    m.weight.copy_(t(g[f"{prefix}_{tag}_W{k}"]))
                    m.bias.copy_(t(g[f"{prefix}_{tag}_b{k}"]))
                    k += 1
                elif isinstance(m, torch.nn.BatchNorm1d):
                    j = k - 1
                    m.weight.copy_(t(g[f"{prefix}_{tag}_bnw{j}"]))
                    m.bias.copy_(t(g[f"{prefix}_{tag}_bnb{j}"]))
                    m.running_mean.copy_(t(g[f"{prefix}_{tag}_bnm{j}"]))
                    m.running_var.copy_(t(g[f"{prefix}_{tag}_bnv{j}"]))
    for p in ae.parameters():
        p.requires_grad = False
    return ae.to(device).eval()


def load_fixture_generator(g, prefix, repr_str, device="cpu"):
    from symode_amd.lie import LieGenerator
    gen = LieGenerator(repr=repr_str, group_idx="0", sigma_init=1, gan_st_thres=0.3, keep_center=True, device=device)
    with torch.no_grad():
        gen.Li[0].copy_(t(g[f"{prefix}_Li"]))
        gen.sigma[0].copy_(t(g[f"{prefix}_sigma"]))
    for p in gen.parameters():
        p.requires_grad = False
    return gen.to(device).eval()
