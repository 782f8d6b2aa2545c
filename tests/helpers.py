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


F11_GENERATORS = {"so2": [[0.0, 1.0], [-1.0, 0.0]], "scaling2": [[2.0, 0.0], [0.0, 1.0]]}


def f11_case(g, tag):
    """One record of tests/golden/f11_lbfgs_noisy.npz (the reference's train_SIGED_lbfgs on noisy, GP-smoothed data) as a dict."""
    d, order, cc = [int(v) for v in g[f"{tag}_cfg"]]
    lr, st_freq, thr, epochs = [float(v) for v in g[f"{tag}_hp"]]
    c = dict(tag=tag, d=d, order=order, constrain_constant=bool(cc), lr=lr, st_freq=int(st_freq), thr=thr, epochs=int(epochs),
             x=t(g[f"{tag}_x"]), dx=t(g[f"{tag}_dx"]), mask_final=g[f"{tag}_mask_final"], Xi_final=g[f"{tag}_Xi_final"],
             loss_hist=g[f"{tag}_loss_hist"], thr_epoch=g[f"{tag}_thr_epoch"], thr_Xi=g[f"{tag}_thr_Xi"],
             thr_mask_before=g[f"{tag}_thr_mask_before"], L=None, trace_Xi=t(g[f"{tag}_trace_Xi"]), trace_mask=t(g[f"{tag}_trace_mask"]))
    if f"{tag}_L" in g.files:
        c.update(L=t(g[f"{tag}_L"]), Q=t(g[f"{tag}_Q"]), beta0=t(g[f"{tag}_init_beta"]), const0=t(g[f"{tag}_init_const"]),
                 use_kron=bool(g[f"{tag}_use_kron"]))
    else:
        c["Xi0"] = t(g[f"{tag}_init_Xi"])
    # the reference's own near-threshold record: live coefficients within 1e-4 of the threshold at its thresholding events
    near = []
    for Xi, m in zip(c["thr_Xi"], c["thr_mask_before"]):
        hit = (np.abs(np.abs(Xi) - np.float32(thr)) < 1e-4) & (m > 0)
        near += [(int(i), int(k)) for i, k in zip(*np.nonzero(hit))]
    c["near"] = near
    wide = []                                        # ... within the trainer's stopping ball (1e-3, train.py:643)
    for Xi, m in zip(c["thr_Xi"], c["thr_mask_before"]):
        hit = (np.abs(np.abs(Xi) - np.float32(thr)) < 1e-3) & (m > 0)
        wide += [(int(i), int(k)) for i, k in zip(*np.nonzero(hit))]
    c["near_wide"] = wide
    # the reference itself from 12 starts one unit in the last place away (tools/gen_golden.py::f11_lbfgs_noisy)
    c["ulp_masks"], c["ulp_logged_epochs"] = g[f"{tag}_ulp_masks"], g[f"{tag}_ulp_logged_epochs"]
    c["ulp_thr_epoch"], c["ulp_last_loss"], c["ulp_first_loss"] = g[f"{tag}_ulp_thr_epoch"], g[f"{tag}_ulp_last_loss"], g[f"{tag}_ulp_first_loss"]
    c["unstable"] = [(int(i), int(k)) for i, k in np.argwhere((c["ulp_masks"] != c["mask_final"][None]).any(axis=0))]
    return c


def f11_oracle_regressor(O, c, cls=None, own_Q=False):
    """Oracle regressor at the recorded start of an f11 case (``own_Q``: keep the oracle's own null-space basis instead of the
    reference's -- beta is then re-expressed in it, the start Xi is the same)."""
    cls = cls or O.OracleRegressor
    if c["L"] is None:
        return cls(c["d"], c["order"], threshold=c["thr"], Xi0=c["Xi0"])
    reg = cls(c["d"], c["order"], L_list=[c["L"]], threshold=c["thr"], constrain_constant=c["constrain_constant"],
              beta0=c["beta0"], const0=c["const0"])
    if own_Q:
        with torch.no_grad():
            reg.beta.copy_(reg.Q.T @ (c["Q"] @ c["beta0"]))          # orthonormal bases of the same subspace
    else:
        reg.Q = c["Q"]
    return reg


def compiled(d, order, flags=0):
    """Is library (d, order, flags) in the built libsymode_hip.so?  (d = 4 only with `make ALL=1`.)  No GPU needed."""
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from symode_amd.engine import load_library
    global _LIB
    try:
        _LIB
    except NameError:
        _LIB = load_library()
    return _LIB.symode_lib_size(int(d), int(order), int(flags)) >= 0


def only_compiled(cases):
    """Filter a parametrize list whose entries start with (d, order, ...)."""
    return [c for c in cases if compiled(c[0], c[1])]


CONFIG2_AE = dict(ae_arch="mlp", input_dim=2, hidden_dim=512, latent_dim=2, n_layers=5, n_comps=2, activation="ReLU",
                  activation_args=[], batch_norm=True, ortho_ae=True)
CONFIG2_GEN = dict(repr="(2,1,2)", group_idx="0", sigma_init=1, gan_st_thres=0.3, keep_center=True, n_comps=2)


def make_config2(S, dev):
    """BASELINE config[2] (lv/noise99_eq_isymreg.cfg) at full size: (x, dx) (20 000, 2), frozen autoencoder, generator."""
    from symode_amd.autoencoder import AutoEncoder
    from symode_amd.lie import LieGenerator
    x, dx = S.data.gen_data("lv", 200, dt=0.002, num_steps=10000, noise=0.99, smoothing="gp", seed=0, device=dev)
    x, dx = x.reshape(-1, 2), dx.reshape(-1, 2)
    rows = torch.randperm(x.shape[0], generator=torch.Generator().manual_seed(0))[:20000].to(dev)
    x, dx = x[rows].contiguous(), dx[rows].contiguous()
    torch.manual_seed(11)
    ae = AutoEncoder(**CONFIG2_AE).to(dev)
    gen = LieGenerator(device=dev, **CONFIG2_GEN).to(dev)
    ae.train()
    with torch.no_grad():
        for k in range(4):
            ae(torch.stack([x[k::4], x[k::4] + 0.1 * dx[k::4]], dim=1))
    ae.eval()
    gen.eval()
    for p in list(ae.parameters()) + list(gen.parameters()):
        p.requires_grad = False
    return x, dx, ae, gen
