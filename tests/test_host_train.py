"""Host-side surface on CPU: argument parsing, stock modules, and the trainer's control flow
driven by the oracle-backed test engine against the reference's recorded L-BFGS runs."""
import os

import numpy as np
import pytest
import torch

import symode_amd
from symode_amd import parser_utils
from symode_amd.sindy import SINDyRegression
from symode_amd.train import train_SIGED_lbfgs, train_SINDy
from tests.helpers import TinyAE, load_fixture_autoencoder, load_fixture_generator, t
from tests.oracle_engine import OracleEngine

torch.set_num_threads(4)


# ------------------------------------------------------------------------------- parser
def test_parser_defaults_and_flag_names():
    a = parser_utils.get_args(argv=[])
    assert (a.task, a.batch_size, a.num_epochs, a.lr_sindy, a.poly_order, a.st_freq, a.threshold) == ("rd", 256, 1000, 1e-3, 2, 100, 0.1)
    assert (a.sindy_optimizer, a.lbfgs_subsample, a.sym_reg_type, a.w_sym_reg, a.int_t, a.int_dt) == ("adam", 1.0, "i", 0.0, 0.1, 0.01)
    assert (a.repr, a.group_idx, a.ae_arch, a.hidden_dim, a.n_layers, a.seed, a.gpu) == ("(1,so2)", "0", "mlp", 512, 5, 42, 0)
    assert a.eq_constraint is False and a.activation_args == [] and str(a.device) in ("cpu", "cuda:0")
    assert len(parser_utils._MAIN_ARGS) == 79       # the reference parser's 76 flags (+ --config, --help = its 78 actions) + --lstsq_driver, --device_lbfgs, --torch_lbfgs
    assert a.lstsq_driver is None                   # = torch.linalg.lstsq's default on the device the data lives on
    s = parser_utils.get_sindy_args(argv=[])
    assert (s.lr, s.reg_type, s.w_reg, s.seq_thres_freq, s.batch_size, s.save_dir) == (1e-3, "l1", 0.1, 100, 64, "sindy-test")


def test_config_file_precedence(tmp_path, monkeypatch):
    cfg_dir = tmp_path / "run_configs" / "dosc"
    cfg_dir.mkdir(parents=True)
    (cfg_dir / "x.cfg").write_text("--task dosc\n--noise 0.2 --smoothing gp\n--lr_sindy 0.1\n--poly_order 2\n--print_eq\n--threshold 5e-2")
    monkeypatch.chdir(tmp_path)
    a = parser_utils.get_args(argv=["--config", "dosc/x.cfg", "--seed", "3", "--lr_sindy", "0.5", "--poly_order", "2"])
    assert a.task == "dosc" and a.noise == 0.2 and a.smoothing == "gp" and a.print_eq is True and a.threshold == 0.05
    assert a.seed == 3 and a.lr_sindy == 0.5                   # CLI value differs from the default -> wins
    b = parser_utils.get_args(argv=["--config", "dosc/x.cfg", "--lr_sindy", "0.001"])
    assert b.lr_sindy == 0.1                                   # CLI value == parser default -> the file wins (reference rule)
    assert parser_utils.parse_config(str(cfg_dir / "x.cfg"))[:4] == ["--task", "dosc", "--noise", "0.2"]


# ------------------------------------------------------------------------ stock modules
@pytest.mark.parametrize("tag,act,rep", [("relu_sim2", "ReLU", "(2,sim2)"), ("tanh_learn", "Tanh", "(2,1,2)")])
def test_stock_autoencoder_and_generator_match_fixture(golden, tag, act, rep):
    g = golden("f6_symreg")
    ae = load_fixture_autoencoder(g, tag, act)
    ref = TinyAE(g, tag, act)
    x = torch.randn(9, 2, 2)
    assert torch.allclose(ae.encode(x), ref.encode(x), atol=1e-6) and torch.allclose(ae.decode(x), ref.decode(x), atol=1e-6)
    assert torch.allclose(ae.encoder[-2].bias, ref.z_mean)
    gen = load_fixture_generator(g, tag, rep)
    assert np.allclose(torch.stack(gen.get_full_basis_list()).numpy(), g[f"{tag}_basis"])
    for scale, key in [(1.0, "gelems"), (0.01, "gelems_r")]:
        ge = torch.stack([e.reshape(e.shape[-2:]) for e in gen.get_deterministic_group_elems(scale=scale)])
        assert np.allclose(ge.numpy(), g[f"{tag}_{key}"], rtol=1e-6, atol=1e-7)
    assert set(gen.state_dict().keys()) == {"Li.0", "sigma.0", "struct_const.0"}
    keys = set(ae.state_dict().keys())
    assert {"encoder.0.weight", "encoder.2.running_mean", "encoder.5.0.weight", "decoder.0.weight"} <= keys


def test_generator_fixed_groups_and_repr_parsing():
    from symode_amd.lie import LieGenerator, parse_repr
    assert parse_repr("(2,sim2)+(1)") == [("2", "sim2"), ("1",)]
    g = LieGenerator(repr="(1,so2)", group_idx="0")
    assert torch.equal(g.get_full_basis_list()[0], torch.tensor([[0.0, 1.0], [-1.0, 0.0]]))
    g3 = LieGenerator(repr="(1,so3)", group_idx="0")
    assert len(g3.get_full_basis_list()) == 3 and g3.n_dims == 3
    with pytest.raises(ValueError):
        LieGenerator(repr="(1,so2)+(1,so2)", group_idx="0")


# ------------------------------------------------------------------------------ trainer
def _regressor(g, tag, d, order, thr):
    if f"{tag}_init_Xi" in g.files:
        r = SINDyRegression(d, order, False, False, threshold=thr, device="cpu", engine=OracleEngine())
        r.Xi.data = t(g[f"{tag}_init_Xi"])
    else:
        r = SINDyRegression(d, order, False, False, L_list=[torch.tensor([[0.0, 1.0], [-1.0, 0.0]])], threshold=thr,
                            device="cpu", constrain_constant=False, engine=OracleEngine())
        r.Q = t(g[f"{tag}_Q"])
        r.beta.data, r.const.data = t(g[f"{tag}_init_beta"]), t(g[f"{tag}_init_const"])
    return r


@pytest.mark.parametrize("tag", ["dosc_sindy", "dosc_esindy", "selkov_sindy"])
def test_lbfgs_trainer_reproduces_reference_run(golden, tag, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    g = golden("f4_lbfgs")
    d, order = [int(v) for v in g[f"{tag}_cfg"]]
    lr, st_freq, thr, epochs = g[f"{tag}_hp"]
    x, dx = t(g[f"{tag}_x"]), t(g[f"{tag}_dx"])
    r = _regressor(g, tag, d, order, float(thr))
    ident = torch.nn.Identity()
    train_SIGED_lbfgs(train_loader=[(x, dx)], test_loader=[], num_epochs=int(epochs), device="cpu", log_interval=10 ** 9,
                      save_interval=10 ** 9, save_dir="t", autoencoder=ident, generator=ident, regressor=r,
                      regressor_dst=None, use_latent=False, distill_latent=False, lr_sindy=float(lr), w_sindy_z=0.0,
                      w_sindy_x=1.0, sindy_reg_type="l1", w_sindy_reg=0.0, sym_reg_type="i", w_sym_reg=0.0,
                      st_freq=int(st_freq), threshold=float(thr), int_t=0.1, int_dt=0.01, print_eq=False)
    assert np.array_equal(r.mask.numpy(), g[f"{tag}_mask_final"])                      # identical sparsity pattern
    want = g[f"{tag}_Xi_final"]
    assert np.allclose(r.get_Xi().detach().numpy(), want, rtol=1e-4, atol=2e-5 * np.abs(want).max())
    assert any(f.startswith("regressor_") for f in os.listdir(tmp_path / "saved_models" / "t"))   # final-convergence checkpoint


def test_train_sindy_loop(golden, capsys):
    g = golden("f3_stlsq")
    x, dx = t(g["dosc_clean_x"]), t(g["dosc_clean_dx"])
    r = SINDyRegression(2, 3, False, False, threshold=0.05, device="cpu", engine=OracleEngine())
    train_SINDy(r, x, dx, num_epochs=8, device="cpu", log_interval=1, save_interval=100, save_dir="t", w_sindy_reg=0.0, threshold=0.05)
    assert np.array_equal(r.mask.numpy(), g["dosc_clean_masks"][-1])
    assert "Final convergence reached at iteration 1" in capsys.readouterr().out


def test_unknown_tasks_and_autoencoders_raise_like_the_reference():
    from symode_amd.autoencoder import AutoEncoder
    from symode_amd.dataset import get_dataset
    with pytest.raises(NotImplementedError):                   # dataset.py:56
        get_dataset({"task": "nope", "noise": 0.0, "smoothing": None})
    with pytest.raises(NotImplementedError):
        AutoEncoder(ae_arch="stick_cnn")


@pytest.mark.parametrize("act", ["ReLU", "Tanh", "Sigmoid", "SiLU", "ELU", "LeakyReLU", "Softplus"])
@pytest.mark.parametrize("bn,ortho", [(False, False), (True, True)])
def test_forward_mode_mlp_jvp_equals_functional_jvp(act, bn, ortho):
    """model_utils.mlp_jvp (analytic forward mode through the autoencoder's layers) against torch.autograd.functional.jvp
    -- values, tangents, and their gradients -- in fp64; the group transform g(x) = dec(g (enc(x) - mu) + mu) likewise."""
    from functools import partial
    from torch.autograd.functional import jvp
    from symode_amd.autoencoder import AutoEncoder
    from symode_amd.model_utils import _group_transform, _group_transform_jvp, mlp_jvp
    torch.manual_seed(0)
    ae = AutoEncoder(ae_arch="mlp", input_dim=3, hidden_dim=16, latent_dim=2, n_layers=3, n_comps=2, activation=act,
                     activation_args=[], batch_norm=bn, ortho_ae=ortho).double()
    for m in ae.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.3)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.7, 1.3)
            m.bias.data.normal_(0, 0.2)
    ae.eval()
    z = torch.randn(7, 2, 2, dtype=torch.float64, requires_grad=True)
    v = torch.randn(7, 2, 2, dtype=torch.float64, requires_grad=True)
    y, ty = mlp_jvp(ae.decoder, z, v)
    yr, tyr = jvp(ae.decoder, z, v=v, create_graph=True)
    assert torch.allclose(y, yr) and torch.allclose(ty, tyr, atol=1e-12)
    ga = torch.autograd.grad((ty ** 2).sum() + (y ** 3).sum(), [z, v], allow_unused=True)
    gb = torch.autograd.grad((tyr ** 2).sum() + (yr ** 3).sum(), [z, v], allow_unused=True)
    for a, b in zip(ga, gb):
        a = torch.zeros_like(z) if a is None else a
        b = torch.zeros_like(z) if b is None else b
        assert torch.allclose(a, b, atol=1e-10)
    x, e = torch.randn(9, 3, dtype=torch.float64), torch.randn(9, 3, dtype=torch.float64)
    g = torch.matrix_exp(torch.randn(4, 4, dtype=torch.float64) * 0.1)
    zmean = torch.randn(2, dtype=torch.float64) * 0.1
    for nrm in ("global", "in_batch"):
        tr = partial(_group_transform, autoencoder=ae, g=g, normalize=nrm, z_mean=zmean)
        a, b = _group_transform_jvp(x, e, ae, g, nrm, zmean), jvp(tr, x, v=e)
        assert torch.allclose(a[0], b[0], atol=1e-12) and torch.allclose(a[1], b[1], atol=1e-10)
    ae.train()                                                  # train-mode batch norm couples the samples: not covered
    assert (mlp_jvp(ae.encoder, torch.randn(5, 2, 3, dtype=torch.float64), torch.randn(5, 2, 3, dtype=torch.float64)) is None) == bn


@pytest.mark.parametrize("tag,act,rep", [("relu_sim2", "ReLU", "(2,sim2)"), ("tanh_learn", "Tanh", "(2,1,2)")])
def test_constant_half_of_the_symmetry_regularisers(golden, tag, act, rep):
    """S2 / S3 with the Xi-independent x component computed once per batch (x_const=...) equal the all-at-once evaluation
    (values here; gradients in tests/test_gpu_train.py), and the cache follows the batch tensor."""
    from symode_amd import model_utils as MU
    g = golden("f6_symreg")
    ae = load_fixture_autoencoder(g, tag, act)
    gen = load_fixture_generator(g, tag, rep)
    for p in list(ae.parameters()) + list(gen.parameters()):
        p.requires_grad = False
    d, order, sine, exp = [int(v) for v in g[f"{tag}_cfg"]]
    r = SINDyRegression(d, order, bool(sine), bool(exp), threshold=0.05, device="cpu", engine=OracleEngine())
    r.Xi.data = t(g[f"{tag}_Xi"]).clone()
    flow = MU._EulerFlow(r, 0.05, 0.01)
    x = t(g[f"{tag}_x"])[:256].contiguous()
    for kind, fn in (("i", MU.symmreg_i), ("f", MU.symmreg_f)):
        for batch in (x, (x * 1.1).contiguous()):                  # a second batch must not reuse the first one's half
            with torch.no_grad():
                x_fx = torch.stack([batch, flow(batch)], dim=1)
                a = fn(x_fx, ae, gen, f=flow, x_const=batch).item()
                b = fn(x_fx, ae, gen, f=flow).item()
            assert a == pytest.approx(b, rel=2e-5), kind
            assert MU._CONST_HALF[kind][0]() is batch


def test_mlp_split_autoencoder_layout():
    """'mlp_split' (reference model.py:62-70): two independent MLPs on the halves of the last axis, state_dict prefixes
    encoder.model{1,2}.layers.* / decoder.model{1,2}.layers.*."""
    from symode_amd.autoencoder import AutoEncoder
    ae = AutoEncoder(ae_arch="mlp_split", input_dim=6, hidden_dim=8, latent_dim=2, n_layers=2, n_comps=2, activation="ReLU",
                     activation_args=[], batch_norm=False, ortho_ae=False)
    keys = set(ae.state_dict())
    assert {"encoder.model1.layers.0.weight", "encoder.model2.layers.5.0.weight", "decoder.model1.layers.0.weight"} <= keys
    x = torch.randn(5, 2, 12)
    z, xhat = ae(x)
    assert z.shape == (5, 2, 4) and xhat.shape == (5, 2, 12)
    x2 = x.clone()
    x2[..., 6:] += 1.0                                           # the second half only reaches the second model
    assert torch.equal(ae.encode(x2)[..., :2], z[..., :2]) and not torch.equal(ae.encode(x2)[..., 2:], z[..., 2:])


def test_reference_config_names_parse(monkeypatch):
    """Every config / script name a user of the reference types exists here and parses -- through the reference's
    ``--config`` rules (parser_utils.py:100-118, 183-186) -- to the effective settings of SURVEY.md Appendix A."""
    import importlib.util
    import os
    import symode_amd
    from symode_amd import parser_utils
    pkg = os.path.dirname(os.path.abspath(symode_amd.__file__))
    spec = importlib.util.spec_from_file_location("write_run_configs", os.path.join(os.path.dirname(pkg), "tools", "write_run_configs.py"))
    W = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(W)
    monkeypatch.chdir(pkg)                                     # the package directory holds run_configs/
    defaults = vars(parser_utils.get_args(construct_parser=True).parse_args([]))
    for rel, row in W.CONFIGS.items():
        assert os.path.exists(os.path.join("run_configs", rel)), rel
        got = vars(parser_utils.get_args(argv=["--config", rel]))
        for key, value in row.items():
            assert got[key] == value, (rel, key, got[key], value)
        for key, value in defaults.items():                    # everything the row does not set stays at the parser default
            if key not in row and key != "config":
                assert got[key] == value, (rel, key)
    for name, (module, cfg) in W.SCRIPTS.items():
        text = open(os.path.join("run_scripts", name + ".sh")).read()
        assert f"--config {cfg}" in text and f"symode_amd.{module}" in text and "{0..49}" in text
    # Appendix A, spelled out for the five BASELINE configurations
    a = vars(parser_utils.get_args(argv=["--config", "dosc/noise20_sindy.cfg", "--seed", "7"]))
    assert (a["task"], a["noise"], a["smoothing"], a["sindy_optimizer"], a["lbfgs_subsample"], a["lr_sindy"], a["poly_order"],
            a["st_freq"], a["threshold"], a["num_epochs"], a["seed"]) == ("dosc", 0.2, "gp", "lbfgs", 0.5, 0.1, 2, 50, 5e-2, 200, 7)
    a = vars(parser_utils.get_args(argv=["--config", "dosc/noise20_esindy.cfg"]))
    assert a["eq_constraint"] and a["repr"] == "(1,so2)" and a["ae_arch"] == "none" and (a["lr_sindy"], a["st_freq"], a["threshold"]) == (1.0, 100, 1e-2)
    a = vars(parser_utils.get_args(argv=["--config", "lv/noise99_eq_isymreg.cfg"]))
    assert a["sym_reg_type"] == "i" and a["include_exp"] and a["poly_order"] == 2 and a["fix_laligan"] and a["load_laligan"] == "laligan-noise99-lv"
    assert (a["lbfgs_subsample"], a["w_sym_reg"], a["threshold"], a["int_t"], a["int_dt"], a["n_comps"], a["repr"]) == (0.01, 0.1, 0.15, 0.1, 0.01, 2, "(2,1,2)")
    a = vars(parser_utils.get_args(argv=["--config", "selkov/noise20_eq_sindy.cfg"]))
    assert (a["poly_order"], a["lr_sindy"], a["st_freq"], a["threshold"], a["lbfgs_subsample"], a["num_epochs"]) == (3, 1.0, 50, 7.5e-2, 0.5, 200)
    a = vars(parser_utils.get_args(argv=["--config", "rd/sym_eq.cfg"]))
    assert a["task"] == "mt_rd" and a["include_sindy"] and a["eq_constraint"] and a["constrain_constant"] and a["w_sindy_x"] == 0.0
    assert (a["w_sindy_z"], a["batch_size"], a["lr_ae"], a["gan_st_thres"], a["latent_dim"], a["threshold"], a["w_sindy_reg"]) == (0.1, 64, 3e-4, 0.05, 2, 0.1, 0.1)


def test_symmetry_caches_see_parameter_and_offset_changes():
    """The Xi-independent halves are cached per live batch; a changed latent offset, a load_state_dict or an in-place
    update of the frozen autoencoder / generator must invalidate them, and a wrong x_const must raise."""
    import torch
    import pytest
    from symode_amd import model_utils as MU
    from symode_amd.autoencoder import AutoEncoder
    from symode_amd.lie import LieGenerator
    torch.manual_seed(0)
    ae = AutoEncoder(ae_arch="mlp", input_dim=2, hidden_dim=8, latent_dim=2, n_layers=2, n_comps=2, activation="Tanh",
                     activation_args=[], batch_norm=True, ortho_ae=False).eval()    # the latent offset is the last BatchNorm's bias
    gen = LieGenerator(repr="(2,sim2)", group_idx="0").eval()
    for p in list(ae.parameters()) + list(gen.parameters()):
        p.requires_grad = False
    x = torch.randn(32, 2)
    f = lambda a: a + 0.1 * torch.tanh(a)  # noqa: E731
    x_fx = torch.stack([x, f(x)], dim=1)

    def both(z_mean=None):
        a = MU.symmreg_f(x_fx, ae, gen, f, z_mean=z_mean, x_const=x)
        b = MU.symmreg_f(x_fx, ae, gen, f, z_mean=z_mean)
        return a.item(), b.item()
    a, b = both()
    assert abs(a - b) <= 1e-5 * abs(b)
    zm = torch.full((2,), 0.3)
    a, b = both(zm)                                             # same batch, new latent offset
    assert abs(a - b) <= 1e-5 * abs(b)
    with torch.no_grad():
        ae.decoder[0].weight.mul_(1.5)                          # in-place update of a "frozen" module
    a, b = both(zm)
    assert abs(a - b) <= 1e-5 * abs(b)
    with pytest.raises(ValueError):
        MU.symmreg_f(x_fx, ae, gen, f, x_const=x + 1.0)


def test_plain_lbfgs_is_bit_identical_to_torch_optim_lbfgs():
    """train._PlainLBFGS (the trainers' default: torch.optim.LBFGS's tensor-op sequence without the Optimizer base and its
    torch._dynamo import) against torch.optim.LBFGS itself: bit-equal parameters and returned losses after every step --
    several parameter tensors, an ill-conditioned least-squares objective, a history that wraps (size 5), runs that stop
    on the tolerances, a closure with an L1 term."""
    from symode_amd.train import _PlainLBFGS
    torch.manual_seed(3)
    A = torch.randn(400, 12) * torch.logspace(0, -2.5, 12)
    w_true = torch.randn(12, 2)
    Y = A @ w_true + 0.01 * torch.randn(400, 2)
    for lr, hist, l1, steps in [(1.0, 100, 0.0, 6), (0.3, 5, 0.0, 8), (0.1, 100, 0.05, 5), (1.0, 3, 0.0, 12)]:
        runs = []
        for cls in (torch.optim.LBFGS, _PlainLBFGS):
            torch.manual_seed(11)
            W = torch.randn(10, 2, requires_grad=True)
            b = torch.randn(2, requires_grad=True)
            c = torch.randn(2, 1, requires_grad=True)
            opt = cls([W, b, c], lr=lr, history_size=hist)

            def closure():
                opt.zero_grad()
                pred = A[:, :10] @ W + A[:, 10:] @ torch.stack([b, c[:, 0]])
                loss = ((pred - Y) ** 2).mean() + l1 * (W.abs().sum() + b.abs().sum())
                loss.backward()
                return loss
            trace = []
            for _ in range(steps):
                out = opt.step(closure)
                trace.append((out.detach().clone(), W.detach().clone(), b.detach().clone(), c.detach().clone()))
            runs.append(trace)
        for (la, Wa, ba, ca), (lb, Wb, bb, cb) in zip(*runs):
            assert torch.equal(la, lb) and torch.equal(Wa, Wb) and torch.equal(ba, bb) and torch.equal(ca, cb), (lr, hist, l1)
    # a start at the optimum returns before the first move, like torch's
    x = torch.zeros(3, requires_grad=True)
    o = _PlainLBFGS([x], lr=1.0)

    def at_optimum():
        o.zero_grad()
        l = (x ** 2).sum()
        l.backward()
        return l
    assert float(o.step(at_optimum)) == 0.0 and o.n_iter == 0 and o.func_evals == 1
