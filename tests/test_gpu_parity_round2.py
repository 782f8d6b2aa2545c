"""GPU parity tests that back the loose end-of-run tolerances with tight intermediate checks:

  * BASELINE config[2]'s real path at full size: lv 200 x 10^4 @ 1 % = 20 000 points, noise 0.99 (GP-smoothed), order 2 +
    exp, ``sym_reg_type i`` with the K = 10 Euler flow -- closure value and d/dXi through symode_euler_jvp /
    symode_euler_jvp_vjp against the oracle's model_utils.py:8-67 restatement run live on the same frozen autoencoder;
  * the L-BFGS trainer: the HIP closure agrees with the oracle at EVERY point of the recorded optimisation trajectory
    (the oracle run itself is pinned to the reference's per-epoch loss history, tests/test_oracle_golden.py), and the GPU
    trainer's own per-epoch ``loss_sindy_x`` follows the reference's recorded history;
  * no near-threshold coefficient (| |coef| - thr | < 1e-4) occurs in any pinned run;
  * ``train_SIGED`` (mini-batch Adam; both branches incl. the linear-latent regulariser S1) against an oracle-side
    restatement of the same loop, and the single-seed driver ``main.main``.
"""
import copy
import os

import numpy as np
import pytest
import torch

from oracle import sindy_oracle as O
from tests.helpers import load_fixture_autoencoder, load_fixture_generator, t

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
torch.set_num_threads(8)


@pytest.fixture(scope="module")
def S():
    import symode_amd
    assert torch.cuda.is_available()
    return symode_amd


def _scaled_err(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)


# ------------------------------------------------------------------------------------------------ 1a
@pytest.fixture(scope="module")
def config2(S):
    """lv/noise99_eq_isymreg.cfg at full size: data by the reference's recipe (99 % noise, GP-smoothed), the 1 % L-BFGS
    subsample, a frozen 512 x 5 orthogonal + batch-norm autoencoder with n_comps 2 and the (2,1,2) generator.  The
    authors' LaLiGAN checkpoint is not shipped, so the autoencoder is a seeded random one whose BatchNorm statistics
    were set by a few training-mode passes over the data."""
    from tests.helpers import make_config2
    return make_config2(S, DEV)


@pytest.mark.parametrize("split", [True, False], ids=["x_const", "whole_batch"])
def test_config2_lv_infinitesimal_regulariser_full_size(S, config2, split):
    """Closure of train.py:663-679 with sym_reg_type 'i': MSE + 0.1 * symmreg_i through the K = 10 Euler flow.
    Tolerances: fp32 end to end on both sides, through a 512 x 5 MLP and its decoder tangent (hipBLASLt on the GPU,
    MKL on the host): MSE 1e-5, regulariser value and gradient 3e-5 of their scale (measured on MI355X: 6.6e-6 and
    3.6e-6; the relative loss is a ratio of two batch means)."""
    from symode_amd import model_utils as MU
    x, dx, ae, gen = config2
    assert x.shape == (20000, 2)
    r = S.SINDyRegression(2, 2, False, True, threshold=0.15, device=DEV)
    torch.manual_seed(3)
    Xi0 = torch.randn(2, 8) * 0.3
    r.Xi.data = Xi0.to(DEV)
    int_t, int_dt = 0.1, 0.01
    sym = MU.make_symmreg_pttrain(ae, gen)
    flow = MU._EulerFlow(r, int_t, int_dt)
    x_fx = torch.stack([x, flow(x)], dim=1)
    l_mse = r.mse_loss(x, dx)
    l_sym = sym(x_fx, f=flow, x_const=x) if split else sym(x_fx, f=flow)
    (l_mse + 0.1 * l_sym).backward()

    # the oracle on the host: the same stock modules (copied to the CPU), the reference's op sequence for everything else
    ae_c = copy.deepcopy(ae).cpu().eval()
    xc, dxc = x.cpu(), dx.cpu()
    reg = O.OracleRegressor(2, 2, False, True, Xi0=Xi0)
    f = lambda a: O.odeint(reg, a, int_t, int_dt)  # noqa: E731
    xfx_c = torch.stack([xc, f(xc)], dim=1)
    basis = [b.detach().cpu() for b in gen.get_full_basis_list()]
    lo_mse = torch.nn.functional.mse_loss(reg(xc), dxc)
    lo_sym = O.symreg_infinitesimal(xfx_c, ae_c.encode, ae_c.decode, ae_c.encoder[-2].bias, basis, f)
    (lo_mse + 0.1 * lo_sym).backward()
    gw = reg.Xi.grad.numpy()
    e_mse = abs(l_mse.item() - lo_mse.item()) / abs(lo_mse.item())
    e_sym = abs(l_sym.item() - lo_sym.item()) / abs(lo_sym.item())
    e_grad = _scaled_err(r.Xi.grad.cpu().numpy(), gw)
    print(f"config2 sym_reg_type i (split={split}): mse rel err {e_mse:.2e}, sym rel err {e_sym:.2e}, grad scaled err {e_grad:.2e}")
    assert e_mse <= 1e-5, e_mse
    assert e_sym <= 3e-5, e_sym
    assert e_grad <= 3e-5, e_grad


# ------------------------------------------------------------------------------------------------ 1b
class _Recorder(O.OracleRegressor):
    """Oracle regressor that notes (Xi, mask) at every closure evaluation of the L-BFGS run."""

    def __call__(self, x):
        self.trace.append((self.get_Xi().detach().clone(), self.mask.clone()))
        return super().__call__(x)


def _oracle_run(g, tag):
    d, order = [int(v) for v in g[f"{tag}_cfg"]]
    lr, st_freq, thr, epochs = g[f"{tag}_hp"]
    x, dx = t(g[f"{tag}_x"]), t(g[f"{tag}_dx"])
    if f"{tag}_init_Xi" in g.files:
        reg = _Recorder(d, order, threshold=float(thr), Xi0=t(g[f"{tag}_init_Xi"]))
    else:
        reg = _Recorder(d, order, L_list=[torch.tensor([[0.0, 1.0], [-1.0, 0.0]])], threshold=float(thr),
                        beta0=t(g[f"{tag}_init_beta"]), const0=t(g[f"{tag}_init_const"]))
        reg.Q = t(g[f"{tag}_Q"])
    reg.trace = []
    hist = O.lbfgs_fit(reg, x, dx, int(epochs), float(lr), st_freq=int(st_freq), threshold=float(thr))
    return reg, hist, x, dx, order


@pytest.mark.parametrize("tag", ["dosc_sindy", "dosc_esindy", "selkov_sindy"])
def test_hip_closure_along_the_recorded_lbfgs_trajectory(S, golden, tag):
    """Every closure point (Xi, mask) the pinned oracle run visits -- from the random start to the converged sparse
    model -- evaluated by ONE batched launch of the fused kernel: loss within rtol 1e-5 of the oracle's; gradient within
    1e-5 of the magnitude of the operands it is formed from, 2/(N d) sum_n (|Theta_n| |w_j| + |dx_nj|) |Theta_nk| --
    the forward-error yardstick of an fp32 evaluation: at the end of a noise-free fit the residual is 1e-4 of its
    operands, so the gradient is a cancellation residue whose own size says nothing (the oracle's fp32 gradient is itself
    1.5e-4 of its summands away from the fp64 value there, and 2e-7 on this yardstick) -- and, wherever the gradient
    loses less than one digit to that cancellation (max |g| >= 0.1 of the yardstick), within 2e-5 of its own max-norm.
    This is what
    licenses the looser end-of-run coefficient tolerance of the trainer tests: both runs follow the same map, evaluated
    to 1e-5, until L-BFGS's own stopping ball (update norm < 1e-3, train.py:705)."""
    g = golden("f4_lbfgs")
    reg, hist, x, dx, order = _oracle_run(g, tag)
    trace = reg.trace
    assert len(trace) == hist["n_closure"] and len(trace) >= 10
    eng = S.get_engine()
    n = len(trace)
    Xi = torch.stack([a for a, _ in trace]).to(DEV)
    M = torch.stack([b for _, b in trace]).to(DEV)
    X = x.to(DEV)[None].expand(n, -1, -1).contiguous()
    DX = dx.to(DEV)[None].expand(n, -1, -1).contiguous()
    loss, grad = eng.loss_grad(X, DX, Xi, M, order)
    loss, grad = loss.cpu().numpy(), grad.cpu().numpy()
    worst_l = worst_g = worst_rel = 0.0
    th = O.theta(x, order).double().abs()
    for k, (a, b) in enumerate(trace):
        wl, wg = O.mse_loss_and_grad(x, dx, a, b, order)
        wl, wg = wl.item(), wg.numpy()
        operands = th @ (a * b).double().abs().T + dx.double().abs()            # (N, d): what the residual is the difference of
        yard = (2.0 / operands.numel()) * (operands.T @ th).numpy()             # (d, p)
        # losses at the 1e-9 floor of a noise-free fit are sums of squared fp32 rounding errors of the residual itself
        worst_l = max(worst_l, abs(loss[k] - wl) / max(abs(wl), 1e-7))
        live = b.numpy() > 0
        worst_g = max(worst_g, (np.abs(grad[k] - wg)[live] / yard[live]).max())
        if np.abs(wg).max() >= 0.1 * yard[live].max():                          # less than one digit lost to cancellation
            worst_rel = max(worst_rel, np.abs(grad[k] - wg).max() / np.abs(wg).max())
    print(f"{tag}: {n} closure points, worst loss rel err {worst_l:.2e}, worst grad err vs operand magnitude {worst_g:.2e}, "
          f"vs max-norm (points with little cancellation) {worst_rel:.2e}")
    assert worst_l <= 1e-5, worst_l
    assert worst_g <= 1e-5, worst_g
    assert worst_rel <= 2e-5, worst_rel


@pytest.mark.parametrize("tag", ["dosc_sindy", "dosc_esindy", "selkov_sindy"])
def test_gpu_trainer_follows_the_reference_loss_history(S, golden, tag, tmp_path, monkeypatch):
    """Per-epoch ``loss_sindy_x`` logged by the GPU trainer vs the reference's recorded run (tests/golden/f4_lbfgs.npz,
    ``*_loss_hist``): the same number of logged epochs (= the same sequence of convergence / thresholding events); the
    FIRST epoch -- 20 L-BFGS iterations from the same start -- to rtol 1e-5 (measured 7e-7 on MI355X); the later
    pre-threshold epochs to 15 %.  Why not tighter there: torch's L-BFGS takes unit steps without a line search and the
    loss falls ~100x per epoch, so the value logged at the end of an epoch is a steep function of where along that
    descent the 20th iteration lands; a 1e-6 difference in one closure moves it by percents while both runs converge to
    the same minimiser.  On the selkov library (cond 9e3, lr 1.0) that amplification is chaotic from the first epoch
    (69 % apart after 20 iterations, with closures that agree to 1e-6) until both runs sit on the same
    plateau: there the last pre-threshold epoch is compared (1e-3; measured 2e-4).  The tight statement about the map
    both runs iterate is the replay test above (every closure point, 1e-5); this one pins the trajectory's shape, its
    events and its end."""
    from tests.test_gpu_train import _regressor
    monkeypatch.chdir(tmp_path)
    g = golden("f4_lbfgs")
    d, order = [int(v) for v in g[f"{tag}_cfg"]]
    lr, st_freq, thr, epochs = g[f"{tag}_hp"]
    x, dx = t(g[f"{tag}_x"]), t(g[f"{tag}_dx"])
    r = _regressor(S, g, tag, d, order, float(thr))
    logged = []
    monkeypatch.setattr(S.train.wandb, "log", lambda dct, *a, **k: logged.append(dict(dct)), raising=False)
    ident = torch.nn.Identity()
    S.train.train_SIGED_lbfgs(train_loader=[(x, dx)], test_loader=[], num_epochs=int(epochs), device=DEV,
                              log_interval=10 ** 9, save_interval=10 ** 9, save_dir="t", autoencoder=ident, generator=ident,
                              regressor=r, regressor_dst=None, use_latent=False, distill_latent=False, lr_sindy=float(lr),
                              w_sindy_z=0.0, w_sindy_x=1.0, sindy_reg_type="l1", w_sindy_reg=0.0, sym_reg_type="i",
                              w_sym_reg=0.0, st_freq=int(st_freq), threshold=float(thr), int_t=0.1, int_dt=0.01, print_eq=False)
    got = np.array([l["loss_sindy_x"] for l in logged])
    want = g[f"{tag}_loss_hist"]
    assert len(got) == len(want), (got, want)                                   # same number of logged epochs = same events
    # epochs before the first thresholding event of the oracle run (which reproduces the reference's history)
    _, hist, *_ = _oracle_run(g, tag)
    first_event = min([e for e, _ in hist["events"]] + [len(want)])
    n_pre = max(1, min(first_event + 1, len(want)))
    rel = np.abs(got[:n_pre] - want[:n_pre]) / np.maximum(np.abs(want[:n_pre]), 1e-30)
    print(f"{tag}: pre-threshold epochs {n_pre}, rel err of the per-epoch loss {rel}")
    above_floor = want[:n_pre] > 1e-7                                            # below: sums of squared fp32 rounding errors
    if tag == "selkov_sindy":
        assert rel[n_pre - 1] <= 1e-3, rel                                       # the plateau both runs reach before thresholding
    else:
        assert rel[0] <= 1e-5 or want[0] < 1e-8, rel                             # (esindy starts at the rounding floor)
        assert np.all(rel[above_floor] <= 0.15), rel
    assert r.near_threshold == [], r.near_threshold
    assert np.array_equal(r.mask.cpu().numpy(), g[f"{tag}_mask_final"])


# ------------------------------------------------------------------------------------------------ 1c
def test_no_near_threshold_coefficient_in_the_pinned_stlsq_runs(S, golden):
    """The selkov ridge fixture (the one train_SINDy is pinned on) and the 64-seed sweep at full size."""
    g = golden("f3_stlsq")
    x, dx = t(g["selkov_ridge_x"]).to(DEV), t(g["selkov_ridge_dx"]).to(DEV)
    r = S.SINDyRegression(2, 3, False, False, threshold=0.075, device=DEV, lstsq_driver="gelsy")
    S.train.train_SINDy(r, x, dx, num_epochs=8, device=DEV, log_interval=100, save_interval=100, save_dir="t", w_sindy_reg=0.1,
                        threshold=0.075)
    assert r.near_threshold == [], r.near_threshold
    xs, dxs = S.data.make_dataset("selkov", 10, 10000, dt=0.002, noise=0.0, seed=2, device=DEV)
    sw = S.sweep.SeedSweepSTLSQ(xs[0], dxs[0], 3, n_seeds=64, subsample=0.5, seed0=0)
    sw.solve(0.0, 0.075, lstsq_driver="gels")
    assert sw.near_threshold == [], sw.near_threshold


def test_near_threshold_report_fires(S):
    """A coefficient placed 5e-5 above the threshold is listed, one 2e-4 above is not."""
    r = S.SINDyRegression(2, 2, False, False, threshold=0.1, device=DEV)
    Xi = torch.zeros(2, 6)
    Xi[0, 1], Xi[1, 2], Xi[1, 4] = 0.10005, -0.1002, 0.5
    r.Xi.data = Xi.to(DEV)
    r.set_threshold(0.1)
    assert [c["index"] for c in r.near_threshold] == [(0, 1)]
    assert r.mask.cpu()[0, 1] == 1 and r.mask.cpu()[1, 2] == 1 and r.mask.cpu()[0, 0] == 0


# ------------------------------------------------------------------------------------------------ 1d
def _adam_oracle(reg, batches, epochs, lr, closure):
    opt = torch.optim.Adam(reg.parameters(), lr=lr)
    for _ in range(epochs):
        for xb, dxb in batches:
            loss = closure(xb, dxb)
            opt.zero_grad()
            loss.backward()
            opt.step()


def test_train_SIGED_adam_with_infinitesimal_regulariser_matches_oracle_loop(S, golden, tmp_path, monkeypatch):
    """train.py:382-614, non-latent branch: mini-batch Adam on  w_x MSE + w_sym symmreg_i(K-step Euler flow) + w_reg L1
    with a frozen fixture autoencoder -- against the same loop written with the oracle's pieces on the host."""
    monkeypatch.chdir(tmp_path)
    g = golden("f6_symreg")
    tag, act, rep = "relu_sim2", "ReLU", "(2,sim2)"
    ae, gen = load_fixture_autoencoder(g, tag, act, DEV), load_fixture_generator(g, tag, rep, DEV)
    d, order, sine, exp = [int(v) for v in g[f"{tag}_cfg"]]
    x, Xi0 = t(g[f"{tag}_x"]), t(g[f"{tag}_Xi"])
    dx = O.forward(x, Xi0 * 0.5, torch.ones_like(Xi0), order, bool(sine), bool(exp)).detach()
    batches = [(x[i:i + 128], dx[i:i + 128]) for i in range(0, 512, 128)]
    int_t, int_dt, w_sym, w_reg, lr, epochs = 0.03, 0.01, 0.1, 0.01, 1e-2, 3
    r = S.SINDyRegression(d, order, bool(sine), bool(exp), threshold=0.05, device=DEV)
    r.Xi.data = Xi0.to(DEV)
    ident = torch.nn.Identity()
    S.train.train_SIGED(train_loader=batches, test_loader=batches[:1], num_epochs=epochs, device=DEV, log_interval=1,
                        save_interval=10 ** 9, save_dir="t", autoencoder=ae, discriminator=ident, generator=gen, lr_ae=0, lr_d=0,
                        lr_g=0, w_recon=0, w_gan=0, w_reg_norm=0, w_reg_ortho=0, w_reg_closure=0, use_original_x=False,
                        gan_st_freq=0, gan_st_thres=0.0, ae_arch="mlp", regressor=r, use_latent=False, lr_sindy=lr, w_sindy_z=0.0,
                        w_sindy_x=1.0, sindy_reg_type="l1", w_sindy_reg=w_reg, w_sym_reg=w_sym, st_freq=2, threshold=0.05,
                        int_t=int_t, int_dt=int_dt)
    from tests.helpers import TinyAE
    tae = TinyAE(g, tag, act)
    basis = [t(b) for b in g[f"{tag}_basis"]]
    reg = O.OracleRegressor(d, order, bool(sine), bool(exp), threshold=0.05, Xi0=Xi0)

    def closure(xb, dxb):
        f = lambda a: O.odeint(reg, a, int_t, int_dt)  # noqa: E731
        x_fx = torch.stack([xb, f(xb)], dim=1)
        return (torch.nn.functional.mse_loss(reg(xb), dxb) + w_sym * O.symreg_infinitesimal(x_fx, tae.encode, tae.decode, tae.z_mean, basis, f)
                + w_reg * sum(torch.norm(p, 1) for p in reg.parameters()))
    # the oracle loop with the trainer's epoch event: threshold every st_freq epochs (train.py:545-546)
    opt = torch.optim.Adam(reg.parameters(), lr=lr)
    for epoch in range(epochs):
        for xb, dxb in batches:
            loss = closure(xb, dxb)
            opt.zero_grad()
            loss.backward()
            opt.step()
        if (epoch + 1) % 2 == 0:
            reg.set_threshold(0.05)
    assert np.array_equal(r.mask.cpu().numpy(), reg.mask.numpy())
    err = _scaled_err(r.Xi.detach().cpu().numpy(), reg.Xi.detach().numpy())
    print(f"train_SIGED (i-regulariser): coefficient scaled err after {epochs * len(batches)} Adam steps {err:.2e}")
    assert err <= 2e-3, err                     # Adam normalises every step by sqrt(v): sign-level agreement of small gradients


def test_train_SIGED_latent_branch_with_linear_regulariser_S1(S, golden, tmp_path, monkeypatch):
    """train.py:382-614 with use_latent: w_z MSE(h(z), dz) + w_x MSE(J_dec dz_pred, dx) + w_sym * S1 (train.py:502-507, the
    linear-latent regulariser as ONE fused kernel over the generators) + L1, identity autoencoder (z = x)."""
    monkeypatch.chdir(tmp_path)
    g = golden("f4_lbfgs")
    x, dx = t(g["dosc_sindy_x"]), t(g["dosc_sindy_dx"])
    batches = [(x[i:i + 500], dx[i:i + 500]) for i in range(0, 2000, 500)]
    ae = S.autoencoder.AutoEncoder(ae_arch="none").to(DEV)
    gen = S.lie.LieGenerator(repr="(1,so2)", group_idx="0", device=DEV).to(DEV)
    torch.manual_seed(4)
    Xi0 = torch.randn(2, 6) * 0.3
    r = S.SINDyRegression(2, 2, False, False, threshold=0.05, device=DEV)
    r.Xi.data = Xi0.to(DEV)
    w_z, w_x, w_sym, w_reg, lr, epochs = 1.0, 0.5, 0.05, 0.01, 1e-2, 3
    ident = torch.nn.Identity()
    S.train.train_SIGED(train_loader=batches, test_loader=[], num_epochs=epochs, device=DEV, log_interval=10 ** 9,
                        save_interval=10 ** 9, save_dir="t", autoencoder=ae, discriminator=ident, generator=gen, lr_ae=0, lr_d=0,
                        lr_g=0, w_recon=0, w_gan=0, w_reg_norm=0, w_reg_ortho=0, w_reg_closure=0, use_original_x=False,
                        gan_st_freq=0, gan_st_thres=0.0, ae_arch="none", regressor=r, use_latent=True, lr_sindy=lr, w_sindy_z=w_z,
                        w_sindy_x=w_x, sindy_reg_type="l1", w_sindy_reg=w_reg, w_sym_reg=w_sym, st_freq=0, threshold=0.05,
                        int_t=0.1, int_dt=0.01)
    basis = [b.detach().cpu() for b in gen.get_full_basis_list()]
    reg = O.OracleRegressor(2, 2, threshold=0.05, Xi0=Xi0)

    def closure(zb, dzb):                         # identity autoencoder: z = x, dz = dx, J_dec = I; the reference weights each
        pred = reg(zb)                            # MSE term by its w (train.py:497-498)
        return (w_z * torch.nn.functional.mse_loss(pred, dzb) + w_x * torch.nn.functional.mse_loss(pred, dzb)
                + w_sym * O.symreg_linear_latent(zb, basis, reg, dz_pred=pred) + w_reg * sum(torch.norm(p, 1) for p in reg.parameters()))
    _adam_oracle(reg, batches, epochs, lr, closure)
    err = _scaled_err(r.Xi.detach().cpu().numpy(), reg.Xi.detach().numpy())
    print(f"train_SIGED (latent, S1): coefficient scaled err after {epochs * len(batches)} Adam steps {err:.2e}")
    assert err <= 2e-3, err


def test_main_driver_single_seed_end_to_end(S, tmp_path, monkeypatch, capsys):
    """``python -m symode_amd.main --seed 0 --config dosc/noise20_esindy.cfg`` (reference main.py:18-140) in a scratch
    directory: the data set is generated by the reference's recipe, the constrained L-BFGS fit runs on the HIP path,
    and the reference's outputs appear under saved_models/ and eval_results/ with its keys."""
    import shutil
    import symode_amd.main as M
    monkeypatch.chdir(tmp_path)
    src = os.path.join(os.path.dirname(os.path.abspath(M.__file__)), "run_configs")
    shutil.copytree(src, tmp_path / "run_configs")
    reg = M.main(["--seed", "0", "--config", "dosc/noise20_esindy.cfg", "--num_epochs", "40", "--gpu", "0"])
    out = capsys.readouterr().out
    assert "=== Evaluation ===" in out and "Near-threshold coefficients" in out
    sd = tmp_path / "saved_models" / "esindy-noise20-dosc"
    for name in ("autoencoder.pt", "discriminator.pt", "generator.pt", "generator_mask.pt", "regressor.pt", "regressor_lie_list.pt"):
        assert (sd / name).exists(), name
    state = torch.load(sd / "regressor.pt", weights_only=True)
    assert set(state) == {"beta", "const"}                                     # the reference's state_dict keys under the constraint
    ev = np.load(tmp_path / "eval_results" / "esindy-noise20-dosc" / "seed0.npz")
    assert set(ev.files) == {"coefficients", "correct_form", "mse", "correct_form_all", "mse_all"}
    assert ev["coefficients"].shape == (2, 6)
    mask = reg.mask.cpu().numpy() != 0
    truth = O.SINDY_TRUTH["dosc"] != 0
    assert np.array_equal(ev["correct_form"], [float(np.all(mask[i] == truth[i])) for i in range(2)])
    # so2-equivariant library on 20 %-noise data: the recovered linear part is the damped rotation
    assert np.allclose(ev["coefficients"][:, 1:3], [[-0.1, -1.0], [1.0, -0.1]], atol=0.05), ev["coefficients"]
    assert os.path.exists(tmp_path / "data" / "dosc-train-noise20-gp-x.pt")     # the reference's data file naming
