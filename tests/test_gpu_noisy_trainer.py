"""The L-BFGS trainer at the reference's REAL operating point: noisy, GP-smoothed data with the shipped configs'
hyper-parameters (tests/golden/f11_lbfgs_noisy.npz, recorded from the reference by tools/gen_golden.py::f11_lbfgs_noisy;
the oracle is pinned to the same records in tests/test_oracle_golden.py::test_lbfgs_fit_noisy).

  * replay: the HIP closure at EVERY closure point the recorded run visits, loss and gradient to 1e-5;
  * the trainer on its default path (optimiser + epoch logic as device kernels) and on torch's own optimiser: identical
    final masks -- or every differing entry explained by a listed near-threshold coefficient --, the same thresholding
    epochs, the near-threshold list EQUAL to the reference's own (non-empty in the edge case built for it), coefficients
    within the optimiser's stopping ball;
  * constrained cases run with the PRODUCT's constraint builder (no injected Q), compared through Xi.
"""
import numpy as np
import pytest
import torch

from oracle import sindy_oracle as O
from tests.helpers import f11_case, f11_oracle_regressor
from tests.test_oracle_golden import F11_CASES

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
torch.set_num_threads(8)


@pytest.fixture(scope="module")
def S():
    import symode_amd
    assert torch.cuda.is_available()
    return symode_amd


@pytest.mark.parametrize("tag", F11_CASES)
def test_hip_closure_along_the_noisy_trajectory(S, golden, tag):
    """Every closure point (Xi, mask) the REFERENCE's recorded run visits -- noted by a forward hook on its regressor
    when the fixture was made -- through ONE batched launch of the fused kernel, against the oracle's closure at the same
    point: loss rtol 1e-5; gradient 1e-5 of the operand magnitude always (see test_gpu_parity_round2.py for the
    yardstick) and 2e-5 of its own max-norm wherever it is not a cancellation residue.  (The trace is the recorded one,
    not an oracle run made here: on selkov -- cond 1e4, lr 1.0, no line search -- the reference's own op sequence lands
    on a different mask on a different host CPU; the closure both iterate is what can be pinned.)"""
    c = f11_case(golden("f11_lbfgs_noisy"), tag)
    eng = S.get_engine()
    x, dx, order, n = c["x"], c["dx"], c["order"], len(c["trace_Xi"])
    assert n >= 9
    loss, grad = eng.loss_grad(x.to(DEV)[None].expand(n, -1, -1).contiguous(), dx.to(DEV)[None].expand(n, -1, -1).contiguous(),
                               c["trace_Xi"].to(DEV), c["trace_mask"].to(DEV), order)
    loss, grad = loss.cpu().numpy(), grad.cpu().numpy()
    th = O.theta(x, order).double().abs()
    worst_l = worst_g = worst_rel = 0.0
    n_rel = 0
    for k, (a, b) in enumerate(zip(c["trace_Xi"], c["trace_mask"])):
        wl, wg = O.mse_loss_and_grad(x, dx, a, b, order)
        wl, wg = wl.item(), wg.numpy()
        operands = th @ (a * b).double().abs().T + dx.double().abs()
        yard = (2.0 / operands.numel()) * (operands.T @ th).numpy()
        live = b.numpy() > 0
        worst_l = max(worst_l, abs(loss[k] - wl) / abs(wl))
        worst_g = max(worst_g, (np.abs(grad[k] - wg)[live] / yard[live]).max())
        if np.abs(wg).max() >= 0.1 * yard[live].max():
            n_rel += 1
            worst_rel = max(worst_rel, np.abs(grad[k] - wg).max() / np.abs(wg).max())
    print(f"{tag}: {n} closure points, loss rel err {worst_l:.2e}, grad vs operand magnitude {worst_g:.2e}, "
          f"vs max-norm {worst_rel:.2e} ({n_rel} points with < 1 digit of cancellation)")
    assert worst_l <= 1e-5 and worst_g <= 1e-5 and worst_rel <= 2e-5, (worst_l, worst_g, worst_rel)


def _product_regressor(S, c):
    """The product's regressor at the recorded start -- under the constraint with ITS OWN null-space basis (the start
    beta is re-expressed in it: both bases are orthonormal bases of the same subspace, so the start Xi is the same)."""
    if c["L"] is None:
        r = S.SINDyRegression(c["d"], c["order"], False, False, threshold=c["thr"], device=DEV)
        r.Xi.data = c["Xi0"].to(DEV)
        return r
    r = S.SINDyRegression(c["d"], c["order"], False, False, L_list=[c["L"]], threshold=c["thr"], device=DEV,
                          constrain_constant=c["constrain_constant"])
    Qp = r.Q.detach().cpu()
    assert Qp.shape == c["Q"].shape and bool(r.use_kron_product) == c["use_kron"]
    assert torch.allclose(Qp @ Qp.T, c["Q"] @ c["Q"].T, atol=2e-5)               # same subspace, gauge aside
    r.beta.data = (Qp.T @ (c["Q"] @ c["beta0"])).to(DEV)
    r.const.data = c["const0"].to(DEV)
    return r


@pytest.mark.parametrize("mode", ["device_trainer", "torch_lbfgs"])
@pytest.mark.parametrize("tag", F11_CASES)
def test_trainer_on_the_noisy_reference_runs(S, golden, tag, mode, tmp_path, monkeypatch, capsys):
    """The fit from the recorded start on the default path (optimiser + epoch logic as device kernels) and on torch's own
    optimiser.  What the reference itself reproduces under a ONE-ULP change of its start (12 recorded runs per case) is
    what is asserted exactly; what it does not reproduce is asserted to stay inside the reference's own spread:
      * mask: identical to the recorded one, except at coefficients (a) whose mask bit the reference's own 1-ulp runs do
        not agree on (selkov: cond 1e4, lr 1.0 -- none of the 12 reaches the recorded mask; the edge case), or (b) listed
        within the trainer's stopping ball (1e-3) of the threshold by either run;
      * BASELINE.md section 3 list (1e-4 band): empty where the reference's is; the edge case's (0, 8) is in the product's
        stopping-ball list;
      * logged epochs (= events): within the range of the reference's 1-ulp runs (+-1);
      * first-epoch loss 1e-5 (or 4x the reference's own 1-ulp spread); last logged loss 1e-4 and coefficients within 3e-3
        (a few stopping balls) when the mask is the same."""
    monkeypatch.chdir(tmp_path)
    c = f11_case(golden("f11_lbfgs_noisy"), tag)
    r = _product_regressor(S, c)
    logged = []
    monkeypatch.setattr(S.train.wandb, "log", lambda dct, *a, **k: logged.append(dict(dct)), raising=False)
    ident = torch.nn.Identity()
    S.train.train_SIGED_lbfgs(train_loader=[(c["x"], c["dx"])], test_loader=[], num_epochs=c["epochs"], device=DEV,
                              log_interval=10 ** 9, save_interval=10 ** 9, save_dir="t", autoencoder=ident, generator=ident,
                              regressor=r, regressor_dst=None, use_latent=False, distill_latent=False, lr_sindy=c["lr"],
                              w_sindy_z=0.0, w_sindy_x=1.0, sindy_reg_type="l1", w_sindy_reg=0.0, sym_reg_type="i",
                              w_sym_reg=0.0, st_freq=c["st_freq"], threshold=c["thr"], int_t=0.1, int_dt=0.01, print_eq=False,
                              torch_lbfgs=mode == "torch_lbfgs")
    out = capsys.readouterr().out
    assert "Final convergence reached" in out
    got_mask, want_mask = r.mask.cpu().numpy(), c["mask_final"]
    near = sorted(set(tuple(e["index"]) for e in r.near_threshold))
    near_ball = sorted(set(tuple(e["index"]) for e in r.near_threshold_within(1e-3)))
    differing = [(int(i), int(k)) for i, k in np.argwhere(got_mask != want_mask)]
    print(f"{tag} [{mode}]: near-threshold (1e-4) {near} (reference {sorted(c['near'])}); within the stopping ball (1e-3) "
          f"{near_ball} (reference {sorted(set(c['near_wide']))}); mask bits the reference's 1-ulp runs disagree on "
          f"{c['unstable']}; differing from the recorded mask {differing}")
    explained = set(c["unstable"]) | set(near_ball) | set(c["near_wide"])
    assert all(ix in explained for ix in differing), (differing, sorted(explained))
    if tag == "dosc_n20_o3_edge":
        assert set(c["near"]) <= set(near_ball), (c["near"], near_ball)
    elif not c["unstable"]:
        assert near == [] and c["near"] == []
        assert np.array_equal(got_mask, want_mask)
    got_hist = np.array([l["loss_sindy_x"] for l in logged])
    lo = min(int(c["ulp_logged_epochs"].min()), len(c["loss_hist"])) - 1
    hi = max(int(c["ulp_logged_epochs"].max()), len(c["loss_hist"])) + 1
    print(f"   logged epochs {len(got_hist)} (reference {len(c['loss_hist'])}, its 1-ulp runs {sorted(set(c['ulp_logged_epochs'].tolist()))})")
    assert lo <= len(got_hist) <= hi, (len(got_hist), lo, hi)
    # first epoch = 20 iterations from the same start: 1e-5 -- or four times what ONE ULP in the start does to the
    # reference's own first-epoch loss where that is more (lr 1.0 without a line search: selkov 5e-4, growth 1e-5)
    rel0 = abs(got_hist[0] - c["loss_hist"][0]) / abs(c["loss_hist"][0])
    spread0 = np.abs(c["ulp_first_loss"] - c["loss_hist"][0]).max() / abs(c["loss_hist"][0])
    print(f"   first-epoch loss rel err {rel0:.1e} (the reference's 1-ulp runs: {spread0:.1e})")
    assert rel0 <= max(1e-5, 4 * spread0), (rel0, spread0)
    if np.array_equal(got_mask, want_mask):
        rel_last = abs(got_hist[-1] - c["loss_hist"][-1]) / abs(c["loss_hist"][-1])
        got = r.get_Xi().detach().cpu().numpy()
        err = np.abs((got - c["Xi_final"]) * want_mask).max() / np.abs(c["Xi_final"]).max()
        print(f"   first-epoch loss rel err {rel0:.1e}, last logged loss rel err {rel_last:.1e}, final coefficients scaled err {err:.2e}")
        assert rel_last <= 1e-4, rel_last            # noisy data: the loss sits on a plateau, not at a rounding floor
        # both runs stop when the parameters moved < 1e-3 over an epoch, twice (train.py:705-714): a few of those balls apart
        assert err <= 3e-3, err
