"""GPU parity of the operator layer (SINDyRegression / solve_SINDy*) against golden vectors
and the oracle, through the real HIP engine."""
import numpy as np
import pytest
import torch

from oracle import sindy_oracle as O
from tests.helpers import t

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import symode_amd
    assert torch.cuda.is_available()
    return symode_amd


def make(S, d, order, sine=False, exp=False, L_list=(), thr=0.05, cc=False):
    # the golden vectors were made by the reference on the CPU (torch.linalg.lstsq -> gelsy): pin that driver
    return S.SINDyRegression(d, order, sine, exp, L_list=list(L_list), threshold=thr, device="cuda:0", constrain_constant=cc,
                             lstsq_driver="gelsy")


def test_stlsq_golden_on_gpu(S, golden):
    g = golden("f3_stlsq")
    for tag in g["cases"]:
        d, order, sine, exp = [int(v) for v in g[f"{tag}_cfg"]]
        gamma, thr = [float(v) for v in g[f"{tag}_hp"]]
        x, dx = t(g[f"{tag}_x"]).cuda(), t(g[f"{tag}_dx"]).cuda()
        r = make(S, d, order, bool(sine), bool(exp), thr=thr)
        for wm, wx, wc in zip(g[f"{tag}_masks"], g[f"{tag}_xis"], g[f"{tag}_conv"]):
            res, c = S.solve_SINDy_one_step(r, x, dx, gamma, thr)
            assert np.array_equal(r.mask.cpu().numpy(), wm), tag                 # identical sparsity mask
            assert np.allclose(r.Xi.detach().cpu().numpy(), wx, rtol=1e-5, atol=2e-5 * np.abs(wx).max()), tag
            assert bool(c) == bool(wc) and res.is_cuda and torch.isfinite(res)
        r2 = make(S, d, order, bool(sine), bool(exp), thr=thr)
        S.solve_SINDy(r2, x, dx, gamma, thr)
        assert np.array_equal(r2.mask.cpu().numpy(), g[f"{tag}_solve_mask"]), tag


@pytest.mark.parametrize("tag", ["solve_dosc_so2", "solve_dosc_so2_o3_cc", "solve_growth_scaling2", "solve_growth_scaling2_ac"])
def test_constrained_stlsq_golden_on_gpu(S, golden, tag):
    g = golden("f5_constraint")
    d, order, cc = [int(v) for v in g[f"{tag}_cfg"]]
    gamma, thr = [float(v) for v in g[f"{tag}_hp"]]
    x, dx = t(g[f"{tag}_x"]).cuda(), t(g[f"{tag}_dx"]).cuda()
    r = make(S, d, order, L_list=[t(g[f"{tag}_L"])], thr=thr, cc=bool(cc))
    r.Q = t(g[f"{tag}_Q"]).cuda()
    for wm, wx in zip(g[f"{tag}_masks"], g[f"{tag}_xis"]):
        S.solve_SINDy_one_step(r, x, dx, gamma, thr)
        assert np.array_equal(r.mask.cpu().numpy(), wm)
        assert np.allclose(r.get_Xi().detach().cpu().numpy(), wx, rtol=1e-5, atol=2e-5 * np.abs(wx).max())


@pytest.mark.parametrize("tag", ["solve_dosc_so2", "solve_dosc_so2_o3_cc", "solve_growth_scaling2", "solve_growth_scaling2_ac"])
def test_constrained_stlsq_with_the_products_own_Q(S, golden, tag):
    """Same recorded reference runs, but the null-space basis Q is the one the product's constraint builder makes (not
    the fixture's): product Q -> constrained solve -> Xi, compared through Xi (gauge-free) and the masks."""
    g = golden("f5_constraint")
    d, order, cc = [int(v) for v in g[f"{tag}_cfg"]]
    gamma, thr = [float(v) for v in g[f"{tag}_hp"]]
    x, dx = t(g[f"{tag}_x"]).cuda(), t(g[f"{tag}_dx"]).cuda()
    r = make(S, d, order, L_list=[t(g[f"{tag}_L"])], thr=thr, cc=bool(cc))
    Qp, Qr = r.Q.detach().cpu().double(), t(g[f"{tag}_Q"]).double()
    assert Qp.shape == Qr.shape and bool(r.use_kron_product) == bool(g[f"{tag}_use_kron"])
    assert torch.allclose(Qp @ Qp.T, Qr @ Qr.T, atol=2e-5)                      # same subspace
    for wm, wx in zip(g[f"{tag}_masks"], g[f"{tag}_xis"]):
        S.solve_SINDy_one_step(r, x, dx, gamma, thr)
        got = r.mask.cpu().numpy()
        if thr > 1e-6:
            assert np.array_equal(got, wm)
        else:
            # threshold 1e-9: the coefficients the constraint forces to zero come out as rounding noise of Q (1e-8, a
            # different basis: different noise), on either side of such a threshold -- a mask bit may differ only there
            assert np.all((got == wm) | (np.abs(wx) < 1e-6)), (got, wm)
        assert np.allclose(r.get_Xi().detach().cpu().numpy(), wx, rtol=1e-5, atol=2e-5 * np.abs(wx).max())
    assert (r.near_threshold == []) == (thr > 1e-6), r.near_threshold


@pytest.mark.parametrize("tag", ["o3", "o2e", "d3o2s"])
def test_forward_autograd_and_fused_mse(S, golden, tag):
    g = golden("f2_fwd_loss_grad")
    d, order, sine, exp = [int(v) for v in g[f"{tag}_cfg"]]
    r = make(S, d, order, bool(sine), bool(exp))
    r.Xi.data = t(g[f"{tag}_Xi"]).cuda()
    r.mask = t(g[f"{tag}_mask"]).cuda()
    x, dx = t(g[f"{tag}_x"]).cuda(), t(g[f"{tag}_dx"]).cuda()
    # generic autograd route: forward kernel + vjp kernel, loss assembled by torch
    loss = torch.nn.MSELoss()(r(x), dx) + 0.05 * sum(torch.norm(p, 1) for p in r.parameters())
    loss.backward()
    assert np.isclose(loss.item(), float(g[f"{tag}_loss"]) + 0.05 * float(g[f"{tag}_l1"]), rtol=1e-5)
    scale = np.abs(g[f"{tag}_grad_total"]).max()
    assert np.allclose(r.Xi.grad.cpu().numpy(), g[f"{tag}_grad_total"], rtol=1e-4, atol=2e-5 * scale)
    # fused route
    r.Xi.grad = None
    lf = r.mse_loss(x, dx)
    lf.backward()
    assert np.isclose(lf.item(), float(g[f"{tag}_loss"]), rtol=1e-5)
    assert np.allclose(r.Xi.grad.cpu().numpy(), g[f"{tag}_grad_mse"], rtol=1e-4, atol=2e-5 * np.abs(g[f"{tag}_grad_mse"]).max())


def test_vjp_and_jvp_against_torch_autograd(S):
    torch.manual_seed(4)
    from tests.helpers import only_compiled
    for d, order, sine, exp in only_compiled([(2, 3, False, False), (2, 5, False, False), (3, 2, True, True), (1, 4, True, False), (4, 3, False, True)]):
        r = make(S, d, order, sine, exp)
        r.mask = (torch.rand_like(r.mask) > 0.2).float()
        x = (torch.randn(777, d) * 0.7)
        gout, v = torch.randn(777, d), torch.randn(777, d)
        xg = x.cuda().requires_grad_(True)
        out = r(xg)
        out.backward(gout.cuda())
        # oracle in fp64
        xo = x.double().requires_grad_(True)
        Xi = r.Xi.detach().cpu().double().requires_grad_(True)
        oo = O.forward(xo, Xi, r.mask.cpu().double(), order, sine, exp)
        oo.backward(gout.double())
        assert torch.allclose(out.detach().cpu().double(), oo.detach(), rtol=1e-4, atol=1e-5 * oo.abs().max().item())
        assert torch.allclose(xg.grad.cpu().double(), xo.grad, rtol=1e-4, atol=1e-5 * xo.grad.abs().max().item())
        assert torch.allclose(r.Xi.grad.cpu().double(), Xi.grad, rtol=1e-4, atol=1e-5 * Xi.grad.abs().max().item())
        _, jv = torch.autograd.functional.jvp(lambda a: O.forward(a, Xi.detach(), r.mask.cpu().double(), order, sine, exp), x.double(), v.double())
        got = r.jvp(x.cuda(), v.cuda()).cpu().double()
        assert torch.allclose(got, jv, rtol=1e-4, atol=1e-5 * jv.abs().max().item())


def test_constrained_forward_gradients_reach_beta(S, golden):
    g = golden("f4_lbfgs")
    tag = "dosc_esindy"
    x, dx = t(g[f"{tag}_x"]).cuda(), t(g[f"{tag}_dx"]).cuda()
    r = make(S, 2, 2, L_list=[torch.tensor([[0.0, 1.0], [-1.0, 0.0]])], thr=0.01)
    r.Q = t(g[f"{tag}_Q"]).cuda()
    r.beta.data, r.const.data = t(g[f"{tag}_init_beta"]).cuda(), t(g[f"{tag}_init_const"]).cuda()
    loss = r.mse_loss(x, dx)
    loss.backward()
    reg = O.OracleRegressor(2, 2, L_list=[torch.tensor([[0.0, 1.0], [-1.0, 0.0]])], beta0=t(g[f"{tag}_init_beta"]), const0=t(g[f"{tag}_init_const"]))
    reg.Q = t(g[f"{tag}_Q"])
    lo = torch.nn.functional.mse_loss(reg(t(g[f"{tag}_x"])), t(g[f"{tag}_dx"]))
    lo.backward()
    assert np.isclose(loss.item(), lo.item(), rtol=1e-5)
    assert torch.allclose(r.beta.grad.cpu(), reg.beta.grad, rtol=1e-4, atol=1e-6)
    assert torch.allclose(r.const.grad.cpu(), reg.const.grad, rtol=1e-4, atol=1e-6)


def test_eval_theta_and_leading_dims(S):
    r = make(S, 2, 3)
    x = torch.randn(5, 7, 2)
    assert torch.equal(r.eval_Theta_at(x.cuda()).cpu(), O.theta(x, 3))
    assert r(x.cuda()).shape == (5, 7, 2)
    with pytest.raises(S.SymodeError):
        r(x)            # CPU input: no fallback


def test_gather_gram_and_seed_sweep_on_gpu(S, golden):
    """64-seed sweep (BASELINE config 4 shape, reduced): one gather launch + host solves vs the oracle per seed."""
    from symode_amd.sweep import SeedSweepSTLSQ
    g = golden("f3_stlsq")
    x, dx = t(g["selkov_ridge_x"]), t(g["selkov_ridge_dx"])
    sw = SeedSweepSTLSQ(x.cuda(), dx.cuda(), 3, n_seeds=64, subsample=0.5, seed0=0)
    G = sw.grams()
    for s in (0, 17, 63):
        rows = sw.idx[s].long().cpu()
        A = torch.cat([O.theta(x[rows], 3), dx[rows]], 1).double()
        assert np.allclose(G[s], (A.T @ A).numpy(), rtol=1e-12, atol=0)
    Xi, mask, passes = sw.solve(0.1, 0.075, lstsq_driver="gelsy")
    for s in (0, 5, 17, 40, 63):
        rows = sw.idx[s].long().cpu()
        reg = O.OracleRegressor(2, 3, threshold=0.075, Xi0=torch.zeros(2, 10))
        hist = O.stlsq_until_converged(reg, x[rows], dx[rows], 10, 0.1, 0.075)
        assert np.array_equal(mask[s].numpy(), reg.mask.numpy()), s          # identical sparsity mask per seed
        assert len(hist) == passes[s]
        want = reg.Xi.detach().numpy()
        assert np.allclose(Xi[s].numpy(), want, rtol=1e-5, atol=2e-5 * np.abs(want).max())


def test_lstsq_residual_is_differentiable_wrt_data(S):
    """train_lassi's SINDy branch back-propagates the least-squares residual into the encoder (train.py:166-174):
    d residual / d(z, dz) from the HIP kernels vs autograd through an fp64 ridge solve on the CPU."""
    torch.manual_seed(0)
    n, d, order, gamma = 64, 2, 2, 0.1                       # the rd latent batch: 64 points x 2 dims
    z0, dz0 = torch.randn(n, d) * 0.7, torch.randn(n, d)

    def reference(z, dz, mask):
        th = O.theta(z, order)
        p = th.shape[1]
        total = 0.0
        cols = 0
        for j in range(d):
            idx = mask[j].bool()
            A = torch.cat([th[:, idx], gamma * torch.eye(p, dtype=z.dtype)[:, idx]], 0)
            B = torch.cat([dz[:, j], torch.zeros(p, dtype=z.dtype)])
            w = torch.linalg.solve(A.T @ A, A.T @ B)
            total = total + ((A @ w - B) ** 2).sum()
            cols += 1
        return total

    for partial in (False, True):
        r = S.SINDyRegression(d, order, False, False, threshold=1e-9, device="cuda:0", lstsq_driver="gels")
        mask = torch.ones(d, 6)
        if partial:
            mask[0, 3] = 0.0
            mask[1, 5] = 0.0
        r.mask = mask.cuda()
        z, dz = z0.cuda().requires_grad_(True), dz0.cuda().requires_grad_(True)
        res, _ = S.solve_SINDy_one_step(r, z, dz, gamma, 1e-9)
        res.backward()
        zr, dzr = z0.double().requires_grad_(True), dz0.double().requires_grad_(True)
        want = reference(zr, dzr, mask) / (n * (1 if partial else d))
        want.backward()
        assert np.isclose(res.item(), want.item(), rtol=1e-4)
        assert torch.allclose(dz.grad.cpu().double(), dzr.grad, rtol=1e-3, atol=1e-5 * dzr.grad.abs().max().item())
        assert torch.allclose(z.grad.cpu().double(), zr.grad, rtol=1e-3, atol=1e-4 * zr.grad.abs().max().item())


def test_eval_ltp_accuracy_rollout_matches_chained_steps():
    """evaluation.eval_ltp_accuracy (reference eval_ltp.py): RK4 roll-out of the learned model from x[:, 0] -- the fused
    full-trajectory kernel against the oracle's step-by-step fp32 integration, with and without an autoencoder."""
    import symode_amd
    from symode_amd.evaluation import eval_ltp_accuracy
    x, _ = symode_amd.data.gen_data("dosc", 6, dt=0.2, num_steps=60, seed=1, device="cuda:0")
    r = symode_amd.SINDyRegression(2, 2, False, False, threshold=0.05, device="cuda:0")
    Xi = torch.tensor([[0.0, -0.11, -0.97, 0.0, 0.01, 0.0], [0.0, 1.02, -0.1, 0.0, 0.0, -0.01]])
    r.Xi.data = Xi.cuda()
    res = eval_ltp_accuracy(r, None, x, task="dosc")
    assert res["x_pred"].shape == (6, 59, 2) and res["error"].shape == (6, 59) and np.allclose(res["t"], np.arange(1, 60) * 0.2)
    f = lambda a: O.forward(a, Xi, torch.ones(2, 6), 2)  # noqa: E731
    want = O.odeint(f, x[:, 0].cpu(), 59 * 0.2, 0.2, "rk4", full_traj=True).transpose(0, 1)
    assert np.allclose(res["x_pred"], want.numpy(), rtol=1e-4, atol=1e-5)
    assert np.allclose(res["error"], ((x[:, 1:].cpu() - want) ** 2).mean(-1).numpy(), rtol=1e-3, atol=1e-7)
    assert res["error"][:, :5].max() < 1e-2                      # a near-true model tracks the orbit at first

    class Scale(torch.nn.Module):                                # a linear "autoencoder": z = 2 x
        def encode(self, a):
            return 2.0 * a

        def decode(self, z):
            return 0.5 * z
    r2 = symode_amd.SINDyRegression(2, 2, False, False, threshold=0.05, device="cuda:0")
    r2.Xi.data = torch.tensor([[0.0, -0.11, -0.97, 0.0, 0.005, 0.0], [0.0, 1.02, -0.1, 0.0, 0.0, -0.005]]).cuda()   # same ODE in z = 2x
    res2 = eval_ltp_accuracy(r2, Scale(), x, task="mt_dosc")
    assert np.allclose(res2["x_pred"], res["x_pred"], rtol=1e-3, atol=1e-4)
