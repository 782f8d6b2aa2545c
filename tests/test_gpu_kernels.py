"""GPU parity tests proper: HIP kernels (through the C ABI) vs the CPU oracle and golden vectors.

Tolerances (north_star: coefficients within rtol 1e-5 fp32, masks bit-exact):
  * polynomial library columns: bit-exact (same left-to-right fp32 products as the reference);
  * sin / exp columns: 2 ulp-level relative tolerance (device libm vs torch CPU SLEEF);
  * reductions (loss, gradient): rtol 1e-5 relative to the gradient's scale;
  * fp64 Gram: rtol 1e-12 against an fp64 host product of the same fp32 library.
"""
import os
import re

import numpy as np
import pytest
import torch

from oracle import sindy_oracle as O
from tests.helpers import TinyAE, only_compiled, t

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import symode_amd
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return symode_amd.get_engine()


def dev(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float32).cuda()


def flags_of(sine, exp):
    return (1 if sine else 0) | (2 if exp else 0)


def assert_close_scaled(got, want, rtol=1e-5, what=""):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    scale = max(np.abs(want).max(), 1e-30)
    err = np.abs(got - want).max() / scale
    assert err <= rtol, f"{what}: max scaled error {err:.3e} > {rtol}"


# ----------------------------------------------------------------------------------- theta
def test_theta_golden_bit_exact(eng, golden):
    g = golden("f1_theta")
    for key in g["cases"]:
        d, order, sine, exp = (int(key[1]), int(key[4]), key[7] == "1", key[10] == "1")
        x = dev(g[f"x_d{d}"])
        got = eng.theta(x, order, flags_of(sine, exp)).cpu().numpy()
        want = g["theta_" + key]
        npoly = O.term_count(d, order)
        assert got.shape == want.shape
        assert np.array_equal(got[:, :npoly], want[:, :npoly]), key      # bit-exact monomials
        if want.shape[1] > npoly:
            assert np.allclose(got[:, npoly:], want[:, npoly:], rtol=3e-7, atol=1e-7), key
    got = eng.theta(dev(g["probe_x"]), 3).cpu().numpy()
    assert got[0].tolist() == [1, 2, 3, 4, 6, 9, 8, 12, 18, 27]


@pytest.mark.parametrize("d,order", only_compiled([(1, 5), (2, 4), (2, 5), (3, 4), (4, 3)]))
def test_theta_high_order_vs_oracle(eng, d, order):
    """Orders 4-5 extend the reference ordering (parity unpinned by the reference itself)."""
    torch.manual_seed(d * 10 + order)
    for n in (1, 2, 3, 5, 64, 257, 1000):
        x = torch.randn(n, d)
        got = eng.theta(x.cuda(), order).cpu()
        assert torch.equal(got, O.theta(x, order)), (d, order, n)


def test_theta_leading_dims_and_empty(eng):
    x = torch.randn(7, 5, 2)
    assert torch.equal(eng.theta(x.cuda(), 3).cpu(), O.theta(x, 3))
    assert eng.theta(torch.empty(0, 2).cuda(), 3).shape == (0, 10)


# --------------------------------------------------------------------------------- forward
@pytest.mark.parametrize("tag", ["o3", "o2e", "d3o2s"])
def test_forward_loss_grad_golden(eng, golden, tag):
    g = golden("f2_fwd_loss_grad")
    d, order, sine, exp = [int(v) for v in g[f"{tag}_cfg"]]
    fl = flags_of(sine, exp)
    x, dx, Xi, mask = (dev(g[f"{tag}_{k}"]) for k in ("x", "dx", "Xi", "mask"))
    pred = eng.forward(x, Xi, mask, order, fl).cpu().numpy()
    assert_close_scaled(pred, g[f"{tag}_pred"], 2e-6, "forward")
    loss, grad = eng.loss_grad(x, dx, Xi, mask, order, fl)
    assert np.isclose(loss.item(), float(g[f"{tag}_loss"]), rtol=1e-5)
    assert_close_scaled(grad.cpu().numpy(), g[f"{tag}_grad_mse"], 1e-5, "grad")
    # masked entries get exactly zero gradient, as autograd through Xi*mask gives
    assert np.all(grad.cpu().numpy()[g[f"{tag}_mask"] == 0] == 0.0)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 63, 64, 65, 255, 1023, 4097, 125000])
@pytest.mark.parametrize("d,order,fl", only_compiled([(2, 3, 0), (2, 5, 0), (1, 3, 1), (3, 2, 2), (4, 2, 0), (4, 3, 0), (3, 4, 1), (3, 3, 3), (4, 2, 3)]))
def test_loss_grad_ragged_sizes_vs_oracle(eng, n, d, order, fl):
    torch.manual_seed(n + 7 * d + order)
    x, dx = torch.randn(n, d) * 0.8, torch.randn(n, d)
    p = O.term_count(d, order, bool(fl & 1), bool(fl & 2))
    Xi = torch.randn(d, p) * 0.5
    mask = (torch.rand(d, p) > 0.25).float()
    wl, wg = O.mse_loss_and_grad(x.double(), dx.double(), Xi.double(), mask.double(), order, bool(fl & 1), bool(fl & 2))
    loss, grad = eng.loss_grad(x.cuda(), dx.cuda(), Xi.cuda(), mask.cuda(), order, fl)
    assert np.isclose(loss.item(), wl.item(), rtol=2e-5), (n, d, order)
    assert_close_scaled(grad.cpu().numpy(), wg.numpy(), 2e-5, f"grad n={n}")


def test_loss_grad_unaligned_base_pointer(eng):
    """x[1:] starts 8 bytes into the allocation: the kernel must take its scalar-load path."""
    torch.manual_seed(3)
    x, dx = torch.randn(1001, 2), torch.randn(1001, 2)
    Xi = torch.randn(2, 10)
    xs, dxs = x.cuda()[1:], dx.cuda()[1:]
    assert xs.data_ptr() % 16 != 0
    loss, grad = eng.loss_grad(xs, dxs, Xi.cuda(), None, 3)
    wl, wg = O.mse_loss_and_grad(x[1:].double(), dx[1:].double(), Xi.double(), torch.ones(2, 10).double(), 3)
    assert np.isclose(loss.item(), wl.item(), rtol=2e-5)
    assert_close_scaled(grad.cpu().numpy(), wg.numpy(), 2e-5)


@pytest.mark.parametrize("S,n", [(3, 1000), (5, 999), (64, 2500), (300, 50)])
def test_loss_grad_batched_problems(eng, S, n):
    """Independent (trajectory, seed) problems in one launch, own Xi / mask each (odd n: ragged rows)."""
    torch.manual_seed(S)
    d, order, p = 2, 3, 10
    x, dx = torch.randn(S, n, d) * 0.7, torch.randn(S, n, d)
    Xi, mask = torch.randn(S, d, p), (torch.rand(S, d, p) > 0.3).float()
    loss, grad = eng.loss_grad(x.cuda(), dx.cuda(), Xi.cuda(), mask.cuda(), order)
    assert loss.shape == (S,) and grad.shape == (S, d, p)
    for s in range(0, S, max(1, S // 7)):
        wl, wg = O.mse_loss_and_grad(x[s].double(), dx[s].double(), Xi[s].double(), mask[s].double(), order)
        assert np.isclose(loss[s].item(), wl.item(), rtol=2e-5)
        assert_close_scaled(grad[s].cpu().numpy(), wg.numpy(), 2e-5)


@pytest.mark.parametrize("S,n", [(4, 4096), (3, 10000), (2, 1028)])
def test_loss_grad_batched_three_dimensional_states(eng, S, n):
    """d = 3 (12-byte points): whole waves take the coalesced-tile + LDS-redistribution path, ragged ones the strided loads."""
    torch.manual_seed(S + n)
    d, order = 3, 3
    p = O.term_count(d, order)
    x, dx = torch.randn(S, n, d) * 0.7, torch.randn(S, n, d)
    Xi, mask = torch.randn(S, d, p) * 0.5, (torch.rand(S, d, p) > 0.3).float()
    loss, grad = eng.loss_grad(x.cuda(), dx.cuda(), Xi.cuda(), mask.cuda(), order)
    for s in range(S):
        wl, wg = O.mse_loss_and_grad(x[s].double(), dx[s].double(), Xi[s].double(), mask[s].double(), order)
        assert np.isclose(loss[s].item(), wl.item(), rtol=2e-5)
        assert_close_scaled(grad[s].cpu().numpy(), wg.numpy(), 2e-5)


@pytest.mark.parametrize("d,order,fl,S,n", only_compiled([(4, 3, 0, 3, 5000), (3, 4, 0, 4, 4099), (4, 3, 2, 2, 777)]))
def test_loss_grad_large_libraries_row_per_wave(eng, d, order, fl, S, n):
    """d*p > 64: the row-per-wave kernel (a workgroup = d waves over the same points, wave j owns row j of Xi), batched."""
    torch.manual_seed(S + n)
    p = O.term_count(d, order, bool(fl & 1), bool(fl & 2))
    assert d * p > 64
    x, dx = torch.randn(S, n, d) * 0.6, torch.randn(S, n, d)
    Xi, mask = torch.randn(S, d, p) * 0.3, (torch.rand(S, d, p) > 0.3).float()
    loss, grad = eng.loss_grad(x.cuda(), dx.cuda(), Xi.cuda(), mask.cuda(), order, fl)
    again = eng.loss_grad(x.cuda(), dx.cuda(), Xi.cuda(), mask.cuda(), order, fl)
    assert torch.equal(loss, again[0]) and torch.equal(grad, again[1])               # deterministic
    for s in range(S):
        wl, wg = O.mse_loss_and_grad(x[s].double(), dx[s].double(), Xi[s].double(), mask[s].double(), order, bool(fl & 1), bool(fl & 2))
        assert np.isclose(loss[s].item(), wl.item(), rtol=2e-5)
        assert_close_scaled(grad[s].cpu().numpy(), wg.numpy(), 2e-5)


def test_loss_grad_is_deterministic(eng):
    torch.manual_seed(0)
    x, dx, Xi = torch.randn(125000, 2).cuda(), torch.randn(125000, 2).cuda(), torch.randn(2, 21).cuda()
    a = eng.loss_grad(x, dx, Xi, None, 5)
    a = (a[0].clone(), a[1].clone())
    for _ in range(3):
        b = eng.loss_grad(x, dx, Xi, None, 5)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_loss_grad_inv_count_sharding_identity(eng):
    """Two half-shards with inv_count = 1/(N_global d) add up to the full-batch result."""
    torch.manual_seed(1)
    x, dx, Xi = torch.randn(5000, 2).cuda(), torch.randn(5000, 2).cuda(), torch.randn(2, 10).cuda()
    full = eng.loss_grad(x, dx, Xi, None, 3)
    full = (full[0].clone(), full[1].clone())
    inv = 1.0 / (5000 * 2)
    a = eng.loss_grad(x[:2000], dx[:2000], Xi, None, 3, inv_count=inv)
    a = (a[0].clone(), a[1].clone())
    b = eng.loss_grad(x[2000:], dx[2000:], Xi, None, 3, inv_count=inv)
    assert torch.allclose(a[0] + b[0], full[0], rtol=1e-5)
    assert torch.allclose(a[1] + b[1], full[1], rtol=1e-4, atol=1e-6)


# ---------------------------------------------------------------------------------- odeint
@pytest.mark.parametrize("tag", ["relu_sim2", "tanh_learn"])
def test_odeint_golden(eng, golden, tag):
    g = golden("f6_symreg")
    d, order, sine, exp = [int(v) for v in g[f"{tag}_cfg"]]
    fl = flags_of(sine, exp)
    x, Xi, mask = dev(g[f"{tag}_x"]), dev(g[f"{tag}_Xi"]), dev(g[f"{tag}_mask"])
    K, dt = int(g[f"{tag}_K"]), float(g[f"{tag}_dt"])
    got = eng.odeint(x, Xi, mask, order, fl, K, dt, "euler").cpu().numpy()
    assert np.allclose(got, g[f"{tag}_euler"], rtol=1e-5, atol=1e-6)
    got = eng.odeint(x, Xi, mask, order, fl, K, dt, "rk4").cpu().numpy()
    assert np.allclose(got, g[f"{tag}_rk4"], rtol=1e-5, atol=1e-6)
    assert torch.equal(eng.odeint(x, Xi, mask, order, fl, 0, dt), x)


@pytest.mark.parametrize("n", [1, 255, 256, 1027, 4096, 125000])
@pytest.mark.parametrize("d,order,fl", only_compiled([(3, 3, 0), (3, 2, 3), (1, 4, 0), (2, 3, 0), (4, 2, 0), (2, 5, 0), (2, 4, 0)]))
def test_forward_and_odeint_chunked_streams_vs_oracle(eng, n, d, order, fl):
    """Streaming map kernels at ragged and full sizes for every chunk layout (d = 1: 4 points per 16 bytes, d = 2: 2, d = 3:
    coalesced 192-vector tiles redistributed through LDS for whole waves / strided loads for ragged ones, d = 4: 1)."""
    torch.manual_seed(n + d)
    p = O.term_count(d, order, bool(fl & 1), bool(fl & 2))
    x, Xi = (torch.randn(n, d) * 0.3).clamp(-0.6, 0.6), torch.randn(d, p) * 0.2
    mask = (torch.rand(d, p) > 0.3).float()
    want = O.forward(x, Xi, mask, order, bool(fl & 1), bool(fl & 2))
    got = eng.forward(x.cuda(), Xi.cuda(), mask.cuda(), order, fl)
    assert_close_scaled(got.cpu().numpy(), want.numpy(), 2e-6, f"forward n={n} d={d}")
    f = lambda a: O.forward(a, Xi, mask, order, bool(fl & 1), bool(fl & 2))  # noqa: E731
    want = O.odeint(f, x, 5 * 0.02 + 0.01, 0.02, "rk4")
    got = eng.odeint(x.cuda(), Xi.cuda(), mask.cuda(), order, fl, 5, 0.02, "rk4")
    assert np.allclose(got.cpu().numpy(), want.numpy(), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("order", [3, 5])
@pytest.mark.parametrize("cap", [None, 3])
def test_forward_slabs_of_the_order_4_5_map_every_point_once(eng, order, cap):
    """The forward map gives every workgroup contiguous slabs of four rounds of chunks at order 4-5 (kernels.hpp,
    map_rounds) and one chunk per lane below: sizes around the slab boundaries (2048 points), a ragged last slab, and a
    grid capped to 3 workgroups (SYMODE_MAP_GRID: every workgroup walks several slabs) -- each output row against fp64."""
    torch.manual_seed(order)
    p = O.term_count(2, order)
    Xi, mask = (torch.randn(2, p) * 0.2).cuda(), (torch.rand(2, p) > 0.3).float().cuda()
    with _env(**({} if cap is None else {"SYMODE_MAP_GRID": cap})):
        for n in (2046, 2048, 2050, 4096 + 2, 3 * 2048 + 514, 100003):
            x = (torch.randn(n, 2) * 0.3).clamp(-0.6, 0.6).cuda()
            got = eng.forward(x, Xi, mask, order, 0).double().cpu()
            want = O.theta(x.cpu(), order).double() @ (Xi * mask).double().cpu().T
            assert_close_scaled(got.numpy(), want.numpy(), 2e-6, f"forward order={order} n={n} cap={cap}")


@pytest.mark.parametrize("d,order,fl,method", only_compiled([(2, 3, 0, "rk4"), (2, 2, 2, "rk4"), (2, 5, 0, "euler"), (3, 2, 1, "rk4"), (1, 3, 0, "euler"), (4, 2, 0, "rk4")]))
def test_odeint_full_trajectory_vs_oracle_steps(eng, d, order, fl, method):
    """symode_odeint_traj: the state after EVERY step equals the oracle's chained fp32 steps (odeint(..., full_traj=True));
    its last row is what symode_odeint returns."""
    torch.manual_seed(d * 10 + order)
    n, K, dt = 333, 40, 0.02
    p = O.term_count(d, order, bool(fl & 1), bool(fl & 2))
    x, Xi = (torch.randn(n, d) * 0.3).clamp(-0.6, 0.6), torch.randn(d, p) * 0.2
    mask = (torch.rand(d, p) > 0.3).float()
    for j in range(d):                                          # linear damping keeps every orbit bounded over the window
        Xi[j, 1 + j], mask[j, 1 + j] = -1.0, 1.0
    f = lambda a: O.forward(a, Xi, mask, order, bool(fl & 1), bool(fl & 2))  # noqa: E731
    want = O.odeint(f, x, K * dt + 0.5 * dt, dt, method, full_traj=True)
    assert want.abs().max() < 2.0
    got = eng.odeint_traj(x.cuda(), Xi.cuda(), mask.cuda(), order, fl, K, dt, method)
    assert got.shape == (K, n, d)
    # 40 chained steps of a random polynomial field: rounding differences of the p-term sums grow along the orbit
    assert np.allclose(got.cpu().numpy(), want.numpy(), rtol=3e-4, atol=2e-5)
    assert np.allclose(got[:5].cpu().numpy(), want[:5].numpy(), rtol=2e-5, atol=2e-6)
    assert torch.equal(got[-1], eng.odeint(x.cuda(), Xi.cuda(), mask.cuda(), order, fl, K, dt, method))
    assert eng.odeint_traj(x.cuda(), Xi.cuda(), mask.cuda(), order, fl, 0, dt, method).shape == (0, n, d)


# ------------------------------------------------------------------------------------ gram
@pytest.mark.parametrize("d,order,fl", only_compiled([(2, 2, 0), (2, 3, 0), (2, 5, 0), (2, 2, 2), (3, 3, 0), (1, 4, 3), (4, 3, 0), (3, 4, 1)]))
@pytest.mark.parametrize("n", [1, 63, 256, 257, 5000])
def test_aug_gram_vs_fp64_host(eng, d, order, fl, n):
    torch.manual_seed(n + order)
    x, dx = torch.randn(n, d) * 0.9, torch.randn(n, d)
    A = torch.cat([O.theta(x, order, bool(fl & 1), bool(fl & 2)), dx], dim=1)
    got = eng.aug_gram(x.cuda(), dx.cuda(), order, fl).cpu()
    npoly = O.term_count(d, order)
    # polynomial + dx block: the fp32 features are bit-identical, so only fp64 summation order differs
    idx = list(range(npoly)) + list(range(A.shape[1] - d, A.shape[1]))
    want = A.double().T @ A.double()
    sub_g, sub_w = got[idx][:, idx], want[idx][:, idx]
    assert torch.allclose(sub_g, sub_w, rtol=1e-12, atol=1e-12 * sub_w.abs().max().item())
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-6 * want.abs().max().item())     # sin/exp: libm ulps
    assert torch.equal(got, got.T)


def test_aug_gram_batched_and_asymmetric_tiles(eng):
    """Off-diagonal 16x16 tiles (p+d = 23 > 16) catch a transposed MFMA C/D mapping."""
    torch.manual_seed(5)
    S, n, d, order = 6, 777, 2, 5
    x, dx = torch.rand(S, n, d) + 0.5, torch.randn(S, n, d)
    got = eng.aug_gram(x.cuda(), dx.cuda(), order).cpu()
    for s in range(S):
        A = torch.cat([O.theta(x[s], order), dx[s]], dim=1).double()
        assert torch.allclose(got[s], A.T @ A, rtol=1e-12, atol=0)


def test_gram_reproduces_loss_and_grad_at_full_size(eng):
    """Size-independent property at the BASELINE shape (50 x 2500 x 2, order 5):
    loss and gradient are quadratic forms of the augmented Gram matrix."""
    torch.manual_seed(11)
    N, d, order, p = 125000, 2, 5, 21
    x, dx = (torch.randn(N, d) * 0.6).cuda(), torch.randn(N, d).cuda()
    Xi, mask = torch.randn(d, p).cuda() * 0.3, (torch.rand(d, p) > 0.2).float().cuda()
    loss, grad = eng.loss_grad(x, dx, Xi, mask, order)
    G = eng.aug_gram(x, dx, order)
    W = (Xi * mask).double()
    Gtt, Gty, Gyy = G[:p, :p], G[:p, p:], G[p:, p:]
    want_loss = (torch.trace(W @ Gtt @ W.T) - 2 * torch.trace(W @ Gty) + torch.trace(Gyy)) / (N * d)
    want_grad = 2.0 / (N * d) * (W @ Gtt - Gty.T) * mask.double()
    assert np.isclose(loss.item(), want_loss.item(), rtol=1e-5)
    assert_close_scaled(grad.cpu().numpy(), want_grad.cpu().numpy(), 1e-5)


# --------------------------------------------------------------------------------- sym-reg
@pytest.mark.parametrize("tag,act", [("relu_sim2", "ReLU"), ("tanh_learn", "Tanh")])
def test_symreg_linear_and_reversed_golden(eng, golden, tag, act):
    g = golden("f6_symreg")
    d, order, sine, exp = [int(v) for v in g[f"{tag}_cfg"]]
    fl = flags_of(sine, exp)
    x, Xi, mask = dev(g[f"{tag}_x"]), dev(g[f"{tag}_Xi"]), dev(g[f"{tag}_mask"])
    # S1
    loss, grad = eng.symreg_linear(x, Xi, mask, dev(g[f"{tag}_s1_L"]), order, fl)
    assert np.isclose(loss.item(), float(g[f"{tag}_s1_loss"]), rtol=2e-5)
    assert_close_scaled(grad.cpu().numpy(), g[f"{tag}_s1_grad"], 5e-5, "s1 grad")
    # S4 with (g(x), J_g(x)) precomputed by the oracle from the frozen tiny autoencoder
    ae = TinyAE(g, tag, act)
    gel = [t(e) for e in g[f"{tag}_gelems_r"]]
    gx, Jgx = O.precompute_group_jacobians(t(g[f"{tag}_x"]), ae.encode, ae.decode, ae.z_mean, gel)
    loss, grad = eng.symreg_reversed(x, torch.stack(gx).cuda(), torch.stack(Jgx).cuda(), Xi, mask, order, fl)
    assert np.isclose(loss.item(), float(g[f"{tag}_s4_loss"]), rtol=1e-3)
    assert_close_scaled(grad.cpu().numpy(), g[f"{tag}_s4_grad"], 5e-3, "s4 grad")
    # and tightly against the oracle's own precomputed formulation (same inputs)
    reg = O.OracleRegressor(d, order, bool(sine), bool(exp), Xi0=t(g[f"{tag}_Xi"]))
    reg.mask = t(g[f"{tag}_mask"])
    ol = O.symreg_reversed_precomputed(t(g[f"{tag}_x"]), gx, Jgx, reg)
    ol.backward()
    assert np.isclose(loss.item(), ol.item(), rtol=2e-5)
    assert_close_scaled(grad.cpu().numpy(), reg.Xi.grad.numpy(), 5e-5, "s4 grad vs oracle")


# ------------------------------------------------------------------------- error behaviour
def test_no_cpu_fallback_and_argument_errors(eng):
    import symode_amd
    with pytest.raises(symode_amd.SymodeError):
        eng.theta(torch.randn(4, 2), 3)                      # CPU tensor: refuse, never fall back
    with pytest.raises(symode_amd.SymodeError):
        eng.theta(torch.randn(4, 5).cuda(), 3)               # d = 5 not compiled in
    with pytest.raises(symode_amd.SymodeError):
        eng.loss_grad(torch.randn(4, 2).cuda(), torch.randn(4, 2).cuda(), torch.randn(2, 9).cuda(), None, 3)
    with pytest.raises(symode_amd.SymodeError):
        eng.theta(torch.randn(4, 2).double().cuda(), 3)


@pytest.mark.parametrize("name", ["dosc", "selkov", "lv", "growth"])
def test_rk4_trajectory_kernel_matches_fp64_reference_integrator(eng, name):
    """symode_rk4_traj (one trajectory per thread, fp64) vs the oracle's numpy RK4 of the reference RHS."""
    from symode_amd import data
    rhs = {"dosc": O.rhs_dosc, "selkov": O.rhs_selkov, "lv": O.rhs_lv, "growth": O.rhs_growth}[name]
    ics = {"dosc": O.ics_dosc, "selkov": O.ics_selkov, "lv": O.ics_lv, "growth": O.ics_growth}[name]
    x0 = ics(37, np.random.RandomState(3))
    dt = data.SYSTEMS[name][2]
    xs, dxs = O.rk4_trajectories(rhs, x0, dt, 500)
    for sub in (1, 7):
        gx, gdx = data.rk4_trajectories_fused(name, torch.from_numpy(x0).cuda(), dt, 500, subsample=sub)
        assert gx.shape == (37, (500 + sub - 1) // sub, 2)
        assert np.allclose(gx.cpu().numpy(), xs[:, ::sub].astype(np.float32), rtol=2e-6, atol=1e-7)
        assert np.allclose(gdx.cpu().numpy(), dxs[:, ::sub].astype(np.float32), rtol=2e-6, atol=1e-7)
    a, b = data.make_dataset(name, 4, 300, seed=5, device="cuda", fused=True)
    c, d_ = data.make_dataset(name, 4, 300, seed=5, device="cuda", fused=False)
    assert torch.allclose(a, c, rtol=2e-6, atol=1e-7) and torch.allclose(b, d_, rtol=2e-6, atol=1e-7)


def test_roofline_shape_2_27_points_single_problem(eng):
    """SURVEY H1 shape: one problem of N = 2^27 points (2 GiB of x + dx, far beyond the 256 MiB Infinity Cache).
    Size-independent property: loss and gradient equal the quadratic forms of the fp64 Gram of the same data."""
    N, d, order, p = 1 << 27, 2, 3, 10
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(N, d, device="cuda", generator=g) * 0.6
    dx = torch.randn(N, d, device="cuda", generator=g)
    Xi = (torch.randn(d, p, generator=torch.Generator().manual_seed(1)) * 0.3).cuda()
    loss, grad = eng.loss_grad(x, dx, Xi, None, order)
    G = eng.aug_gram(x, dx, order)
    W = Xi.double()
    want_loss = (torch.trace(W @ G[:p, :p] @ W.T) - 2 * torch.trace(W @ G[:p, p:]) + torch.trace(G[p:, p:])) / (N * d)
    want_grad = 2.0 / (N * d) * (W @ G[:p, :p] - G[:p, p:].T)
    assert np.isclose(loss.item(), want_loss.item(), rtol=2e-5)
    assert_close_scaled(grad.cpu().numpy(), want_grad.cpu().numpy(), 2e-5)
    assert abs(G[0, 0].item() - N) < 0.5                      # the constant column counts the points exactly


def test_plain_c_caller_of_the_abi_runs(tmp_path):
    """examples/capi_demo.c -- a C99 program with no Python and no torch -- drives symode_loss_grad through the C ABI."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from symode_amd import engine
    libdir, rocm = os.path.dirname(engine.LIB_PATH), os.environ.get("ROCM_PATH", "/opt/rocm")
    exe = str(tmp_path / "capi_demo")
    subprocess.run(["gcc", "-std=c99", "-D__HIP_PLATFORM_AMD__", f"-I{rocm}/include", f"-I{root}/include", f"{root}/examples/capi_demo.c",
                    f"-L{libdir}", "-lsymode_hip", f"-L{rocm}/lib", "-lamdhip64", f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{rocm}/lib",
                    "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    loss, grad = [float(v) for v in re.findall(r"[-+]?\d\.\d+e[-+]\d+", out)]
    i = np.arange(125000)
    b = (-1.0 + 2.0 * (i % 613).astype(np.float32) / np.float32(613.0)).astype(np.float64)
    assert "10 terms" in out
    assert np.isclose(loss, np.mean((0.05 * b) ** 2) / 2, rtol=1e-4)              # only dx1 is off: residual 0.05 * x1, mean over n*d
    assert np.isclose(grad, 2 * np.mean(0.05 * b * b) / 2, rtol=1e-4)             # d loss / d Xi[1][x1 column]


# ------------------------------------------------------------- one-launch finalisation (tickets)
class _env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update({k: str(v) for k, v in self.kv.items()})
        _reload_env()                                  # the library reads its SYMODE_* variables once; re-read on request

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        _reload_env()


def _reload_env():
    import symode_amd
    symode_amd.engine.reload_env()


@pytest.mark.parametrize("S,n,d,order", [(1, 125000, 2, 5), (1, 999, 2, 3), (7, 4097, 2, 3), (1, 50000, 3, 2), (64, 20000, 2, 2),
                                         (1, 1 << 22, 2, 3)])
def test_fused_finalize_bit_identical_to_two_launches(eng, S, n, d, order):
    """The last-workgroup-done epilogue adds the partial rows in the order finalize_kernel does: same bits;
    repeated launches on the same workspace keep giving them (the tickets reset themselves).  Both in-launch hand-offs:
    the default (write-through sc1 rows, no fence) and SYMODE_FUSED_FINALIZE=2 (plain rows behind an agent-scope release,
    read behind an agent-scope acquire)."""
    torch.manual_seed(S + n)
    x = (torch.randn(S, n, d) * 0.7).cuda()
    dx = torch.randn(S, n, d).cuda()
    p = eng.lib_size(d, order, 0)
    xi = (torch.randn(S, d, p) * 0.3).cuda()
    mask = (torch.rand(S, d, p) > 0.3).float().cuda()
    with _env(SYMODE_FUSED_FINALIZE=0):
        l0, g0 = eng.loss_grad(x, dx, xi, mask, order)
    with _env(SYMODE_FUSED_FINALIZE=1):
        for _ in range(3):
            l1, g1 = eng.loss_grad(x, dx, xi, mask, order)
            assert torch.equal(l0, l1) and torch.equal(g0, g1)
    with _env(SYMODE_FUSED_FINALIZE=2):                 # the fenced hand-off (agent-scope release / acquire): same bits again
        for _ in range(3):
            l2, g2 = eng.loss_grad(x, dx, xi, mask, order)
            assert torch.equal(l0, l2) and torch.equal(g0, g2)
    # fp64 sums of the oracle's fp32 library: at 4 M random points the oracle's own fp32 matmul is off by 4e-4
    th = O.theta(x[0].cpu(), order).double()
    w = (xi[0] * mask[0]).cpu().double()
    r = th @ w.T - dx[0].cpu().double()
    want_l = (r * r).mean()
    want_g = 2.0 / r.numel() * (r.T @ th) * mask[0].cpu().double()
    assert np.isclose(l1[0].item(), want_l.item(), rtol=2e-5)
    assert_close_scaled(g1[0].cpu(), want_g, 2e-5, "fused finalize grad")


def test_fused_finalize_other_reductions(eng, golden):
    """vjp / jvp_vjp / euler_jvp_vjp / symreg_linear / symreg_reversed share the epilogue."""
    torch.manual_seed(3)
    n, d, order = 30011, 2, 3
    p = eng.lib_size(d, order, 0)
    x, v, g = (torch.randn(n, d).cuda() * 0.5 for _ in range(3))
    xi = (torch.randn(d, p) * 0.3).cuda()
    L = torch.tensor([[[0.0, 1.0], [-1.0, 0.0]]]).cuda()
    gx = x[None] + 0.01 * torch.randn(1, n, d).cuda()
    jgx = torch.eye(d).cuda().expand(1, n, d, d).contiguous() + 0.01 * torch.randn(1, n, d, d).cuda()

    def run():
        return (eng.vjp(x, g, xi, None, order)[1], eng.jvp_vjp(x, v, g, g, xi, None, order)[2],
                eng.euler_jvp_vjp(x, v, g, g, xi, None, order, 0, 5, 0.01)[2],
                *eng.symreg_linear(x, xi, None, L, order), *eng.symreg_reversed(x, gx, jgx, xi, None, order))
    with _env(SYMODE_FUSED_FINALIZE=0):
        a = run()
    with _env(SYMODE_FUSED_FINALIZE=1):
        b = run()
        c = run()
    for u, w, z in zip(a, b, c):
        assert torch.equal(u, w) and torch.equal(u, z)


def test_uninitialised_workspace_gives_nan_not_stale_numbers(eng):
    """A workspace that never saw symode_workspace_init must not produce plausible output."""
    import ctypes
    n, d, order = 5000, 2, 3
    p = eng.lib_size(d, order, 0)
    x, dx = torch.randn(n, d).cuda(), torch.randn(n, d).cuda()
    xi = torch.randn(d, p).cuda()
    need = eng.lib.symode_workspace_bytes(d, order, 0, 1, n)
    ws = torch.full((need // 8 + 1,), 7.0, dtype=torch.float64).cuda()          # garbage header
    loss, grad = torch.zeros(1).cuda(), torch.zeros(d, p).cuda()
    vp = lambda a: ctypes.c_void_p(a.data_ptr())  # noqa: E731
    with _env(SYMODE_FUSED_FINALIZE=1):
        rc = eng.lib.symode_loss_grad(vp(x), vp(dx), 1, n, d, order, 0, vp(xi), None, 1.0 / (n * d), vp(loss), vp(grad), vp(ws),
                                      ws.numel() * 8, None)
        torch.cuda.synchronize()
        assert rc == 0 and torch.isnan(loss).all() and torch.isnan(grad[0, 0])
        assert eng.lib.symode_workspace_init(vp(ws), ws.numel() * 8, None) == 0
        rc = eng.lib.symode_loss_grad(vp(x), vp(dx), 1, n, d, order, 0, vp(xi), None, 1.0 / (n * d), vp(loss), vp(grad), vp(ws),
                                      ws.numel() * 8, None)
        torch.cuda.synchronize()
    want_l, want_g = O.mse_loss_and_grad(x.cpu(), dx.cpu(), xi.cpu(), torch.ones(d, p), order)
    assert rc == 0 and np.isclose(loss.item(), want_l.item(), rtol=2e-5)
    assert_close_scaled(grad.cpu(), want_g, 2e-5, "grad after init")


def test_zero_copy_closure_matches_copy_path(eng):
    """_HostShadow: Xi read from / [loss | grad] written to pinned host memory by ONE launch == the copy path."""
    from symode_amd.sindy import SINDyRegression
    from symode_amd.train import _HostShadow
    torch.manual_seed(0)
    x, dx = torch.randn(125000, 2).cuda(), torch.randn(125000, 2).cuda()
    reg = SINDyRegression(2, 3, False, False, threshold=0.05, device="cuda:0")
    sh = _HostShadow(reg, x, dx, numpy_vars=False, use_graph=False, zero_copy=True)
    assert sh.zero_copy, "pinned host memory is not device-visible on this box?"
    _, vals, grads = sh.evaluate()
    sh2 = _HostShadow(reg, x, dx, numpy_vars=False, use_graph=True, zero_copy=False)
    _, vals2, grads2 = sh2.evaluate()
    assert torch.equal(vals[0], vals2[0]) and torch.equal(grads[0], grads2[0])


# ------------------------------------------------------------- batched reversed symmetry regulariser
@pytest.mark.parametrize("S,n,n_g,d,order,fl", [c for c in [(1, 20000, 1, 2, 2, 2), (3, 4096, 2, 2, 3, 0), (2, 1001, 1, 2, 5, 0), (1, 777, 3, 3, 2, 1),
                                                            (4, 512, 1, 1, 4, 0), (2, 300, 2, 4, 2, 0)] if only_compiled([c[3:5]])])
def test_symreg_reversed_batched_vs_oracle(eng, S, n, n_g, d, order, fl):
    """symode_symreg_reversed_batched (16-byte non-temporal chunk loads of x, g(x), J_g; ragged tails and unaligned slabs
    fall back to per-point loads) against the oracle's model_utils.py:166-168 with the explicit matvec."""
    torch.manual_seed(S * 100 + n)
    p = eng.lib_size(d, order, fl)
    x = torch.randn(S, n, d) * 0.5
    gx = x[:, None] + 0.05 * torch.randn(S, n_g, n, d)
    jgx = torch.eye(d) + 0.05 * torch.randn(S, n_g, n, d, d)
    xi = torch.randn(S, d, p) * 0.3
    mask = (torch.rand(S, d, p) > 0.25).float()
    loss, grad = eng.symreg_reversed(x.cuda(), gx.cuda(), jgx.cuda(), xi.cuda(), mask.cuda(), order, fl)
    assert loss.shape == (S,) and grad.shape == (S, d, p)
    for s in range(S):
        reg = O.OracleRegressor(d, order, bool(fl & 1), bool(fl & 2), Xi0=xi[s])
        reg.mask = mask[s]
        want = O.symreg_reversed_precomputed(x[s], list(gx[s]), list(jgx[s]), reg)
        want.backward()
        assert np.isclose(loss[s].item(), want.item(), rtol=2e-5), (s, loss[s].item(), want.item())
        assert_close_scaled(grad[s].cpu(), reg.Xi.grad * mask[s], 3e-5, f"symreg_reversed grad, problem {s}")
    # the one-problem form (and an unaligned view of it) gives the same numbers as row 0 of the batch
    l1, g1 = eng.symreg_reversed(x[0].cuda(), gx[0].cuda(), jgx[0].cuda(), xi[0].cuda(), mask[0].cuda(), order, fl)
    assert np.isclose(l1.item(), loss[0].item(), rtol=1e-6)
    pad = torch.zeros(n * d + 1).cuda()
    xu = pad[1:].view(n, d)
    xu.copy_(x[0])
    l2, g2 = eng.symreg_reversed(xu, gx[0].cuda(), jgx[0].cuda(), xi[0].cuda(), mask[0].cuda(), order, fl)
    assert np.isclose(l2.item(), l1.item(), rtol=1e-5)
    assert_close_scaled(g2.cpu(), g1.cpu(), 1e-5, "unaligned x")


def test_bench_starts_its_own_ranks(tmp_path):
    """``python bench.py --gpus 2`` with no torchrun environment spawns two ranks itself; on this one-GPU box both use
    cuda:0 and the collectives go through gloo (--rehearse_gloo).  The JSON line must say two ranks ran."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse_gloo", "--problems", "64", "--steps", "3",
                          "--warmup", "1", "--no_cpu_baseline", "--profile", "--config3", "--master_port", "29533"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["ranks_observed"] == 2 and rec["warmup"] == 1 and rec["steps"] == 3
    assert rec["metric"].endswith("sym-reg") and len(rec["roofline_legs"]) == 1 and rec["value"] > 0      # ONE fused closure kernel per step
    assert "x read once" in rec["roofline"]["kernel"] and rec["roofline"]["bytes_per_point"] == 40
    # BASELINE config[3]'s sharded work: 64-seed index-table Gram on each rank's half of every subsample + ONE fp64
    # all-reduce of the (64, 12, 12) stack, checked against the single-rank Gram of the whole subsample
    c3 = rec["config3_gram_allreduce"]
    assert "error" not in c3, c3
    assert c3["ranks"] == 2 and c3["ranks_observed"] == 2 and c3["allreduce_bytes"] == 64 * 144 * 8
    assert c3["gram_kernel_us"] > 0 and c3["allreduce_us"] > 0 and c3["allreduced_vs_single_rank_max_rel_err"] < 1e-12


def test_bench_collective_path_on_rccl_with_one_rank(tmp_path):
    """RCCL refuses two ranks on one GPU ("Duplicate GPU detected"), so on this box the real backend can carry ONE rank
    only: ``bench.py --force_dist`` initialises the "nccl" process group and sends the step's packed [loss | grad] buffers
    and the config[3] Gram stack through RCCL's all-reduce on the kernels' stream -- the value must equal the
    collective-free run's (an all-reduce over one rank is the identity) and the line must say the collective ran."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force_dist", "--problems", "64", "--steps", "3", "--warmup", "1",
                          "--no_cpu_baseline", "--profile", "--config3", "--master_port", "29541"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["config"]["ranks_observed"] == 1 and rec["value"] > 0
    c3 = rec["config3_gram_allreduce"]
    assert "error" not in c3, c3
    assert c3["allreduce_us"] is not None and c3["allreduce_us"] > 0 and "gloo" not in str(c3.get("allreduce_backend", ""))
    assert c3["allreduced_vs_single_rank_max_rel_err"] == 0.0


@pytest.mark.parametrize("d,order,fl,K,n", only_compiled([(2, 2, 2, 10, 20000), (2, 3, 0, 3, 4097), (3, 2, 0, 5, 3001), (1, 4, 1, 16, 1000),
                                                          (2, 5, 0, 20, 2500), (4, 2, 0, 4, 999)]))
def test_euler_reverse_sweep_state_stack_equals_recompute(eng, d, order, fl, K, n):
    """euler_jvp_vjp keeps the K step states in an LDS column (K <= 16 at d = 2) or recomputes them (larger K,
    SYMODE_EULER_STACK=0): the same adjoints either way, and both match torch autograd through the stepwise flow."""
    torch.manual_seed(K * 7 + n)
    p = eng.lib_size(d, order, fl)
    x, v, gx_, gt_ = ((torch.randn(n, d) * 0.4).cuda() for _ in range(4))
    xi = (torch.randn(d, p) * 0.2).cuda()
    mask = (torch.rand(d, p) > 0.2).float().cuda()
    with _env(SYMODE_EULER_STACK=0):
        a = eng.euler_jvp_vjp(x, v, gx_, gt_, xi, mask, order, fl, K, 0.01)
    with _env(SYMODE_EULER_STACK=1):
        b = eng.euler_jvp_vjp(x, v, gx_, gt_, xi, mask, order, fl, K, 0.01)
    for u, w_, nm in zip(a, b, ("grad_x", "grad_v", "grad_xi")):
        assert_close_scaled(w_.cpu(), u.cpu(), 2e-6, f"stack vs recompute {nm}")
    # reference: fp64 autograd through K explicit steps of the oracle's forward
    X = x.cpu().double().requires_grad_(True)
    V = v.cpu().double().requires_grad_(True)
    W = xi.cpu().double().requires_grad_(True)
    M = mask.cpu().double()
    f = lambda a_: O.forward(a_, W, M, order, bool(fl & 1), bool(fl & 2))  # noqa: E731
    xs, ts = X, V
    for _ in range(K):
        h, jt = torch.autograd.functional.jvp(f, xs, ts, create_graph=True)
        xs, ts = xs + 0.01 * h, ts + 0.01 * jt
    ((xs * gx_.cpu().double()).sum() + (ts * gt_.cpu().double()).sum()).backward()
    for got, want, nm in zip(b, (X.grad, V.grad, W.grad * M), ("grad_x", "grad_v", "grad_xi")):
        assert_close_scaled(got.cpu(), want, 3e-5, f"euler_jvp_vjp {nm} vs fp64 autograd")
    xo, to = eng.euler_jvp(x, v, xi, mask, order, fl, K, 0.01)
    assert_close_scaled(xo.cpu(), xs.detach(), 1e-5, "euler_jvp x_K")
    assert_close_scaled(to.cpu(), ts.detach(), 1e-5, "euler_jvp t_K")


@pytest.mark.parametrize("T,K,d,order,fl", [c for c in [(2000, 50, 2, 3, 0), (777, 50, 2, 2, 2), (8000, 50, 2, 5, 0), (300, 7, 3, 2, 1),
                                                        (64, 128, 1, 4, 0), (1000, 20, 4, 3, 0)] if only_compiled([c[2:4]])])
def test_weak_gram_fused_contraction(eng, T, K, d, order, fl):
    """symode_weak_gram: G = V Theta(x), b = -V' x accumulated by the fp64 matrix cores vs an fp64 host product of the
    same fp32 library and test functions (exact products, fp64 sums: 1e-12 like the Gram)."""
    torch.manual_seed(T + K)
    x = torch.randn(T, d) * 0.6
    V, Vd = torch.randn(K, T) * 0.1, torch.randn(K, T)
    G, b = eng.weak_gram(x.cuda(), V.cuda(), Vd.cuda(), order, fl)
    th = O.theta(x, order, bool(fl & 1), bool(fl & 2))
    th_gpu = eng.theta(x.cuda(), order, fl).cpu()                  # sin / exp columns: the device's own fp32 values
    want_G, want_b = V.double() @ th_gpu.double(), -(Vd.double() @ x.double())
    assert G.shape == (K, th.shape[1]) and b.shape == (K, d)
    assert torch.allclose(G.cpu(), want_G, rtol=1e-12, atol=1e-12 * want_G.abs().max().item())
    assert torch.allclose(b.cpu(), want_b, rtol=1e-12, atol=1e-12 * want_b.abs().max().item())


@pytest.mark.parametrize("S,n,d,order,fl", [c for c in [(1, 125000, 2, 3, 0), (5, 4099, 2, 2, 2), (3, 1000, 2, 2, 0), (2, 777, 1, 5, 2),
                                                        (1, 3000, 3, 1, 0), (2, 5000, 2, 5, 0), (2, 3001, 2, 4, 0), (1, 2050, 2, 4, 3),
                                                        (3, 999, 3, 2, 0), (2, 1500, 4, 2, 0), (1, 70001, 3, 3, 0), (1, 8, 2, 5, 0)]
                                            if only_compiled([c[2:4]])])
def test_gram_vector_pipe_and_matrix_core_forms_agree(eng, S, n, d, order, fl):
    """Small libraries (F = p + d <= 12) take the fp64 vector-pipe Gram (one fma per distinct entry), 12 < F <= 24 the 4x4-tile
    matrix-core form (gram_m4.hpp: F = 13, 17, 19, 21, 23 here; SYMODE_GRAM_M4=0: the split vector-pipe form, the triangle
    in 2, 3 or 4 runs over sibling workgroups), larger ones and SYMODE_GRAM_VALU=0 the 16x16-tile MFMA form: all are fp64
    sums of exact products -- equal to 1e-12, and to the host's."""
    torch.manual_seed(n)
    x, dx = (torch.randn(S, n, d) * 0.7).cuda(), torch.randn(S, n, d).cuda()
    forms = [dict(SYMODE_GRAM_M4=1, SYMODE_GRAM_VALU=1), dict(SYMODE_GRAM_M4=0, SYMODE_GRAM_VALU=1), dict(SYMODE_GRAM_M4=0, SYMODE_GRAM_VALU=0)]
    got = []
    for env in forms:
        with _env(**env):
            got.append(eng.aug_gram(x, dx, order, fl))
    th = eng.theta(x.reshape(-1, d), order, fl).reshape(S, n, -1)
    A = torch.cat([th, dx], dim=2).double().cpu()
    want = A.transpose(1, 2) @ A
    for g in got:
        assert torch.allclose(g.cpu(), want, rtol=1e-12, atol=1e-12 * want.abs().max().item())
        assert torch.equal(g, g.transpose(1, 2))
    # the index-table form (seed sweeps) through all of them
    idx = torch.stack([torch.randperm(n)[: n // 2] for _ in range(4)]).int().cuda()
    gat = []
    for env in forms:
        with _env(**env):
            gat.append(eng.aug_gram_gather(x[0], dx[0], idx, order, fl))
    Ai = A[0][idx[1].long().cpu()]
    for g in gat:
        assert torch.allclose(g, gat[-1], rtol=1e-12, atol=1e-12 * gat[-1].abs().max().item())
        assert torch.allclose(g[1].cpu(), Ai.T @ Ai, rtol=1e-12, atol=1e-12 * want.abs().max().item())


@pytest.mark.parametrize("S,n,n_g,d,order,fl", [(1, 20000, 1, 2, 2, 2), (3, 4096, 2, 2, 3, 0), (2, 125000, 1, 2, 5, 0), (1, 777, 3, 3, 2, 1), (2, 301, 1, 1, 4, 0)])
def test_fused_closure_mse_plus_reversed_regulariser(eng, S, n, n_g, d, order, fl):
    """symode_loss_grad_reversed: residual and regulariser in one pass (x read once) == the two separate launches, and
    the oracle's MSE + w * symreg_reversed_precomputed with its autograd gradient."""
    torch.manual_seed(S + n)
    p = eng.lib_size(d, order, fl)
    x = torch.randn(S, n, d) * 0.5
    dx = torch.randn(S, n, d)
    gx = x[:, None] + 0.05 * torch.randn(S, n_g, n, d)
    jgx = torch.eye(d) + 0.05 * torch.randn(S, n_g, n, d, d)
    xi = torch.randn(S, d, p) * 0.3
    mask = (torch.rand(S, d, p) > 0.25).float()
    w = 0.37
    c = [a.cuda() for a in (x, dx, gx, jgx, xi, mask)]
    loss2, grad = eng.loss_grad_reversed(c[0], c[1], c[2], c[3], c[4], c[5], order, fl, w_sym=w)
    l_m, g_m = eng.loss_grad(c[0], c[1], c[4], c[5], order, fl)
    l_s, g_s = eng.symreg_reversed(c[0], c[2], c[3], c[4], c[5], order, fl)
    assert loss2.shape == (S, 2) and grad.shape == (S, d, p)
    assert torch.allclose(loss2[:, 0], l_m, rtol=2e-6) and torch.allclose(loss2[:, 1], l_s, rtol=2e-6)
    assert_close_scaled(grad.cpu(), (g_m + w * g_s).cpu(), 3e-6, "fused vs two launches")
    for s in range(S):
        reg = O.OracleRegressor(d, order, bool(fl & 1), bool(fl & 2), Xi0=xi[s])
        reg.mask = mask[s]
        mse = torch.nn.functional.mse_loss(reg(x[s]), dx[s])
        sym = O.symreg_reversed_precomputed(x[s], list(gx[s]), list(jgx[s]), reg)
        (mse + w * sym).backward()
        assert np.isclose(loss2[s, 0].item(), mse.item(), rtol=2e-5) and np.isclose(loss2[s, 1].item(), sym.item(), rtol=2e-5)
        assert_close_scaled(grad[s].cpu(), reg.Xi.grad * mask[s], 3e-5, f"fused closure grad, problem {s}")


def test_ticket_handoff_under_uneven_load_every_word(eng):
    """The one-launch finalisation hands partial rows across workgroups without fences (write-through stores, drained,
    one ticket add per workgroup, L1-bypassing loads).  Hammer it the way such hand-offs fail when they fail: thousands
    of launches on ONE workspace whose partial rows change every launch (two alternating coefficient sets, so a stale row
    from the previous launch would change the result), several grid shapes, a second stream keeping the memory system
    busy -- every output word compared bit for bit with the two-launch form."""
    torch.manual_seed(5)
    cases = [(1, 125000, 2, 5), (1, 20000, 2, 3), (16, 9000, 2, 3), (1, 700000, 2, 2)]
    side = torch.cuda.Stream()
    big_a = torch.randn(1 << 26, device="cuda")
    big_b = torch.empty_like(big_a)
    for S, n, d, order in cases:
        p = eng.lib_size(d, order, 0)
        x, dx = (torch.randn(S, n, d) * 0.7).cuda(), torch.randn(S, n, d).cuda()
        xis = [(torch.randn(S, d, p) * 0.3).cuda() for _ in range(2)]
        with _env(SYMODE_FUSED_FINALIZE=0):
            want = [tuple(t_.clone() for t_ in eng.loss_grad(x, dx, xi, None, order)) for xi in xis]
        outs = [(torch.empty(S, device="cuda"), torch.empty(S, d, p, device="cuda")) for _ in range(2)]
        bad = torch.zeros((), dtype=torch.int64, device="cuda")
        with _env(SYMODE_FUSED_FINALIZE=1):
            for it in range(1500):
                if it % 50 == 0:                           # uneven background load from another stream
                    with torch.cuda.stream(side):
                        big_b.copy_(big_a)
                k = it & 1
                eng.loss_grad(x, dx, xis[k], None, order, out=outs[k])
                bad += (outs[k][0] != want[k][0]).sum() + (outs[k][1] != want[k][1]).sum()
        torch.cuda.synchronize()
        assert int(bad.item()) == 0, (S, n, order, int(bad.item()))


def test_dpp_wave_sum_is_bit_identical_to_the_shuffle_butterfly(eng):
    """reduce.hpp::wave_sum_dpp (v_permlane32/16_swap + DPP row / quad permutes; the L-BFGS kernels' dot products) pairs the
    same lanes in the same order as the __shfl_xor butterfly: every lane of every wave must hold the same bits, on
    random data, data spanning 30 binades (where association order shows) and a one-hot pattern per lane (routing)."""
    import ctypes
    torch.manual_seed(0)
    onehot = torch.eye(64, device="cuda").reshape(-1) * torch.arange(1, 65, device="cuda").repeat_interleave(64).float()
    wide = torch.randn(64 * 300, device="cuda") * torch.exp2(torch.randint(-15, 15, (64 * 300,), device="cuda").float())
    for data in (torch.randn(64 * 500, device="cuda"), wide, onehot):
        a, b = torch.empty_like(data), torch.empty_like(data)
        rc = eng.lib.symode_selftest_wave_sum(ctypes.c_void_p(data.data_ptr()), ctypes.c_void_p(a.data_ptr()),
                                              ctypes.c_void_p(b.data_ptr()), data.numel() // 64,
                                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
        assert torch.allclose(a.view(-1, 64)[:, 0], data.view(-1, 64).double().sum(1).float(), rtol=1e-4, atol=1e-4 * data.abs().max().item())


@pytest.mark.parametrize("n", [299_999, 300_001, 1_500_001, 2_000_003])
def test_single_problem_launch_size_classes_vs_fp64(eng, n):
    """The workgroup cap of a single-problem reduction launch changes with the problem's size class (capi.hip:
    small_grid_cap / single_problem_grid; class edges at 300 K, 1.5 M, 2 M points).  Either side of every edge: loss and
    gradient of the closure, the fused closure, the regulariser alone, vjp (both outputs) and the linear-latent
    regulariser against an fp64 evaluation of the oracle's op sequence (torch ops on the GPU: same formulas, wider type)."""
    d, order, p = 2, 3, 10
    gen = torch.Generator(device="cuda").manual_seed(n)
    x = torch.randn(n, d, device="cuda", generator=gen) * 0.6
    dx = torch.randn(n, d, device="cuda", generator=gen)
    g = torch.randn(n, d, device="cuda", generator=gen)
    gx = x + 0.05 * torch.randn(n, d, device="cuda", generator=gen)
    jg = torch.eye(d, device="cuda") + 0.05 * torch.randn(n, d, d, device="cuda", generator=gen)
    Xi = (torch.randn(d, p, generator=torch.Generator().manual_seed(3)) * 0.3).cuda()
    L = torch.tensor([[[0.0, 1.0], [-1.0, 0.0]]], device="cuda")

    W = Xi.double().requires_grad_(True)
    xd = x.double().requires_grad_(True)
    h = lambda a: O.theta(a, order) @ W.T  # noqa: E731
    hx = h(xd)
    mse = ((hx - dx.double()) ** 2).mean()
    sym = ((torch.einsum("bij,bj->bi", jg.double(), hx) - h(gx.double())) ** 2).mean()
    g_mse, = torch.autograd.grad(mse, W, retain_graph=True)
    g_sym, = torch.autograd.grad(sym, W, retain_graph=True)
    gW, gX = torch.autograd.grad((hx * g.double()).sum(), (W, xd))

    loss, grad = eng.loss_grad(x, dx, Xi, None, order)
    assert np.isclose(loss.item(), mse.item(), rtol=2e-5)
    assert_close_scaled(grad.cpu().numpy(), g_mse.cpu().numpy(), 2e-5, "loss_grad")
    l2, gf = eng.loss_grad_reversed(x[None], dx[None], gx[None, None], jg[None, None], Xi[None], None, order, 0, w_sym=0.7)
    assert np.isclose(l2[0, 0].item(), mse.item(), rtol=2e-5) and np.isclose(l2[0, 1].item(), sym.item(), rtol=2e-5)
    assert_close_scaled(gf[0].cpu().numpy(), (g_mse + 0.7 * g_sym).cpu().numpy(), 2e-5, "fused closure")
    ls, gs = eng.symreg_reversed(x, gx[None], jg[None], Xi, None, order)
    assert np.isclose(ls.item(), sym.item(), rtol=2e-5)
    assert_close_scaled(gs.cpu().numpy(), g_sym.cpu().numpy(), 2e-5, "symreg_reversed")
    gxv, gxi = eng.vjp(x, g, Xi, None, order)
    assert_close_scaled(gxi.cpu().numpy(), gW.cpu().numpy(), 2e-5, "vjp grad_xi")
    assert_close_scaled(gxv.cpu().numpy(), gX.cpu().numpy(), 1e-5, "vjp grad_x")
    _, gxi2 = eng.vjp(x, g, Xi, None, order, need_grad_x=False)
    assert_close_scaled(gxi2.cpu().numpy(), gW.cpu().numpy(), 2e-5, "vjp grad_xi (no grad_x)")
    del hx, mse, sym, gW, gX
    W2 = Xi.double().requires_grad_(True)
    s1 = O.symreg_linear_latent(x.double(), [L[0].double()], lambda a: O.theta(a, order) @ W2.T)
    g1, = torch.autograd.grad(s1, W2)
    l1, gr1 = eng.symreg_linear(x, Xi, None, L, order)
    assert np.isclose(l1.item(), s1.item(), rtol=2e-5)
    assert_close_scaled(gr1.cpu().numpy(), g1.cpu().numpy(), 2e-5, "symreg_linear")


# ------------------------------------------------------------------------ seeded subsamples (seed sweeps)
def _subsample_keys(seed, n):
    """numpy restatement of subsample.hip::subsample_key (unsigned arithmetic wraps)"""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0x632BE59BD9B4E019)
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
        a, b = np.uint32(int(z) & 0xFFFFFFFF), np.uint32(int(z) >> 32)
        h = (np.arange(n, dtype=np.uint32) ^ a) * np.uint32(0x9E3779B1) + b
        h ^= h >> np.uint32(16)
        h *= np.uint32(0x85EBCA6B)
        h ^= h >> np.uint32(13)
        h *= np.uint32(0xC2B2AE35)
        h ^= h >> np.uint32(16)
    return h


@pytest.mark.parametrize("n,m", [(1, 1), (2, 1), (1000, 1), (1000, 999), (1024, 512), (1025, 3), (100000, 50000), (300001, 123457)])
def test_seeded_subsamples_are_the_m_smallest_keys_in_row_order(eng, n, m):
    """symode_seeded_subsamples: per seed the m rows with the smallest key(seed, row) (ties: the lower rows), ascending;
    a seed's rows do not depend on the seeds beside it."""
    seeds = [0, 1, 7, 1_000_003 * 5 + 2, 2 ** 40 + 17]
    got = eng.seeded_subsamples(n, m, seeds, "cuda").cpu().numpy()
    assert got.shape == (len(seeds), m) and got.dtype == np.int32
    for row, seed in zip(got, seeds):
        keys = _subsample_keys(seed, n)
        want = np.sort(np.argsort(keys, kind="stable")[:m])
        assert np.array_equal(row, want), (seed, n, m)
    alone = eng.seeded_subsamples(n, m, [seeds[3]], "cuda").cpu().numpy()
    assert np.array_equal(alone[0], got[3])


def test_seeded_subsamples_large_table_and_inclusion_statistics(eng):
    """2^22 rows (the key is a bijection of the row: no ties); every row is in about half of 64 seeds' half-size subsets, and
    two seeds' subsets overlap like independent draws."""
    n, m = 1 << 22, 1 << 21
    got = eng.seeded_subsamples(n, m, [3], "cuda").cpu().numpy()[0]
    keys = _subsample_keys(3, n)
    assert len(np.unique(keys)) == n
    assert np.array_equal(got, np.sort(np.argsort(keys, kind="stable")[:m]))
    tab = eng.seeded_subsamples(20000, 10000, list(range(64)), "cuda")
    counts = torch.zeros(20000, device="cuda").index_add_(0, tab.reshape(-1).long(), torch.ones(tab.numel(), device="cuda"))
    assert abs(counts.mean().item() - 32.0) < 1e-6 and 10 < counts.min().item() and counts.max().item() < 54
    assert abs(counts.std().item() - 4.0) < 0.3                              # binomial(64, 1/2)
    member = torch.zeros(64, 20000, device="cuda")
    member.scatter_(1, tab.long(), 1.0)
    overlap = (member @ member.T).cpu().numpy()                              # pairs of seeds: hypergeometric, mean 5000, sd 35
    off = overlap[~np.eye(64, dtype=bool)]
    assert abs(off.mean() - 5000.0) < 20.0 and np.abs(off - 5000.0).max() < 250.0, (off.mean(), off.min(), off.max())


def test_index_table_is_checked_once_per_table_and_again_after_a_change(eng):
    """aug_gram_gather trusts a table it has checked (the check is a reduction + a sync, dearer than the kernel at config[3]'s
    size) -- by tensor object and in-place version: an out-of-range row is refused on the first call, and a table spoiled in
    place AFTER a good call is refused as well."""
    import symode_amd
    x, dx = torch.randn(1000, 2).cuda(), torch.randn(1000, 2).cuda()
    idx = torch.arange(0, 1000, 2, dtype=torch.int32).repeat(3, 1).cuda()
    bad = idx.clone()
    bad[1, 7] = 1000
    with pytest.raises(symode_amd.engine.SymodeError):
        eng.aug_gram_gather(x, dx, bad, 3)
    a = eng.aug_gram_gather(x, dx, idx, 3)
    b = eng.aug_gram_gather(x, dx, idx, 3)                       # the remembered table
    assert torch.equal(a, b)
    idx[2, 0] = -1
    with pytest.raises(symode_amd.engine.SymodeError):
        eng.aug_gram_gather(x, dx, idx, 3)
    other = torch.arange(1, 1000, 2, dtype=torch.int32).repeat(3, 1).cuda()
    c = eng.aug_gram_gather(x, dx, other, 3)
    A = torch.cat([eng.theta(x[1::2], 3), dx[1::2]], dim=1).double()
    assert torch.allclose(c[0], A.T @ A, rtol=1e-12, atol=0)
