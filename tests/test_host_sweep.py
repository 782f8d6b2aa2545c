"""Batched L-BFGS seed sweep vs the sequential reference loop (oracle lbfgs_fit), on CPU tensors
with the oracle-backed engine: same masks, same coefficients, per seed."""
import numpy as np
import torch

import symode_amd  # noqa: F401
from oracle import sindy_oracle as O
from symode_amd.batched import BatchedClosure
from symode_amd.constraint import constraint_Q
from symode_amd.sweep import BatchedLBFGS, SeedSweepLBFGS
from tests.helpers import t
from tests.oracle_engine import OracleEngine

torch.set_num_threads(4)


def test_batched_lbfgs_equals_torch_lbfgs_on_quadratics():
    """Pure optimiser check: S random convex quadratics, BatchedLBFGS vs torch.optim.LBFGS step by step."""
    torch.manual_seed(0)
    S, n = 6, 7
    A = torch.randn(S, n, n)
    A = A @ A.transpose(1, 2) + 0.5 * torch.eye(n)
    b = torch.randn(S, n)
    P = torch.randn(S, n)
    ref = [P[s].clone().requires_grad_(True) for s in range(S)]
    opts = [torch.optim.LBFGS([ref[s]], lr=0.3) for s in range(S)]

    def batched(Pm):
        AP = torch.einsum("sij,sj->si", A, Pm)
        return 0.5 * (Pm * AP).sum(1) - (b * Pm).sum(1), AP - b

    opt = BatchedLBFGS(P, 0.3)
    for step in range(4):
        opt.step(batched)
        for s in range(S):
            def cl():
                opts[s].zero_grad()
                l = 0.5 * ref[s] @ A[s] @ ref[s] - b[s] @ ref[s]
                l.backward()
                return l
            opts[s].step(cl)
        got = P.numpy()
        want = np.stack([r.detach().numpy() for r in ref])
        assert np.allclose(got, want, rtol=2e-3, atol=2e-4), step


def _sweep(x, dx, order, inits, lr, st_freq, thr, epochs, Q=None):
    S = inits.shape[0]
    X, DX = x[None].expand(S, -1, -1).contiguous(), dx[None].expand(S, -1, -1).contiguous()
    clos = BatchedClosure(X, DX, order, Q=Q, use_kron_product=True, allow_constant=True, engine=OracleEngine())
    return SeedSweepLBFGS(clos, lr, thr, st_freq).fit(inits, epochs)


def test_lbfgs_sweep_matches_sequential_runs_unconstrained(golden):
    g = golden("f4_lbfgs")
    x, dx = t(g["dosc_sindy_x"]), t(g["dosc_sindy_dx"])
    torch.manual_seed(1)
    inits = torch.cat([t(g["dosc_sindy_init_Xi"]).reshape(1, -1), torch.randn(3, 20)])
    out = _sweep(x, dx, 3, inits, 0.1, 50, 0.05, 60)
    assert np.array_equal(out["mask"][0].numpy(), g["dosc_sindy_mask_final"])      # seed 0 = the reference's recorded run
    for s in range(4):
        reg = O.OracleRegressor(2, 3, threshold=0.05, Xi0=inits[s].view(2, 10))
        hist = O.lbfgs_fit(reg, x, dx, 60, 0.1, st_freq=50, threshold=0.05)
        assert torch.equal(out["mask"][s], reg.mask), s
        want = (reg.Xi * reg.mask).detach().numpy()
        assert np.allclose((out["Xi"][s] * out["mask"][s]).numpy(), want, rtol=2e-3, atol=2e-4), s
        assert bool(out["finished"][s]) == any(e[1] == "final" for e in hist["events"])


def test_lbfgs_sweep_matches_sequential_runs_constrained(golden):
    g = golden("f4_lbfgs")
    x, dx = t(g["dosc_esindy_x"]), t(g["dosc_esindy_dx"])
    Q = t(g["dosc_esindy_Q"])
    torch.manual_seed(2)
    first = torch.cat([t(g["dosc_esindy_init_beta"]), t(g["dosc_esindy_init_const"]).reshape(-1)])[None]
    inits = torch.cat([first, torch.randn(2, Q.shape[1] + 2)])
    out = _sweep(x, dx, 2, inits, 1.0, 100, 0.01, 40, Q=Q)
    assert np.array_equal(out["mask"][0].numpy(), g["dosc_esindy_mask_final"])
    so2 = torch.tensor([[0.0, 1.0], [-1.0, 0.0]])
    for s in range(3):
        reg = O.OracleRegressor(2, 2, L_list=[so2], threshold=0.01, beta0=inits[s, :Q.shape[1]], const0=inits[s, Q.shape[1]:].view(2, 1))
        reg.Q = Q
        O.lbfgs_fit(reg, x, dx, 40, 1.0, st_freq=100, threshold=0.01)
        assert torch.equal(out["mask"][s], reg.mask), s
        want = (reg.get_Xi() * reg.mask).detach().numpy()
        assert np.allclose((out["Xi"][s] * out["mask"][s]).numpy(), want, rtol=2e-3, atol=2e-4), s


def test_lbfgs_sweep_with_constrained_constant_matches_sequential_runs():
    """growth/noise05_esindy flags: scaling2 constraint with --constrain_constant, i.e. ``const`` stays a parameter that the
    model never reads (sindy.py:60, 173-175): its gradient is zero and it must not break the flat parameter layout."""
    xs, dxs = O.rk4_trajectories(O.rhs_growth, O.ics_growth(6, np.random.RandomState(4)), 0.02, 150)
    x, dx = torch.from_numpy(xs.reshape(-1, 2)).float(), torch.from_numpy(dxs.reshape(-1, 2)).float()
    scaling2 = torch.tensor([[2.0, 0.0], [0.0, 1.0]])
    Q, use_kron = constraint_Q([scaling2], 2, 2)
    torch.manual_seed(5)
    inits = torch.randn(3, Q.shape[1] + 2)
    S = inits.shape[0]
    clos = BatchedClosure(x.expand(S, -1, -1).contiguous(), dx.expand(S, -1, -1).contiguous(), 2, Q=Q, use_kron_product=use_kron,
                          allow_constant=False, engine=OracleEngine())
    out = SeedSweepLBFGS(clos, 1.0, 0.05, 100).fit(inits, 40)
    truth = O.SINDY_TRUTH["growth"] != 0
    for s in range(S):
        reg = O.OracleRegressor(2, 2, L_list=[scaling2], threshold=0.05, constrain_constant=True, beta0=inits[s, :Q.shape[1]],
                                const0=inits[s, Q.shape[1]:].view(2, 1))
        O.lbfgs_fit(reg, x, dx, 40, 1.0, st_freq=100, threshold=0.05)
        assert torch.equal(out["mask"][s], reg.mask), s
        assert np.allclose((out["Xi"][s] * out["mask"][s]).numpy(), (reg.get_Xi() * reg.mask).detach().numpy(), rtol=2e-3, atol=2e-4), s
        assert torch.equal(out["params"][s, Q.shape[1]:], inits[s, Q.shape[1]:])      # the unread constant never moves
    assert np.array_equal(out["mask"][0].numpy() > 0, truth)


def test_nan_gradient_at_finite_parameters_ends_the_run_with_the_nan_flag():
    """torch.optim.LBFGS.step tests ``flat_grad.abs().max() <= tolerance_grad`` at its top: a NaN gradient does not stop
    the step, the parameters go to NaN and the trainer's NaN guard (train.py:697) ends the run.  A problem whose closure
    returns NaN gradients at finite parameters must therefore come out with nan = True, finished = False -- not frozen at
    its finite start and reported converged."""
    class NaNClosure:                        # the interface SeedSweepLBFGS uses of BatchedClosure, two problems
        S, d, p, Q, distributed = 2, 1, 3, None, False

        def evaluate(self, beta, const=None, mask=None):
            g = 2.0 * beta.reshape(self.S, -1)
            loss = (beta.reshape(self.S, -1) ** 2).sum(1)
            g = g.clone()
            g[1] = float("nan")              # problem 1: NaN gradient (and a finite loss) at finite parameters
            return loss, g.reshape(self.S, self.d, self.p), None

        def xi_from(self, beta, const=None):
            return beta

    P0 = torch.tensor([[1.0, -2.0, 0.5], [0.3, 0.2, -0.1]])
    out = SeedSweepLBFGS(NaNClosure(), 0.1, 0.05, 50).fit(P0, 10)
    assert out["nan"].tolist() == [False, True] and out["finished"].tolist()[1] is False
    assert int(out["epochs"][1]) == 1                     # the guard fires after the first epoch
    assert torch.isnan(out["params"][1]).all() and torch.isfinite(out["params"][0]).all()
    # torch's own optimiser on the same closure: NaN parameters after one step as well
    w = P0[1].clone().requires_grad_(True)
    opt = torch.optim.LBFGS([w], lr=0.1)

    def cl():
        opt.zero_grad()
        loss = (w ** 2).sum()
        w.grad = torch.full_like(w, float("nan"))
        return loss
    opt.step(cl)
    assert torch.isnan(w).all()
