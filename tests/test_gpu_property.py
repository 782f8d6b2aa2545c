"""Randomised (hypothesis) parity of the HIP entry points against the CPU oracle over the whole compiled
library set: random (d, order, sine, exp), sizes, masks, batching and pointer alignment."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import sindy_oracle as O

pytestmark = pytest.mark.gpu

LIBS = [(d, o) for d in (1, 2) for o in range(1, 6)] + [(3, o) for o in range(1, 5)] + [(4, o) for o in range(1, 4)]


@pytest.fixture(scope="module")
def eng():
    import symode_amd
    assert torch.cuda.is_available()
    return symode_amd.get_engine()


common = dict(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])


@settings(**common)
@given(lib=st.sampled_from(LIBS), sine=st.booleans(), exp=st.booleans(), n=st.integers(1, 3000), off=st.integers(0, 3),
       seed=st.integers(0, 2 ** 20), use_mask=st.booleans())
def test_loss_grad_forward_theta_random(eng, lib, sine, exp, n, off, seed, use_mask):
    d, order = lib
    fl = (1 if sine else 0) | (2 if exp else 0)
    g = torch.Generator().manual_seed(seed)
    p = O.term_count(d, order, sine, exp)
    xa, dxa = torch.randn(n + off, d, generator=g) * 0.6, torch.randn(n + off, d, generator=g)
    Xi = torch.randn(d, p, generator=g) * 0.4
    mask = (torch.rand(d, p, generator=g) > 0.3).float() if use_mask else torch.ones(d, p)
    x, dx = xa[off:], dxa[off:]                                    # off > 0: base pointer not 16-byte aligned
    xg, dxg = xa.cuda()[off:], dxa.cuda()[off:]
    th = eng.theta(xg, order, fl).cpu()
    want_th = O.theta(x, order, sine, exp)
    npoly = O.term_count(d, order)
    assert torch.equal(th[:, :npoly], want_th[:, :npoly])
    assert torch.allclose(th[:, npoly:], want_th[:, npoly:], rtol=3e-7, atol=1e-7)
    mg = mask.cuda() if use_mask else None
    fw = eng.forward(xg, Xi.cuda(), mg, order, fl).cpu().double()
    want_fw = O.forward(x.double(), Xi.double(), mask.double(), order, sine, exp)
    assert torch.allclose(fw, want_fw, rtol=1e-4, atol=2e-5 * max(want_fw.abs().max().item(), 1e-3))
    loss, grad = eng.loss_grad(xg, dxg, Xi.cuda(), mg, order, fl)
    wl, wg = O.mse_loss_and_grad(x.double(), dx.double(), Xi.double(), mask.double(), order, sine, exp)
    assert np.isclose(loss.item(), wl.item(), rtol=3e-5)
    assert (grad.cpu().double() - wg).abs().max().item() <= 3e-5 * max(wg.abs().max().item(), 1e-6)


@settings(**common)
@given(lib=st.sampled_from(LIBS), sine=st.booleans(), exp=st.booleans(), n=st.integers(1, 1500), S=st.integers(1, 5),
       seed=st.integers(0, 2 ** 20))
def test_gram_and_batched_loss_random(eng, lib, sine, exp, n, S, seed):
    d, order = lib
    fl = (1 if sine else 0) | (2 if exp else 0)
    g = torch.Generator().manual_seed(seed)
    p = O.term_count(d, order, sine, exp)
    x, dx = torch.randn(S, n, d, generator=g) * 0.6, torch.randn(S, n, d, generator=g)
    Xi = torch.randn(S, d, p, generator=g) * 0.4
    G = eng.aug_gram(x.cuda(), dx.cuda(), order, fl).cpu()
    loss, grad = eng.loss_grad(x.cuda(), dx.cuda(), Xi.cuda(), None, order, fl)
    for s in range(S):
        A = torch.cat([O.theta(x[s], order, sine, exp), dx[s]], 1).double()
        want = A.T @ A
        assert torch.allclose(G[s], want, rtol=1e-5, atol=1e-6 * want.abs().max().item())
        W = Xi[s].double()
        wl = (torch.trace(W @ want[:p, :p] @ W.T) - 2 * torch.trace(W @ want[:p, p:]) + torch.trace(want[p:, p:])) / (n * d)
        wg = 2.0 / (n * d) * (W @ want[:p, :p] - want[:p, p:].T)
        assert np.isclose(loss[s].item(), wl.item(), rtol=1e-4, atol=1e-6)
        assert (grad[s].cpu().double() - wg).abs().max().item() <= 1e-4 * max(wg.abs().max().item(), 1e-6)
