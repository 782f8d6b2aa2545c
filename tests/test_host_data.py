"""Noisy-data recipe of the reference's generators (data_utils/ode.py:30-49): GP smoothing pinned by the reference's
own ``num_diff_gp`` outputs (tests/golden/f10_gp_smoothing.npz), the finite-difference rule, noise conventions."""
import numpy as np
import pytest
import torch

import symode_amd  # noqa: F401
from symode_amd import data as synth
from symode_amd import dataset as D


@pytest.mark.parametrize("tag", ["dosc", "lv"])
def test_gp_smooth_matches_reference_num_diff_gp(golden, tag):
    g = golden("f10_gp_smoothing")
    x = torch.from_numpy(g[f"{tag}_x"])
    dX, X = synth.gp_smooth(x, float(g[f"{tag}_dt"]), float(g[f"{tag}_noise"]), g[f"{tag}_std"], float(g[f"{tag}_sigma_in"]))
    assert np.allclose(X.numpy(), g[f"{tag}_X"], rtol=1e-9, atol=1e-11)
    assert np.allclose(dX.numpy(), g[f"{tag}_dX"], rtol=1e-7, atol=1e-8)          # forward difference over 1e-3: 1e3 x the error of X


def test_gen_data_recipe():
    clean_x, clean_dx = synth.gen_data("dosc", 4, dt=0.01, num_steps=300, subsample_rate=3, noise=0.0, seed=5, fused=False)
    assert clean_x.shape == (4, 100, 2) and clean_x.dtype == torch.float32
    assert torch.allclose(clean_dx[..., 0], -0.1 * clean_x[..., 0] - clean_x[..., 1], atol=1e-6)      # exact RHS when noise-free
    x, dx = synth.gen_data("dosc", 4, dt=0.01, num_steps=300, subsample_rate=1, noise=0.2, seed=5, fused=False)
    assert torch.allclose(dx[:, :-1], (x[:, 1:] - x[:, :-1]) / 0.01, rtol=2e-3, atol=2e-3)            # forward differences of the noisy series
    assert torch.allclose(dx[:, -1, 0], (-0.1 * (cx := synth.gen_data("dosc", 4, dt=0.01, num_steps=300, seed=5, fused=False)[0])[:, -1, 0]
                                          - cx[:, -1, 1]), atol=1e-5)                                  # last sample: RHS of the clean state
    full = synth.gen_data("dosc", 4, dt=0.01, num_steps=300, seed=5, fused=False)[0].double()
    resid = (x.double() - full)
    want = 0.2 * full.std(dim=(0, 1), unbiased=False)
    assert torch.allclose(resid.std(dim=(0, 1)), want, rtol=0.1)                                      # additive, relative to each dimension's spread
    gx, _ = synth.gen_data("growth", 6, dt=0.002, num_steps=200, noise=0.1, multiplicative_noise=True, seed=2, fused=False)
    g0, _ = synth.gen_data("growth", 6, dt=0.002, num_steps=200, seed=2, fused=False)
    assert torch.allclose((gx / g0 - 1).std(), torch.tensor(0.1), rtol=0.1)                           # multiplicative for growth
    sx, sdx = synth.gen_data("dosc", 4, dt=0.01, num_steps=300, noise=0.2, smoothing="gp", gp_sigma_in=0.1, seed=5, fused=False)
    assert (sx.double() - full).abs().mean() < 0.5 * resid.abs().mean()                               # smoothing removes most of the noise
    assert (sdx - clean_dx_full(full)).abs().mean() < 0.1 * (dx - clean_dx_full(full)).abs().mean()    # and rescues the derivative
    with pytest.raises(NotImplementedError):
        synth.gen_data("dosc", 2, num_steps=10, noise=0.1, smoothing="spline", fused=False)


def clean_dx_full(x):
    return torch.stack([-0.1 * x[..., 0] - x[..., 1], x[..., 0] - 0.1 * x[..., 1]], -1).float()


def test_ode_dataset_fallback_honours_smoothing(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    monkeypatch.setitem(D._RECIPES, "dosc", (3, 2, 400, 4, 0.005))
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    ds = D.ODEDataset(path=str(tmp_path), ode_name="dosc", mode="train", noise=0.2, smoothing="gp")
    raw = D.ODEDataset(path=str(tmp_path), ode_name="dosc", mode="train", noise=0.2, smoothing=None)
    assert (tmp_path / "dosc-train-noise20-gp-x.pt").exists() and (tmp_path / "dosc-train-noise20-x.pt").exists()
    assert ds.x.shape == (300, 2) and raw.x.shape == (300, 2)
    truth = clean_dx_full(ds.x)
    assert (ds.dx - truth).abs().mean() < 0.25 * (raw.dx - clean_dx_full(raw.x)).abs().mean()
    again = D.ODEDataset(path=str(tmp_path), ode_name="dosc", mode="train", noise=0.2, smoothing="gp")    # now from the files
    assert torch.equal(again.x, ds.x)


def test_device_batches_serve_dataloader_semantics(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    monkeypatch.setitem(D.RD_SYNTH, "n", 6)
    monkeypatch.setitem(D.RD_SYNTH, "n_samples", 40)
    np.random.seed(0)
    mt = D.MultiTimestepReactionDiffusionDataset(mode="train")
    arrays, n, window = mt.device_arrays()
    fast = D.DeviceBatches(arrays, n, batch_size=7, shuffle=False, device="cpu", window=window)
    slow = torch.utils.data.DataLoader(mt, batch_size=7, shuffle=False)
    assert len(fast) == len(slow) == 5
    for (a, b), (c, d) in zip(fast, slow):
        assert torch.equal(a, c) and torch.equal(b, d)
    torch.manual_seed(0)
    shuffled = D.DeviceBatches(arrays, n, batch_size=7, shuffle=True, device="cpu", window=window)
    seen = torch.cat([a[:, 0, :1] for a, _ in shuffled]).flatten()
    assert sorted(seen.tolist()) == sorted(mt.x[:n, 0].tolist()) and not torch.equal(seen, mt.x[:n, 0])   # a permutation of every item
    monkeypatch.setitem(D._RECIPES, "dosc", (3, 2, 200, 2, 0.005))
    ds = D.ODEDataset(path=str(tmp_path), ode_name="dosc", mode="train", noise=0.0)
    assert isinstance(D.make_loader(ds, 64, True, "cpu"), torch.utils.data.DataLoader)                        # host runs keep DataLoader
    fb = D.DeviceBatches(*ds.device_arrays()[:1], ds.device_arrays()[1], 128, False, "cpu")
    x0, dx0 = next(iter(fb))
    assert torch.equal(x0, ds.x[:128]) and torch.equal(dx0, ds.dx[:128]) and len(fb) == 3
