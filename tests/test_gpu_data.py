"""Data preparation on the GPU: the reference's GP smoothing (rocSOLVER Cholesky through torch) and the noisy-data
recipe on top of the HIP RK4 kernel."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("tag", ["dosc", "lv"])
def test_gp_smooth_on_gpu_matches_reference_fixture(tag):
    from symode_amd import data as synth
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "f10_gp_smoothing.npz"))
    x = torch.from_numpy(g[f"{tag}_x"]).to(DEV)
    dX, X = synth.gp_smooth(x, float(g[f"{tag}_dt"]), float(g[f"{tag}_noise"]), g[f"{tag}_std"], float(g[f"{tag}_sigma_in"]))
    assert X.is_cuda and np.allclose(X.cpu().numpy(), g[f"{tag}_X"], rtol=1e-8, atol=1e-10)
    assert np.allclose(dX.cpu().numpy(), g[f"{tag}_dX"], rtol=1e-6, atol=1e-7)


def test_gen_data_gp_on_gpu_equals_cpu_recipe_on_the_same_noisy_series():
    """T = 2000 samples per trajectory: HIP RK4 orbits + GP solve on the GPU vs the same recipe evaluated on the CPU."""
    from symode_amd import data as synth
    x, dx = synth.gen_data("selkov", 6, dt=0.002, num_steps=2000, subsample_rate=1, noise=0.2, smoothing=None, seed=3, device=DEV)
    assert x.is_cuda and x.shape == (6, 2000, 2)
    xt = x.double().transpose(0, 1).contiguous()
    std = xt.std(dim=(0, 1), unbiased=False)
    d_gpu, s_gpu = synth.gp_smooth(xt, 0.002, 0.2, std, 0.1)
    d_cpu, s_cpu = synth.gp_smooth(xt.cpu(), 0.002, 0.2, std.cpu(), 0.1)
    assert torch.allclose(s_gpu.cpu(), s_cpu, rtol=1e-7, atol=1e-9)
    assert torch.allclose(d_gpu.cpu(), d_cpu, rtol=1e-4, atol=1e-5)
    truth = torch.stack([0.75 - 0.1 * s_cpu[..., 0] - s_cpu[..., 0] * s_cpu[..., 1] ** 2,
                         -s_cpu[..., 1] + 0.1 * s_cpu[..., 0] + s_cpu[..., 0] * s_cpu[..., 1] ** 2], -1)
    fd_err = (dx.double().transpose(0, 1).cpu() - truth).abs().mean()
    assert (d_cpu - truth).abs().mean() < 0.05 * fd_err            # the point of smoothing: a usable derivative
