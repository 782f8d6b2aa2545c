"""N > 1 path through the HIP engine, rehearsed on the one-GPU box: two processes, both on cuda:0, collectives over gloo
(RCCL needs one GPU per rank; the 8-GPU scaling run belongs to the driver).  What is asserted is what the sharded design
promises: an N-rank run fits exactly the problems of the 1-rank run (same per-seed subsample, split over the ranks) and
differs from it by summation order only -- identical sparsity masks, coefficients to 1e-5 -- for

  * BASELINE config[3]: selkov, 64-seed sequential-threshold sweep, ONE all-reduce of the (64, 12, 12) fp64 Gram stack;
  * BASELINE config[1]: damped oscillator, poly-order 5, so2-constrained L-BFGS sweep, the packed [loss | grad] of all
    seeds all-reduced at every closure.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank(rank, world, port, cwd, argv):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.chdir(cwd)
    import torch.distributed as dist
    import symode_amd  # noqa: F401
    from symode_amd import main_sweep
    res = main_sweep.main(list(argv), backend="gloo", one_gpu=True)
    assert (res is None) == (rank != 0)
    dist.barrier()
    dist.destroy_process_group()


def _sweep(tmp_path, world, argv, save_dir):
    argv = list(argv) + ["--save_dir", save_dir]
    if world == 1:
        cwd = os.getcwd()
        os.chdir(tmp_path)
        try:
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
                os.environ.pop(k, None)
            from symode_amd import main_sweep
            main_sweep.main(argv)
        finally:
            os.chdir(cwd)
    else:
        mp.spawn(_rank, args=(world, _free_port(), str(tmp_path), argv), nprocs=world, join=True)
    files = sorted((tmp_path / "eval_results" / save_dir).iterdir(), key=lambda f: int(f.name[4:-4]))
    return np.stack([np.load(f)["coefficients"] for f in files])


COMMON = ["--noise", "0.2", "--smoothing", "gp", "--sindy_optimizer", "lbfgs", "--w_sindy_z", "0.0", "--w_sindy_x", "1.0", "--w_sindy_reg", "0.0",
          "--w_sym_reg", "0.0", "--seed", "0"]


def test_config3_selkov_stlsq_sweep_two_ranks_equals_one_rank(tmp_path):
    argv = COMMON + ["--task", "selkov", "--lbfgs_subsample", "0.5", "--poly_order", "3", "--threshold", "7.5e-2", "--num_epochs", "10",
                     "--n_seeds", "64", "--method", "stlsq"]
    one = _sweep(tmp_path, 1, argv, "sk1")
    two = _sweep(tmp_path, 2, argv, "sk2")
    assert one.shape == two.shape == (64, 2, 10)
    assert np.array_equal(one != 0, two != 0)                                   # identical masks, seed by seed
    assert np.allclose(one, two, rtol=1e-5, atol=1e-5 * np.abs(one).max())     # fp64 Gram sums differ in the last bits only


def test_config1_dosc_order5_constrained_lbfgs_sweep_two_ranks_equals_one_rank(tmp_path):
    argv = COMMON + ["--task", "dosc", "--n_comps", "1", "--repr", "(1,so2)", "--group_idx", "0", "--latent_dim", "2", "--ae_arch", "none",
                     "--eq_constraint", "--lbfgs_subsample", "0.5", "--lr_sindy", "1.0", "--poly_order", "5", "--st_freq", "100",
                     "--threshold", "1e-2", "--num_epochs", "40", "--n_seeds", "8", "--method", "lbfgs"]
    one = _sweep(tmp_path, 1, argv, "do1")
    two = _sweep(tmp_path, 2, argv, "do2")
    assert one.shape == two.shape == (8, 2, 21)
    assert np.array_equal(one != 0, two != 0)                                   # identical masks
    # un-line-searched L-BFGS amplifies the fp32 summation-order difference of the shards up to its stopping ball
    assert np.allclose(one, two, rtol=1e-3, atol=1e-4)
