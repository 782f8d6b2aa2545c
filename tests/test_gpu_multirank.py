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

from oracle import sindy_oracle as O

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


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE config[2]'s closure on point shards (model_utils.py:56-62, 118-121): the relative
# regularisers are ratios of batch means, so per generator the numerator, the denominator and their d/dXi are summed
# locally and cross the ranks in ONE packed all-reduce; ratio and quotient-rule gradient are formed after it.
# ---------------------------------------------------------------------------------------------------------------------
def _config2_closure(MU, reg, kind, ae, gen, x, dx, group=None):
    fn = MU.symmreg_i if kind == "i" else MU.symmreg_f
    flow = MU._EulerFlow(reg, 0.1, 0.01)                                   # K = 10 Euler steps (lv/noise99_eq_isymreg.cfg)
    reg.Xi.grad = None
    x_fx = torch.stack([x, flow(x)], dim=1)
    mse = reg.mse_loss(x, dx)
    if group is None:
        sym = fn(x_fx, ae, gen, f=flow, x_const=x, require_grad=True)
    else:
        n_loc = float(x.numel())
        sym, red = fn(x_fx, ae, gen, f=flow, x_const=x, require_grad=True, group=group,
                      also=[mse * n_loc, torch.tensor(n_loc, device=x.device)])
        mse = red[0] / red[1]
    (mse + 0.1 * sym).backward()
    return np.array([mse.item(), sym.item()]), reg.Xi.grad.detach().cpu().numpy().copy()


def _config2_rank(rank, world, port, blob_path, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    import symode_amd
    from symode_amd import model_utils as MU
    from symode_amd.autoencoder import AutoEncoder
    from symode_amd.lie import LieGenerator
    from tests.helpers import CONFIG2_AE, CONFIG2_GEN
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = "cuda:0"
    blob = torch.load(blob_path, weights_only=True)
    ae = AutoEncoder(**CONFIG2_AE).to(dev)
    gen = LieGenerator(device=dev, **CONFIG2_GEN).to(dev)
    ae.load_state_dict(blob["ae"])
    gen.load_state_dict(blob["gen"])
    ae.eval()
    gen.eval()
    for p in list(ae.parameters()) + list(gen.parameters()):
        p.requires_grad = False
    x, dx = blob["x"].to(dev), blob["dx"].to(dev)
    n = x.shape[0]
    cut = n // 2 + 12                                                    # uneven shards of whole 16-byte chunks
    lo, hi = (0, cut) if rank == 0 else (cut, n)
    xs, dxs = x[lo:hi].contiguous(), dx[lo:hi].contiguous()
    reg = symode_amd.SINDyRegression(2, 2, False, True, threshold=0.15, device=dev)
    reg.Xi.data = blob["Xi0"].to(dev)
    res = {}
    for kind in ("i", "f"):
        res[f"{kind}_vals"], res[f"{kind}_grad"] = _config2_closure(MU, reg, kind, ae, gen, xs, dxs, group=dist.group.WORLD)
    np.savez(os.path.join(out_dir, f"config2_rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


def test_config2_relative_regulariser_closure_two_ranks_equals_one_rank(tmp_path):
    """20 000 points, K = 10, 512 x 5 autoencoder: value and gradient of MSE + 0.1 * symmreg_{i,f} from two point shards
    (HIP engine, both ranks on cuda:0, gloo) against the one-process closure -- 1e-5."""
    import symode_amd
    from symode_amd import model_utils as MU
    from tests.helpers import make_config2
    dev = "cuda:0"
    x, dx, ae, gen = make_config2(symode_amd, dev)
    torch.manual_seed(3)
    Xi0 = torch.randn(2, 8) * 0.3
    reg = symode_amd.SINDyRegression(2, 2, False, True, threshold=0.15, device=dev)
    reg.Xi.data = Xi0.to(dev)
    one = {kind: _config2_closure(MU, reg, kind, ae, gen, x, dx) for kind in ("i", "f")}
    blob = tmp_path / "config2.pt"
    torch.save({"x": x.cpu(), "dx": dx.cpu(), "Xi0": Xi0, "ae": {k: v.cpu() for k, v in ae.state_dict().items()},
                "gen": {k: v.cpu() for k, v in gen.state_dict().items()}}, blob)
    mp.spawn(_config2_rank, args=(2, _free_port(), str(blob), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = [np.load(tmp_path / f"config2_rank{r}.npz") for r in range(2)]
    for kind in ("i", "f"):
        vals, grad = one[kind]
        assert np.array_equal(r0[f"{kind}_vals"], r1[f"{kind}_vals"]) and np.array_equal(r0[f"{kind}_grad"], r1[f"{kind}_grad"])
        e_val = np.abs(r0[f"{kind}_vals"] - vals) / np.abs(vals)
        e_grad = np.abs(r0[f"{kind}_grad"] - grad).max() / np.abs(grad).max()
        print(f"config2 sym_reg_type {kind}, 2 ranks vs 1: mse {e_val[0]:.2e}, regulariser {e_val[1]:.2e}, gradient {e_grad:.2e}")
        assert e_val.max() <= 1e-5 and e_grad <= 1e-5, (kind, e_val, e_grad)


# ---------------------------------------------------------------------------------------------------------------------
# train_SIGED_lbfgs(group=...) on the DEFAULT path (device-resident trainer): the ranks' partial [loss | grad] are summed
# between the closure launch and the optimiser launch of every inner iteration (device_lbfgs.DeviceTrainer._epoch_sharded).
# ---------------------------------------------------------------------------------------------------------------------
def _trainer_rank(rank, world, port, out_dir, kind):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.chdir(out_dir)
    import torch.distributed as dist
    import symode_amd
    from oracle import sindy_oracle as O
    from tests.helpers import load_fixture_autoencoder, load_fixture_generator
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = "cuda:0"
    xs, dxs = O.rk4_trajectories(O.rhs_dosc, O.ics_dosc(12, np.random.RandomState(5)), 0.02, 500)
    x = torch.from_numpy(xs.reshape(-1, 2)).float().to(dev)
    dx = torch.from_numpy(dxs.reshape(-1, 2)).float().to(dev)
    n = x.shape[0]
    cut = n // 2 + 10                                                   # uneven shards of whole chunks
    lo, hi = (0, n) if world == 1 else ((0, cut) if rank == 0 else (cut, n))
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "f6_symreg.npz"))
    ae = load_fixture_autoencoder(g, "tanh_learn", "Tanh", device=dev)
    gen = load_fixture_generator(g, "tanh_learn", "(2,1,2)", device=dev)
    for p in list(ae.parameters()) + list(gen.parameters()):
        p.requires_grad = False
    r = symode_amd.SINDyRegression(2, 3, False, False, threshold=0.05, device=dev)
    r.Xi.data = (torch.randn(2, 10, generator=torch.Generator().manual_seed(1)) * 0.1).to(dev)
    symode_amd.train.train_SIGED_lbfgs(
        train_loader=[(x[lo:hi].contiguous(), dx[lo:hi].contiguous())], test_loader=[], num_epochs=60, device=dev, log_interval=10 ** 9,
        save_interval=10 ** 9, save_dir=f"t{world}", autoencoder=ae, generator=gen, regressor=r, regressor_dst=None, use_latent=False,
        distill_latent=False, lr_sindy=0.1, w_sindy_z=0.0, w_sindy_x=1.0, sindy_reg_type="l1", w_sindy_reg=0.0, sym_reg_type="r",
        w_sym_reg=0.05 if kind == "r" else 0.0, st_freq=50, threshold=0.05, int_t=0.1, int_dt=0.01, print_eq=False,
        group=dist.group.WORLD if world > 1 else None)
    np.savez(os.path.join(out_dir, f"trainer_{kind}_{world}_{rank}.npz"), Xi=r.get_Xi().detach().cpu().numpy(), mask=r.mask.cpu().numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["plain", "r"])
def test_device_trainer_on_two_point_shards_equals_one_rank(tmp_path, kind):
    """dosc, order 3, 6 000 points, with and without the reversed regulariser (tiny fixture autoencoder): two ranks through the
    HIP engine (uneven shards, gloo) recover the mask of the one-rank fit; coefficients within the optimiser's stopping ball."""
    mp.spawn(_trainer_rank, args=(2, _free_port(), str(tmp_path), kind), nprocs=2, join=True)
    mp.spawn(_trainer_rank, args=(1, _free_port(), str(tmp_path), kind), nprocs=1, join=True)
    a, b = [np.load(tmp_path / f"trainer_{kind}_2_{r}.npz") for r in range(2)]
    one = np.load(tmp_path / f"trainer_{kind}_1_0.npz")
    assert np.array_equal(a["Xi"], b["Xi"]) and np.array_equal(a["mask"], b["mask"])          # replicated decisions
    assert np.array_equal(a["mask"], one["mask"])
    want = np.zeros((2, 10))
    want[:, :6] = O.SINDY_TRUTH["dosc"] != 0
    if kind == "plain":
        assert np.array_equal(one["mask"] > 0, want > 0)                                       # the damped oscillator is recovered
    assert np.abs(a["Xi"] - one["Xi"]).max() <= 1e-3 * np.abs(one["Xi"]).max()


# ---------------------------------------------------------------------------------------------------------------------
# the single-seed driver under several ranks: python -m torch.distributed.run --nproc-per-node N -m symode_amd.main ...
# ---------------------------------------------------------------------------------------------------------------------
def _main_rank(rank, world, port, cwd, argv):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.chdir(cwd)
    import torch.distributed as dist
    import symode_amd  # noqa: F401
    from symode_amd import main as M
    M.main(list(argv), backend="gloo", one_gpu=True)
    dist.barrier()
    dist.destroy_process_group()


def test_main_driver_two_ranks_equals_one_rank(tmp_path):
    """`symode_amd.main` on dosc at 20 % noise (order 2, the shipped config's hyper-parameters) with the L-BFGS batch sharded over
    two ranks: with the data files on disk both runs draw the same batch, rank 0 writes the reference's files, and they hold
    the one-process run's mask and -- to summation order through the optimiser -- its coefficients."""
    argv = COMMON + ["--task", "dosc", "--lbfgs_subsample", "0.5", "--lr_sindy", "0.1", "--poly_order", "2", "--st_freq", "50",
                     "--threshold", "5e-2", "--num_epochs", "100", "--ae_arch", "none"]
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        from symode_amd import main as M
        M.main(argv + ["--save_dir", "m0"])                  # makes ./data/*.pt (drawing random numbers on the way)
        M.main(argv + ["--save_dir", "m1"])                  # data on disk: the batch every later run draws, one rank or two
    finally:
        os.chdir(cwd)
    mp.spawn(_main_rank, args=(2, _free_port(), str(tmp_path), argv + ["--save_dir", "m2"]), nprocs=2, join=True)
    one = np.load(tmp_path / "eval_results" / "m1" / "seed0.npz")
    two = np.load(tmp_path / "eval_results" / "m2" / "seed0.npz")
    assert np.array_equal(one["coefficients"] != 0, two["coefficients"] != 0)
    assert np.allclose(one["coefficients"], two["coefficients"], rtol=1e-3, atol=1e-4)
    assert bool(one["correct_form_all"]) == bool(two["correct_form_all"])
    assert sorted(os.listdir(tmp_path / "saved_models" / "m2")) == sorted(os.listdir(tmp_path / "saved_models" / "m1"))
