"""N > 1 path on CPU: two gloo ranks, each holding half of every problem's points; the packed
[loss | grad] (and Gram) all-reduce must reproduce the single-process full-batch result."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import sindy_oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_chunks, constrained, out_dir, with_sym=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import symode_amd  # noqa: F401
    from symode_amd.batched import BatchedClosure
    from symode_amd.constraint import constraint_Q
    from tests.oracle_engine import OracleEngine
    g = torch.Generator().manual_seed(0)
    S, n, d, order = 5, 600, 2, 3
    x, dx = torch.randn(S, n, d, generator=g) * 0.7, torch.randn(S, n, d, generator=g)
    lo, hi = rank * n // world, (rank + 1) * n // world
    Q = None
    if constrained:
        Q, uk = constraint_Q([torch.tensor([[0.0, 1.0], [-1.0, 0.0]])], d, order)
    sym = None
    if with_sym:                                  # reversed symmetry regulariser on precomputed (g(x), J_g(x)), sharded like x
        R = torch.matrix_exp(0.05 * torch.tensor([[0.0, 1.0], [-1.0, 0.0]]))
        gx = (x @ R.T + 0.01 * torch.randn(S, n, d, generator=g)).unsqueeze(1)
        jgx = (R + 0.01 * torch.randn(S, 1, n, d, d, generator=g))
        sym = (gx[:, :, lo:hi].contiguous(), jgx[:, :, lo:hi].contiguous(), 0.1)
    clos = BatchedClosure(x[:, lo:hi].contiguous(), dx[:, lo:hi].contiguous(), order, Q=Q, use_kron_product=True,
                          allow_constant=True, group=dist.group.WORLD, n_chunks=n_chunks, engine=OracleEngine(), reversed_sym=sym)
    if constrained:
        beta, const = torch.randn(S, Q.shape[1], generator=g), torch.randn(S, d, 1, generator=g)
        loss, gb, gc = clos.evaluate(beta, const)
        res = dict(loss=loss.numpy(), gb=gb.numpy(), gc=gc.numpy())
    else:
        Xi = torch.randn(S, d, 10, generator=g)
        mask = (torch.rand(S, d, 10, generator=g) > 0.3).float()
        loss, grad, _ = clos.evaluate(Xi, mask=mask)
        res = dict(loss=loss.numpy(), grad=grad.numpy(), gram=clos.aug_gram().numpy())
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


def _run(tmp_path, n_chunks, constrained, with_sym=False):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_chunks, constrained, str(tmp_path), with_sym), nprocs=2, join=True)
    return [np.load(tmp_path / f"rank{r}.npz") for r in range(2)]


def test_point_sharded_closure_with_reversed_symmetry_term(tmp_path):
    """MSE + 0.1 * reversed sym-reg per problem, points (and their g(x), J_g(x)) split over two ranks: the all-reduced
    packed buffer equals the full-batch closure of the oracle."""
    r0, r1 = _run(tmp_path, n_chunks=2, constrained=False, with_sym=True)
    g = torch.Generator().manual_seed(0)
    S, n, d, order = 5, 600, 2, 3
    x, dx = torch.randn(S, n, d, generator=g) * 0.7, torch.randn(S, n, d, generator=g)
    R = torch.matrix_exp(0.05 * torch.tensor([[0.0, 1.0], [-1.0, 0.0]]))
    gx = (x @ R.T + 0.01 * torch.randn(S, n, d, generator=g)).unsqueeze(1)
    jgx = (R + 0.01 * torch.randn(S, 1, n, d, d, generator=g))
    Xi = torch.randn(S, d, 10, generator=g)
    mask = (torch.rand(S, d, 10, generator=g) > 0.3).float()
    assert np.array_equal(r0["loss"], r1["loss"]) and np.array_equal(r0["grad"], r1["grad"])
    for s in range(S):
        reg = O.OracleRegressor(d, order, Xi0=Xi[s])
        reg.mask = mask[s]
        loss = torch.nn.functional.mse_loss(reg(x[s]), dx[s]) + 0.1 * O.symreg_reversed_precomputed(x[s], list(gx[s]), list(jgx[s]), reg)
        loss.backward()
        assert np.isclose(r0["loss"][s], loss.item(), rtol=1e-5)
        assert np.allclose(r0["grad"][s], (reg.Xi.grad * mask[s]).numpy(), rtol=1e-4, atol=1e-6)


def test_point_sharded_closure_matches_full_batch(tmp_path):
    r0, r1 = _run(tmp_path, n_chunks=2, constrained=False)
    g = torch.Generator().manual_seed(0)
    S, n, d, order = 5, 600, 2, 3
    x, dx = torch.randn(S, n, d, generator=g) * 0.7, torch.randn(S, n, d, generator=g)
    Xi = torch.randn(S, d, 10, generator=g)
    mask = (torch.rand(S, d, 10, generator=g) > 0.3).float()
    for k in ("loss", "grad", "gram"):
        assert np.array_equal(r0[k], r1[k])                          # every rank ends with the same sums
    for s in range(S):
        wl, wg = O.mse_loss_and_grad(x[s], dx[s], Xi[s], mask[s], order)
        assert np.isclose(r0["loss"][s], wl.item(), rtol=1e-5)
        assert np.allclose(r0["grad"][s], wg.numpy(), rtol=1e-4, atol=1e-6)
        A = torch.cat([O.theta(x[s], order), dx[s]], 1).double()
        assert np.allclose(r0["gram"][s], (A.T @ A).numpy(), rtol=1e-12)


def test_sharded_constrained_closure_gradients(tmp_path):
    r0, r1 = _run(tmp_path, n_chunks=1, constrained=True)
    assert np.array_equal(r0["gb"], r1["gb"])
    from symode_amd.constraint import constraint_Q
    g = torch.Generator().manual_seed(0)
    S, n, d, order = 5, 600, 2, 3
    x, dx = torch.randn(S, n, d, generator=g) * 0.7, torch.randn(S, n, d, generator=g)
    Q, _ = constraint_Q([torch.tensor([[0.0, 1.0], [-1.0, 0.0]])], d, order)
    beta, const = torch.randn(S, Q.shape[1], generator=g), torch.randn(S, d, 1, generator=g)
    for s in range(S):
        reg = O.OracleRegressor(d, order, L_list=[torch.tensor([[0.0, 1.0], [-1.0, 0.0]])], beta0=beta[s], const0=const[s])
        reg.Q = Q
        loss = torch.nn.functional.mse_loss(reg(x[s]), dx[s])
        loss.backward()
        assert np.isclose(r0["loss"][s], loss.item(), rtol=1e-5)
        assert np.allclose(r0["gb"][s], reg.beta.grad.numpy(), rtol=1e-4, atol=1e-6)
        assert np.allclose(r0["gc"][s], reg.const.grad.numpy(), rtol=1e-4, atol=1e-6)


def _sweep_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import symode_amd  # noqa: F401
    from symode_amd.batched import BatchedClosure
    from symode_amd.sweep import SeedSweepLBFGS, SeedSweepSTLSQ
    from tests.oracle_engine import OracleEngine
    x, dx = O.rk4_trajectories(O.rhs_dosc, O.ics_dosc(8, np.random.RandomState(3)), 0.02, 300)
    x = torch.from_numpy(x.reshape(-1, 2)).float()
    dx = torch.from_numpy(dx.reshape(-1, 2)).float()
    n = x.shape[0]
    lo, hi = rank * n // world, (rank + 1) * n // world
    # (1) STLSQ sweep: per-rank Gram of the rank's subsample, all-reduced, solved on every rank
    sw = SeedSweepSTLSQ(x[lo:hi], dx[lo:hi], 3, n_seeds=4, subsample=0.5, seed0=0, group=dist.group.WORLD,
                        engine=OracleEngine())
    Xi, mask, passes = sw.solve(0.0, 5e-2)
    # (2) L-BFGS sweep over point shards: every inner iteration all-reduces the packed [loss | grad]
    S = 3
    xs, dxs = x[lo:hi].expand(S, -1, -1).contiguous(), dx[lo:hi].expand(S, -1, -1).contiguous()
    clos = BatchedClosure(xs, dxs, 3, group=dist.group.WORLD, engine=OracleEngine())
    g = torch.Generator().manual_seed(7)
    P0 = torch.randn(S, 20, generator=g) * 0.1
    out = SeedSweepLBFGS(clos, 0.1, 5e-2, 50).fit(P0, 60)
    np.savez(os.path.join(out_dir, f"sweep{rank}.npz"), Xi=Xi.numpy(), mask=mask.numpy(), n_points=sw.n_points,
             lb_mask=out["mask"].numpy(), lb_Xi=out["Xi"].numpy(), lb_done=out["finished"].numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_point_sharded_seed_sweeps_agree_on_every_rank_and_recover_dosc(tmp_path):
    port = _free_port()
    mp.spawn(_sweep_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = [np.load(tmp_path / f"sweep{r}.npz") for r in range(2)]
    for k in ("Xi", "mask", "lb_mask", "lb_Xi"):
        assert np.array_equal(r0[k], r1[k]), k                        # replicated solve: identical on every rank
    want = np.zeros((2, 10), dtype=bool)
    want[:, :6] = O.SINDY_TRUTH["dosc"] != 0                           # dx0 = -0.1 x0 - x1 ; dx1 = x0 - 0.1 x1
    assert int(r0["n_points"]) == 2 * (1200 // 2)                      # 2 ranks x 50 % of 1200 local points
    for s in range(4):
        assert np.array_equal(r0["mask"][s] > 0, want)
        assert np.allclose(r0["Xi"][s][want], [-0.1, -1.0, 1.0, -0.1], atol=2e-3)
    assert r0["lb_done"].all()
    for s in range(3):
        assert np.array_equal(r0["lb_mask"][s] > 0, want)
        assert np.allclose(r0["lb_Xi"][s][want], [-0.1, -1.0, 1.0, -0.1], atol=5e-3)


def _main_sweep_worker(rank, world, port, out_dir, method):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    os.chdir(out_dir)
    torch.set_num_threads(1)
    import symode_amd  # noqa: F401
    from symode_amd import dataset as D, main_sweep
    from tests.oracle_engine import OracleEngine
    D._RECIPES["dosc"] = (6, 2, 600, 3, 0.01)                       # 6 trajectories x 200 samples, noise-free
    argv = ["--task", "dosc", "--noise", "0.0", "--ae_arch", "none", "--sindy_optimizer", "lbfgs", "--lbfgs_subsample", "0.5",
            "--lr_sindy", "0.1", "--w_sindy_x", "1.0", "--w_sindy_z", "0.0", "--w_sindy_reg", "0.0", "--w_sym_reg", "0.0",
            "--poly_order", "2", "--st_freq", "50", "--threshold", "5e-2", "--num_epochs", "60", "--save_dir", f"sweep-{method}",
            "--n_seeds", "4", "--method", method, "--seed", "0"]
    res = main_sweep.main(argv, engine=OracleEngine(), backend="gloo")
    assert (res is None) == (rank != 0)                             # only rank 0 writes and aggregates
    dist.barrier()
    dist.destroy_process_group()


def test_main_sweep_under_two_ranks_lbfgs_and_stlsq(tmp_path):
    """The multi-GPU drivers of BASELINE config 3 (and of the L-BFGS sweeps), rehearsed on two gloo ranks: every rank holds
    half of the trajectories, the Gram / [loss | grad] all-reduce makes all ranks take the same decisions, rank 0 writes
    the reference's eval_results files -- every seed recovers the damped oscillator."""
    for method in ("stlsq", "lbfgs"):
        port = _free_port()
        mp.spawn(_main_sweep_worker, args=(2, port, str(tmp_path), method), nprocs=2, join=True)
        files = sorted((tmp_path / "eval_results" / f"sweep-{method}").iterdir())
        assert [f.name for f in files] == [f"seed{s}.npz" for s in range(4)]
        for f in files:
            r = np.load(f)
            assert bool(r["correct_form_all"]) and float(r["mse_all"]) < 1e-4
            assert np.allclose(r["coefficients"][:, 1:3], [[-0.1, -1.0], [1.0, -0.1]], atol=5e-3)


# ---------------------------------------------------------------------------------------------------------------------
# relative regularisers S2 / S3 on point shards (model_utils.py:56-62, 118-121): the loss is a ratio
# of two batch means per generator, so numerator, denominator and their d/dXi cross the ranks separately -- one packed
# all-reduce -- and the ratio and its quotient-rule gradient are formed after the collective.
# ---------------------------------------------------------------------------------------------------------------------
def _symreg_setup(tag="tanh_learn", act="Tanh", rep="(2,1,2)", n=512):
    from symode_amd import model_utils as MU
    from symode_amd.sindy import SINDyRegression
    from tests.helpers import load_fixture_autoencoder, load_fixture_generator, t
    from tests.oracle_engine import OracleEngine
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "f6_symreg.npz"))
    ae = load_fixture_autoencoder(g, tag, act)
    gen = load_fixture_generator(g, tag, rep)
    for p in list(ae.parameters()) + list(gen.parameters()):
        p.requires_grad = False
    d, order, sine, exp = [int(v) for v in g[f"{tag}_cfg"]]
    r = SINDyRegression(d, order, bool(sine), bool(exp), threshold=0.05, device="cpu", engine=OracleEngine())
    r.Xi.data = t(g[f"{tag}_Xi"]).clone()
    x = t(g[f"{tag}_x"])[:n].contiguous()
    dx = (r(x).detach() + 0.05 * torch.randn(x.shape, generator=torch.Generator().manual_seed(5))).contiguous()
    return MU, ae, gen, r, x, dx


def _closure_terms(MU, kind, ae, gen, r, x, dx, group=None, relative=True):
    """(mse, regulariser, d(mse + 0.1 reg)/dXi) of train.py:663-679 on the points given (this rank's shard when group is set)."""
    fn = MU.symmreg_i if kind == "i" else MU.symmreg_f
    flow = MU._EulerFlow(r, 0.05, 0.01)
    r.Xi.grad = None
    x_fx = torch.stack([x, flow(x)], dim=1)
    mse = r.mse_loss(x, dx)
    if group is None:
        sym = fn(x_fx, ae, gen, f=flow, x_const=x, require_grad=True, relative=relative)
    else:
        n_loc = float(x.numel())
        sym, red = fn(x_fx, ae, gen, f=flow, x_const=x, require_grad=True, relative=relative, group=group,
                      also=[mse * n_loc, torch.tensor(n_loc)])
        mse = red[0] / red[1]
    (mse + 0.1 * sym).backward()
    return mse.item(), sym.item(), r.Xi.grad.clone().numpy()


def _symreg_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import symode_amd  # noqa: F401
    MU, ae, gen, r, x, dx = _symreg_setup()
    n = x.shape[0]
    lo, hi = rank * n // world, (rank + 1) * n // world + (7 if rank == 0 else 0)       # UNEVEN shards: 263 / 249 points
    lo = lo + (7 if rank == 1 else 0)
    xs, dxs = x[lo:hi].contiguous(), dx[lo:hi].contiguous()
    res = {}
    for kind in ("i", "f"):
        for rel in (True, False):
            mse, sym, grad = _closure_terms(MU, kind, ae, gen, r, xs, dxs, group=dist.group.WORLD, relative=rel)
            res[f"{kind}{int(rel)}_mse"], res[f"{kind}{int(rel)}_sym"], res[f"{kind}{int(rel)}_grad"] = mse, sym, grad
    np.savez(os.path.join(out_dir, f"symreg{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


def test_relative_regularisers_on_point_shards_equal_the_full_batch(tmp_path):
    mp.spawn(_symreg_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = [np.load(tmp_path / f"symreg{r}.npz") for r in range(2)]
    import symode_amd  # noqa: F401
    MU, ae, gen, r, x, dx = _symreg_setup()
    for kind in ("i", "f"):
        for rel in (True, False):
            k = f"{kind}{int(rel)}"
            for q in ("mse", "sym", "grad"):
                assert np.array_equal(r0[f"{k}_{q}"], r1[f"{k}_{q}"]), (k, q)        # every rank holds the same numbers
            mse, sym, grad = _closure_terms(MU, kind, ae, gen, r, x, dx, relative=rel)
            assert np.isclose(r0[f"{k}_mse"], mse, rtol=1e-5), k
            assert np.isclose(r0[f"{k}_sym"], sym, rtol=1e-5), k
            assert np.abs(r0[f"{k}_grad"] - grad).max() <= 1e-5 * np.abs(grad).max(), k


def _sharded_trainer_worker(rank, world, port, out_dir, kind):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.chdir(out_dir)
    torch.set_num_threads(1)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import symode_amd  # noqa: F401
    from symode_amd.train import train_SIGED_lbfgs
    MU, ae, gen, r, x, dx = _symreg_setup(n=256)
    n = x.shape[0]
    lo, hi = rank * n // world, (rank + 1) * n // world
    train_SIGED_lbfgs(train_loader=[(x[lo:hi].contiguous(), dx[lo:hi].contiguous())], test_loader=[0], num_epochs=3, device="cpu",
                      log_interval=1, save_interval=10 ** 9, save_dir=f"t{world}", autoencoder=ae, generator=gen, regressor=r,
                      regressor_dst=None, use_latent=False, distill_latent=False, lr_sindy=0.05, w_sindy_z=0.0, w_sindy_x=1.0,
                      sindy_reg_type="l1", w_sindy_reg=1e-3, sym_reg_type=kind, w_sym_reg=0.1, st_freq=2, threshold=0.02,
                      int_t=0.05, int_dt=0.01, print_eq=False, group=dist.group.WORLD if world > 1 else None)
    np.savez(os.path.join(out_dir, f"trainer_{kind}_{world}_{rank}.npz"), Xi=r.get_Xi().detach().numpy(), mask=r.mask.numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_lbfgs_trainer_on_point_shards_with_relative_regulariser(tmp_path):
    """train_SIGED_lbfgs(group=...) with sym_reg_type 'i' / 'f' / 'r': two ranks, half of the batch each, against the one-process
    fit of the whole batch -- same masks after a thresholding event, coefficients to 1e-3 of their scale (60 iterations of
    un-line-searched L-BFGS with an L1 term amplify sums that differ in their last bits: measured 2e-4; the closure
    itself agrees to 1e-5, previous test)."""
    for kind in ("i", "f", "r"):               # r: a plain batch mean per group element, summed over the ranks with the residual
        mp.spawn(_sharded_trainer_worker, args=(2, _free_port(), str(tmp_path), kind), nprocs=2, join=True)
        mp.spawn(_sharded_trainer_worker, args=(1, _free_port(), str(tmp_path), kind), nprocs=1, join=True)
        a, b = [np.load(tmp_path / f"trainer_{kind}_2_{r}.npz") for r in range(2)]
        one = np.load(tmp_path / f"trainer_{kind}_1_0.npz")
        assert np.array_equal(a["Xi"], b["Xi"]) and np.array_equal(a["mask"], b["mask"])
        assert np.array_equal(a["mask"], one["mask"])
        assert np.abs(a["Xi"] - one["Xi"]).max() <= 1e-3 * np.abs(one["Xi"]).max()


def _odd_sweep_worker(rank, world, port, out_dir, method):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    os.chdir(out_dir)
    torch.set_num_threads(1)
    import symode_amd  # noqa: F401
    from symode_amd import dataset as D, main_sweep
    from tests.oracle_engine import OracleEngine
    D._RECIPES["dosc"] = (6, 2, 600, 3, 0.01)                       # 1200 samples; 41.75 % of them = 501 rows per seed
    argv = ["--task", "dosc", "--noise", "0.0", "--ae_arch", "none", "--sindy_optimizer", "lbfgs", "--lbfgs_subsample", "0.4175",
            "--lr_sindy", "0.1", "--w_sindy_x", "1.0", "--w_sindy_z", "0.0", "--w_sindy_reg", "0.0", "--w_sym_reg", "0.0",
            "--poly_order", "2", "--st_freq", "50", "--threshold", "5e-2", "--num_epochs", "60", "--save_dir", f"odd-{method}-{world}",
            "--n_seeds", "3", "--method", method, "--seed", "0"]
    main_sweep.main(argv, engine=OracleEngine(), backend="gloo")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_main_sweep_with_an_odd_subsample_fits_the_same_problems_on_one_two_and_three_ranks(tmp_path):
    """501 rows per seed do not divide by 2 or 3: the shards differ by a row, the global count is summed over the ranks, and
    every world size fits exactly the 1-rank problems -- same masks, coefficients to summation order (the old
    `m -= m % world` dropped rows instead)."""
    res = {}
    for method in ("stlsq", "lbfgs"):
        for world in (1, 2, 3):
            mp.spawn(_odd_sweep_worker, args=(world, _free_port(), str(tmp_path), method), nprocs=world, join=True)
            files = sorted((tmp_path / "eval_results" / f"odd-{method}-{world}").iterdir())
            res[method, world] = np.stack([np.load(f)["coefficients"] for f in files])
        for world in (2, 3):
            assert np.array_equal(res[method, world] != 0, res[method, 1] != 0), (method, world)
            tol = 1e-5 if method == "stlsq" else 1e-3            # L-BFGS: the optimiser's own stopping ball (see test_gpu_multirank)
            assert np.abs(res[method, world] - res[method, 1]).max() <= tol * np.abs(res[method, 1]).max(), (method, world)
