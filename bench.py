#!/usr/bin/env python3
"""Headline benchmark: trajectory-points/s through Theta-build + residual (+ gradient) + symmetry regulariser.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): damped oscillator, n_ics=50 x steps=2500 x dim=2 fp32, poly-order 5 (p = 21),
EquivSINDy-c (so2 equivariance constraint, Xi = reshape(Q beta) + const), batched over S independent
(trajectory, seed) problems per GPU that are resident in HBM.  One *step* = one closure evaluation of every problem:
Xi from (beta, const); the fused Theta + residual + loss + dloss/dXi kernel (Theta never materialised); the fused
Lie-symmetry regulariser on precomputed (g(x), J_g(x)) -- reversed form, model_utils.py:126-170 -- with its gradient;
projection of the summed gradient onto (beta, const); and -- for N > 1 -- the RCCL all-reduce of the packed
[loss | grad] partials (every rank holds its own shard of each problem's trajectories: weak scaling).

``--gpus N`` without a torchrun environment starts the N ranks itself (``python -m torch.distributed.run``) BEFORE
anything touches the GPU and relays rank 0's line.

Prints ONE JSON line (rank 0) following the driver's contract, plus
  roofline      : the dominant kernel of the step priced against the HBM roof (algorithmic bytes / live HIP-event time;
                  PMC traffic when profiles/pmc_traffic.json holds it),
  roofline_legs : the same for every kernel of the step and for the Euler-flow pair of the infinitesimal regulariser
                  (timed after the K steps on a slice of the resident points; not part of ``value``),
  cpu_baseline  : the CPU oracle ("port" of the reference op sequence) timed on the host cores on a bounded sample of
                  the same step (rank 0, N = 1 only); ``legs``: BASELINE.md section 3's table -- Theta only, closure, one
                  sequential-threshold pass with full / partial mask, all cores and one thread, median of 20,
  config3_gram_allreduce : BASELINE config[3]'s sharded work -- 64-seed index-table Gram on this rank's point shard + ONE
                  fp64 all-reduce of the (64, 12, 12) stack: kernel us, collective us, ranks observed (every N),
  single_problem: the bare 50x2500x2 shape, kernel time and closure wall time.
  seed_sweeps: wall time of the 64-seed L-BFGS and sequential-threshold sweeps at BASELINE config[3]'s size (informational).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_GOPS = 256 * 4 * 32 * 2.4      # lane-instructions / ns: 256 CUs x 4 SIMD-32 x 2.4 GHz (fp32 FMA = 2 flop each)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--problems", type=int, default=8192, help="(trajectory, seed) problems per GPU")
    ap.add_argument("--n_ics", type=int, default=50)
    ap.add_argument("--n_steps", type=int, default=2500)
    ap.add_argument("--poly_order", type=int, default=5)
    ap.add_argument("--sym_reg", choices=["r", "none"], default="r",
                    help="'r': the step includes the fused reversed symmetry regulariser (default); 'none': Theta + residual only")
    ap.add_argument("--w_sym_reg", type=float, default=0.1)
    ap.add_argument("--two_launch_sym", action="store_true",
                    help="A/B: residual and regulariser as two launches (x read twice) instead of the one fused closure kernel")
    ap.add_argument("--chunks", type=int, default=0, help="problem chunks per step; chunk c's all-reduce overlaps chunk c+1's kernels (default: 1 on one GPU, 2 when the [loss|grad] buffer is all-reduced)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--cpu_seconds", type=float, default=12.0)
    ap.add_argument("--config3", action="store_true", help="run the config[3] Gram + all-reduce leg even in a --profile run")
    ap.add_argument("--no_sweeps", action="store_true", help="skip the informational seed-sweep timings (N = 1 only)")
    ap.add_argument("--shard", choices=["points", "seeds"], default="points",
                    help="N > 1: 'points' = every rank holds a shard of each problem's trajectories and the packed "
                         "[loss|grad] partials are all-reduced (RCCL, default); 'seeds' = every rank owns whole problems, "
                         "no data-path collective")
    ap.add_argument("--rehearse_gloo", action="store_true",
                    help="N > 1 rehearsal on a ONE-GPU box: every rank uses cuda:0 and the collectives go through gloo "
                         "(exercises the sharded code path; the number it prints is not a scaling measurement)")
    ap.add_argument("--force_dist", action="store_true",
                    help="initialise the RCCL process group and run the collective path even with one rank (self-test)")
    ap.add_argument("--profile", action="store_true",
                    help="profiling run: only the batched steps (no legs, no single-problem loop, no CPU baseline), so that "
                         "rocprofv3 --stats averages the step's launches alone")
    ap.add_argument("--master_port", type=int, default=0, help="rendezvous port when bench.py starts the ranks itself (0 = pick a free one)")
    return ap.parse_args(argv)


def spawn_ranks(a):
    """``python bench.py --gpus N`` with no torchrun environment: start N fresh rank processes (one per GPU) through
    torch.distributed.run and relay rank 0's JSON line.  This process never touches the GPU."""
    port = a.master_port
    if not port:                                       # back-to-back runs (N = 2, 4, 8) must not trip over a port in TIME_WAIT
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line)
    sys.exit(proc.returncode if proc.returncode != 0 or line is not None else 1)


class _StdoutToStderr:
    """fd-level redirect: RCCL prints a version banner on stdout when its communicator is created;
    the driver wants exactly one JSON line there."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def cpu_baseline(x_s, dx_s, Xi_s, order, seconds, sym=None):
    """Oracle closure body (Theta cat-of-products, matmul, MSE [+ the reversed regulariser on the same precomputed
    (g(x), J_g)], autograd backward) on the host cores, one problem after the other."""
    import torch
    from oracle import sindy_oracle as O
    n_prob, n_pts = x_s.shape[0], x_s.shape[1]
    # the box exposes every host core but a 1-GPU job owns a 16-core share: more threads only thrash
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(max(1, min(16, avail)))
    regs = [O.OracleRegressor(2, order, Xi0=Xi_s[s].clone()) for s in range(n_prob)]

    def closure(s):
        reg = regs[s]
        reg.Xi.grad = None
        loss = torch.nn.functional.mse_loss(reg(x_s[s]), dx_s[s])
        if sym is not None:
            gx, jgx, w = sym
            loss = loss + w * O.symreg_reversed_precomputed(x_s[s], list(gx[s]), list(jgx[s]), reg)
        loss.backward()

    for s in range(min(2, n_prob)):
        closure(s)
    t0, calls = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        closure(calls % n_prob)
        calls += 1
    dt = time.perf_counter() - t0
    what = "Theta-build + residual + reversed sym-reg + backward" if sym is not None else "Theta-build + residual + backward"
    return {"value": calls * n_pts / dt, "unit": "points/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{calls} closure evaluations ({what}) over {n_prob} of the problems, {n_pts} points each, "
                      f"{dt:.1f} s of CPU work, torch {torch.__version__} CPU fp32"}


class _KernelTimer:
    """HIP events around every call of one engine method (on torch's current stream, where the kernels are launched)."""

    def __init__(self, eng, name):
        import torch
        self.torch, self.eng, self.name, self.orig, self.events = torch, eng, name, getattr(eng, name), []

    def __enter__(self):
        def timed(*args, **kw):
            e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
            e0.record()
            out = self.orig(*args, **kw)
            e1.record()
            self.events.append((e0, e1))
            return out
        setattr(self.eng, self.name, timed)
        return self

    def __exit__(self, *exc):
        setattr(self.eng, self.name, self.orig)

    def ms(self):
        return sorted(e0.elapsed_time(e1) for e0, e1 in self.events)


def _leg(kernel, bytes_per_launch, ms_sorted, traffic=None, **extra):
    if not ms_sorted:
        return None
    avg = sum(ms_sorted) / len(ms_sorted)
    ach = bytes_per_launch / (avg * 1e-3) / 1e9
    leg = {"kernel": kernel, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
           "traffic": traffic, "kernel_ms": avg, "kernel_ms_min": ms_sorted[0], "kernel_ms_median": ms_sorted[len(ms_sorted) // 2],
           "kernel_ms_max": ms_sorted[-1], "bytes_per_launch": bytes_per_launch, "launches": len(ms_sorted)}
    leg.update(extra)
    return leg


def _pmc_traffic(kernel_key, points_per_launch):
    """HBM bytes per launch from the committed PMC pass (profiles/pmc_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE, separate
    passes, gfx950 correction), scaled per point when it was taken on another launch size (streaming kernels)."""
    for name in ("pmc_traffic.json", "pmc_traffic_two_launch.json"):
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", name))).get("kernels", {}).get(kernel_key)
            if rec is not None and rec.get("bytes_per_point") is not None:
                return rec["bytes_per_point"] * points_per_launch
        except Exception:
            pass
    return None


def seed_sweeps(eng, dev):
    """Informational (not part of ``value``): wall time of the two seed-sweep drivers at BASELINE config[3]'s size --
    64 seeds x 50 000 points, order 3 -- best of 3: the L-BFGS sweep (closure kernel + ONE optimiser launch per inner
    iteration) and the sequential-threshold sweep (device subsample draw, index-table Gram, host solves)."""
    import torch
    out = {"shape": "64 seeds x 50000 points x 2, order 3"}
    try:
        from symode_amd import data
        from symode_amd.batched import BatchedClosure
        from symode_amd.sweep import SeedSweepLBFGS, SeedSweepSTLSQ
        X, DX = data.make_dataset("dosc", 50, 1000, dt=0.02, noise=0.0, seed=10, device=dev, n_problems=64)
        torch.manual_seed(0)
        inits = torch.randn(64, 20, device=dev)
        sw = SeedSweepLBFGS(BatchedClosure(X, DX, 3, engine=eng), 0.1, 0.05, 50)
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fit = sw.fit(inits, 60)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out["lbfgs_sweep_ms"] = best * 1e3
        out["lbfgs_sweep_epochs_max"] = int(fit["epochs"].max())
        out["lbfgs_optimiser_launches_per_iteration"] = 1
        xs, dxs = X[0].reshape(-1, 2).contiguous(), DX[0].reshape(-1, 2).contiguous()
        xs, dxs = xs.repeat(2, 1)[:100000].contiguous(), dxs.repeat(2, 1)[:100000].contiguous()
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = SeedSweepSTLSQ(xs, dxs, 3, n_seeds=64, subsample=0.5, seed0=0, engine=eng)
            _, _, passes = st.solve(0.0, 0.05, max_iter=10)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out["stlsq_sweep_ms"] = best * 1e3
        out["stlsq_passes"] = int(passes.sum())
    except Exception as e:                                   # never let an informational leg fail the bench line
        out["error"] = f"{type(e).__name__}: {e}"
    return out


def _median_ms(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def cpu_baseline_legs(x1, dx1, Xi1, order, eng=None, dev=None, reps=20):
    """BASELINE.md section 3's protocol on ONE 50x2500x2 problem: the reference's op sequence (oracle) for (i) Theta only,
    (ii) closure body forward + backward, (iii) one sequential-threshold pass with the full mask and with a partial one
    (the dense block_diag system of sindy.py:270-273) -- each with all host cores of this job and with ONE thread, 3
    warm-ups, median of ``reps`` (>= 20) repetitions; beside every leg the product's time for the same operation on the
    GPU (wall time with a synchronisation per call, same protocol)."""
    import torch
    from oracle import sindy_oracle as O
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    all_cores = max(1, min(16, avail))                   # a 1-GPU job owns a 16-core share of the box
    n_pts, d = x1.shape
    p = O.term_count(d, order)
    g = torch.Generator().manual_seed(5)
    partial = (torch.rand(d, p, generator=g) > 0.4).float()
    partial[:, 1:3] = 1.0

    def cpu_ops():
        reg = O.OracleRegressor(d, order, Xi0=Xi1.clone())

        def closure():
            reg.Xi.grad = None
            torch.nn.functional.mse_loss(reg(x1), dx1).backward()

        def stlsq(mask):
            def run():
                r2 = O.OracleRegressor(d, order, Xi0=Xi1.clone())
                r2.mask = mask.clone()
                O.stlsq_one_step(r2, x1, dx1, 0.05, 0.05, lstsq=torch.linalg.lstsq)      # the reference's own solver call
            return run
        return {"theta": lambda: O.theta(x1, order), "closure": closure, "stlsq_full_mask": stlsq(torch.ones(d, p)),
                "stlsq_partial_mask": stlsq(partial)}

    legs = {k: {"points": n_pts} for k in ("theta", "closure", "stlsq_full_mask", "stlsq_partial_mask")}
    for label, nt in (("all_cores", all_cores), ("one_thread", 1)):
        torch.set_num_threads(nt)
        for name, fn in cpu_ops().items():
            ms = _median_ms(fn, reps=reps)
            legs[name][f"cpu_ms_{label}"] = ms
            legs[name][f"cpu_points_per_s_{label}"] = n_pts / (ms * 1e-3)
    torch.set_num_threads(all_cores)
    if eng is not None:
        from symode_amd.sindy import SINDyRegression, solve_SINDy_one_step
        xg, dxg, Xig = x1.to(dev), dx1.to(dev), Xi1.to(dev)
        out = (torch.empty(1, device=dev), torch.empty(d, p, device=dev))

        def sync(fn):
            def run():
                fn()
                torch.cuda.synchronize()
            return run

        def gpu_stlsq(mask):
            def run():
                r = gpu_stlsq.reg
                r.mask.copy_(mask.to(dev))
                solve_SINDy_one_step(r, xg, dxg, 0.05, 0.05)
            return run
        gpu_stlsq.reg = SINDyRegression(d, order, False, False, threshold=0.05, device=dev)
        for name, fn in (("theta", lambda: eng.theta(xg, order)), ("closure", lambda: eng.loss_grad(xg, dxg, Xig, None, order, out=out)),
                         ("stlsq_full_mask", gpu_stlsq(torch.ones(d, p))), ("stlsq_partial_mask", gpu_stlsq(partial))):
            ms = _median_ms(sync(fn), reps=reps)
            legs[name]["gpu_ms"] = ms
            legs[name]["gpu_points_per_s"] = n_pts / (ms * 1e-3)
    return {"protocol": f"one problem of {n_pts} points x {d}, order {order} (p = {p}); 3 warm-ups, median of {reps}; CPU = the "
                        "oracle's restatement of the reference's op sequence (torch.linalg.lstsq on the ridge-augmented / "
                        "block-diagonal system), GPU = the product's call for the same operation incl. launch and synchronisation",
            "os_cpu_count": os.cpu_count(), "cores_available": avail, "torch_num_threads_all_cores": all_cores, "legs": legs}


def config3_leg(eng, dev, rank, world, group, gloo, reps=20):
    """BASELINE config[3]: selkov n_ics=10 x 10 000 steps, 64 seeds x 50 % subsample, order 3 -- every seed's subsample is
    drawn once (seeded by the seed alone), rank r gathers rows [r m / W, (r+1) m / W) of every draw with ONE index-table Gram
    launch (symode_aug_gram_gather, fp64), and ONE all-reduce sums the (64, 12, 12) fp64 stack over the ranks (RCCL over
    xGMI; gloo in a rehearsal); the 64 threshold loops then run on the host from the summed matrices (identical on every rank).
    Reports the kernel time (HIP events on the launch stream), the collective time, and the whole sweep's wall time; rank 0
    also forms the full-subsample Gram alone and checks the all-reduced stack against it."""
    import torch
    import torch.distributed as dist
    from symode_amd import data
    from symode_amd.sweep import SeedSweepSTLSQ, seeded_subsamples
    S, order = 64, 3
    xs, dxs = data.make_dataset("selkov", 10, 10000, dt=0.002, noise=0.2, seed=3, device=dev)
    x_all, dx_all = xs[0].reshape(-1, 2).contiguous(), dxs[0].reshape(-1, 2).contiguous()
    n_all = x_all.shape[0]
    m = n_all // 2
    rows = seeded_subsamples(n_all, m, list(range(S)), dev)                  # (S, m): the same table on every rank
    lo, hi = rank * m // world, (rank + 1) * m // world
    idx = rows[:, lo:hi].to(torch.int32).contiguous()
    for _ in range(3):
        G = eng.aug_gram_gather(x_all, dx_all, idx, order, 0)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in ev:
        e0.record()
        G = eng.aug_gram_gather(x_all, dx_all, idx, order, 0)
        e1.record()
    torch.cuda.synchronize()
    kern = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
    out = {"shape": f"selkov 10 x 10000 x 2, {S} seeds x {m} points (50 % subsample), order 3; this rank gathers {hi - lo} rows per seed",
           "gram_kernel_us": kern[len(kern) // 2], "gram_kernel_us_min": kern[0], "gram_bytes_per_launch": S * (hi - lo) * (16 + 4),
           "allreduce_us": None, "allreduce_bytes": S * 12 * 12 * 8, "ranks": world}
    out["gram_GBs"] = out["gram_bytes_per_launch"] / (out["gram_kernel_us"] * 1e-6) / 1e9
    if group is not None:
        coll = []
        for _ in range(reps + 3):
            Gc = G.clone()
            torch.cuda.synchronize()
            dist.barrier(group=group)
            t0 = time.perf_counter()
            dist.all_reduce(Gc, op=dist.ReduceOp.SUM, group=group)
            torch.cuda.synchronize()
            coll.append((time.perf_counter() - t0) * 1e6)
        coll = sorted(coll[3:])
        out["allreduce_us"] = coll[len(coll) // 2]
        out["allreduce_us_min"] = coll[0]
        out["allreduce_backend"] = "gloo (rehearsal)" if gloo else "nccl (RCCL)"
        counted = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(counted, group=group)
        out["ranks_observed"] = int(round(counted.item()))
    # the whole sweep as main_sweep --method stlsq runs it: Gram launch + collective + 64 host threshold loops
    # (first call = first-touch costs of the process: reported apart; then the median of five)
    walls = []
    for _ in range(6):
        torch.cuda.synchronize()
        if group is not None:
            dist.barrier(group=group)
        t0 = time.perf_counter()
        sw = SeedSweepSTLSQ(x_all, dx_all, order, n_seeds=S, group=group, engine=eng, idx=idx, idx_sorted=True)
        Xi, mask, passes = sw.solve(0.0, 0.075, max_iter=10)
        walls.append((time.perf_counter() - t0) * 1e3)
    out["sweep_wall_ms_first_call"] = walls[0]
    out["sweep_wall_ms"] = sorted(walls[1:])[2]
    out["stlsq_passes"] = int(passes.sum())
    out["masks_sha"] = __import__("hashlib").sha256(mask.numpy().astype("uint8").tobytes()).hexdigest()[:16]
    if rank == 0:
        full = eng.aug_gram_gather(x_all, dx_all, rows.to(torch.int32).contiguous(), order, 0).cpu().numpy()
        got = sw.grams()
        out["allreduced_vs_single_rank_max_rel_err"] = float(abs(got - full).max() / abs(full).max())
    return out


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)                                   # never returns
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or drop the torchrun environment "
                         f"and let bench.py start them)")
    if a.rehearse_gloo:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or a.force_dist
    ranks_observed = 1
    if use_dist:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(a.master_port or 29517), RANK="0", WORLD_SIZE="1")
        with _StdoutToStderr():
            if a.rehearse_gloo:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)
            warm = torch.ones(1, device=dev)
            dist.all_reduce(warm)                      # creates the RCCL communicator (banner goes to stderr)
            torch.cuda.synchronize()
        ranks_observed = int(round(warm.item()))       # what the collective itself counted

    import symode_amd
    from symode_amd import data
    from symode_amd.batched import BatchedClosure
    from symode_amd.constraint import constraint_Q

    eng = symode_amd.get_engine()
    S, d, order = a.problems, 2, a.poly_order
    n_pts = a.n_ics * a.n_steps

    # ---- synthetic inputs, made directly in HBM (dt 0.02 so the spiral decays over the window) ----
    x, dx = data.make_dataset("dosc", a.n_ics, a.n_steps, dt=0.02, noise=0.2, seed=1234 + rank, device=dev, n_problems=S)
    so2 = torch.tensor([[0.0, 1.0], [-1.0, 0.0]])
    Q, use_kron = constraint_Q([so2], d, order)
    Q = Q.to(dev)
    sym = None
    if a.sym_reg == "r":
        # (g(x), J_g(x)) of ONE group element exp(0.01 * so2) -- what precompute_symmreg_r leaves resident for a frozen
        # autoencoder (model_utils.py:172-211); here the decoder/encoder pair is the identity, so g is the rotation itself
        R = torch.matrix_exp(0.01 * so2).to(dev)
        gx = (x @ R.T).unsqueeze(1).contiguous()                                # (S, 1, N, d)
        jgx = R.expand(S, 1, n_pts, d, d).contiguous()                          # (S, 1, N, d, d)
        sym = (gx, jgx, a.w_sym_reg)
    n_chunks = a.chunks or (2 if (use_dist and a.shard == "points") else 1)
    clos = BatchedClosure(x, dx, order, Q=Q, use_kron_product=use_kron, allow_constant=True,
                          group=dist.group.WORLD if (use_dist and a.shard == "points") else None, n_chunks=n_chunks, engine=eng,
                          reversed_sym=sym, fuse_sym=not a.two_launch_sym)
    g = torch.Generator(device=dev)
    g.manual_seed(7 + rank)
    beta = torch.randn(S, Q.shape[1], generator=g, device=dev) * 0.3
    const = torch.randn(S, d, 1, generator=g, device=dev) * 0.1
    if use_dist:                        # all ranks optimise the same coefficients
        dist.broadcast(beta, 0)
        dist.broadcast(const, 0)

    def step():
        return clos.evaluate(beta, const)

    for _ in range(a.warmup):           # exactly W untimed warm-up steps (default 20: clocks settle within ~10)
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()

    # ---- timed region: exactly K steps; HIP events bracket every fused-kernel launch ----
    with _KernelTimer(eng, "loss_grad") as t_lg, _KernelTimer(eng, "symreg_reversed") as t_sr, _KernelTimer(eng, "loss_grad_reversed") as t_fc:
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = step()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    assert torch.isfinite(out[0]).all()

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    launches_per_step = max(max(len(t_lg.events), len(t_fc.events)) // max(a.steps, 1), 1)
    pts_per_launch = (S / launches_per_step) * n_pts
    bpp_sym = 4 * d + 1 * (4 * d + 4 * d * d)                  # x + per group element (g(x), J_g(x)): 32 B/point at d = 2
    legs = [_leg(f"loss_grad_kernel<Library<2,{order},0>> (Theta + residual + loss + grad; last workgroup finalises)",
                 pts_per_launch * (2 * 4 * d), t_lg.ms(), _pmc_traffic("loss_grad", pts_per_launch), bytes_per_point=2 * 4 * d,
                 in_step=True),
            _leg(f"symreg_reversed_kernel<Library<2,{order},0>,false> (S4 on precomputed g(x), J_g(x))", pts_per_launch * bpp_sym,
                 t_sr.ms(), _pmc_traffic("symreg_reversed", pts_per_launch), bytes_per_point=bpp_sym, in_step=True),
            # the fused closure reads x ONCE for both terms: dx + x + g(x) + J_g(x) = 40 B/point at d = 2
            _leg(f"symreg_reversed_kernel<Library<2,{order},0>,true> (fused closure: Theta + residual + S4 + grad, x read once)",
                 pts_per_launch * (4 * d + bpp_sym), t_fc.ms(), _pmc_traffic("closure_reversed", pts_per_launch),
                 bytes_per_point=4 * d + bpp_sym, in_step=True)]

    # ---- Euler-flow pair of the infinitesimal regulariser (S2): K = 10 steps + tangent, and its reverse ----
    single = {}
    if not a.profile:
        S2 = min(S, 512)
        n2 = S2 * n_pts
        x2 = x[:S2].reshape(n2, d)
        v2 = torch.randn(n2, d, device=dev) * 0.5
        g1, g2 = torch.randn(n2, d, device=dev), torch.randn(n2, d, device=dev)
        o2, f2, K2, dt2 = 2, 2, 10, 0.01                       # lv/noise99_eq_isymreg.cfg: order 2 + exp, int_t 0.1 / int_dt 0.01
        xi2 = torch.randn(d, eng.lib_size(d, o2, f2), device=dev) * 0.05
        for _ in range(2):
            eng.euler_jvp(x2, v2, xi2, None, o2, f2, K2, dt2)
            eng.euler_jvp_vjp(x2, v2, g1, g2, xi2, None, o2, f2, K2, dt2)
        with _KernelTimer(eng, "euler_jvp") as t_ej, _KernelTimer(eng, "euler_jvp_vjp") as t_ev:
            for _ in range(5):
                eng.euler_jvp(x2, v2, xi2, None, o2, f2, K2, dt2)
                eng.euler_jvp_vjp(x2, v2, g1, g2, xi2, None, o2, f2, K2, dt2)
            torch.cuda.synchronize()
        legs.append(_leg("euler_jvp_kernel<Library<2,2,2>> (S2: K = 10 Euler steps + tangent in registers)", n2 * 4 * 4 * d, t_ej.ms(),
                         bytes_per_point=4 * 4 * d, in_step=False, limited_by="valu (10 library + tangent evaluations per 32 bytes)"))
        legs.append(_leg("euler_jvp_vjp_kernel<Library<2,2,2>> (S2 reverse)", n2 * 6 * 4 * d, t_ev.ms(), bytes_per_point=6 * 4 * d,
                         in_step=False, limited_by="valu (reverse sweep over the K steps)"))
        del x2, v2, g1, g2

        # ---- single problem (the 50x2500x2 shape on its own is launch-latency bound) ----
        x1, dx1 = x[0], dx[0]
        Xi1 = clos.xi_from(beta, const)[0].contiguous()
        l1, gr1 = torch.empty(1, device=dev), torch.empty(d, clos.p, device=dev)
        ws1 = eng.new_workspace(dev, eng.lib.symode_workspace_bytes(d, order, 0, 1, n_pts))
        for _ in range(20):
            eng.loss_grad(x1, dx1, Xi1, None, order, out=(l1, gr1), ws=ws1)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            eng.loss_grad(x1, dx1, Xi1, None, order, out=(l1, gr1), ws=ws1)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph, per = torch.cuda.CUDAGraph(), 20
        with torch.cuda.graph(graph):
            for _ in range(per):                                    # back-to-back launches, no host in between
                eng.loss_grad(x1, dx1, Xi1, None, order, out=(l1, gr1), ws=ws1)
        graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        kernel_us = e0.elapsed_time(e1) * 1e3 / (10 * per)
        # closure as the L-BFGS trainer evaluates it: Xi in, [loss | grad] out, one launch, one sync (train._HostShadow)
        from symode_amd.sindy import SINDyRegression
        from symode_amd.train import _HostShadow
        reg = SINDyRegression(d, order, False, False, threshold=0.01, device=dev)
        sh = _HostShadow(reg, x1, dx1, numpy_vars=False)
        with torch.no_grad():
            for _ in range(20):
                sh.evaluate()
            tw = time.perf_counter()
            for _ in range(200):
                sh.evaluate()
            closure_us = (time.perf_counter() - tw) * 1e6 / 200
        single = {"shape": f"{a.n_ics}x{a.n_steps}x2", "launches_per_closure": 1 if sh.zero_copy else 2,
                  "kernel_us": kernel_us, "closure_us": closure_us, "latency_us": kernel_us,
                  "points_per_s": n_pts / (closure_us * 1e-6), "points_per_s_kernel": n_pts / (kernel_us * 1e-6),
                  "note": "kernel_us: GPU time per closure launch (Theta + residual + loss + grad, finalised by the last workgroup), "
                          "20 back-to-back launches replayed from a HIP graph; closure_us: host wall time of one closure of the "
                          "L-BFGS trainer (coefficients in, [loss | grad] out through pinned memory, one launch, one sync)"}

    # ---- BASELINE config[3]: 64-seed index-table Gram on point shards + ONE fp64 all-reduce of the (64, 12, 12) stack ----
    c3 = None
    if (not a.profile and not a.no_sweeps) or a.config3:
        try:
            c3 = config3_leg(eng, dev, rank, world, dist.group.WORLD if use_dist else None, a.rehearse_gloo)
        except Exception as e:                               # an informational leg never fails the bench line
            c3 = {"error": f"{type(e).__name__}: {e}"}

    if rank != 0:
        dist.destroy_process_group()
        return

    total_points = float(S) * n_pts * world * a.steps
    value = total_points / elapsed
    legs = [leg for leg in legs if leg is not None]
    in_step = [leg for leg in legs if leg.get("in_step")]
    dominant = max(in_step, key=lambda leg: leg["kernel_ms"] * leg["launches"])
    parts = "Xi from beta, " + ("ONE fused Theta+residual+reversed-sym-reg+loss+grad kernel" if (sym is not None and not a.two_launch_sym)
                                else "fused Theta+residual+loss+grad kernel" + (", fused reversed sym-reg kernel" if sym is not None else "")) + ", grad->beta"
    metric = "trajectory-points/sec through Theta-build+residual+sym-reg" if sym is not None else \
        "trajectory-points/sec through Theta-build+residual (no sym-reg leg: --sym_reg none)"
    res = {
        "metric": metric,
        "value": value, "unit": "points/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"damped_oscillator n_ics={a.n_ics} steps={a.n_steps} dim=2 poly-order={order} "
                               f"EquivSINDy-c (so2), {S} (trajectory,seed) problems per GPU resident in HBM; "
                               f"step = closure ({parts}"
                               f"{(', gloo (REHEARSAL on one GPU) all-reduce of [loss|grad]' if a.rehearse_gloo else ', RCCL all-reduce of [loss|grad]') if (use_dist and a.shard == 'points') else ''})",
                   "points_per_step_per_gpu": S * n_pts, "library_terms": clos.p, "sym_reg": a.sym_reg,
                   "parallelism": f"{a.shard[:-1]}-shard x{world}" if world > 1 else "single", "ranks_observed": ranks_observed},
        "roofline": {k: v for k, v in dominant.items() if k != "in_step"},
        "roofline_legs": legs,
    }
    if single:
        res["single_problem"] = single
    if c3 is not None:
        res["config3_gram_allreduce"] = c3
    if world == 1 and not a.profile and not a.no_sweeps:
        res["seed_sweeps"] = seed_sweeps(eng, dev)
    if world == 1 and not a.no_cpu_baseline and not a.profile:
        ns = min(S, 4)
        Xi_s = clos.xi_from(beta, const)[:ns].cpu()
        sym_c = None if sym is None else (sym[0][:ns].cpu(), sym[1][:ns].cpu(), sym[2])
        res["cpu_baseline"] = cpu_baseline(x[:ns].cpu(), dx[:ns].cpu(), Xi_s, order, a.cpu_seconds, sym_c)
        if single:
            # north_star's own comparison: Theta-build + residual (+ backward) at 50x2500x2, reference op sequence on the
            # host cores vs one closure of the GPU trainer (wall time, launch and read-back included)
            cpu1 = cpu_baseline(x[:1].cpu(), dx[:1].cpu(), Xi_s[:1], order, min(4.0, a.cpu_seconds), None)
            single["cpu_points_per_s"] = cpu1["value"]
            single["cpu_cores"] = cpu1["cores"]
            single["speedup_vs_cpu"] = single["points_per_s"] / cpu1["value"]
        # Theta-only / closure / STLSQ full + partial mask, all cores and one thread, median of >= 20
        res["cpu_baseline"].update(cpu_baseline_legs(x[0].cpu(), dx[0].cpu(), torch.randn(d, clos.p) * 0.3, order, eng, dev))
    print(json.dumps(res))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
