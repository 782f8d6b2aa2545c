#!/usr/bin/env python3
"""Headline benchmark: trajectory-points/s through the fused Theta-build + residual (+ gradient) pass.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): damped oscillator, n_ics=50 x steps=2500 x dim=2 fp32,
poly-order 5 (p = 21), EquivSINDy-c (so2 equivariance constraint, Xi = reshape(Q beta) + const),
batched over S independent (trajectory, seed) problems per GPU that are resident in HBM.
One *step* = one closure evaluation of every problem: Xi from (beta, const), the fused HIP
kernel (loss + dloss/dXi, Theta never materialised), projection of the gradient onto
(beta, const) and -- for N > 1 -- the RCCL all-reduce of the packed [loss | grad] partials
(every rank holds its own shard of each problem's trajectories: weak scaling).

Prints ONE JSON line (rank 0) following the driver's contract, plus
  roofline     : the dominant kernel priced against the HBM roof (algorithmic bytes / live
                 HIP-event time; PMC traffic when profiles/pmc_traffic.json is present),
  cpu_baseline : the CPU oracle ("port" of the reference op sequence) timed on the host cores
                 on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--problems", type=int, default=8192, help="(trajectory, seed) problems per GPU")
    ap.add_argument("--n_ics", type=int, default=50)
    ap.add_argument("--n_steps", type=int, default=2500)
    ap.add_argument("--poly_order", type=int, default=5)
    ap.add_argument("--chunks", type=int, default=0, help="problem chunks per step; chunk c's all-reduce overlaps chunk c+1's kernel (default: 1 on one GPU, 2 when the [loss|grad] buffer is all-reduced)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--cpu_seconds", type=float, default=12.0)
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
                    help="profiling run: only the batched steps (no single-problem loop, no CPU baseline), so that "
                         "rocprofv3 --stats averages the headline launches alone")
    return ap.parse_args()


def cpu_baseline(x_s, dx_s, Xi_s, order, seconds):
    """Oracle closure body (Theta cat-of-products, matmul, MSE, autograd backward) on host cores."""
    from oracle import sindy_oracle as O
    mask = torch.ones_like(Xi_s[0])
    n_prob, n_pts = x_s.shape[0], x_s.shape[1]
    # the box exposes every host core but a 1-GPU job owns a 16-core share: more threads only thrash
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(max(1, min(16, avail)))
    for s in range(min(2, n_prob)):
        O.mse_loss_and_grad(x_s[s], dx_s[s], Xi_s[s], mask, order)
    t0, calls = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        s = calls % n_prob
        O.mse_loss_and_grad(x_s[s], dx_s[s], Xi_s[s], mask, order)
        calls += 1
    dt = time.perf_counter() - t0
    return {"value": calls * n_pts / dt, "unit": "points/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{calls} closure evaluations (Theta-build + residual + backward) over {n_prob} of the "
                      f"problems, {n_pts} points each, {dt:.1f} s of CPU work, torch {torch.__version__} CPU fp32"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus or world == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    if a.rehearse_gloo:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or a.force_dist
    if use_dist:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", RANK="0", WORLD_SIZE="1")
        with _StdoutToStderr():
            if a.rehearse_gloo:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)
            warm = torch.zeros(1, device=dev)
            dist.all_reduce(warm)                      # creates the RCCL communicator (banner goes to stderr)
            torch.cuda.synchronize()

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
    n_chunks = a.chunks or (2 if (use_dist and a.shard == "points") else 1)
    clos = BatchedClosure(x, dx, order, Q=Q, use_kron_product=use_kron, allow_constant=True,
                          group=dist.group.WORLD if (use_dist and a.shard == "points") else None, n_chunks=n_chunks, engine=eng)
    g = torch.Generator(device=dev)
    g.manual_seed(7 + rank)
    beta = torch.randn(S, Q.shape[1], generator=g, device=dev) * 0.3
    const = torch.randn(S, d, 1, generator=g, device=dev) * 0.1
    if use_dist:                        # all ranks optimise the same coefficients
        dist.broadcast(beta, 0)
        dist.broadcast(const, 0)

    def step():
        return clos.evaluate(beta, const)

    # W untimed warm-up steps as asked, topped up to >= 20 so that a small W does not leave the first timed launches on
    # ramping clocks (the timed region below is exactly K steps either way)
    for _ in range(max(a.warmup, 20)):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()

    # ---- timed region: exactly K steps; HIP events bracket every fused-kernel launch ----
    events = []
    orig = eng.loss_grad

    def timed_loss_grad(*args, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(*args, **kw)
        e1.record()
        events.append((e0, e1))
        return out

    eng.loss_grad = timed_loss_grad
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    eng.loss_grad = orig
    assert torch.isfinite(out[0]).all()

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    kern_all = sorted(e0.elapsed_time(e1) for e0, e1 in events)
    kern_ms = sum(kern_all) / max(len(kern_all), 1)
    launches_per_step = len(events) // max(a.steps, 1)
    bytes_per_launch = (S / launches_per_step) * n_pts * (2 * 4 * d)        # read x and dx once: 16 B/point at d=2
    achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9

    # ---- single-problem latency (the 50x2500x2 shape on its own is launch-latency bound) ----
    single_us = float("nan")
    if not a.profile:
        x1, dx1 = x[0], dx[0]
        Xi1 = clos.xi_from(beta, const)[0].contiguous()
        for _ in range(20):
            eng.loss_grad(x1, dx1, Xi1, None, order)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 200
        e0.record()
        for _ in range(reps):
            eng.loss_grad(x1, dx1, Xi1, None, order)
        e1.record()
        torch.cuda.synchronize()
        single_us = e0.elapsed_time(e1) * 1e3 / reps

    if rank != 0:
        dist.destroy_process_group()
        return

    total_points = float(S) * n_pts * world * a.steps
    value = total_points / elapsed
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            rec = json.load(open(pmc))
            per_point = rec.get("loss_grad_bytes_per_point")
            if per_point is not None and abs(rec.get("points_per_launch", 0) - (S / launches_per_step) * n_pts) < 1:
                traffic = rec["loss_grad_bytes_per_launch"]          # PMC pass taken on this very launch shape
            elif per_point is not None:
                traffic = per_point * (S / launches_per_step) * n_pts  # scaled from the profiled launch (streaming kernel)
        except Exception:
            traffic = None
    res = {
        "metric": "trajectory-points/sec through Theta-build+residual+sym-reg",
        "value": value, "unit": "points/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"damped_oscillator n_ics={a.n_ics} steps={a.n_steps} dim=2 poly-order={order} "
                               f"EquivSINDy-c (so2), {S} (trajectory,seed) problems per GPU resident in HBM; "
                               f"step = closure (Xi from beta, fused Theta+residual+loss+grad kernel, grad->beta"
                               f"{', RCCL all-reduce of [loss|grad]' if (use_dist and a.shard == 'points') else ''})",
                   "points_per_step_per_gpu": S * n_pts, "library_terms": clos.p,
                   "parallelism": f"{a.shard[:-1]}-shard x{world}" if world > 1 else "single"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "loss_grad_kernel<Library<2,5,0>> (+ finalize)", "kernel_ms": kern_ms, "kernel_ms_min": kern_all[0], "kernel_ms_median": kern_all[len(kern_all) // 2],
                     "kernel_ms_max": kern_all[-1],
                     "bytes_per_launch": bytes_per_launch, "launches_per_step": launches_per_step},
        "single_problem": {"shape": f"{a.n_ics}x{a.n_steps}x2", "latency_us": single_us,
                           "points_per_s": n_pts / (single_us * 1e-6)},
    }
    if world == 1 and not a.no_cpu_baseline and not a.profile:
        ns = min(S, 8)
        Xi_s = clos.xi_from(beta, const)[:ns].cpu()
        res["cpu_baseline"] = cpu_baseline(x[:ns].cpu(), dx[:ns].cpu(), Xi_s, order, a.cpu_seconds)
        res["speedup_vs_cpu_batched"] = value / res["cpu_baseline"]["value"]
        res["speedup_vs_cpu_single_problem"] = res["single_problem"]["points_per_s"] / res["cpu_baseline"]["value"]
    print(json.dumps(res))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
