"""Seed sweep of an equation-discovery config in ONE process per GPU (the reference runs
``for i in {0..49}; do python main.py --seed $i --config ...; done``, run_scripts/*.sh).

    python -m symode_amd.main_sweep --config dosc/noise20_sindy.cfg --seed 0 --n_seeds 50
    python -m symode_amd.main_sweep --config selkov/noise20_eq_sindy.cfg --n_seeds 64 --method stlsq
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m symode_amd.main_sweep \
        --config selkov/noise20_eq_sindy.cfg --n_seeds 64 --method stlsq          # BASELINE config 3: 8 x MI355X

Every seed gets its own initial coefficients and its own ``--lbfgs_subsample`` draw of the data set
(main.py:36-38).  ``--method lbfgs`` (default): all seeds are optimised in lockstep by sweep.SeedSweepLBFGS on
the batched fused closure.  ``--method stlsq``: sequential-threshold least squares per seed (train.py:872-887) from
ONE gather-Gram launch over the (seed, point) index table (sweep.SeedSweepSTLSQ).  Under ``torch.distributed``
(one process per GPU, backend nccl = RCCL over xGMI) every rank holds a contiguous block of trajectories; the per-seed
``[loss | grad]`` vectors / Gram matrices are all-reduced and every rank takes identical decisions; rank 0 writes the
reference's ``eval_results/<save_dir>/seed{n}.npz`` so that ``evaluation.aggregate_results`` works unchanged.
Plain / constrained SINDy only (no autoencoder terms).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.distributed as dist

from .batched import BatchedClosure
from .dataset import get_dataset
from .evaluation import aggregate_results, sindy_truth
from .lie import LieGenerator
from .parser_utils import get_args
from .sindy import SINDyRegression
from .sweep import SeedSweepLBFGS, SeedSweepSTLSQ, seeded_subsamples


def _pop(argv, flag, default, cast):
    if flag in argv:
        i = argv.index(flag)
        value = cast(argv[i + 1])
        del argv[i:i + 2]
        return value
    return default


def _write_results(args, seeds, Xi, mask, truth):
    """eval_results/<save_dir>/seed{n}.npz per seed (evaluation/eval_eq.py:7-34, main.py:128-138)."""
    eval_dir = f'eval_results/{args["save_dir"]}'
    os.makedirs(eval_dir, exist_ok=True)
    tmask = truth != 0
    for k, s in enumerate(seeds):
        coef = np.where(mask[k], Xi[k], 0.0)
        cf = np.array([float(np.all(mask[k, i] == tmask[i])) for i in range(truth.shape[0])])
        mse = np.array([np.mean((coef[i, tmask[i]] - truth[i, tmask[i]]) ** 2) for i in range(truth.shape[0])])
        np.savez(f'{eval_dir}/seed{s}.npz', coefficients=coef, correct_form=cf, mse=mse, correct_form_all=np.all(cf),
                 mse_all=np.mean(mse))


def main(argv=None, engine=None, backend='nccl', one_gpu=False):
    """``engine`` / ``backend`` exist for the CPU rehearsal of the multi-rank path in tests (gloo + the test engine);
    ``one_gpu``: every rank uses cuda:0 (rehearsal of the HIP path with several ranks on a one-GPU box, gloo collectives)."""
    argv = list(sys.argv[1:] if argv is None else argv)
    n_seeds = _pop(argv, '--n_seeds', 50, int)
    method = _pop(argv, '--method', 'lbfgs', str)
    if method not in ('lbfgs', 'stlsq'):
        raise SystemExit(f'--method {method}: lbfgs or stlsq')
    args = vars(get_args(argv=argv))
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    if engine is None:
        if str(args['device']) == 'cpu':
            raise SystemExit('symode_amd runs the SINDy path on the GPU only (no CPU fallback): a HIP device is required')
        if world > 1:
            local = 0 if one_gpu else int(os.environ.get('LOCAL_RANK', '0'))
            torch.cuda.set_device(local)
            args['device'] = torch.device('cuda', local)
    if args['sindy_optimizer'] != 'lbfgs' or args['use_latent'] or args['w_sym_reg'] > 0:
        raise SystemExit('main_sweep covers the L-BFGS SINDy / EquivSINDy-c configs (no latent / symmetry-regulariser terms)')
    dev = args['device']
    group = None
    if world > 1:
        if not dist.is_initialized():
            dist.init_process_group(backend, **({'device_id': dev} if backend == 'nccl' else {}))
        group = dist.group.WORLD
        if rank == 0:                                               # one rank makes the data files, the others read them
            train_dataset, _, args = get_dataset(args)
        dist.barrier()
        if rank != 0:
            train_dataset, _, args = get_dataset(args)
    else:
        train_dataset, _, args = get_dataset(args)
    # Every seed draws ONE subsample of the whole flattened data set (main.py:36-38: the first batch of a shuffled loader),
    # seeded by the seed alone; rank r works on rows [r m / W, (r+1) m / W) of that draw.  The union over the ranks is
    # the single-process subsample whatever the world size, so an N-rank run fits the same problems as a 1-rank run and
    # differs from it by summation order only.  (The data sets of the reference are a few MB: every rank keeps all of
    # x, dx resident and gathers its rows; the COMPUTE is what is sharded.)
    x_all, dx_all = train_dataset.x.to(dev), train_dataset.dx.to(dev)
    n_all = x_all.shape[0]
    m = int(n_all * args['lbfgs_subsample'])
    lo, hi = rank * m // world, (rank + 1) * m // world  # shards may differ by a row: counts are summed over the ranks
    seeds = list(range(args['seed'], args['seed'] + n_seeds))
    gens = [torch.Generator().manual_seed(s) for s in seeds]

    truth = sindy_truth[args['task']]

    def padded_truth(p, template):
        if truth.shape[1] < p and not (template.include_sine or template.include_exp):
            return np.concatenate([truth, np.zeros((truth.shape[0], p - truth.shape[1]))], axis=1)
        return truth

    if method == 'stlsq':
        if args['eq_constraint']:
            raise SystemExit('--method stlsq sweeps the unconstrained library (use --method lbfgs for EquivSINDy-c)')
        idx = seeded_subsamples(n_all, m, seeds, dev)[:, lo:hi]
        sw = SeedSweepSTLSQ(x_all, dx_all, args['poly_order'], args['include_sine'], args['include_exp'], n_seeds=n_seeds,
                            subsample=args['lbfgs_subsample'], seed0=args['seed'], group=group, engine=engine, idx=idx, idx_sorted=True)
        Xi, mask, passes = sw.solve(args['w_sindy_reg'], args['threshold'], max_iter=max(1, args['num_epochs']),
                                    lstsq_driver=args.get('lstsq_driver'))
        if rank == 0:
            class _T:                                               # library flags for the truth-table padding
                include_sine, include_exp = args['include_sine'], args['include_exp']
            _write_results(args, seeds, Xi.numpy(), mask.numpy().astype(bool), padded_truth(mask.shape[-1], _T))
            print(f'{n_seeds} seeds x {sw.n_points} points (over {world} rank(s)), STLSQ passes {int(passes.min())}-{int(passes.max())}')
            print(f'near-threshold coefficients (| |coef| - thr | < 1e-4): {sw.near_threshold if sw.near_threshold else "none"}')
            return aggregate_results(args['save_dir'], min_seed=seeds[0], max_seed=seeds[-1] + 1)
        return None

    # one template regressor fixes the library / constraint; per-seed draws follow the constructor's order
    if args['eq_constraint']:
        gen = LieGenerator(**args)
        L_list = gen.get_full_basis_list()
        rd = L_list[0].shape[-1] // args['n_comps']
        args['L_list'] = [L[:rd, :rd].detach().cpu() for L in L_list]
    template = SINDyRegression(**args, **({'engine': engine} if engine is not None else {})).to(dev)
    inits, xs, dxs = [], [], []
    all_rows = seeded_subsamples(n_all, m, seeds, dev)[:, lo:hi]
    for (s, g), rows in zip(zip(seeds, gens), all_rows):
        if template.constraint:
            beta = torch.randn(template.Q.shape[1], generator=g)
            const = torch.randn(template.latent_dim, generator=g)
            inits.append(torch.cat([beta, const]))
        else:
            inits.append(torch.randn(template.latent_dim * template.get_term_num(), generator=g))
        xs.append(x_all[rows])
        dxs.append(dx_all[rows])
    X, DX = torch.stack(xs).contiguous(), torch.stack(dxs).contiguous()
    clos = BatchedClosure(X, DX, template.poly_order, template.include_sine, template.include_exp,
                          Q=template.Q if template.constraint else None,
                          use_kron_product=getattr(template, 'use_kron_product', True),
                          allow_constant=getattr(template, 'allow_constant', True), group=group,
                          **({'engine': engine} if engine is not None else {}))
    sweep = SeedSweepLBFGS(clos, args['lr_sindy'], args['threshold'], args['st_freq'], w_sindy_x=args['w_sindy_x'],
                           sindy_reg_type=args['sindy_reg_type'], w_sindy_reg=args['w_sindy_reg'])
    out = sweep.fit(torch.stack(inits).to(dev), args['num_epochs'])

    if rank != 0:
        return None
    Xi, mask = out['Xi'].cpu().numpy(), out['mask'].cpu().numpy().astype(bool)
    _write_results(args, seeds, Xi, mask, padded_truth(mask.shape[-1], template))
    print(f'{n_seeds} seeds, epochs used {int(out["epochs"].min())}-{int(out["epochs"].max())}, '
          f'finished {int(out["finished"].sum())}, NaN {int(out["nan"].sum())}, '
          f'seeds with near-threshold coefficients {[seeds[i] for i in torch.nonzero(out["near_threshold"]).flatten().tolist()] or "none"}')
    return aggregate_results(args['save_dir'], min_seed=seeds[0], max_seed=seeds[-1] + 1)


if __name__ == '__main__':
    main()
