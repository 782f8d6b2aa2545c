"""Seed sweep of an equation-discovery config in ONE process (the reference runs
``for i in {0..49}; do python main.py --seed $i --config ...; done``, run_scripts/*.sh).

    python -m symode_amd.main_sweep --config dosc/sindy_lbfgs.cfg --seed 0 --n_seeds 50

Every seed gets its own initial coefficients and its own ``--lbfgs_subsample`` draw of the data set
(main.py:36-38); all seeds are optimised in lockstep by sweep.SeedSweepLBFGS on the batched fused
closure, and each seed leaves the reference's ``eval_results/<save_dir>/seed{n}.npz`` so that
``evaluation.aggregate_results`` works unchanged.  Plain / constrained L-BFGS SINDy only (no
autoencoder terms).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from .batched import BatchedClosure
from .dataset import get_dataset
from .evaluation import aggregate_results, sindy_truth
from .lie import LieGenerator
from .parser_utils import get_args
from .sindy import SINDyRegression
from .sweep import SeedSweepLBFGS


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    n_seeds = 50
    if "--n_seeds" in argv:
        i = argv.index("--n_seeds")
        n_seeds = int(argv[i + 1])
        del argv[i:i + 2]
    args = vars(get_args(argv=argv))
    if str(args['device']) == 'cpu':
        raise SystemExit('symode_amd runs the SINDy path on the GPU only (no CPU fallback): a HIP device is required')
    if args['sindy_optimizer'] != 'lbfgs' or args['use_latent'] or args['w_sym_reg'] > 0:
        raise SystemExit('main_sweep covers the L-BFGS SINDy / EquivSINDy-c configs (no latent / symmetry-regulariser terms)')
    dev = args['device']
    train_dataset, _, args = get_dataset(args)
    x_all, dx_all = train_dataset.x.to(dev), train_dataset.dx.to(dev)
    m = int(len(train_dataset) * args['lbfgs_subsample'])
    seeds = list(range(args['seed'], args['seed'] + n_seeds))

    # one template regressor fixes the library / constraint; per-seed draws follow the constructor's order
    if args['eq_constraint']:
        gen = LieGenerator(**args)
        L_list = gen.get_full_basis_list()
        rd = L_list[0].shape[-1] // args['n_comps']
        args['L_list'] = [L[:rd, :rd].detach().cpu() for L in L_list]
    template = SINDyRegression(**args).to(dev)
    inits, xs, dxs = [], [], []
    for s in seeds:
        g = torch.Generator().manual_seed(s)
        if template.constraint:
            beta = torch.randn(template.Q.shape[1], generator=g)
            const = torch.randn(template.latent_dim, generator=g)
            inits.append(torch.cat([beta, const]))
        else:
            inits.append(torch.randn(template.latent_dim * template.get_term_num(), generator=g))
        rows = torch.randperm(len(train_dataset), generator=g)[:m].to(dev)
        xs.append(x_all[rows])
        dxs.append(dx_all[rows])
    X, DX = torch.stack(xs).contiguous(), torch.stack(dxs).contiguous()
    clos = BatchedClosure(X, DX, template.poly_order, template.include_sine, template.include_exp,
                          Q=template.Q if template.constraint else None,
                          use_kron_product=getattr(template, 'use_kron_product', True),
                          allow_constant=getattr(template, 'allow_constant', True))
    sweep = SeedSweepLBFGS(clos, args['lr_sindy'], args['threshold'], args['st_freq'], w_sindy_x=args['w_sindy_x'],
                           sindy_reg_type=args['sindy_reg_type'], w_sindy_reg=args['w_sindy_reg'])
    out = sweep.fit(torch.stack(inits).to(dev), args['num_epochs'])

    truth = sindy_truth[args['task']]
    p = out['mask'].shape[-1]
    if truth.shape[1] < p and not (template.include_sine or template.include_exp):
        truth = np.concatenate([truth, np.zeros((truth.shape[0], p - truth.shape[1]))], axis=1)
    eval_dir = f'eval_results/{args["save_dir"]}'
    os.makedirs(eval_dir, exist_ok=True)
    Xi, mask = out['Xi'].cpu().numpy(), out['mask'].cpu().numpy().astype(bool)
    tmask = truth != 0
    for k, s in enumerate(seeds):                                   # evaluation/eval_eq.py:7-34 per seed
        coef = np.where(mask[k], Xi[k], 0.0)
        cf = np.array([float(np.all(mask[k, i] == tmask[i])) for i in range(truth.shape[0])])
        mse = np.array([np.mean((coef[i, tmask[i]] - truth[i, tmask[i]]) ** 2) for i in range(truth.shape[0])])
        np.savez(f'{eval_dir}/seed{s}.npz', coefficients=coef, correct_form=cf, mse=mse, correct_form_all=np.all(cf),
                 mse_all=np.mean(mse))
    print(f'{n_seeds} seeds, epochs used {int(out["epochs"].min())}-{int(out["epochs"].max())}, '
          f'finished {int(out["finished"].sum())}, NaN {int(out["nan"].sum())}')
    return aggregate_results(args['save_dir'], min_seed=seeds[0], max_seed=seeds[-1] + 1)


if __name__ == '__main__':
    main()
