"""Driver with the reference's main.py flag / config surface for the equation-discovery runs:

    python -m symode_amd.main --seed 0 --config dosc/noise20_sindy.cfg        (from a directory holding run_configs/)

args -> seed -> dataset -> loaders -> autoencoder / generator -> optional LaLiGAN load ->
regressor (with the equivariance constraint when --eq_constraint) -> train fn -> checkpoints ->
eval_results/<save_dir>/seed{seed}.npz, as main.py:18-140 of the reference.  Multi-timestep tasks
(mt_rd, mt_lv, mt_selkov) go to train_lassi (autoencoder + LieGAN on stock PyTorch, latent SINDy on HIP).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import train as T
from .autoencoder import AutoEncoder
from .dataset import get_dataset, make_loader
from .evaluation import eval_sindy_regressor, sindy_truth
from .lie import Discriminator, LieGenerator
from .parser_utils import get_args
from .sindy import SINDyRegression


def main(argv=None, backend='nccl', one_gpu=False):
    """``WORLD_SIZE`` > 1 (python -m torch.distributed.run --nproc-per-node N -m symode_amd.main ...): the L-BFGS fits shard
    their fixed batch over the ranks' GPUs by points -- every rank draws the SAME batch (same seed) and keeps rows
    [r m / W, (r+1) m / W); loss, gradient and, for the relative regularisers, numerators and denominators are summed over
    the ranks inside ``train_SIGED_lbfgs(group=...)``; rank 0 alone prints and writes.  ``backend`` / ``one_gpu``: gloo
    rehearsal with every rank on cuda:0 (tests)."""
    args = get_args(argv=argv)
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    T.wandb.init(project='anonym', entity='anonym', name=args.wandb_name, config=args)
    seed = args.seed
    torch.manual_seed(seed)
    np.random.seed(seed)
    args = vars(args)
    if str(args['device']) == 'cpu':
        raise SystemExit('symode_amd runs the SINDy path on the GPU only (no CPU fallback): a HIP device is required')
    group = None
    if world > 1:
        import contextlib
        import io
        import torch.distributed as dist
        if args['mt_data'] or args['sindy_optimizer'] != 'lbfgs' or args['use_latent']:
            raise SystemExit('multi-rank runs of symode_amd.main cover the non-latent L-BFGS fits (point shards)')
        local = 0 if one_gpu else int(os.environ.get('LOCAL_RANK', '0'))
        torch.cuda.set_device(local)
        args['device'] = torch.device('cuda', local)
        if not dist.is_initialized():
            dist.init_process_group(backend, **({'device_id': args['device']} if backend == 'nccl' else {}))
        group = dist.group.WORLD
        if rank != 0:                                       # one rank makes the data files and talks
            dist.barrier()
            with contextlib.redirect_stdout(io.StringIO()):
                return _run(args, seed, group, rank, world)
        train_dataset_first = get_dataset(args)             # (rank 0 writes ./data/*.pt if they are missing)
        dist.barrier()
        return _run(args, seed, group, rank, world, train_dataset_first)
    return _run(args, seed, group, rank, world)


def _run(args, seed, group, rank, world, datasets=None):
    train_dataset, val_dataset, args = datasets if datasets is not None else get_dataset(args)
    if group is not None:
        # rank 0 may just have GENERATED the data files (drawing random numbers) where the others read them: from here on
        # every rank must draw the same batch and the same initial coefficients
        torch.manual_seed(seed)
        np.random.seed(seed)
    # DataLoader semantics (main.py:33-39), served from device-resident arrays: one gather per batch
    if args['sindy_optimizer'] != 'lbfgs':
        train_loader = make_loader(train_dataset, args['batch_size'], True, args['device'])
    else:
        data_size = int(len(train_dataset) * args['lbfgs_subsample'])
        train_loader = make_loader(train_dataset, data_size, True, args['device'])
    val_loader = make_loader(val_dataset, args['batch_size'], False, args['device'])

    autoencoder = AutoEncoder(**args).to(args['device'])
    discriminator = Discriminator(**args).to(args['device'])
    generator = LieGenerator(**args).to(args['device'])

    laligan_path = args['load_laligan']
    if laligan_path is not None:
        autoencoder.load_state_dict(torch.load(f'saved_models/{laligan_path}/autoencoder.pt', weights_only=True))
        saved = torch.load(f'saved_models/{laligan_path}/generator.pt', weights_only=True)
        current = generator.state_dict()
        for name, param in current.items():                       # tolerate older generator files (main.py:52-60)
            saved.setdefault(name, param)
        generator.load_state_dict({k: v for k, v in saved.items() if k in current})
        masks = torch.load(f'saved_models/{laligan_path}/generator_mask.pt', weights_only=True)
        generator.masks = [m.to(args['device']) if m is not None else None for m in masks]
    if args['fix_laligan']:
        for module in (autoencoder, generator, discriminator):
            for param in module.parameters():
                param.requires_grad = False

    if args['eq_constraint']:
        L_list = generator.get_full_basis_list()
        repr_dim = L_list[0].shape[-1] // args['n_comps']
        args['L_list'] = [L[:repr_dim, :repr_dim].detach().cpu() for L in L_list]      # main.py:72-76
    regressor = SINDyRegression(**args).to(args['device'])
    if args['distill_latent']:
        args_distill = dict(args, eq_constraint=False, use_latent=False, L_list=[])
        regressor_dst = SINDyRegression(**args_distill).to(args['device'])
    else:
        regressor_dst = None

    if args['mt_data']:
        train_fn = T.train_lassi
    elif args['sindy_optimizer'] == 'lbfgs':
        train_fn = T.train_SIGED_lbfgs
    else:
        train_fn = T.train_SIGED
    if group is not None:
        x, dx = next(iter(train_loader))                    # the fit's ONE fixed batch (train.py:626): identical on every rank
        m = x.shape[0]
        lo, hi = rank * m // world, (rank + 1) * m // world
        train_loader = [(x[lo:hi].contiguous(), dx[lo:hi].contiguous())]
        args = dict(args, group=group)
    train_fn(autoencoder=autoencoder, discriminator=discriminator, generator=generator, regressor=regressor,
             regressor_dst=regressor_dst, train_loader=train_loader, test_loader=val_loader, **args)
    if rank != 0:
        return regressor

    out = f'saved_models/{args["save_dir"]}'
    os.makedirs(out, exist_ok=True)
    torch.save(autoencoder.state_dict(), f'{out}/autoencoder.pt')
    torch.save(discriminator.state_dict(), f'{out}/discriminator.pt')
    torch.save(generator.state_dict(), f'{out}/generator.pt')
    torch.save(generator.masks, f'{out}/generator_mask.pt')
    torch.save(regressor.state_dict(), f'{out}/regressor.pt')
    torch.save(regressor.mask, f'{out}/regressor_mask.pt')             # additive: the reference drops the mask
    torch.save(regressor.L_list, f'{out}/regressor_lie_list.pt')
    if regressor_dst is not None:
        torch.save(regressor_dst.state_dict(), f'{out}/regressor.pt')  # overwrites, as main.py:116-117 does

    if args['mt_data']:                                                # discovery runs have no truth table (main.py:120)
        T.wandb.finish()
        return regressor
    print('\n=== Evaluation ===\n')
    true_eq = sindy_truth[args['task']]
    regressor_eval = regressor_dst if args['distill_latent'] else regressor
    n_terms = regressor_eval.mask.shape[1]
    if true_eq.shape[1] < n_terms and not (regressor_eval.include_sine or regressor_eval.include_exp):
        # a higher polynomial order only appends columns: the truth table extends with zeros
        true_eq = np.concatenate([true_eq, np.zeros((true_eq.shape[0], n_terms - true_eq.shape[1]))], axis=1)
    coef, cf, mse, cf_all, mse_all = eval_sindy_regressor(regressor_eval, true_eq)
    print(f'Near-threshold coefficients (| |coef| - thr | < 1e-4): {regressor_eval.near_threshold or "none"}')
    print(f'Correct form: {cf}')
    print(f'MSE: {np.where(cf, mse, 0.0)}')
    print(f'MSE (any): {mse}')
    eval_save_dir = f'eval_results/{args["save_dir"]}'
    os.makedirs(eval_save_dir, exist_ok=True)
    np.savez(f'{eval_save_dir}/seed{seed}.npz', coefficients=coef, correct_form=cf, mse=mse, correct_form_all=cf_all,
             mse_all=mse_all)
    T.wandb.finish()
    return regressor_eval


if __name__ == '__main__':
    main()
