"""Command-line / config-file surface of the reference drivers (parser_utils.py), flag for flag.

Flag names, types and defaults are the interface (run_configs/*.cfg and run_scripts/*.sh are
written against them) and are kept verbatim; the parsing rules are the reference's:
``--config X`` reads ``run_configs/X``, splits it on whitespace into argv tokens
(parser_utils.py:183-186) and parses it with the same parser; a value given on the command
line wins over the file iff it differs from the parser default (parser_utils.py:100-114).
"""
from __future__ import annotations

import argparse
import os

import torch

RUN_CONFIG_DIR = 'run_configs'

_F, _I, _S = float, int, str
_FLAG = "flag"          # action='store_true'

# (name, kind, default[, nargs])  -- order and values follow parser_utils.py:9-94
_MAIN_ARGS = [
    # dataset
    ("task", _S, "rd"), ("mt_data", _FLAG, None), ("noise", _F, 0.0), ("smoothing", _S, None),
    # hyper-parameters
    ("batch_size", _I, 256), ("num_epochs", _I, 1000), ("lr_ae", _F, 1e-3), ("lr_d", _F, 1e-3), ("lr_g", _F, 1e-3),
    ("lr_sindy", _F, 1e-3), ("w_recon", _F, 1), ("w_gan", _F, 1), ("w_reg_norm", _F, 1e-2), ("w_reg_sim", _F, 1e-2),
    ("w_reg_ortho", _F, 0.0), ("w_reg_closure", _F, 0.0), ("w_sindy_z", _F, 1e-3), ("w_sindy_x", _F, 1e-1),
    ("sindy_reg_type", _S, "l1"), ("w_sindy_reg", _F, 1e-1), ("sym_reg_type", _S, "i"), ("w_sym_reg", _F, 0.0),
    # general model configuration
    ("latent_dim", _I, 2), ("hidden_dim", _I, 512), ("n_layers", _I, 5), ("n_comps", _I, 1), ("activation", _S, "ReLU"),
    ("activation_args", _F, [], "+"), ("load_laligan", _S, None), ("fix_laligan", _FLAG, None),
    # autoencoder
    ("ae_arch", _S, "mlp"), ("ortho_ae", _FLAG, None), ("batch_norm", _FLAG, None),
    # generator
    ("repr", _S, "(1,so2)"), ("group_idx", _S, "0"), ("coef_dist", _S, "normal"), ("g_init", _S, "random"),
    ("sigma_init", _F, 1), ("uniform_max", _F, 1), ("int_param", _FLAG, None), ("int_param_max", _I, 2),
    ("int_param_noise", _F, 0.1), ("gan_st_freq", _I, 5), ("gan_st_thres", _F, 0.3), ("keep_center", _FLAG, None),
    # discriminator
    ("use_original_x", _FLAG, None), ("use_invariant_y", _FLAG, None), ("embed_y", _FLAG, None), ("y_dim", _I, 1),
    ("y_classes", _I, 2), ("y_embed_dim", _I, 16),
    # SINDy
    ("include_sindy", _FLAG, None), ("poly_order", _I, 2), ("include_sine", _FLAG, None), ("include_exp", _FLAG, None),
    ("st_freq", _I, 100), ("threshold", _F, 0.1), ("use_latent", _FLAG, None), ("distill_latent", _FLAG, None),
    ("eq_constraint", _FLAG, None), ("constrain_constant", _FLAG, None), ("int_t", _F, 0.1), ("int_dt", _F, 0.01),
    ("sindy_optimizer", _S, "adam"), ("lbfgs_subsample", _F, 1.0),
    # least-squares driver of the STLSQ solves (new flag): gels = torch.linalg.lstsq on a GPU (default here),
    # gelsy = its rank-truncating CPU default
    ("lstsq_driver", _S, None),
    # L-BFGS fits run optimiser + epoch logic on the device by default; --torch_lbfgs selects torch.optim.LBFGS's own
    # sequence of tensor operations on host-resident variables instead (--device_lbfgs: accepted, the default now)
    ("device_lbfgs", _FLAG, None), ("torch_lbfgs", _FLAG, None),
    # PySR (accepted for config compatibility; that path is out of scope)
    ("pysr_subsample", _F, 1.0), ("pysr_bs", _I, 1000), ("pysr_symmreg", _FLAG, None),
    # run settings
    ("gpu", _I, 0), ("log_interval", _I, 1), ("save_interval", _I, 100), ("print_li", _FLAG, None),
    ("print_eq", _FLAG, None), ("wandb_name", _S, "test"), ("save_dir", _S, "test"), ("seed", _I, 42),
]

# parser_utils.py:122-171
_SINDY_ARGS = [
    ("task", _S, "rd"), ("batch_size", _I, 64), ("num_epochs", _I, 100), ("lr_ae", _F, 1e-3), ("lr", _F, 1e-3),
    ("reg_type", _S, "l1"), ("w_reg", _F, 1e-1), ("rel_loss", _FLAG, None), ("w_sindy_z", _F, 1e-1), ("w_sindy_x", _F, 1e-1),
    ("w_align", _F, 1e-1), ("w_cons", _F, 1e-1),
    ("latent_dim", _I, 2), ("hidden_dim", _I, 512), ("n_layers", _I, 5), ("n_comps", _I, 1), ("activation", _S, "ReLU"),
    ("activation_args", _F, [], "+"),
    ("learn_ae", _FLAG, None), ("ae_arch", _S, "mlp"), ("ortho_ae", _FLAG, None), ("batch_norm", _FLAG, None),
    ("load_ae", _FLAG, None), ("load_Lie", _FLAG, None), ("load_dir", _S, "autoencoder.pt"),
    ("poly_order", _I, 2), ("include_sine", _FLAG, None), ("include_exp", _FLAG, None), ("seq_thres_freq", _I, 100),
    ("threshold", _F, 0.1), ("lstsq_driver", _S, None),
    ("use_delay", _FLAG, None), ("delay_n", _I, 5), ("delay_q", _I, 3), ("delay_p", _I, 2),
    ("gpu", _I, 0), ("log_interval", _I, 1), ("save_interval", _I, 100), ("wandb_name", _S, "sindy-test"),
    ("save_dir", _S, "sindy-test"), ("seed", _I, 42),
]


def _build(spec):
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", help="Path to a configuration file")
    for entry in spec:
        name, kind, default = entry[:3]
        if kind == _FLAG:
            parser.add_argument(f"--{name}", action="store_true")
        elif len(entry) > 3:
            parser.add_argument(f"--{name}", nargs=entry[3], type=kind, default=default)
        else:
            parser.add_argument(f"--{name}", type=kind, default=default)
    return parser


def parse_config(file_path):
    """Whitespace-separated argv tokens of a config file (parser_utils.py:183-186)."""
    with open(file_path, 'r') as f:
        return [item.strip() for item in f.read().split() if item.strip()]


def _device(gpu):
    return torch.device('cuda:{}'.format(gpu) if torch.cuda.is_available() and gpu != -1 else 'cpu')


def get_args(construct_parser=False, argv=None):
    parser = _build(_MAIN_ARGS)
    if construct_parser:
        return parser
    defaults = {a.dest: a.default for a in parser._actions if a.dest != 'help'}
    args, _ = parser.parse_known_args(argv)
    provided = {k: v for k, v in vars(args).items() if v != defaults[k]}
    if args.config:
        cfg = parser.parse_args(parse_config(os.path.join(RUN_CONFIG_DIR, args.config)))
        for key, value in vars(cfg).items():
            if key not in provided:                       # CLI wins iff it differs from the default
                setattr(args, key, value)
    else:
        args = parser.parse_args(argv)
    args.device = _device(args.gpu)
    return args


def get_sindy_args(argv=None):
    parser = _build(_SINDY_ARGS)
    args, _ = parser.parse_known_args(argv)
    if args.config:
        args = parser.parse_args(parse_config(args.config))   # no run_configs/ prefix here (parser_utils.py:174-175)
    else:
        args = parser.parse_args(argv)
    args.device = _device(args.gpu)
    return args
