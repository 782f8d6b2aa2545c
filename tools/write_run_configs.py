#!/usr/bin/env python3
"""Writes symmetry-ode-discovery_amd/run_configs/**.cfg and run_scripts/*.sh under the names a user of the
reference types (``--config dosc/noise20_sindy.cfg``, ``bash run_scripts/dosc_noise20_sindy.sh``).

The table below holds the EFFECTIVE settings of every experiment (SURVEY.md Appendix A + the reference's
run_configs read as data: hyper-parameters, save-dir names); each file is emitted from it in this parser's own
flag order, one line per parser section -- parser_utils.parse_config splits the file on whitespace
(reference parser_utils.py:183-186).  tests/test_host_train.py::test_reference_config_names_parse checks that
every file parses back to its row.  Out of scope and not written: the PySR configs (``*_pysr*.cfg``).

    python tools/write_run_configs.py          # rewrite the files in place
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "symmetry-ode-discovery_amd")
sys.path.insert(0, ROOT)

# ---- shared blocks -----------------------------------------------------------------------------------
GP = dict(smoothing="gp")
LBFGS = dict(sindy_optimizer="lbfgs", w_sindy_z=0.0, w_sindy_x=1.0, w_sindy_reg=0.0)
LOG10 = dict(log_interval=1, save_interval=10)


def named(stem):
    return dict(save_dir=stem, wandb_name=stem)


# frozen LaLiGAN halves (autoencoder + generator) the symmetry-regularised runs load
LV_LALIGAN = dict(n_comps=2, repr="(2,1,2)", group_idx="0", latent_dim=2, ae_arch="mlp", ortho_ae=True, batch_norm=True)
SK_LALIGAN = dict(n_comps=2, repr="(2,sim2)", group_idx="0", latent_dim=2, ae_arch="mlp", ortho_ae=True, batch_norm=True,
                  n_layers=4, hidden_dim=128)


def lv_symreg(kind, stem):
    """EquivSINDy-r on Lotka-Volterra, 99 % noise: regulariser type i (parser default) / f / r."""
    cfg = dict(task="lv", noise=0.99, **GP, **LV_LALIGAN, load_laligan="laligan-noise99-lv", fix_laligan=True, w_recon=0.0,
               w_gan=0.0, w_reg_norm=0.0, **LBFGS, lbfgs_subsample=0.01, lr_sindy=0.1, w_sym_reg=0.1, include_exp=True,
               st_freq=100, threshold=0.15, **LOG10, **named(stem), print_eq=True, num_epochs=100)
    if kind != "i":
        cfg["sym_reg_type"] = kind
    return cfg


def selkov_symreg(kind, stem):
    cfg = dict(task="selkov", noise=0.2, **GP, **SK_LALIGAN, load_laligan="laligan-noise20-selkov", fix_laligan=True, **LBFGS,
               lbfgs_subsample=0.5, lr_sindy=1.0, w_sym_reg=0.1, int_t=0.03, sym_reg_type=kind, poly_order=3, st_freq=50,
               threshold=7.5e-2, **LOG10, **named(stem), print_eq=True, num_epochs=200)
    return cfg


CONFIGS = {
    # ---- damped oscillator, 20 % noise (BASELINE configs 1-2) ----------------------------------------
    "dosc/noise20_sindy.cfg": dict(task="dosc", noise=0.2, **GP, **LBFGS, lbfgs_subsample=0.5, lr_sindy=0.1, w_sym_reg=0.0,
                                   poly_order=2, st_freq=50, threshold=5e-2, **LOG10, **named("sindy-noise20-dosc"),
                                   print_eq=True, num_epochs=200),
    "dosc/noise20_esindy.cfg": dict(task="dosc", noise=0.2, **GP, n_comps=1, repr="(1,so2)", group_idx="0", latent_dim=2,
                                    ae_arch="none", **LBFGS, lbfgs_subsample=0.5, lr_sindy=1.0, eq_constraint=True, poly_order=2,
                                    w_sym_reg=0.0, st_freq=100, threshold=1e-2, **LOG10, **named("esindy-noise20-dosc"),
                                    print_eq=True, batch_size=8192, num_epochs=100),
    "dosc/noise20_wsindy.cfg": dict(task="dosc", noise=0.2, **GP, w_sindy_reg=0.0, threshold=5e-2, **LOG10,
                                    **named("wsindy-noise20-dosc"), print_eq=True, num_epochs=10),
    # extension (not in the reference): BASELINE.json configs[1] quotes the constrained fit at poly-order 5
    "dosc/noise20_esindy_order5.cfg": dict(task="dosc", noise=0.2, **GP, n_comps=1, repr="(1,so2)", group_idx="0", latent_dim=2,
                                           ae_arch="none", **LBFGS, lbfgs_subsample=0.5, lr_sindy=1.0, eq_constraint=True,
                                           poly_order=5, w_sym_reg=0.0, st_freq=100, threshold=1e-2, **LOG10,
                                           **named("esindy-order5-noise20-dosc"), print_eq=True, batch_size=8192, num_epochs=100),
    # ---- growth, 5 % multiplicative noise ------------------------------------------------------------
    "growth/noise05_sindy.cfg": dict(task="growth", noise=0.05, **GP, **LBFGS, lbfgs_subsample=0.5, lr_sindy=1.0, w_sym_reg=0.0,
                                     poly_order=2, st_freq=50, threshold=5e-2, **LOG10, **named("sindy-noise05-growth"),
                                     print_eq=True, num_epochs=200),
    "growth/noise05_esindy.cfg": dict(task="growth", noise=0.05, **GP, n_comps=1, repr="(1,scaling2)", group_idx="0", latent_dim=2,
                                      ae_arch="none", **LBFGS, lbfgs_subsample=0.5, lr_sindy=1.0, eq_constraint=True,
                                      constrain_constant=True, poly_order=2, w_sym_reg=0.0, st_freq=100, threshold=5e-2, **LOG10,
                                      **named("esindy-noise05-growth"), print_eq=True, batch_size=8192, num_epochs=100),
    "growth/noise05_wsindy.cfg": dict(task="growth", noise=0.05, **GP, w_sindy_reg=0.05, poly_order=2, threshold=5e-2, **LOG10,
                                      **named("wsindy-noise05-growth"), print_eq=True, num_epochs=10),
    # ---- Lotka-Volterra, 99 % noise (BASELINE config 3 = noise99_eq_isymreg after noise99_sym) ------------
    "lv/noise99_sym.cfg": dict(task="mt_lv", mt_data=True, noise=0.99, **GP, n_comps=2, repr="(2,1,2)", group_idx="0", latent_dim=2,
                               ae_arch="mlp", ortho_ae=True, w_gan=0.01, w_reg_norm=0.01, log_interval=1, save_interval=5,
                               **named("laligan-noise99-lv"), batch_size=8192, batch_norm=True, print_li=True, seed=43,
                               num_epochs=15),
    "lv/noise99_eq_isymreg.cfg": lv_symreg("i", "symreg2-noise99-lv"),
    # named by the reference's run_scripts (lv_noise99_eq_freg.sh / _rreg.sh) but absent from its run_configs:
    # the same run with the finite / reversed regulariser
    "lv/noise99_eq_fsymreg.cfg": lv_symreg("f", "fsymreg-noise99-lv"),
    "lv/noise99_eq_rsymreg.cfg": lv_symreg("r", "rsymreg-noise99-lv"),
    "lv/noise99_eq_sindy_2.cfg": dict(task="lv", noise=0.99, **GP, **LBFGS, lbfgs_subsample=0.01, lr_sindy=0.1, w_sym_reg=0.0,
                                      include_exp=True, st_freq=20, threshold=0.15, **LOG10, **named("sindy2-noise99-lv"),
                                      print_eq=True, num_epochs=100),
    "lv/noise99_eq_wsindy.cfg": dict(task="lv", noise=0.99, **GP, w_sindy_reg=0.0, include_exp=True, threshold=0.15, **LOG10,
                                     **named("wsindy-noise99-lv"), print_eq=True, num_epochs=10),
    # ---- Selkov, 20 % noise (BASELINE config 4 = noise20_eq_sindy swept over seeds) -------------------------
    "selkov/noise20_sym.cfg": dict(task="mt_selkov", mt_data=True, noise=0.2, **GP, n_comps=2, repr="(2,sim2)", group_idx="0",
                                   latent_dim=2, ae_arch="mlp", n_layers=4, hidden_dim=128, ortho_ae=True, lr_ae=3e-4, w_gan=0.2,
                                   w_reg_norm=0.0, gan_st_thres=0.0, log_interval=1, save_interval=5,
                                   **named("laligan-noise20-selkov"), batch_size=8192, batch_norm=True, num_epochs=50),
    "selkov/noise20_eq_sindy.cfg": dict(task="selkov", noise=0.2, **GP, **LBFGS, lbfgs_subsample=0.5, lr_sindy=1.0, w_sym_reg=0.0,
                                        poly_order=3, st_freq=50, threshold=7.5e-2, log_interval=1, save_interval=100,
                                        **named("sindy-noise20-selkov"), num_epochs=200),
    "selkov/noise20_eq_symreg.cfg": selkov_symreg("i", "symreg-noise20-selkov"),
    # run_scripts/selkov_noise20_eq_symreg2.sh / 3.sh name configs the reference does not hold: f and r variants
    "selkov/noise20_eq_symreg2.cfg": selkov_symreg("f", "symreg2-noise20-selkov"),
    "selkov/noise20_eq_symreg3.cfg": selkov_symreg("r", "symreg3-noise20-selkov"),
    "selkov/noise20_eq_wsindy.cfg": dict(task="selkov", noise=0.2, **GP, w_sindy_reg=0.0, poly_order=3, threshold=7.5e-2, **LOG10,
                                         **named("wsindy-noise20-selkov"), print_eq=True, num_epochs=10),
    # ---- reaction-diffusion (BASELINE config 5 = sym_eq) ---------------------------------------------------
    "rd/sym.cfg": dict(n_comps=2, task="mt_rd", repr="(2,1,2)", lr_ae=3e-4, num_epochs=100, batch_size=64, batch_norm=True,
                       w_gan=0.01, w_reg_norm=0.0, w_reg_sim=0.1, log_interval=10, save_dir="laligan-rd", save_interval=10,
                       ortho_ae=True, print_li=True, keep_center=True, gan_st_thres=0.05),
    "rd/sym_eq.cfg": dict(n_comps=2, task="mt_rd", repr="(2,1,2)", lr_ae=3e-4, num_epochs=100, batch_size=64, batch_norm=True,
                          w_gan=0.01, w_reg_norm=0.0, w_reg_sim=0.1, include_sindy=True, eq_constraint=True,
                          constrain_constant=True, w_sindy_z=0.1, w_sindy_x=0.0, log_interval=10, save_dir="laligan-sindy-rd-2",
                          save_interval=10, ortho_ae=True, print_li=True, keep_center=True, gan_st_thres=0.05),
}

# run_scripts/<name>.sh -> (driver module, config): the reference's seed loops, `for i in {0..49}`
SCRIPTS = {
    "dosc_noise20_sindy": ("main", "dosc/noise20_sindy.cfg"),
    "dosc_noise20_esindy": ("main", "dosc/noise20_esindy.cfg"),
    "dosc_noise20_wsindy": ("main_wsindy", "dosc/noise20_wsindy.cfg"),
    "growth_noise05_sindy": ("main", "growth/noise05_sindy.cfg"),
    "growth_noise05_esindy": ("main", "growth/noise05_esindy.cfg"),
    "growth_noise05_wsindy": ("main_wsindy", "growth/noise05_wsindy.cfg"),
    "lv_noise99_eq_ireg": ("main", "lv/noise99_eq_isymreg.cfg"),
    "lv_noise99_eq_freg": ("main", "lv/noise99_eq_fsymreg.cfg"),
    "lv_noise99_eq_rreg": ("main", "lv/noise99_eq_rsymreg.cfg"),
    "lv_noise99_eq_sindy": ("main", "lv/noise99_eq_sindy_2.cfg"),
    "lv_noise99_eq_wsindy": ("main_wsindy", "lv/noise99_eq_wsindy.cfg"),
    "selkov_noise20_eq_sindy": ("main", "selkov/noise20_eq_sindy.cfg"),
    "selkov_noise20_eq_symreg": ("main", "selkov/noise20_eq_symreg.cfg"),
    "selkov_noise20_eq_symreg2": ("main", "selkov/noise20_eq_symreg2.cfg"),
    "selkov_noise20_eq_symreg3": ("main", "selkov/noise20_eq_symreg3.cfg"),
    "selkov_noise20_eq_wsindy": ("main_wsindy", "selkov/noise20_eq_wsindy.cfg"),
}


def _fmt(value):
    if isinstance(value, float):
        text = repr(value)
        return text
    return str(value)


# the parser's sections (parser_utils._MAIN_ARGS is declared in this order): one line of the file per section
SECTIONS = [("task", "smoothing"), ("batch_size", "w_sym_reg"), ("latent_dim", "fix_laligan"), ("ae_arch", "batch_norm"),
            ("repr", "keep_center"), ("use_original_x", "y_embed_dim"), ("include_sindy", "torch_lbfgs"),
            ("pysr_subsample", "pysr_symmreg"), ("gpu", "seed")]


def render(settings):
    """Flags in the parser's declaration order, one line per parser section (the file is split on whitespace, so the
    line structure is free); store_true flags bare."""
    import symode_amd.parser_utils as P
    order = [e[0] for e in P._MAIN_ARGS]
    kinds = {e[0]: e[1] for e in P._MAIN_ARGS}
    unknown = set(settings) - set(order)
    assert not unknown, unknown
    lines = []
    for first, last in SECTIONS:
        toks = []
        for name in order[order.index(first):order.index(last) + 1]:
            if name not in settings:
                continue
            if kinds[name] == P._FLAG:
                assert settings[name] is True
                toks.append(f"--{name}")
            else:
                toks.append(f"--{name} {_fmt(settings[name])}")
        if toks:
            lines.append("  ".join(toks))
    covered = [n for f, l in SECTIONS for n in order[order.index(f):order.index(l) + 1]]
    assert sorted(covered) == sorted(order), "SECTIONS must cover every flag exactly once"
    return "\n".join(lines) + "\n"


def main():
    for rel, settings in CONFIGS.items():
        path = os.path.join(PKG, "run_configs", rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(render(settings))
    sdir = os.path.join(PKG, "run_scripts")
    os.makedirs(sdir, exist_ok=True)
    for name, (module, cfg) in SCRIPTS.items():
        st = CONFIGS[cfg]
        # main_sweep takes the L-BFGS SINDy / EquivSINDy-c configs only (main_sweep.py: no latent, no symmetry regulariser)
        sweepable = (module == "main" and st.get("sindy_optimizer", "lbfgs") == "lbfgs" and not st.get("use_latent", False)
                     and float(st.get("w_sym_reg", 0.0)) == 0.0)
        hint = ("# PYTHONPATH.  The whole loop as ONE process per GPU: python -m symode_amd.main_sweep --config "
                f"{cfg} --n_seeds 50\n") if sweepable else "# PYTHONPATH.\n"
        with open(os.path.join(sdir, name + ".sh"), "w") as f:
            f.write("#!/bin/bash\n"
                    f"# Seeds 0-49 of {cfg}, one process per seed like the reference's script of the same name.\n"
                    "# Run from a directory that holds run_configs/ (this package directory does) with the repository root on\n"
                    + hint +
                    "for i in {0..49}; do\n"
                    '    echo "Running seed $i"\n'
                    f'    python -m symode_amd.{module} --seed "$i" --config {cfg}\n'
                    "done\n")
    print(f"wrote {len(CONFIGS)} configs, {len(SCRIPTS)} scripts")


if __name__ == "__main__":
    main()
