#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE implementation (CPU) in the build container.

Runs only where /root/reference exists (never on the GPU box).  It imports the reference
read-only (``sys.path`` insert, no bytecode written, cwd outside the tree), installs a stub
``wandb`` module (the real one is not installed and the reference only calls
init/log/finish), drives the reference's own functions on seeded inputs and stores
inputs + outputs as small .npz fixtures.  No reference source is copied anywhere.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py [--only f1,f2,...]
"""
import argparse
import io
import os
import sys
import types
import contextlib

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

REF = os.environ.get("SYMODE_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

import numpy as np
import torch

torch.set_num_threads(4)

# ---- stub wandb before the reference's train.py is imported -------------------------------
_wandb_log = []
wandb = types.ModuleType("wandb")
wandb.init = lambda *a, **k: None
wandb.finish = lambda *a, **k: None
wandb.log = lambda d, *a, **k: _wandb_log.append(dict(d))
sys.modules["wandb"] = wandb

sys.path.insert(0, REF)
_cwd = os.getcwd()
os.makedirs("/tmp/symode_golden_cwd", exist_ok=True)
os.chdir("/tmp/symode_golden_cwd")          # reference writes saved_models/ relative to cwd
with contextlib.redirect_stdout(io.StringIO()):
    import sindy as ref_sindy
    import model_utils as ref_mu
    import train as ref_train
    import autoencoder as ref_ae
    import gan as ref_gan
    from evaluation import eval_eq as ref_eval
    from data_utils import damped_oscillator as ref_dosc, selkov as ref_selkov, lotka as ref_lv, growth as ref_growth
    from data_utils import ode as ref_ode
    from data_utils import smoothing as ref_smoothing


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        return fn(*a, **k)


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.1f} KiB)")


def make_regressor(d, order, sine=False, exp=False, L_list=(), threshold=0.05, constrain_constant=False):
    return quiet(ref_sindy.SINDyRegression, d, order, sine, exp, L_list=list(L_list), threshold=threshold,
                 device="cpu", constrain_constant=constrain_constant)


# -------------------------------------------------------------------------------------------
def f1_theta():
    """A1/A2: eval_Theta_at for d in {1,2,3}, order 1..3, sine/exp on/off; get_term_num."""
    g = torch.Generator().manual_seed(101)
    arrays = {}
    cases = []
    for d in (1, 2, 3):
        x = torch.randn(129, d, generator=g) * 1.7
        arrays[f"x_d{d}"] = x
        for order in (1, 2, 3):
            for sine in (False, True):
                for exp in (False, True):
                    r = make_regressor(d, order, sine, exp)
                    key = f"d{d}_o{order}_s{int(sine)}_e{int(exp)}"
                    arrays["theta_" + key] = r.eval_Theta_at(x)
                    arrays["p_" + key] = np.int64(r.get_term_num())
                    cases.append(key)
    # the probe point of SURVEY section 8: x = (2, 3), order 3
    r = make_regressor(2, 3)
    arrays["probe_x"] = torch.tensor([[2.0, 3.0]])
    arrays["probe_theta"] = r.eval_Theta_at(arrays["probe_x"])
    arrays["cases"] = np.array(cases)
    save("f1_theta", **arrays)


def f2_fwd_loss_grad():
    """A3/A6: regressor(x), MSELoss, dloss/dXi, L1 term -- order 3 and order 2 + exp."""
    g = torch.Generator().manual_seed(202)
    arrays = {}
    for tag, (d, order, sine, exp, n) in {"o3": (2, 3, False, False, 4096), "o2e": (2, 2, False, True, 4096),
                                          "d3o2s": (3, 2, True, False, 1024)}.items():
        x = torch.randn(n, d, generator=g)
        dx = torch.randn(n, d, generator=g)
        r = make_regressor(d, order, sine, exp)
        Xi0 = torch.randn(d, r.get_term_num(), generator=g)
        mask = (torch.rand(d, r.get_term_num(), generator=g) > 0.3).float()
        r.Xi.data = Xi0.clone()
        r.mask = mask.clone()
        pred = r(x)
        loss = torch.nn.MSELoss()(pred, dx)
        l1 = sum(torch.norm(p, 1) for p in r.parameters())
        (loss + 0.05 * l1).backward()
        arrays.update({f"{tag}_x": x, f"{tag}_dx": dx, f"{tag}_Xi": Xi0, f"{tag}_mask": mask,
                       f"{tag}_pred": pred, f"{tag}_loss": loss, f"{tag}_l1": l1,
                       f"{tag}_grad_total": r.Xi.grad.clone()})
        r.Xi.grad = None
        torch.nn.MSELoss()(r(x), dx).backward()
        arrays[f"{tag}_grad_mse"] = r.Xi.grad.clone()
        arrays[f"{tag}_cfg"] = np.array([d, order, int(sine), int(exp)])
    save("f2_fwd_loss_grad", **arrays)


def _system_data(name, n_ics, num_steps, sub, dt, noise, seed):
    """Trajectories from the reference's own generators (noise on x, finite-difference dx)."""
    np.random.seed(seed)
    fn = {"dosc": ref_dosc.get_dosc_data, "selkov": ref_selkov.get_selkov_data,
          "lv": ref_lv.get_lv_data, "growth": ref_growth.get_growth_data}[name]
    kw = dict(n_ics=n_ics, num_steps=num_steps, subsample_rate=sub, dt=dt, noise=noise, smoothing=None)
    if name == "growth":
        kw["multiplicative_noise"] = noise > 0
    x, dx = quiet(fn, **kw)
    x = torch.from_numpy(x).float().reshape(-1, x.shape[-1])
    dx = torch.from_numpy(dx).float().reshape(-1, dx.shape[-1])
    return x, dx


def f3_stlsq():
    """A5/A8/A11: solve_SINDy_one_step iterated (train_SINDy loop) on the four systems."""
    arrays, cases = {}, []
    specs = [
        # tag, system, n_ics, steps, sub, dt, noise, order, sine, exp, gamma, thr
        ("dosc_clean", "dosc", 10, 2000, 10, 0.02, 0.0, 3, False, False, 0.0, 0.05),
        ("dosc_noisy", "dosc", 10, 2000, 10, 0.02, 0.02, 3, False, False, 0.05, 0.05),
        ("selkov_clean", "selkov", 10, 2000, 10, 0.002, 0.0, 3, False, False, 0.0, 0.075),
        ("selkov_ridge", "selkov", 10, 2000, 10, 0.002, 0.0, 3, False, False, 0.1, 0.075),
        ("growth_clean", "growth", 20, 1000, 10, 0.002, 0.0, 2, False, False, 0.0, 0.05),
        ("lv_clean", "lv", 20, 2000, 20, 0.002, 0.0, 2, False, True, 0.0, 0.15),
        ("lv_noisy", "lv", 20, 2000, 20, 0.002, 0.001, 2, False, True, 0.05, 0.15),
    ]
    for tag, sysname, n_ics, steps, sub, dt, noise, order, sine, exp, gamma, thr in specs:
        x, dx = _system_data(sysname, n_ics, steps, sub, dt, noise, seed=7)
        if noise > 0:   # drop the last sample of each trajectory (finite-difference dx is stale there)
            T = steps // sub
            keep = torch.ones(x.shape[0], dtype=torch.bool)
            keep[T - 1::T] = False
            x, dx = x[keep], dx[keep]
        r = make_regressor(2, order, sine, exp, threshold=thr)
        masks, xis, conv = [], [], []
        for it in range(8):
            _, c = ref_sindy.solve_SINDy_one_step(r, x, dx, gamma, thr)
            masks.append(r.mask.clone())
            xis.append(r.Xi.detach().clone())
            conv.append(bool(c))
            if c:
                break
        arrays.update({f"{tag}_x": x, f"{tag}_dx": dx, f"{tag}_masks": torch.stack(masks),
                       f"{tag}_xis": torch.stack(xis), f"{tag}_conv": np.array(conv),
                       f"{tag}_cfg": np.array([2, order, int(sine), int(exp)]),
                       f"{tag}_hp": np.array([gamma, thr])})
        cases.append(tag)
        # solve_SINDy (reset mask, <=5 passes)
        r2 = make_regressor(2, order, sine, exp, threshold=thr)
        r2.mask = (torch.rand_like(r2.mask) > 0.5).float()
        ref_sindy.solve_SINDy(r2, x, dx, gamma, thr)
        arrays[f"{tag}_solve_mask"] = r2.mask.clone()
        arrays[f"{tag}_solve_xi"] = r2.Xi.detach().clone()
    arrays["cases"] = np.array(cases)
    save("f3_stlsq", **arrays)


def f4_lbfgs():
    """A6/A7/A12: train_SIGED_lbfgs end to end with injected batch and Xi0."""
    arrays, cases = {}, []
    specs = [
        # tag, system, order, lr, st_freq, thr, epochs, L (or None), noise
        ("dosc_sindy", "dosc", 3, 0.1, 50, 0.05, 60, None, 0.0),
        ("dosc_esindy", "dosc", 2, 1.0, 100, 0.01, 40, "so2", 0.0),
        ("selkov_sindy", "selkov", 3, 1.0, 50, 0.075, 60, None, 0.0),
    ]
    for tag, sysname, order, lr, st_freq, thr, epochs, Lname, noise in specs:
        if sysname == "dosc":
            x, dx = _system_data("dosc", 10, 2000, 10, 0.02, noise, seed=11)
        else:
            x, dx = _system_data("selkov", 10, 2000, 10, 0.002, noise, seed=11)
        torch.manual_seed(5)
        L_list = [torch.tensor([[0.0, 1.0], [-1.0, 0.0]])] if Lname == "so2" else []
        r = make_regressor(2, order, L_list=L_list, threshold=thr)
        init = {k: v.detach().clone() for k, v in r.state_dict().items()}
        _wandb_log.clear()
        identity = torch.nn.Identity()
        quiet(ref_train.train_SIGED_lbfgs, train_loader=[(x, dx)], test_loader=[], num_epochs=epochs, device="cpu",
              log_interval=10 ** 9, save_interval=10 ** 9, save_dir="golden_tmp", autoencoder=identity, generator=identity,
              regressor=r, regressor_dst=None, use_latent=False, distill_latent=False, lr_sindy=lr, w_sindy_z=0.0,
              w_sindy_x=1.0, sindy_reg_type="l1", w_sindy_reg=0.0, sym_reg_type="i", w_sym_reg=0.0, st_freq=st_freq,
              threshold=thr, int_t=0.1, int_dt=0.01, print_eq=False)
        Xi = r.get_Xi() if r.constraint else r.Xi
        truth = ref_eval.sindy_truth[sysname]
        if truth.shape[1] == Xi.shape[1]:
            coef, cf, mse, cf_all, mse_all = ref_eval.eval_sindy_regressor(r, truth)
            arrays.update({f"{tag}_eval_coef": coef, f"{tag}_eval_cf": cf, f"{tag}_eval_mse": mse,
                           f"{tag}_eval_cf_all": np.array(cf_all), f"{tag}_eval_mse_all": np.array(mse_all)})
        arrays.update({f"{tag}_x": x, f"{tag}_dx": dx, f"{tag}_Xi_final": Xi.detach(), f"{tag}_mask_final": r.mask,
                       f"{tag}_loss_hist": np.array([d["loss_sindy_x"] for d in _wandb_log]),
                       f"{tag}_cfg": np.array([2, order]), f"{tag}_hp": np.array([lr, st_freq, thr, epochs])})
        for k, v in init.items():
            arrays[f"{tag}_init_{k}"] = v
        if r.constraint:
            arrays[f"{tag}_Q"] = r.Q
            arrays[f"{tag}_beta_final"] = r.beta.detach()
            arrays[f"{tag}_const_final"] = r.const.detach()
            arrays[f"{tag}_use_kron"] = np.array(r.use_kron_product)
        cases.append(tag)
    arrays["cases"] = np.array(cases)
    save("f4_lbfgs", **arrays)


def f5_constraint():
    """A4/A10: M_list, Q (projector), use_kron_product, get_Xi; constrained STLSQ."""
    g = torch.Generator().manual_seed(505)
    gens = {
        "so2": torch.tensor([[0.0, 1.0], [-1.0, 0.0]]),
        "scaling2": torch.tensor([[2.0, 0.0], [0.0, 1.0]]),
        "sim2": torch.tensor([[-0.2, 1.0], [-1.0, 0.0]]),
        "singular": torch.tensor([[1.0, 2.0], [0.5, 1.0]]),
        "random": torch.randn(2, 2, generator=g),
        "negdet": torch.tensor([[0.0, 1.0], [1.0, 0.0]]),
    }
    arrays, cases = {}, []
    for name, L in gens.items():
        for order in (2, 3):
            for cc in (False, True):
                torch.manual_seed(9)
                r = make_regressor(2, order, L_list=[L], constrain_constant=cc)
                M = quiet(r.get_M_list)[0]
                key = f"{name}_o{order}_cc{int(cc)}"
                arrays[f"{key}_L"] = L
                arrays[f"{key}_M"] = M
                arrays[f"{key}_Q"] = r.Q
                arrays[f"{key}_use_kron"] = np.array(r.use_kron_product)
                arrays[f"{key}_beta"] = r.beta.detach().clone()
                arrays[f"{key}_const"] = r.const.detach().clone()
                arrays[f"{key}_Xi"] = r.get_Xi().detach().clone()
                cases.append(key)
    # 3-D generator (so3 basis element) at order 2
    L3 = torch.zeros(3, 3)
    L3[1, 0], L3[0, 1] = 1.0, -1.0
    torch.manual_seed(9)
    r = make_regressor(3, 2, L_list=[L3], constrain_constant=False)
    arrays["so3a_o2_L"], arrays["so3a_o2_M"], arrays["so3a_o2_Q"] = L3, quiet(r.get_M_list)[0], r.Q
    arrays["so3a_o2_use_kron"] = np.array(r.use_kron_product)
    # two generators at once (so2 + isotropic scaling)
    torch.manual_seed(9)
    Ls = [gens["so2"], torch.eye(2)]
    r = make_regressor(2, 3, L_list=Ls, constrain_constant=True)
    arrays["pair_o3_L"] = torch.stack(Ls)
    arrays["pair_o3_Q"] = r.Q
    arrays["pair_o3_use_kron"] = np.array(r.use_kron_product)

    # constrained solve_SINDy on clean damped-oscillator data (so2) and growth data (scaling2)
    for tag, sysname, Lname, order, cc, gamma, thr in [
        ("solve_dosc_so2", "dosc", "so2", 2, False, 0.0, 0.01),
        ("solve_dosc_so2_o3_cc", "dosc", "so2", 3, True, 0.05, 1e-9),
        ("solve_growth_scaling2", "growth", "scaling2", 2, True, 0.0, 0.05),
        ("solve_growth_scaling2_ac", "growth", "scaling2", 2, False, 0.0, 0.05),
    ]:
        if sysname == "dosc":
            x, dx = _system_data("dosc", 10, 2000, 10, 0.02, 0.0, seed=13)
        else:
            x, dx = _system_data("growth", 20, 1000, 10, 0.002, 0.0, seed=13)
        torch.manual_seed(3)
        r = make_regressor(2, order, L_list=[gens[Lname]], threshold=thr, constrain_constant=cc)
        masks, xis, conv = [], [], []
        for it in range(6):
            _, c = ref_sindy.solve_SINDy_one_step(r, x, dx, gamma, thr)
            masks.append(r.mask.clone())
            xis.append(r.get_Xi().detach().clone())
            conv.append(bool(c))
            if c:
                break
        arrays.update({f"{tag}_x": x, f"{tag}_dx": dx, f"{tag}_L": gens[Lname], f"{tag}_Q": r.Q,
                       f"{tag}_use_kron": np.array(r.use_kron_product),
                       f"{tag}_masks": torch.stack(masks), f"{tag}_xis": torch.stack(xis), f"{tag}_conv": np.array(conv),
                       f"{tag}_cfg": np.array([2, order, int(cc)]), f"{tag}_hp": np.array([gamma, thr])})
    # Rank-deficient constrained system (so2, order 3, support = linear terms): three of the four
    # columns of A @ Q[mask] are collinear.  torch 2.10's CPU gelsy path is NOT reproducible on
    # it (identical calls return rank 1 or rank 2), so the fixture stores the explicit system,
    # the outcomes of 12 repeated reference-driver calls and the SVD-driver (gelsd) solution.
    x, dx = _system_data("dosc", 10, 2000, 10, 0.02, 0.0, seed=13)
    torch.manual_seed(3)
    r = make_regressor(2, 3, L_list=[gens["so2"]], threshold=0.01, constrain_constant=True)
    th = r.eval_Theta_at(x)
    A = torch.cat([th, 0.05 * torch.eye(10)], 0)
    B = torch.cat([dx, torch.zeros(10, 2)], 0)
    m = torch.zeros(2, 10, dtype=torch.bool)
    m[:, 1] = True
    m[:, 2] = True
    AQ = torch.block_diag(A, A)[:, m.flatten()] @ r.Q[m.flatten()]
    Bf = B.T.reshape(-1)
    outs = []
    for i in range(12):
        _junk = torch.randn(1000 + 37 * i)
        lm = torch.linalg.lstsq(AQ, Bf)
        outs.append(np.concatenate([[float(lm.rank)], lm.solution.numpy()]))
    arrays["rankdef_G"] = AQ.double().T @ AQ.double()
    arrays["rankdef_C"] = AQ.double().T @ Bf.double()
    arrays["rankdef_rows"] = np.array(AQ.shape[0])
    arrays["rankdef_gelsy_outcomes"] = np.stack(outs)
    arrays["rankdef_gelsd"] = torch.linalg.lstsq(AQ, Bf, driver="gelsd").solution
    arrays["cases"] = np.array(cases)
    save("f5_constraint", **arrays)


def _tiny_ae(activation, seed, hidden=16, n_layers=2, n_comps=2):
    torch.manual_seed(seed)
    ae = ref_ae.AutoEncoder(ae_arch="mlp", input_dim=2, hidden_dim=hidden, latent_dim=2, n_layers=n_layers,
                            n_comps=n_comps, activation=activation, activation_args=[], batch_norm=True, ortho_ae=False)
    # non-trivial eval-mode batch-norm statistics
    with torch.no_grad():
        for m in ae.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.3)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.7, 1.3)
                m.bias.normal_(0, 0.2)
    ae.eval()
    return ae


def _ae_arrays(ae, prefix):
    """Flatten encoder/decoder into (Linear, BatchNorm-eval) arrays a test can rebuild."""
    out, i = {}, 0
    def walk(seq, tag):
        nonlocal i
        k = 0
        for m in seq.modules():
            if isinstance(m, torch.nn.Linear):
                out[f"{prefix}_{tag}_W{k}"], out[f"{prefix}_{tag}_b{k}"] = m.weight.detach(), m.bias.detach()
                k += 1
            elif isinstance(m, torch.nn.BatchNorm1d):
                j = k - 1
                out[f"{prefix}_{tag}_bnw{j}"], out[f"{prefix}_{tag}_bnb{j}"] = m.weight.detach(), m.bias.detach()
                out[f"{prefix}_{tag}_bnm{j}"], out[f"{prefix}_{tag}_bnv{j}"] = m.running_mean.detach(), m.running_var.detach()
                out[f"{prefix}_{tag}_bneps{j}"] = np.array(m.eps)
        out[f"{prefix}_{tag}_n"] = np.array(k)
    walk(ae.encoder, "enc")
    walk(ae.decoder, "dec")
    return out


def _generator(repr_str):
    return ref_gan.LieGenerator(repr=repr_str, group_idx="0", uniform_max=1, coef_dist="normal", g_init="random", task="lv",
                                sigma_init=1, int_param=False, int_param_noise=0.1, int_param_max=2, gan_st_thres=0.3,
                                keep_center=True, device="cpu")


def f6_symreg():
    """S1-S4, A9: symmetry losses and their dXi with a tiny frozen AE + generator; odeint."""
    arrays = {}
    g = torch.Generator().manual_seed(606)
    for tag, act, repr_str, order, exp, K, dt in [("relu_sim2", "ReLU", "(2,sim2)", 3, False, 3, 0.01),
                                                  ("tanh_learn", "Tanh", "(2,1,2)", 2, True, 10, 0.01)]:
        ae = _tiny_ae(act, seed=17)
        torch.manual_seed(23)
        gen = _generator(repr_str)
        gen.eval()
        for p_ in list(ae.parameters()) + list(gen.parameters()):
            p_.requires_grad = False
        arrays.update(_ae_arrays(ae, tag))
        basis = gen.get_full_basis_list()
        gel1 = gen.get_deterministic_group_elems()
        gel001 = gen.get_deterministic_group_elems(scale=0.01)
        arrays[f"{tag}_basis"] = torch.stack(basis)
        arrays[f"{tag}_gelems"] = torch.stack([e.reshape(e.shape[-2:]) for e in gel1])
        arrays[f"{tag}_gelems_r"] = torch.stack([e.reshape(e.shape[-2:]) for e in gel001])
        arrays[f"{tag}_Li"] = gen.Li[0].detach()
        arrays[f"{tag}_sigma"] = gen.sigma[0].detach()
        arrays[f"{tag}_zmean"] = ae.encoder[-2].bias.detach()

        x = torch.randn(512, 2, generator=g) * 0.6
        r = make_regressor(2, order, False, exp)
        Xi0 = torch.randn(2, r.get_term_num(), generator=g) * 0.3
        mask = torch.ones_like(Xi0)
        mask[0, 3] = 0.0
        mask[1, 1] = 0.0
        r.Xi.data = Xi0.clone()
        r.mask = mask.clone()
        arrays.update({f"{tag}_x": x, f"{tag}_Xi": Xi0, f"{tag}_mask": mask,
                       f"{tag}_cfg": np.array([2, order, 0, int(exp)]), f"{tag}_K": np.array(K), f"{tag}_dt": np.array(dt)})

        def fstep(xx):
            return ref_mu.odeint(r, xx, K * dt + 1e-9, dt)
        with torch.no_grad():
            arrays[f"{tag}_euler"] = ref_mu.odeint(r, x, K * dt + 1e-9, dt)
            arrays[f"{tag}_euler_traj"] = ref_mu.odeint(r, x, K * dt + 1e-9, dt, full_traj=True)
            arrays[f"{tag}_rk4"] = ref_mu.odeint(r, x, K * dt + 1e-9, dt, method="rk4")

        def run(loss_fn):
            r.Xi.grad = None
            loss = loss_fn()
            loss.backward()
            return loss.detach().clone(), r.Xi.grad.detach().clone()

        # S2 infinitesimal (relative and absolute)
        def s2(relative):
            fx = fstep(x)
            x_fx = torch.stack([x, fx], dim=1)
            return ref_mu.symmreg_i(x_fx, ae, gen, f=fstep, relative=relative, require_grad=True)
        arrays[f"{tag}_s2_loss"], arrays[f"{tag}_s2_grad"] = run(lambda: s2(True))
        arrays[f"{tag}_s2abs_loss"], arrays[f"{tag}_s2abs_grad"] = run(lambda: s2(False))

        # S3 finite
        def s3(relative):
            fx = fstep(x)
            x_fx = torch.stack([x, fx], dim=1)
            return ref_mu.symmreg_f(x_fx, ae, gen, f=fstep, relative=relative, require_grad=True)
        arrays[f"{tag}_s3_loss"], arrays[f"{tag}_s3_grad"] = run(lambda: s3(True))

        # S4 reversed
        arrays[f"{tag}_s4_loss"], arrays[f"{tag}_s4_grad"] = run(
            lambda: ref_mu.symmreg_r(x, ae, gen, h=r, require_grad=True))
        gx_list, Jgx_list = ref_mu.precompute_symmreg_r(x, ae, gen)
        arrays[f"{tag}_s4_gx"] = torch.stack(gx_list)
        arrays[f"{tag}_s4_Jgx"] = torch.stack([J.reshape(J.shape[0], 2, 2) for J in Jgx_list])

        # S1 linear latent (train.py:502-507 with the [1] the shipped line forgets)
        from torch.autograd.functional import jvp
        Ls = [b[:2, :2] for b in basis]
        def s1():
            z = x
            dz_pred = r(z)
            loss = 0.0
            for v in Ls:
                loss = loss + torch.norm(jvp(r, z, torch.einsum('ij, bj->bi', v, z), create_graph=True)[1]
                                         - torch.einsum('ij, bj->bi', v, dz_pred)) ** 2
            return loss
        arrays[f"{tag}_s1_L"] = torch.stack(Ls)
        arrays[f"{tag}_s1_loss"], arrays[f"{tag}_s1_grad"] = run(s1)
    save("f6_symreg", **arrays)


def f7_wsindy():
    """N1 (next row): WSINDyWrapper.solve on one clean trajectory."""
    x, _ = _system_data("dosc", 1, 2000, 1, 0.02, 0.0, seed=21)
    n = x.shape[0]
    t = torch.arange(n) * 0.02
    r = make_regressor(2, 3, threshold=0.05)
    w = ref_sindy.WSINDyWrapper(r, t, n * 0.02, device="cpu")
    masks, xis, conv, res = [], [], [], []
    for it in range(6):
        rs, c = w.solve(x, 0.0, 0.05)
        masks.append(r.mask.clone()); xis.append(r.Xi.detach().clone()); conv.append(bool(c)); res.append(rs)
        if c:
            break
    save("f7_wsindy", x=x, V_head=w.V[:, :48], V_drv_head=w.V_drv[:, :48], masks=torch.stack(masks), xis=torch.stack(xis),
         conv=np.array(conv), tmax=np.array(n * 0.02))


def f8_known_answers():
    """A12 + synthetic-data generator: truth tables, RHS at fixed points, a short RK4 run."""
    rng = np.random.RandomState(808)
    pts = rng.uniform(-1.5, 1.5, (64, 2))
    arrays = {"pts": pts}
    for name, fn in [("dosc", ref_dosc.dosc), ("selkov", ref_selkov.selkov), ("lv", ref_lv.lotka_volterra), ("growth", ref_growth.growth)]:
        arrays[f"rhs_{name}"] = fn(pts)
        arrays[f"truth_{name}"] = ref_eval.sindy_truth[name]
    x0 = rng.uniform(0.5, 1.0, (5, 2))
    xs, dxs = quiet(ref_ode.solve_ode_batch, ref_selkov.selkov, x0, dt=0.002, num_steps=200)
    arrays["rk4_x0"], arrays["rk4_x"], arrays["rk4_dx"] = x0, np.transpose(xs, (1, 0, 2)), np.transpose(dxs, (1, 0, 2))
    # eval_sindy_regressor on a hand-made regressor
    r = make_regressor(2, 2)
    r.Xi.data = torch.tensor([[0.01, -0.11, -0.98, 0.0, 0.2, 0.0], [0.0, 1.02, -0.1, 0.0, 0.0, 0.0]])
    r.mask = torch.tensor([[0., 1, 1, 0, 1, 0], [0., 1, 1, 0, 0, 0]])
    coef, cf, mse, cf_all, mse_all = ref_eval.eval_sindy_regressor(r, ref_eval.sindy_truth["dosc"])
    arrays.update({"eval_Xi": r.Xi.detach(), "eval_mask": r.mask, "eval_coef": coef, "eval_cf": cf, "eval_mse": mse,
                   "eval_cf_all": np.array(cf_all), "eval_mse_all": np.array(mse_all)})
    save("f8_known_answers", **arrays)


def _flat_state(module, prefix):
    return {f"{prefix}/{k}": v.detach().clone() for k, v in module.state_dict().items()}


def f9_lassi():
    """N2 / BASELINE config 5: two epochs of the reference's train_lassi (Adam branch of the latent SINDy model,
    w_sindy_x > 0 -- its lstsq branch returns NaN on the CPU, see f3) on a tiny multi-timestep field; initial and
    final state_dicts of every module, the data and the logged epoch means.  Two variants: 'plain' (no batch norm:
    every quantity is well conditioned) and 'bn' (batch norm as in rd/sym_eq.cfg: the biases in front of a BatchNorm
    have zero true gradient, Adam turns their rounding noise into +-lr steps, so they and the eval-mode statistics
    that depend on them are reproducible only loosely)."""
    rng = np.random.RandomState(909)
    T, N, dt, om = 40, 30, 0.05, 1.3
    A, B = rng.randn(N), rng.randn(N)
    t = np.arange(T + 1) * dt
    field = np.outer(np.cos(om * t), A) + np.outer(np.sin(om * t), B) + 0.1 * np.outer(np.cos(2 * om * t), A * B)
    dfield = om * (-np.outer(np.sin(om * t), A) + np.outer(np.cos(om * t), B)) - 0.2 * om * np.outer(np.sin(2 * om * t), A * B)
    x = torch.tensor(np.stack([field[:-1], field[1:]], 1), dtype=torch.float32)        # (T, 2, N)
    dx = torch.tensor(np.stack([dfield[:-1], dfield[1:]], 1), dtype=torch.float32)
    ds_train = torch.utils.data.TensorDataset(x[:32], dx[:32])
    ds_val = torch.utils.data.TensorDataset(x[32:], dx[32:])
    arrays = {"x": x, "dx": dx, "n_train": np.array(32)}
    for tag, bn in (("plain", False), ("bn", True)):
        args = dict(ae_arch="mlp", input_dim=N, hidden_dim=16, latent_dim=2, n_layers=2, n_comps=2, activation="ReLU",
                    activation_args=[], batch_norm=bn, ortho_ae=False,
                    repr="(2,1,2)", group_idx="0", uniform_max=1, coef_dist="normal", g_init="random", task="mt_rd", sigma_init=1,
                    int_param=False, int_param_noise=0.1, int_param_max=2, gan_st_thres=0.05, keep_center=True, device="cpu",
                    use_original_x=False, use_invariant_y=False)
        torch.manual_seed(9)
        ae = ref_ae.AutoEncoder(**args)
        disc = ref_gan.Discriminator(**args)
        gen = ref_gan.LieGenerator(**args)
        reg = make_regressor(2, 2, threshold=0.1)
        for name, m in (("ae", ae), ("disc", disc), ("gen", gen), ("reg", reg)):
            arrays.update(_flat_state(m, f"{tag}/init_{name}"))
        torch.manual_seed(10)
        tl = torch.utils.data.DataLoader(ds_train, batch_size=8, shuffle=True)
        vl = torch.utils.data.DataLoader(ds_val, batch_size=8, shuffle=False)
        _wandb_log.clear()
        quiet(ref_train.train_lassi, ae, disc, gen, tl, vl, num_epochs=2, lr_ae=1e-3, lr_d=2e-3, lr_g=1e-2, w_recon=1.0, w_gan=0.01,
              w_reg_norm=0.0, w_reg_sim=0.1, w_reg_ortho=0.05, w_reg_closure=0.0, use_original_x=False, gan_st_freq=2,
              gan_st_thres=0.05, ae_arch="mlp", include_sindy=True, regressor=reg, lr_sindy=1e-3, w_sindy_z=0.1, w_sindy_x=0.5,
              sindy_reg_type="l1", w_sindy_reg=1e-3, st_freq=1, threshold=0.1, device="cpu", log_interval=1,
              save_interval=1000, save_dir="golden-lassi", n_comps=2, print_li=False)
        for name, m in (("ae", ae), ("disc", disc), ("gen", gen), ("reg", reg)):
            arrays.update(_flat_state(m, f"{tag}/final_{name}"))
        arrays[f"{tag}/final_reg_mask"] = reg.mask
        arrays[f"{tag}/final_gen_mask"] = gen.masks[0]
        keys = sorted(_wandb_log[0].keys())
        arrays["log_keys"] = np.array(keys)
        arrays[f"{tag}/log_values"] = np.array([[float(e[k]) for k in keys] for e in _wandb_log])
    save("f9_lassi", **arrays)


def f10_gp_smoothing():
    """N4: ``num_diff_gp`` (data_utils/smoothing.py:155-196) on short noisy damped-oscillator and Lotka-Volterra series."""
    arrays = {}
    for tag, fn, init, T, dt, noise, sig_in in (("dosc", ref_dosc.dosc, ref_dosc.generate_random_ics, 150, 0.02, 0.2, 0.1),
                                                ("lv", ref_lv.lotka_volterra, None, 120, 0.01, 0.5, 0.05)):
        np.random.seed(1010)
        x0 = init(5) if init is not None else np.random.uniform(-0.5, 0.5, (5, 2))
        x, _ = quiet(ref_ode.solve_ode_batch, fn, x0, dt=dt, num_steps=T)
        std = np.std(x, axis=(0, 1))
        x = x + np.random.randn(*x.shape) * noise * std
        dX, X = quiet(ref_smoothing.num_diff_gp, x.copy(), dt, noise_level=noise, std_base=std, sigma_in=sig_in)
        arrays.update({f"{tag}_x": x, f"{tag}_std": std, f"{tag}_dt": np.array(dt), f"{tag}_noise": np.array(noise),
                       f"{tag}_sigma_in": np.array(sig_in), f"{tag}_dX": dX, f"{tag}_X": X})
    save("f10_gp_smoothing", **arrays)


def _noisy_gp_data(name, n_ics, num_steps, dt, noise, sigma_in, seed):
    """The reference's own data recipe (data_utils/ode.py:30-49) with noise AND Gaussian-process smoothing, on series
    short enough for the O(T^3) smoother: x, dx flattened as dataset.py:193-194 does, float32 like dataset.py:188-189."""
    np.random.seed(seed)
    fn = {"dosc": ref_dosc.get_dosc_data, "selkov": ref_selkov.get_selkov_data, "growth": ref_growth.get_growth_data}[name]
    kw = dict(n_ics=n_ics, num_steps=num_steps, subsample_rate=1, dt=dt, noise=noise, smoothing="gp", gp_sigma_in=sigma_in)
    if name == "growth":
        kw["multiplicative_noise"] = True                      # data_utils/growth.py:48
    x, dx = quiet(fn, **kw)
    x = torch.from_numpy(x).to(torch.float32).reshape(-1, x.shape[-1])
    dx = torch.from_numpy(dx).to(torch.float32).reshape(-1, dx.shape[-1])
    return x, dx


def _f11_run(x, dx, order, lr, st_freq, thr, epochs, L_list, cc, perturb=None):
    """One reference run of train_SIGED_lbfgs from the seeded start (``perturb``: a generator -- every start parameter
    is multiplied by 1 + u, |u| <= 1.2e-7, i.e. moved by about one unit in the last place)."""
    torch.manual_seed(6)
    r = make_regressor(2, order, L_list=L_list, threshold=thr, constrain_constant=cc)
    if perturb is not None:
        with torch.no_grad():
            for p_ in r.parameters():
                p_.mul_(1 + (torch.rand(p_.shape, generator=perturb) - 0.5) * 2.4e-7)
    init = {k: v.detach().clone() for k, v in r.state_dict().items()}
    _wandb_log.clear()
    thresholded = []                               # (epoch index of the log, Xi, mask before) at every set_threshold call
    plain_set = r.set_threshold

    def noting_set(threshold, _r=r, _plain=plain_set, _rec=thresholded):
        Xi = _r.get_Xi() if _r.constraint else _r.Xi
        _rec.append((len(_wandb_log), Xi.detach().clone(), _r.mask.clone()))
        return _plain(threshold)
    r.set_threshold = noting_set
    trace = []                                     # (Xi, mask) at every closure evaluation of the reference's run
    hook = r.register_forward_pre_hook(lambda m, inp, _t=trace: _t.append(
        ((m.get_Xi() if m.constraint else m.Xi).detach().clone(), m.mask.clone())))
    identity = torch.nn.Identity()
    quiet(ref_train.train_SIGED_lbfgs, train_loader=[(x, dx)], test_loader=[], num_epochs=epochs, device="cpu",
          log_interval=10 ** 9, save_interval=10 ** 9, save_dir="golden_tmp", autoencoder=identity, generator=identity,
          regressor=r, regressor_dst=None, use_latent=False, distill_latent=False, lr_sindy=lr, w_sindy_z=0.0,
          w_sindy_x=1.0, sindy_reg_type="l1", w_sindy_reg=0.0, sym_reg_type="i", w_sym_reg=0.0, st_freq=st_freq,
          threshold=thr, int_t=0.1, int_dt=0.01, print_eq=False)
    hook.remove()
    return r, init, [d["loss_sindy_x"] for d in _wandb_log], thresholded, trace


def f11_lbfgs_noisy():
    """A6/A7/A12 at the reference's real operating point: train_SIGED_lbfgs on NOISY, GP-smoothed data with the
    hyper-parameters of the shipped configs (run_configs/dosc/noise20_sindy.cfg at order 2 and 3,
    selkov/noise20_eq_sindy.cfg, dosc/noise20_esindy.cfg, growth/noise05_esindy.cfg, growth/noise05_sindy.cfg), injected
    batch and start.  Every record holds the per-epoch ``loss_sindy_x`` log, the coefficients and mask at every closure
    evaluation (a forward hook on the reference's regressor), the coefficients going into every thresholding event (a
    wrapper around its set_threshold) and the result -- and the outcome of 12 more runs of the reference from starts
    moved by about ONE UNIT IN THE LAST PLACE: SURVEY H5 made measurable.  Without a line search the epoch at which the
    update norm crosses 1e-3 is a matter of last bits (dosc: the first event between epoch 5 and 8, masks identical),
    and on selkov (cond 1e4, lr 1.0) so is the mask itself."""
    arrays, cases = {}, []
    gens = {"so2": torch.tensor([[0.0, 1.0], [-1.0, 0.0]]), "scaling2": torch.tensor([[2.0, 0.0], [0.0, 1.0]])}
    specs = [
        # tag, system, n_ics, steps, dt, noise, sigma_in, order, lr, st_freq, thr, epochs, L, constrain_constant
        ("dosc_n20_o3", "dosc", 8, 400, 0.02, 0.2, 0.1, 3, 0.1, 50, 0.05, 200, None, False),
        ("dosc_n20_o2", "dosc", 8, 400, 0.02, 0.2, 0.1, 2, 0.1, 50, 0.05, 200, None, False),
        ("selkov_n20_o3", "selkov", 8, 400, 0.02, 0.2, 0.1, 3, 1.0, 50, 0.075, 200, None, False),
        ("dosc_n20_so2", "dosc", 8, 400, 0.02, 0.2, 0.1, 2, 1.0, 100, 0.01, 100, "so2", False),
        ("growth_n05_scaling2", "growth", 16, 200, 0.01, 0.05, 0.05, 2, 1.0, 100, 0.05, 100, "scaling2", True),
        ("growth_n05_o2", "growth", 16, 200, 0.01, 0.05, 0.05, 2, 1.0, 50, 0.05, 200, None, False),
    ]
    # an EDGE case on purpose: dosc_n20_o3 again with the threshold moved to 5e-5 below
    # the smallest coefficient that survives the first thresholding event of that run -- the event happens at the same
    # epoch with the same coefficients (the threshold enters nowhere before it), so the reference's own record holds a
    # coefficient inside the 1e-4 near-threshold band.  The spec is appended once the base run is known.
    todo = list(specs)
    while todo:
        tag, sysname, n_ics, steps, dt, noise, sig_in, order, lr, st_freq, thr, epochs, Lname, cc = todo.pop(0)
        x, dx = _noisy_gp_data(sysname, n_ics, steps, dt, noise, sig_in, seed=1111)
        L_list = [gens[Lname]] if Lname else []
        r, init, loss_hist, thresholded, trace = _f11_run(x, dx, order, lr, st_freq, thr, epochs, L_list, cc)
        Xi = r.get_Xi() if r.constraint else r.Xi
        truth = ref_eval.sindy_truth[sysname]
        if truth.shape[1] == Xi.shape[1]:
            coef, cf, mse, cf_all, mse_all = ref_eval.eval_sindy_regressor(r, truth)
            arrays.update({f"{tag}_eval_coef": coef, f"{tag}_eval_cf": cf, f"{tag}_eval_mse": mse,
                           f"{tag}_eval_cf_all": np.array(cf_all), f"{tag}_eval_mse_all": np.array(mse_all)})
        p = Xi.shape[1]
        arrays.update({f"{tag}_x": x, f"{tag}_dx": dx, f"{tag}_Xi_final": Xi.detach(), f"{tag}_mask_final": r.mask,
                       f"{tag}_loss_hist": np.array(loss_hist),
                       f"{tag}_trace_Xi": torch.stack([a for a, _ in trace]), f"{tag}_trace_mask": torch.stack([m for _, m in trace]),
                       f"{tag}_cfg": np.array([2, order, int(cc)]), f"{tag}_hp": np.array([lr, st_freq, thr, epochs]),
                       f"{tag}_thr_epoch": np.array([e for e, _, _ in thresholded], dtype=np.int64),
                       f"{tag}_thr_Xi": torch.stack([a for _, a, _ in thresholded]) if thresholded else np.zeros((0, 2, p), np.float32),
                       f"{tag}_thr_mask_before": torch.stack([m for _, _, m in thresholded]) if thresholded else np.zeros((0, 2, p), np.float32)})
        for k, v in init.items():
            arrays[f"{tag}_init_{k}"] = v
        if r.constraint:
            arrays[f"{tag}_L"] = gens[Lname]
            arrays[f"{tag}_Q"] = r.Q
            arrays[f"{tag}_beta_final"] = r.beta.detach()
            arrays[f"{tag}_const_final"] = r.const.detach()
            arrays[f"{tag}_use_kron"] = np.array(r.use_kron_product)
        # the reference again from 12 starts one unit in the last place away
        gp = torch.Generator().manual_seed(1212)
        pm, pe, pl, pn, pf = [], [], [], [], []
        for k in range(12):
            r2, _, lh2, th2, _ = _f11_run(x, dx, order, lr, st_freq, thr, epochs, L_list, cc, perturb=gp)
            pm.append(r2.mask.clone())
            ev = [e for e, _, _ in th2][:8]
            pe.append(ev + [-1] * (8 - len(ev)))
            pl.append(lh2[-1])
            pf.append(lh2[0])
            pn.append(len(lh2))
        arrays.update({f"{tag}_ulp_masks": torch.stack(pm), f"{tag}_ulp_thr_epoch": np.array(pe, dtype=np.int64),
                       f"{tag}_ulp_last_loss": np.array(pl), f"{tag}_ulp_first_loss": np.array(pf),
                       f"{tag}_ulp_logged_epochs": np.array(pn, dtype=np.int64)})
        cases.append(tag)
        if tag == "dosc_n20_o3":
            a0, m0 = thresholded[0][1].abs(), thresholded[0][2]
            alive = a0[(m0 > 0) & (a0 > thr)]
            thr_edge = float(np.float32(alive.min().item()) - np.float32(5e-5))
            todo.append(("dosc_n20_o3_edge", sysname, n_ics, steps, dt, noise, sig_in, order, lr, st_freq, thr_edge, epochs, Lname, cc))
        stable = int((torch.stack(pm) == r.mask).all(dim=0).all())
        print(f"  {tag}: {x.shape[0]} points, {len(loss_hist)} logged epochs, {len(thresholded)} thresholding events, "
              f"mask {r.mask.int().tolist()}; 1-ulp starts: logged epochs {sorted(set(pn))}, "
              f"{sum(int(torch.equal(m, r.mask)) for m in pm)}/12 reach the same mask")
    arrays["cases"] = np.array(cases)
    save("f11_lbfgs_noisy", **arrays)


ALL = {"f1": f1_theta, "f2": f2_fwd_loss_grad, "f3": f3_stlsq, "f4": f4_lbfgs, "f5": f5_constraint,
       "f6": f6_symreg, "f7": f7_wsindy, "f8": f8_known_answers, "f9": f9_lassi, "f10": f10_gp_smoothing,
       "f11": f11_lbfgs_noisy}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    names = [s for s in a.only.split(",") if s] or list(ALL)
    for n in names:
        ALL[n]()
