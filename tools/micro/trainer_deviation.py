import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import symode_amd as S
from oracle import sindy_oracle as O
DEV='cuda'
ident = torch.nn.Identity()
def _lbfgs(r, x, dx, lr, st_freq, thr, epochs):
    S.train.train_SIGED_lbfgs(train_loader=[(x, dx)], test_loader=[], num_epochs=epochs, device=DEV, log_interval=10 ** 9,
                              save_interval=10 ** 9, save_dir="bl", autoencoder=ident, generator=ident, regressor=r,
                              regressor_dst=None, use_latent=False, distill_latent=False, lr_sindy=lr, w_sindy_z=0.0,
                              w_sindy_x=1.0, sindy_reg_type="l1", w_sindy_reg=0.0, sym_reg_type="i", w_sym_reg=0.0,
                              st_freq=st_freq, threshold=thr, int_t=0.1, int_dt=0.01, print_eq=False)
os.makedirs('/tmp/devprobe', exist_ok=True); os.chdir('/tmp/devprobe')
for noise in (0.0, 0.05, 0.2):
    x, dx = S.data.make_dataset("dosc", 50, 2500, dt=0.02, noise=noise, seed=0, device=DEV); x, dx = x[0], dx[0]
    torch.manual_seed(0); Xi0 = torch.randn(2, 10)
    r = S.SINDyRegression(2, 3, False, False, threshold=0.05, device=DEV); r.Xi.data = Xi0.to(DEV)
    _lbfgs(r, x, dx, 0.1, 50, 0.05, 60)
    reg = O.OracleRegressor(2, 3, threshold=0.05, Xi0=Xi0)
    O.lbfgs_fit(reg, x.cpu(), dx.cpu(), 60, 0.1, st_freq=50, threshold=0.05)
    got = (r.Xi * r.mask).detach().cpu().numpy(); want = (reg.Xi * reg.mask).detach().numpy()
    live = want != 0
    print(f"config0 noise {noise}: max abs diff {np.abs(got-want).max():.2e}, max rel diff on live coefficients {np.abs((got-want)[live]/want[live]).max():.2e}, mask equal {np.array_equal(got!=0, want!=0)}")
    so2 = torch.tensor([[0.0, 1.0], [-1.0, 0.0]])
    torch.manual_seed(1)
    r = S.SINDyRegression(2, 5, False, False, L_list=[so2], threshold=0.01, device=DEV, constrain_constant=False)
    reg = O.OracleRegressor(2, 5, L_list=[so2], threshold=0.01, beta0=r.beta.detach().cpu(), const0=r.const.detach().cpu())
    reg.Q = r.Q.cpu()
    _lbfgs(r, x, dx, 1.0, 100, 0.01, 60)
    O.lbfgs_fit(reg, x.cpu(), dx.cpu(), 60, 1.0, st_freq=100, threshold=0.01)
    got, want = (r.get_Xi() * r.mask).detach().cpu().numpy(), (reg.get_Xi() * reg.mask).detach().numpy()
    live = np.abs(want) > 1e-6
    print(f"config1 noise {noise}: max abs diff {np.abs(got-want).max():.2e}, max rel diff on live coefficients {np.abs((got-want)[live]/want[live]).max():.2e}")
