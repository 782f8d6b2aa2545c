"""SURVEY H4 / VERDICT r2 #7: would an fp32 Gram (fp32 products and fp32 sums inside blocks of B points -- what v_mfma_f32_32x32x2_f32\nper-block partials give --, fp64 only across blocks) keep the normal-equation coefficients within rtol 1e-5 of an fp64 QR solve?\nCPU experiment on the BASELINE data sets (no GPU needed): python tools/micro/h4_check.py"""
import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import sindy_oracle as O
import symode_amd as S
torch.manual_seed(0)
def run(name, n_ics, steps, dt, order, noise):
    x, dx = S.data.make_dataset(name, n_ics, steps, dt=dt, noise=noise, seed=0, device='cpu')
    x, dx = x[0], dx[0]
    th = O.theta(x, order)                      # fp32 features, as the reference builds them
    A = torch.cat([th, dx], 1)
    A64 = A.double().numpy()
    G64 = A64.T @ A64                           # what the product computes (exact products, fp64 sums)
    # H4: fp32 products, fp32 sums inside blocks of B points, fp64 across blocks
    out = {}
    for B in (64, 256, 1024):
        n = (A.shape[0] // B) * B
        blocks = A[:n].reshape(-1, B, A.shape[1])
        Gb = torch.einsum('nbi,nbj->nij', blocks, blocks)          # fp32 matmul per block
        G32 = Gb.double().sum(0).numpy() + (A64[n:].T @ A64[n:])
        out[B] = G32
    p = th.shape[1]
    def solve(G):
        return np.linalg.solve(G[:p, :p], G[:p, p:])               # normal equations, full mask
    ref = np.linalg.lstsq(A64[:, :p], A64[:, p:], rcond=None)[0]   # fp64 QR/SVD on the fp32 features
    e64 = np.abs(solve(G64) - ref).max() / np.abs(ref).max()
    cond = np.linalg.cond(A64[:, :p])
    msg = f"{name} order {order} noise {noise}: cond(Theta) {cond:.2e}; coefficients vs fp64 lstsq: fp64 Gram {e64:.1e}"
    for B, G in out.items():
        e = np.abs(solve(G) - ref).max() / np.abs(ref).max()
        g = np.abs(G - G64).max() / np.abs(G64).max()
        msg += f" | fp32 blocks of {B}: Gram {g:.1e}, coefficients {e:.1e}"
    print(msg)
run('dosc', 50, 2500, 0.02, 5, 0.0)
run('dosc', 50, 2500, 0.02, 5, 0.2)
run('dosc', 50, 2500, 0.02, 3, 0.0)
run('selkov', 10, 10000, 0.002, 3, 0.0)
run('selkov', 10, 10000, 0.002, 3, 0.2)
run('lv', 200, 10000, 0.002, 2, 0.0)
