"""Equivariance constraint on Xi: null-space basis Q of the linear system the Lie generators impose.

Host-only (tiny matrices, run once per generator update), replacing the reference's sympy
route (sindy.py:123-166) by the closed form of the same matrix: for a monomial
Theta_t = prod z_i^{a_i},

    sum_m dTheta_t/dz_m (L z)_m = sum_{m,n} a_m L[m,n] z^{a - e_m + e_n}

so  J_Theta(z) L z = M Theta(z)  with  M[t, col(a - e_m + e_n)] += a_m L[m,n].
The constraint matrix, SVD rank rule (sigma <= 5e-3) and the Kronecker / non-Kronecker branch
follow sindy.py:85-115 exactly, quirks included (``det(L) < 1e-5`` without abs; the branch flag
of the last generator wins).
"""
from __future__ import annotations

import numpy as np
import torch

from . import library


def constraint_M(L: torch.Tensor, d: int, order: int) -> torch.Tensor:
    exps = library.exponents(d, order)
    col = {e: t for t, e in enumerate(exps)}
    p = len(exps)
    M = np.zeros((p, p))
    Ld = L.detach().cpu().double().numpy()
    for t, a in enumerate(exps):
        for m in range(d):
            if not a[m]:
                continue
            for n in range(d):
                if Ld[m, n] == 0.0:
                    continue
                b = list(a)
                b[m] -= 1
                b[n] += 1
                M[t, col[tuple(b)]] += a[m] * Ld[m, n]
    return torch.from_numpy(M).float()


def constraint_Q(L_list, d: int, order: int):
    """Returns (Q (d*p, r) float32 on CPU, use_kron_product)."""
    blocks, use_kron = [], None
    for L in L_list:
        L = L.detach().cpu().float()
        M = constraint_M(L, d, order)
        if torch.det(L) < 1e-5:                      # sindy.py:90 (no abs -- kept)
            use_kron = False
            MT = M.transpose(0, 1).contiguous()
            C = torch.kron(-MT, torch.eye(L.shape[0])) + torch.kron(torch.eye(MT.shape[0]), L)
        else:
            use_kron = True
            C = torch.kron(L.inverse(), M.T) - torch.eye(M.shape[0] * L.shape[0])
        blocks.append(C)
    C_total = torch.cat(blocks, dim=0)
    # same LAPACK route as the reference (sindy.py:100) -- on ONE thread: a threaded BLAS under gesdd rounds differently
    # with the thread count, and every process of a run (the ranks of a sweep, a 1-rank rerun) must build the same Q to
    # the last bit for its masks to be comparable (the matrix is at most a few hundred rows: microseconds either way)
    n_threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        _, sigma, V = torch.svd(C_total)
    finally:
        torch.set_num_threads(n_threads)
    r = 0
    for r in range(len(sigma)):                      # sindy.py:102-104
        if abs(sigma[-1 - r]) > 5e-3:
            break
    return V[:, -r:], use_kron
