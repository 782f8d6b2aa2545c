"""Host-side least-squares solves on the (tiny) normal equations built from the fp64 Gram.

The reference solves ``min ||A w - b||`` with ``torch.linalg.lstsq(A, b)`` on the tall
ridge-augmented matrix (sindy.py:261-288).  Its default LAPACK driver differs by device:

  * CPU  -> ``gelsy``: QR with column pivoting + incremental condition estimation, columns
    beyond the numerical rank are dropped and the minimum-norm solution returned, with
    torch's default ``rcond = eps(fp32) * max(m, n)``.  For m = 125 010 rows that is 1.5e-2:
    any library with cond(A) > ~67 is solved *rank-truncated* (probe: selkov order 3,
    cond 9e3 -> rank 5 of 10).
  * GPU  -> ``gels``: plain QR, full rank assumed, no truncation.

Both are reproduced here from the Gram matrix ``G = A^T A`` and ``C = A^T b`` alone (fp64):
pivoted Cholesky of G yields the same R factor (up to row signs) and the same pivot order as
xGEQP3 on A; the rank rule is LAPACK's xGELSY loop on xLAIC1 estimates; the truncated
minimum-norm solution follows from R.  ``driver='gelsy'`` is the default so that results match
the reference *CPU* path; ``driver='gels'`` gives what the reference computes on a GPU.
"""
from __future__ import annotations

import ctypes
import math

import numpy as np

EPS32 = float(np.finfo(np.float32).eps)
_EPS64 = float(np.finfo(np.float64).eps)


def _laic1(job: int, x: np.ndarray, sest: float, w: np.ndarray, gamma: float):
    """One step of incremental condition estimation (LAPACK xLAIC1, real case).

    job 1: largest, job 2: smallest singular value of [[L, 0], [w^T, gamma]] given the
    estimate ``sest`` with approximate singular vector ``x``.  Returns (sestpr, s, c).
    """
    eps = _EPS64
    alpha = float(np.dot(x, w))
    absalp, absgam, absest = abs(alpha), abs(gamma), abs(sest)
    sign = lambda a: 1.0 if a >= 0 else -1.0  # noqa: E731
    if job == 1:
        if sest == 0.0:
            s1 = max(absgam, absalp)
            if s1 == 0.0:
                return 0.0, 0.0, 1.0
            s, c = alpha / s1, gamma / s1
            tmp = math.sqrt(s * s + c * c)
            return s1 * tmp, s / tmp, c / tmp
        if absgam <= eps * absest:
            tmp = max(absest, absalp)
            s1, s2 = absest / tmp, absalp / tmp
            return tmp * math.sqrt(s1 * s1 + s2 * s2), 1.0, 0.0
        if absalp <= eps * absest:
            s1, s2 = absgam, absest
            return (s2, 1.0, 0.0) if s1 <= s2 else (s1, 0.0, 1.0)
        if absest <= eps * absalp or absest <= eps * absgam:
            s1, s2 = absgam, absalp
            if s1 <= s2:
                tmp = s1 / s2
                s = math.sqrt(1.0 + tmp * tmp)
                return s2 * s, sign(alpha) / s, (gamma / s2) / s
            tmp = s2 / s1
            c = math.sqrt(1.0 + tmp * tmp)
            return s1 * c, (alpha / s1) / c, sign(gamma) / c
        zeta1, zeta2 = alpha / absest, gamma / absest
        b = (1.0 - zeta1 * zeta1 - zeta2 * zeta2) * 0.5
        c = zeta1 * zeta1
        t = c / (b + math.sqrt(b * b + c)) if b > 0.0 else math.sqrt(b * b + c) - b
        sine, cosine = -zeta1 / t, -zeta2 / (1.0 + t)
        tmp = math.sqrt(sine * sine + cosine * cosine)
        return math.sqrt(t + 1.0) * absest, sine / tmp, cosine / tmp
    # job == 2
    if sest == 0.0:
        if max(absgam, absalp) == 0.0:
            sine, cosine = 1.0, 0.0
        else:
            sine, cosine = -gamma, alpha
        s1 = max(abs(sine), abs(cosine))
        s, c = sine / s1, cosine / s1
        tmp = math.sqrt(s * s + c * c)
        return 0.0, s / tmp, c / tmp
    if absgam <= eps * absest:
        return absgam, 0.0, 1.0
    if absalp <= eps * absest:
        s1, s2 = absgam, absest
        return (s1, 0.0, 1.0) if s1 <= s2 else (s2, 1.0, 0.0)
    if absest <= eps * absalp or absest <= eps * absgam:
        s1, s2 = absgam, absalp
        if s1 <= s2:
            tmp = s1 / s2
            c = math.sqrt(1.0 + tmp * tmp)
            return absest * (tmp / c), -(gamma / s2) / c, sign(alpha) / c
        tmp = s2 / s1
        s = math.sqrt(1.0 + tmp * tmp)
        return absest / s, -sign(gamma) / s, (alpha / s1) / s
    zeta1, zeta2 = alpha / absest, gamma / absest
    norma = max(1.0 + zeta1 * zeta1 + abs(zeta1 * zeta2), abs(zeta1 * zeta2) + zeta2 * zeta2)
    test = 1.0 + 2.0 * (zeta1 - zeta2) * (zeta1 + zeta2)
    if test >= 0.0:
        b = (zeta1 * zeta1 + zeta2 * zeta2 + 1.0) * 0.5
        c = zeta2 * zeta2
        t = c / (b + math.sqrt(abs(b * b - c)))
        sine, cosine = zeta1 / (1.0 - t), -zeta2 / t
        sestpr = math.sqrt(t + 4.0 * eps * eps * norma) * absest
    else:
        b = (zeta2 * zeta2 + zeta1 * zeta1 - 1.0) * 0.5
        c = zeta1 * zeta1
        t = -c / (b + math.sqrt(b * b + c)) if b >= 0.0 else b - math.sqrt(b * b + c)
        sine, cosine = -zeta1 / t, -zeta2 / (1.0 + t)
        sestpr = math.sqrt(1.0 + t + 4.0 * eps * eps * norma) * absest
    tmp = math.sqrt(sine * sine + cosine * cosine)
    return sestpr, sine / tmp, cosine / tmp


def pivoted_cholesky(G: np.ndarray):
    """Upper-triangular R and pivot order with R^T R = G[piv][:, piv] (the xGEQP3 factor of A).

    Pivot rule: largest remaining (Schur-complement) diagonal = largest partial column norm.
    A non-positive pivot ends the factorisation (remaining rows of R are zero).
    """
    n = G.shape[0]
    S = np.array(G, dtype=np.float64, copy=True)
    piv = np.arange(n)
    R = np.zeros((n, n))
    for k in range(n):
        diag = np.diag(S)[k:]
        j = k + int(np.argmax(diag))
        if j != k:
            S[[k, j], :] = S[[j, k], :]
            S[:, [k, j]] = S[:, [j, k]]
            R[:, [k, j]] = R[:, [j, k]]
            piv[[k, j]] = piv[[j, k]]
        dkk = S[k, k]
        if not dkk > 0.0:
            break
        r = math.sqrt(dkk)
        R[k, k] = r
        R[k, k + 1:] = S[k, k + 1:] / r
        S[k + 1:, k + 1:] -= np.outer(R[k, k + 1:], R[k, k + 1:])
        S[k, :] = 0.0
        S[:, k] = 0.0
        S[k, k] = 0.0
    return R, piv


def gelsy_rank(R: np.ndarray, rcond: float) -> int:
    """Numerical rank by LAPACK xGELSY's loop over xLAIC1 estimates of R's leading blocks."""
    n = R.shape[0]
    if n == 0 or abs(R[0, 0]) == 0.0:
        return 0
    xmin, xmax = np.array([1.0]), np.array([1.0])
    smax = smin = abs(R[0, 0])
    rank = 1
    while rank < n:
        w, gamma = R[:rank, rank], R[rank, rank]
        sminpr, s1, c1 = _laic1(2, xmin, smin, w, gamma)
        smaxpr, s2, c2 = _laic1(1, xmax, smax, w, gamma)
        if smaxpr * rcond <= sminpr:
            xmin = np.append(s1 * xmin, c1)
            xmax = np.append(s2 * xmax, c2)
            smin, smax = sminpr, smaxpr
            rank += 1
        else:
            break
    return rank


_NATIVE = None


def _native():
    """libsymode_hip.so's host solver (csrc/host_lstsq.cpp), or False when the library is not built."""
    global _NATIVE
    if _NATIVE is None:
        try:
            from .engine import load_library
            _NATIVE = load_library()
        except Exception:
            _NATIVE = False
    return _NATIVE


def lstsq_normal(G, C, m_rows: int, driver: str = "gelsy", rcond: float | None = None):
    """Solve min ||A w - b|| given G = A^T A (n,n) and C = A^T b (n,k); A had ``m_rows`` rows.
    Native host routine when libsymode_hip.so is present (same algorithm, ~100x less Python), else
    the reference implementation below.  Returns (W (n,k) or (n,), rank)."""
    lib = _native()
    if not lib:
        return lstsq_normal_py(G, C, m_rows, driver, rcond)
    if driver not in ("gelsy", "gels"):
        raise ValueError(f"unknown lstsq driver {driver!r}")
    G = np.ascontiguousarray(G, dtype=np.float64)
    C = np.asarray(C, dtype=np.float64)
    squeeze = C.ndim == 1
    C2 = np.ascontiguousarray(C[:, None] if squeeze else C)
    n, k = G.shape[0], C2.shape[1]
    W = np.zeros((n, k))
    rank = ctypes.c_int(0)
    rc = lib.symode_host_lstsq_normal(G.ctypes.data, C2.ctypes.data, n, k, int(m_rows), 0 if driver == "gelsy" else 1,
                                      -1.0 if rcond is None else float(rcond), W.ctypes.data, ctypes.addressof(rank))
    if rc != 0:
        if driver == "gels":
            return _singular_fallback(G, C, m_rows)
        raise RuntimeError(f"symode_host_lstsq_normal failed with code {rc}")
    return (W[:, 0] if squeeze else W), rank.value


SINGULAR_RCOND = 1e-7      # on the triangular factor, i.e. cond(A^T A) > 1e14: singular to fp64 working accuracy
_warned_singular = False


def _singular_fallback(G, C, m_rows):
    """The full-rank driver met exactly singular normal equations (a latent batch that lies in a subspace, a constraint
    that leaves a column empty).  torch.linalg.lstsq(driver='gels') returns unusable numbers there without saying so;
    here the rank-revealing solve takes over with a rank cut at fp64 working accuracy: the minimum-norm solution."""
    global _warned_singular
    if not _warned_singular:
        import warnings
        warnings.warn("singular normal equations under the full-rank (gels) driver: minimum-norm solution returned instead",
                      RuntimeWarning)
        _warned_singular = True
    if _native():                                   # same algorithm in the native host routine (~100x less Python)
        return lstsq_normal(G, C, m_rows, "gelsy", SINGULAR_RCOND)
    return lstsq_normal_py(G, C, m_rows, "gelsy", SINGULAR_RCOND)


def lstsq_normal_py(G: np.ndarray, C: np.ndarray, m_rows: int, driver: str = "gelsy", rcond: float | None = None):
    """Pure-numpy reference implementation of ``lstsq_normal`` (kept as the readable specification and
    for environments without the built library).

    Returns (W (n,k), rank).  ``rcond=None`` -> torch's default eps(fp32) * max(m, n).
    """
    G = np.asarray(G, dtype=np.float64)
    C = np.asarray(C, dtype=np.float64)
    squeeze = C.ndim == 1
    if squeeze:
        C = C[:, None]
    n = G.shape[0]
    if n == 0:
        W = np.zeros((0, C.shape[1]))
        return (W[:, 0] if squeeze else W), 0
    if driver == "gels":
        try:
            W = np.linalg.solve(G, C)
        except np.linalg.LinAlgError:
            return _singular_fallback(G, C[:, 0] if squeeze else C, m_rows)
        return (W[:, 0] if squeeze else W), n
    if driver != "gelsy":
        raise ValueError(f"unknown lstsq driver {driver!r}")
    if rcond is None:
        rcond = EPS32 * max(m_rows, n)
    R, piv = pivoted_cholesky(G)
    rank = gelsy_rank(R, rcond)
    Cp = C[piv]
    Y = np.zeros((n, C.shape[1]))
    if rank > 0:
        R11 = R[:rank, :rank]
        q1b = _solve_tri(R11.T, Cp[:rank], lower=True)            # (Q^T b)[:rank]
        if rank == n:
            Y = _solve_tri(R11, q1b, lower=False)
        else:
            Wm = R[:rank, :]                                        # [R11 R12]
            Y = Wm.T @ np.linalg.solve(Wm @ Wm.T, q1b)              # minimum-norm solution
    W = np.zeros_like(Y)
    W[piv] = Y
    return (W[:, 0] if squeeze else W), rank


def _solve_tri(T: np.ndarray, B: np.ndarray, lower: bool) -> np.ndarray:
    n = T.shape[0]
    X = np.zeros_like(B, dtype=np.float64)
    rng = range(n) if lower else range(n - 1, -1, -1)
    for i in rng:
        if lower:
            acc = B[i] - T[i, :i] @ X[:i]
        else:
            acc = B[i] - T[i, i + 1:] @ X[i + 1:]
        X[i] = acc / T[i, i]
    return X
