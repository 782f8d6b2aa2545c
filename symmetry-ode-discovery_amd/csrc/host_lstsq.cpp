// Host-side least squares on the normal equations (G = A^T A, C = A^T b), fp64 -- the native twin of
// symmetry-ode-discovery_amd/lstsq.py (see there for the derivation and the LAPACK references):
//   driver 0 "gelsy": pivoted Cholesky of G (= xGEQP3's R factor and pivot order on A), numerical rank by
//                     xGELSY's loop over xLAIC1 estimates with rcond, minimum-norm solution;
//   driver 1 "gels" : full-rank solve.
// Runs on the host; the (p+d)^2 Gram matrix comes from the GPU (symode_aug_gram).
#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

#include "../../include/symode.h"

namespace {

const double EPS = std::numeric_limits<double>::epsilon();
inline double sgn(double a) { return a >= 0 ? 1.0 : -1.0; }

// one step of incremental condition estimation (LAPACK xLAIC1, real case)
void laic1(int job, const std::vector<double>& x, double sest, const double* w, int j, double gamma, double& sestpr,
           double& s, double& c) {
    double alpha = 0;
    for (int i = 0; i < j; ++i) alpha += x[i] * w[i];
    const double absalp = std::fabs(alpha), absgam = std::fabs(gamma), absest = std::fabs(sest);
    if (job == 1) {
        if (sest == 0.0) {
            const double s1 = std::max(absgam, absalp);
            if (s1 == 0.0) { s = 0; c = 1; sestpr = 0; return; }
            s = alpha / s1; c = gamma / s1;
            const double t = std::sqrt(s * s + c * c);
            s /= t; c /= t; sestpr = s1 * t; return;
        }
        if (absgam <= EPS * absest) {
            const double t = std::max(absest, absalp), s1 = absest / t, s2 = absalp / t;
            s = 1; c = 0; sestpr = t * std::sqrt(s1 * s1 + s2 * s2); return;
        }
        if (absalp <= EPS * absest) {
            if (absgam <= absest) { s = 1; c = 0; sestpr = absest; } else { s = 0; c = 1; sestpr = absgam; }
            return;
        }
        if (absest <= EPS * absalp || absest <= EPS * absgam) {
            const double s1 = absgam, s2 = absalp;
            if (s1 <= s2) { const double t = s1 / s2; const double q = std::sqrt(1 + t * t); sestpr = s2 * q; c = (gamma / s2) / q; s = sgn(alpha) / q; }
            else { const double t = s2 / s1; const double q = std::sqrt(1 + t * t); sestpr = s1 * q; s = (alpha / s1) / q; c = sgn(gamma) / q; }
            return;
        }
        const double z1 = alpha / absest, z2 = gamma / absest;
        const double b = (1.0 - z1 * z1 - z2 * z2) * 0.5, cc = z1 * z1;
        const double t = b > 0 ? cc / (b + std::sqrt(b * b + cc)) : std::sqrt(b * b + cc) - b;
        const double sine = -z1 / t, cosine = -z2 / (1.0 + t), nrm = std::sqrt(sine * sine + cosine * cosine);
        s = sine / nrm; c = cosine / nrm; sestpr = std::sqrt(t + 1.0) * absest; return;
    }
    if (sest == 0.0) {
        sestpr = 0;
        double sine, cosine;
        if (std::max(absgam, absalp) == 0.0) { sine = 1; cosine = 0; } else { sine = -gamma; cosine = alpha; }
        const double s1 = std::max(std::fabs(sine), std::fabs(cosine));
        s = sine / s1; c = cosine / s1;
        const double t = std::sqrt(s * s + c * c);
        s /= t; c /= t; return;
    }
    if (absgam <= EPS * absest) { s = 0; c = 1; sestpr = absgam; return; }
    if (absalp <= EPS * absest) {
        if (absgam <= absest) { s = 0; c = 1; sestpr = absgam; } else { s = 1; c = 0; sestpr = absest; }
        return;
    }
    if (absest <= EPS * absalp || absest <= EPS * absgam) {
        const double s1 = absgam, s2 = absalp;
        if (s1 <= s2) { const double t = s1 / s2; const double q = std::sqrt(1 + t * t); sestpr = absest * (t / q); s = -(gamma / s2) / q; c = sgn(alpha) / q; }
        else { const double t = s2 / s1; const double q = std::sqrt(1 + t * t); sestpr = absest / q; c = (alpha / s1) / q; s = -sgn(gamma) / q; }
        return;
    }
    const double z1 = alpha / absest, z2 = gamma / absest;
    const double norma = std::max(1.0 + z1 * z1 + std::fabs(z1 * z2), std::fabs(z1 * z2) + z2 * z2);
    const double test = 1.0 + 2.0 * (z1 - z2) * (z1 + z2);
    double sine, cosine;
    if (test >= 0) {
        const double b = (z1 * z1 + z2 * z2 + 1.0) * 0.5, cc = z2 * z2;
        const double t = cc / (b + std::sqrt(std::fabs(b * b - cc)));
        sine = z1 / (1.0 - t); cosine = -z2 / t;
        sestpr = std::sqrt(t + 4.0 * EPS * EPS * norma) * absest;
    } else {
        const double b = (z2 * z2 + z1 * z1 - 1.0) * 0.5, cc = z1 * z1;
        const double t = b >= 0 ? -cc / (b + std::sqrt(b * b + cc)) : b - std::sqrt(b * b + cc);
        sine = -z1 / t; cosine = -z2 / (1.0 + t);
        sestpr = std::sqrt(1.0 + t + 4.0 * EPS * EPS * norma) * absest;
    }
    const double nrm = std::sqrt(sine * sine + cosine * cosine);
    s = sine / nrm; c = cosine / nrm;
}

}  // namespace

extern "C" int symode_host_lstsq_normal(const double* G, const double* C, int n, int k, long m_rows, int driver,
                                        double rcond, double* W, int* rank_out) {
    if (n < 0 || k < 1 || (driver != 0 && driver != 1)) return SYMODE_E_BADSIZE;
    if (n == 0) { if (rank_out) *rank_out = 0; return SYMODE_OK; }
    if (!G || !C || !W) return SYMODE_E_NULLPTR;
    std::vector<double> S(G, G + (size_t)n * n), R((size_t)n * n, 0.0);
    std::vector<int> piv(n);
    for (int i = 0; i < n; ++i) piv[i] = i;
    auto Sat = [&](int i, int j) -> double& { return S[(size_t)i * n + j]; };
    auto Rat = [&](int i, int j) -> double& { return R[(size_t)i * n + j]; };
    if (driver == 1) {                                    // full rank: Cholesky without pivoting would do; keep pivoting for stability
        rcond = 0.0;
    } else if (rcond < 0) {
        rcond = (double)std::numeric_limits<float>::epsilon() * (double)std::max<long>(m_rows, n);
    }
    int fact = n;
    for (int kk = 0; kk < n; ++kk) {                      // pivoted Cholesky: largest remaining diagonal first
        int j = kk;
        for (int i = kk + 1; i < n; ++i)
            if (Sat(i, i) > Sat(j, j)) j = i;
        if (j != kk) {
            for (int c = 0; c < n; ++c) std::swap(Sat(kk, c), Sat(j, c));
            for (int r = 0; r < n; ++r) std::swap(Sat(r, kk), Sat(r, j));
            for (int r = 0; r < n; ++r) std::swap(Rat(r, kk), Rat(r, j));
            std::swap(piv[kk], piv[j]);
        }
        const double dkk = Sat(kk, kk);
        if (!(dkk > 0.0)) { fact = kk; break; }
        const double r = std::sqrt(dkk);
        Rat(kk, kk) = r;
        for (int c = kk + 1; c < n; ++c) Rat(kk, c) = Sat(kk, c) / r;
        for (int a = kk + 1; a < n; ++a)
            for (int b = kk + 1; b < n; ++b) Sat(a, b) -= Rat(kk, a) * Rat(kk, b);
        for (int c = 0; c < n; ++c) { Sat(kk, c) = 0.0; Sat(c, kk) = 0.0; }
    }
    int rank;
    if (driver == 1) {
        if (fact < n) return SYMODE_E_BADSIZE;            // singular system under the full-rank driver
        rank = n;
    } else {                                              // xGELSY rank loop
        rank = 0;
        if (std::fabs(Rat(0, 0)) != 0.0) {
            std::vector<double> xmin(1, 1.0), xmax(1, 1.0), w(n);
            double smax = std::fabs(Rat(0, 0)), smin = smax;
            rank = 1;
            while (rank < n) {
                for (int i = 0; i < rank; ++i) w[i] = Rat(i, rank);
                double sminpr, s1, c1, smaxpr, s2, c2;
                laic1(2, xmin, smin, w.data(), rank, Rat(rank, rank), sminpr, s1, c1);
                laic1(1, xmax, smax, w.data(), rank, Rat(rank, rank), smaxpr, s2, c2);
                if (smaxpr * rcond <= sminpr) {
                    for (int i = 0; i < rank; ++i) { xmin[i] *= s1; xmax[i] *= s2; }
                    xmin.push_back(c1); xmax.push_back(c2);
                    smin = sminpr; smax = smaxpr; ++rank;
                } else break;
            }
        }
    }
    if (rank_out) *rank_out = rank;
    std::vector<double> Y((size_t)n * k, 0.0), q((size_t)std::max(rank, 1) * k, 0.0);
    // q = R11^{-T} (P^T C)[:rank]
    for (int col = 0; col < k; ++col)
        for (int i = 0; i < rank; ++i) {
            double acc = C[(size_t)piv[i] * k + col];
            for (int t = 0; t < i; ++t) acc -= Rat(t, i) * q[(size_t)t * k + col];
            q[(size_t)i * k + col] = acc / Rat(i, i);
        }
    if (rank == n) {                                      // Y = R^{-1} q
        for (int col = 0; col < k; ++col)
            for (int i = n - 1; i >= 0; --i) {
                double acc = q[(size_t)i * k + col];
                for (int t = i + 1; t < n; ++t) acc -= Rat(i, t) * Y[(size_t)t * k + col];
                Y[(size_t)i * k + col] = acc / Rat(i, i);
            }
    } else if (rank > 0) {                                // minimum norm: Y = Wm^T (Wm Wm^T)^{-1} q, Wm = R[:rank, :]
        std::vector<double> M((size_t)rank * rank, 0.0), Lc((size_t)rank * rank, 0.0), z((size_t)rank * k);
        for (int a = 0; a < rank; ++a)
            for (int b = 0; b < rank; ++b) {
                double acc = 0;
                for (int t = 0; t < n; ++t) acc += Rat(a, t) * Rat(b, t);
                M[(size_t)a * rank + b] = acc;
            }
        for (int a = 0; a < rank; ++a)                    // Cholesky M = Lc Lc^T (SPD: Wm has full row rank)
            for (int b = 0; b <= a; ++b) {
                double acc = M[(size_t)a * rank + b];
                for (int t = 0; t < b; ++t) acc -= Lc[(size_t)a * rank + t] * Lc[(size_t)b * rank + t];
                Lc[(size_t)a * rank + b] = (a == b) ? std::sqrt(acc) : acc / Lc[(size_t)b * rank + b];
            }
        for (int col = 0; col < k; ++col) {
            for (int i = 0; i < rank; ++i) {
                double acc = q[(size_t)i * k + col];
                for (int t = 0; t < i; ++t) acc -= Lc[(size_t)i * rank + t] * z[(size_t)t * k + col];
                z[(size_t)i * k + col] = acc / Lc[(size_t)i * rank + i];
            }
            for (int i = rank - 1; i >= 0; --i) {
                double acc = z[(size_t)i * k + col];
                for (int t = i + 1; t < rank; ++t) acc -= Lc[(size_t)t * rank + i] * z[(size_t)t * k + col];
                z[(size_t)i * k + col] = acc / Lc[(size_t)i * rank + i];
            }
            for (int t = 0; t < n; ++t) {
                double acc = 0;
                for (int a = 0; a < rank; ++a) acc += Rat(a, t) * z[(size_t)a * k + col];
                Y[(size_t)t * k + col] = acc;
            }
        }
    }
    for (int i = 0; i < n; ++i)
        for (int col = 0; col < k; ++col) W[(size_t)piv[i] * k + col] = Y[(size_t)i * k + col];
    return SYMODE_OK;
}


// Sequential-threshold least squares to convergence for S problems from their augmented Gram matrices -- the loop of
// train.py:872-887 around solve_SINDy_one_step (sindy.py:250-315), unconstrained case, as sweep.SeedSweepSTLSQ.solve
// states it in numpy: per pass the ridge system on the current support (full mask: one (p, p) system with d right-hand
// sides; otherwise the reference's block-diagonal, column-selected system in equation-major order, solved as ONE system
// because gelsy's rank decision is joint), coefficients rounded to fp32, strict > threshold on the still-active
// entries, until the mask repeats.  gels meeting a singular system falls back to the rank-revealing solve at fp64 working
// accuracy (lstsq.py::_singular_fallback) and counts it.  Same solver, same inputs, same arithmetic as the Python loop:
// 24 us -> ~3 us per pass.
extern "C" int symode_host_stlsq_sweep(const double* G, int S, int d, int p, long n_points, double gamma, double threshold,
                                       int max_iter, int driver, double near_band, float* xi_out, unsigned char* mask_out,
                                       int* passes_out, int* near_out, int* fallback_out) {
    if (S < 0 || d < 1 || p < 1 || max_iter < 1 || (driver != 0 && driver != 1)) return SYMODE_E_BADSIZE;
    if (S == 0) return SYMODE_OK;
    if (!G || !xi_out || !mask_out || !passes_out) return SYMODE_E_NULLPTR;
    const int F = p + d, dp = d * p;
    const float thr32 = (float)threshold;
    std::vector<double> Gtt((size_t)p * p), Gty((size_t)p * d), W, Gm, cm, w;
    std::vector<int> sel;
    auto solve = [&](const double* A, const double* C, int n, int k, long m_rows, double* out, int& fell) -> int {
        int rank = 0;
        int rc = symode_host_lstsq_normal(A, C, n, k, m_rows, driver, -1.0, out, &rank);
        if (rc != SYMODE_OK && driver == 1) {                      // singular under the full-rank driver
            ++fell;
            rc = symode_host_lstsq_normal(A, C, n, k, m_rows, 0, 1e-7, out, &rank);
        }
        return rc;
    };
    for (int s = 0; s < S; ++s) {
        const double* Gs = G + (size_t)s * F * F;
        for (int i = 0; i < p; ++i) {
            for (int j = 0; j < p; ++j) Gtt[(size_t)i * p + j] = Gs[(size_t)i * F + j] + (i == j ? gamma * gamma : 0.0);
            for (int j = 0; j < d; ++j) Gty[(size_t)i * d + j] = Gs[(size_t)i * F + p + j];
        }
        unsigned char* mask = mask_out + (size_t)s * dp;
        float* xi32 = xi_out + (size_t)s * dp;
        for (int f = 0; f < dp; ++f) mask[f] = 1;
        int near = 0, fell = 0, passes = 0;
        for (int it = 0; it < max_iter; ++it) {
            std::vector<double> xi((size_t)dp, 0.0);
            sel.clear();
            for (int f = 0; f < dp; ++f)
                if (mask[f]) sel.push_back(f);
            const int nnz = (int)sel.size();
            if (nnz == dp) {
                W.assign((size_t)p * d, 0.0);
                const int rc = solve(Gtt.data(), Gty.data(), p, d, n_points + p, W.data(), fell);
                if (rc != SYMODE_OK) return rc;
                for (int j = 0; j < d; ++j)
                    for (int k = 0; k < p; ++k) xi[(size_t)j * p + k] = W[(size_t)k * d + j];
            } else if (nnz > 0) {
                Gm.assign((size_t)nnz * nnz, 0.0);
                cm.assign((size_t)nnz, 0.0);
                w.assign((size_t)nnz, 0.0);
                for (int a = 0; a < nnz; ++a) {
                    const int ja = sel[a] / p, ka = sel[a] % p;
                    cm[a] = Gty[(size_t)ka * d + ja];
                    for (int b = 0; b < nnz; ++b)
                        if (sel[b] / p == ja) Gm[(size_t)a * nnz + b] = Gtt[(size_t)ka * p + sel[b] % p];
                }
                const int rc = solve(Gm.data(), cm.data(), nnz, 1, (long)d * (n_points + p), w.data(), fell);
                if (rc != SYMODE_OK) return rc;
                for (int a = 0; a < nnz; ++a) xi[sel[a]] = w[a];
            }
            bool converged = true;
            for (int f = 0; f < dp; ++f) {
                const float v = (float)xi[f];
                xi32[f] = v;
                const float av = std::fabs(v);
                if (mask[f] && std::fabs((double)av - threshold) < near_band) ++near;
                const unsigned char nm = (mask[f] && av > thr32) ? 1 : 0;
                if (nm != mask[f]) converged = false;
                mask[f] = nm;
            }
            passes = it + 1;
            if (converged) break;
        }
        passes_out[s] = passes;
        if (near_out) near_out[s] = near;
        if (fallback_out) fallback_out[s] = fell;
    }
    return SYMODE_OK;
}
