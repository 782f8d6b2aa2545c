// Compile-time description of the SINDy feature library Theta(x).
//
// Column order (reference sindy.py:68-77, verified column by column in tests/golden/f1_theta):
//   [ 1 | x_0..x_{D-1} | x_i x_j (i<=j) | (x_i x_j) x_k (i<=j<=k) | ... | sin x_i | exp x_i ]
// A polynomial column of degree n with sorted index tuple (i1<=...<=in) is the column of
// its prefix (i1..i_{n-1}) times x_{in}: exactly the reference's left-to-right product
// (sindy.py:14, 20), one fp32 rounding per multiply, so polynomial columns are bit-exact.
// Enumerating "children of every degree n-1 term q, last variable v = last(q)..D-1" visits
// the degree-n tuples in the reference's nested-loop (lexicographic) order, which extends
// the reference's ordering to orders 4-5 (the reference stops at cubic, sindy.py:37).
#pragma once
#include <hip/hip_runtime.h>

namespace symode {

constexpr int FLAG_SINE = 1;   // include_sine  (sindy.py:74-75)
constexpr int FLAG_EXP = 2;    // include_exp   (sindy.py:76-77)

constexpr int MAX_D = 4;

constexpr int MAX_ORDER = 5;

constexpr long binom(int n, int k) {
    long r = 1;
    for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
    return r;
}

// number of polynomial columns including the constant       (sindy.py:179-184)
constexpr int poly_terms(int d, int order) {
    int p = 1;
    for (int n = 1; n <= order; ++n) p += (int)binom(d + n - 1, n);
    return p;
}

constexpr int lib_terms(int d, int order, int flags) {
    return poly_terms(d, order) + ((flags & FLAG_SINE) ? d : 0) + ((flags & FLAG_EXP) ? d : 0);
}

template <int D, int ORDER>
struct PolyTable {
    static constexpr int NP = poly_terms(D, ORDER);
    int parent[NP];   // column of the degree n-1 prefix (0 = the constant column)
    int var[NP];      // last variable of the sorted index tuple
    constexpr PolyTable() : parent{}, var{} {
        parent[0] = -1;
        var[0] = 0;
        int begin = 0, end = 1, t = 1;
        for (int n = 1; n <= ORDER; ++n) {
            for (int q = begin; q < end; ++q)
                for (int v = (q == 0 ? 0 : var[q]); v < D; ++v) {
                    parent[t] = q;
                    var[t] = v;
                    ++t;
                }
            begin = end;
            end = t;
        }
    }
};

template <int D_, int ORDER_, int FLAGS_>
struct Library {
    static constexpr int D = D_;
    static constexpr int ORDER = ORDER_;
    static constexpr int FLAGS = FLAGS_;
    static constexpr int NP = poly_terms(D, ORDER);
    static constexpr int P = lib_terms(D, ORDER, FLAGS);
    static constexpr bool SINE = (FLAGS & FLAG_SINE) != 0;
    static constexpr bool EXP = (FLAGS & FLAG_EXP) != 0;
    static constexpr int SIN0 = NP;                       // first sine column
    static constexpr int EXP0 = NP + (SINE ? D : 0);      // first exp column
    static constexpr PolyTable<D, ORDER> tab{};

    // Theta(x) into th[P].
    static __device__ __forceinline__ void eval(const float (&x)[D], float (&th)[P]) {
        th[0] = 1.0f;
#pragma unroll
        for (int t = 1; t < NP; ++t) th[t] = th[tab.parent[t]] * x[tab.var[t]];
        if constexpr (SINE) {
#pragma unroll
            for (int i = 0; i < D; ++i) th[SIN0 + i] = sinf(x[i]);
        }
        if constexpr (EXP) {
#pragma unroll
            for (int i = 0; i < D; ++i) th[EXP0 + i] = expf(x[i]);
        }
    }

    // Same recurrence in fp64 (offline data generation only: the reference integrates in numpy float64).
    static __device__ __forceinline__ void eval_f64(const double (&x)[D], double (&th)[P]) {
        th[0] = 1.0;
#pragma unroll
        for (int t = 1; t < NP; ++t) th[t] = th[tab.parent[t]] * x[tab.var[t]];
        if constexpr (SINE) {
#pragma unroll
            for (int i = 0; i < D; ++i) th[SIN0 + i] = sin(x[i]);
        }
        if constexpr (EXP) {
#pragma unroll
            for (int i = 0; i < D; ++i) th[EXP0 + i] = exp(x[i]);
        }
    }

    // Theta(x) and the directional derivative dth = J_Theta(x) . v  (product rule along the
    // same recurrence: d(th_q * x_v) = dth_q * x_v + th_q * v_v).
    static __device__ __forceinline__ void eval_jvp(const float (&x)[D], const float (&v)[D], float (&th)[P],
                                                    float (&dth)[P]) {
        th[0] = 1.0f;
        dth[0] = 0.0f;
#pragma unroll
        for (int t = 1; t < NP; ++t) {
            const int q = tab.parent[t], a = tab.var[t];
            th[t] = th[q] * x[a];
            dth[t] = (q == 0) ? v[a] : fmaf(dth[q], x[a], th[q] * v[a]);
        }
        if constexpr (SINE) {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                th[SIN0 + i] = sinf(x[i]);
                dth[SIN0 + i] = cosf(x[i]) * v[i];
            }
        }
        if constexpr (EXP) {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                th[EXP0 + i] = expf(x[i]);
                dth[EXP0 + i] = th[EXP0 + i] * v[i];
            }
        }
    }

    // Reverse mode through the same recurrence: given bar_th = dL/dTheta (consumed), return
    // bar_x = J_Theta(x)^T bar_th.  th must hold Theta(x).
    static __device__ __forceinline__ void vjp(const float (&x)[D], const float (&th)[P], float (&bar)[P],
                                               float (&bx)[D]) {
#pragma unroll
        for (int i = 0; i < D; ++i) bx[i] = 0.0f;
#pragma unroll
        for (int t = NP - 1; t >= 1; --t) {
            const int q = tab.parent[t], a = tab.var[t];
            bx[a] = fmaf(bar[t], th[q], bx[a]);
            if (q != 0) bar[q] = fmaf(bar[t], x[a], bar[q]);
        }
        if constexpr (SINE) {
#pragma unroll
            for (int i = 0; i < D; ++i) bx[i] = fmaf(bar[SIN0 + i], cosf(x[i]), bx[i]);
        }
        if constexpr (EXP) {
#pragma unroll
            for (int i = 0; i < D; ++i) bx[i] = fmaf(bar[EXP0 + i], th[EXP0 + i], bx[i]);
        }
    }

    // Reverse mode of eval_jvp: adjoints bar (on Theta) and dbar (on dTheta), both consumed,
    // give bx = dL/dx (includes the second-order term sum_t dbar_t Hess(Theta_t) v) and bv = dL/dv.
    // th, dth must hold eval_jvp(x, v).
    static __device__ __forceinline__ void vjp_of_jvp(const float (&x)[D], const float (&v)[D], const float (&th)[P],
                                                      const float (&dth)[P], float (&bar)[P], float (&dbar)[P],
                                                      float (&bx)[D], float (&bv)[D]) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            bx[i] = 0.0f;
            bv[i] = 0.0f;
        }
#pragma unroll
        for (int t = NP - 1; t >= 1; --t) {
            const int q = tab.parent[t], a = tab.var[t];
            if (q == 0) {                       // Theta_t = x_a, dTheta_t = v_a
                bx[a] += bar[t];
                bv[a] += dbar[t];
            } else {                            // Theta_t = Theta_q x_a ; dTheta_t = dTheta_q x_a + Theta_q v_a
                bx[a] = fmaf(bar[t], th[q], fmaf(dbar[t], dth[q], bx[a]));
                bv[a] = fmaf(dbar[t], th[q], bv[a]);
                bar[q] = fmaf(bar[t], x[a], fmaf(dbar[t], v[a], bar[q]));
                dbar[q] = fmaf(dbar[t], x[a], dbar[q]);
            }
        }
        if constexpr (SINE) {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const float c = cosf(x[i]), sn = th[SIN0 + i];
                bx[i] += bar[SIN0 + i] * c - dbar[SIN0 + i] * sn * v[i];
                bv[i] = fmaf(dbar[SIN0 + i], c, bv[i]);
            }
        }
        if constexpr (EXP) {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const float e = th[EXP0 + i];
                bx[i] += bar[EXP0 + i] * e + dbar[EXP0 + i] * e * v[i];
                bv[i] = fmaf(dbar[EXP0 + i], e, bv[i]);
            }
        }
    }
};

}  // namespace symode
