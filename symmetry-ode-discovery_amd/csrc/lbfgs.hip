// L-BFGS two-loop recursion for S independent small problems (seed sweeps): one wave per problem.
//
// torch.optim.LBFGS builds its search direction with ~4 tiny tensor ops per history pair
// (torch/optim/lbfgs.py, "compute the approximate (L-BFGS) inverse Hessian multiplied by the
// gradient"); for S problems of n <= 256 parameters that is hundreds of launches per iteration.
// Here problem s is one wavefront: lane l owns components l, l+64, l+128, l+192; the curvature pairs
// live in ring buffers (S, H, n) with per-problem head / count; every dot product is a wave reduction.
#include <hip/hip_runtime.h>

#include "../../include/symode.h"
#include "reduce.hpp"

namespace symode {

constexpr int LB_MAXC = 4;      // components per lane -> n <= 256
constexpr int LB_MAXH = 128;    // history slots

__global__ __launch_bounds__(WAVE) void lbfgs_direction_kernel(const float* __restrict__ g,
                                                               const float* __restrict__ old_dirs,
                                                               const float* __restrict__ old_stps,
                                                               const float* __restrict__ ro,
                                                               const long* __restrict__ head,
                                                               const long* __restrict__ count,
                                                               const float* __restrict__ h_diag, int n, int H,
                                                               float* __restrict__ d_out) {
    __shared__ float al[LB_MAXH];
    const long s = blockIdx.x;
    const int lane = threadIdx.x;
    const int m = (int)count[s], h0 = (int)head[s];
    const float* Y = old_dirs + s * (long)H * n;
    const float* Sx = old_stps + s * (long)H * n;
    const float* R = ro + s * (long)H;
    float q[LB_MAXC];
#pragma unroll
    for (int c = 0; c < LB_MAXC; ++c) {
        const int i = lane + WAVE * c;
        q[c] = i < n ? -g[s * n + i] : 0.0f;
    }
    for (int k = m - 1; k >= 0; --k) {                       // newest -> oldest
        const int slot = (h0 + k) % H;
        float part = 0.0f;
#pragma unroll
        for (int c = 0; c < LB_MAXC; ++c) {
            const int i = lane + WAVE * c;
            if (i < n) part = fmaf(Sx[slot * n + i], q[c], part);
        }
        const float a = wave_sum(part) * R[slot];
        if (lane == 0) al[k] = a;
#pragma unroll
        for (int c = 0; c < LB_MAXC; ++c) {
            const int i = lane + WAVE * c;
            if (i < n) q[c] = fmaf(-a, Y[slot * n + i], q[c]);
        }
    }
    __syncthreads();
    const float hd = h_diag[s];
#pragma unroll
    for (int c = 0; c < LB_MAXC; ++c) q[c] *= hd;            // r = q * H_diag
    for (int k = 0; k < m; ++k) {                            // oldest -> newest
        const int slot = (h0 + k) % H;
        float part = 0.0f;
#pragma unroll
        for (int c = 0; c < LB_MAXC; ++c) {
            const int i = lane + WAVE * c;
            if (i < n) part = fmaf(Y[slot * n + i], q[c], part);
        }
        const float be = wave_sum(part) * R[slot];
        const float coef = al[k] - be;
#pragma unroll
        for (int c = 0; c < LB_MAXC; ++c) {
            const int i = lane + WAVE * c;
            if (i < n) q[c] = fmaf(coef, Sx[slot * n + i], q[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < LB_MAXC; ++c) {
        const int i = lane + WAVE * c;
        if (i < n) d_out[s * n + i] = q[c];
    }
}


// NaN-propagating max over the wave (torch's amax propagates NaN; fmaxf alone would drop it).
__device__ __forceinline__ float wave_amax(float v) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) {
        const float o = __shfl_xor(v, off, WAVE);
        v = (v != v || o != o) ? __builtin_nanf("") : fmaxf(v, o);
    }
    return v;
}

// One inner iteration of torch.optim.LBFGS.step up to (and including) the parameter update, for every problem that is
// still active -- torch/optim/lbfgs.py "compute gradient descent direction" ... "no line search, simply move with fixed
// step" -- as ONE launch: curvature-pair update (ring buffer), two-loop recursion, step length, directional-derivative
// test, x += t d.  act[s] goes in as "active" and comes out as "moved" (the closure has to be re-evaluated there).
//
// STAGED: the recursion is 2m dependent steps (dot product -> axpy), and read from global memory every step pays a
// full memory round trip (measured 42 us per launch at 64 problems, m <= 100, beside a 12 us closure kernel; 19 us now).  The
// m pairs are therefore staged ONCE, all loads in flight together, into LDS in logical order ([k][i], k = 0 oldest) --
// the pair made in this launch straight from registers -- and the loops run out of LDS.  Needs (2 n + 1) H floats
// (<= 60 KB: n = 42, H = 100 is 34 KB); larger problems take the unstaged form (NC = 0).
//
// The staged loops (NC = ceil(n / 64) components per lane, compile time) are written for the dependent chain: the rows
// of the NEXT pair are fetched from LDS while the current dot product reduces (wave_sum_dpp), lanes beyond n are
// zeroed by selects at fetch time instead of branches around every use, and the m alphas stay in two registers (lane k
// keeps alpha_k, read back with v_readlane) instead of going through LDS.
//
// ACCEPT: the launch first finishes the PREVIOUS iteration (what lbfgs_accept_kernel does: take the re-evaluated loss /
// gradient, stopping tests) and carries on into this one if the problem is still active -- between two closure
// evaluations the optimiser is then ONE launch, and the accepted gradient never leaves the registers.
struct AcceptArgs {
    const float* new_loss;   // (S)     closure value at the moved parameters
    const float* new_g;      // (S, n)  its gradient
    float tol_grad;
    int l1;                  // != 0: new_loss / new_g are the bare data term, objective = w_x * data + w_reg * |params|_1
    float w_x, w_reg;
};

template <int NC, bool ACCEPT>
__global__ __launch_bounds__(WAVE) void lbfgs_update_kernel(float* __restrict__ params, float* __restrict__ g,
                                                            float* __restrict__ loss, unsigned char* __restrict__ act,
                                                            long* __restrict__ n_iter, float* __restrict__ d,
                                                            float* __restrict__ t, float* __restrict__ old_dirs,
                                                            float* __restrict__ old_stps, float* __restrict__ ro,
                                                            long* __restrict__ head, long* __restrict__ count,
                                                            float* __restrict__ h_diag, float* __restrict__ prev_g,
                                                            float* __restrict__ prev_loss, int n, int H, float lr,
                                                            float tol_change, AcceptArgs acc) {
    constexpr bool STAGED = NC > 0;
    __shared__ float al[LB_MAXH];
    extern __shared__ float staged[];                        // STAGED: pad [n] | Y [H][n] | S [H][n] | ro [H] | 256 floats of padding
    const long s = blockIdx.x;
    const int lane = threadIdx.x;
    if (!__builtin_amdgcn_readfirstlane((int)act[s])) return;   // wave-uniform: this problem stopped earlier
    float* const ldsY = staged + n;                          // (a row of padding in front: the loops prefetch row -1)
    float* const ldsS = ldsY + H * n;
    float* const ldsR = ldsS + H * n;
    float gv[LB_MAXC], q[LB_MAXC];
    float loss_s;
    if constexpr (ACCEPT) {
        const float tt = t[s];
        float nl = acc.new_loss[s];
        float gmax = 0.0f, dmax = 0.0f, p_l1 = 0.0f;
#pragma unroll
        for (int c = 0; c < LB_MAXC; ++c) {
            const int i = lane + WAVE * c;
            gv[c] = 0.0f;
            if (i < n) {
                float v = acc.new_g[s * n + i];
                if (acc.l1) {
                    const float pv = params[s * n + i];
                    const float sg = (float)(pv > 0.0f) - (float)(pv < 0.0f);
                    v = __fadd_rn(__fmul_rn(acc.w_x, v), __fmul_rn(acc.w_reg, sg));
                    p_l1 += fabsf(pv);
                }
                g[s * n + i] = v;
                gv[c] = v;
                const float av = fabsf(v), ad = fabsf(d[s * n + i] * tt);
                gmax = (av != av || gmax != gmax) ? __builtin_nanf("") : fmaxf(gmax, av);
                dmax = (ad != ad || dmax != dmax) ? __builtin_nanf("") : fmaxf(dmax, ad);
            }
        }
        gmax = wave_amax(gmax);
        dmax = wave_amax(dmax);
        if (acc.l1) nl = __fadd_rn(__fmul_rn(acc.w_x, nl), __fmul_rn(acc.w_reg, wave_sum(p_l1)));
        const bool stop = (gmax <= acc.tol_grad) || (dmax <= tol_change) || (fabsf(nl - prev_loss[s]) < tol_change);
        if (lane == 0) loss[s] = nl;
        if (__builtin_amdgcn_readfirstlane((int)stop)) {
            if (lane == 0) act[s] = 0;
            return;
        }
        loss_s = nl;
    } else {
#pragma unroll
        for (int c = 0; c < LB_MAXC; ++c) {
            const int i = lane + WAVE * c;
            gv[c] = i < n ? g[s * n + i] : 0.0f;
        }
        loss_s = loss[s];
    }
    const long ni = (long)__builtin_amdgcn_readfirstlane((int)n_iter[s]) + 1;      // (scalars: uniform loop bounds below)
    const bool first = ni == 1;
    float* Y = old_dirs + s * (long)H * n;
    float* Sx = old_stps + s * (long)H * n;
    float* R = ro + s * (long)H;
    int m = first ? 0 : __builtin_amdgcn_readfirstlane((int)count[s]);
    int h0 = first ? 0 : __builtin_amdgcn_readfirstlane((int)head[s]);
    float hd = first ? 1.0f : __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(h_diag[s])));
    bool fresh = false;                                      // a pair was stored in this launch (logical slot m - 1)
    if (!first) {                                            // "do lbfgs update (update memory)"
        const float told = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(t[s])));
        float y[LB_MAXC], sv[LB_MAXC], p_ys = 0.0f, p_yy = 0.0f;
#pragma unroll
        for (int c = 0; c < LB_MAXC; ++c) {
            const int i = lane + WAVE * c;
            y[c] = i < n ? gv[c] - prev_g[s * n + i] : 0.0f;
            sv[c] = i < n ? d[s * n + i] * told : 0.0f;
            p_ys = fmaf(y[c], sv[c], p_ys);
            p_yy = fmaf(y[c], y[c], p_yy);
        }
        const float ys = wave_sum_dpp_uniform(p_ys);
        if (ys > 1e-10f) {
            const bool full = m == H;
            const int pos = full ? h0 : (h0 + m) % H;        // overwrite the oldest pair when the memory is full
#pragma unroll
            for (int c = 0; c < LB_MAXC; ++c) {
                const int i = lane + WAVE * c;
                if (i < n) {
                    Y[pos * n + i] = y[c];
                    Sx[pos * n + i] = sv[c];
                }
            }
            if (lane == 0) R[pos] = 1.0f / ys;
            if (full) h0 = (h0 + 1) % H; else m += 1;
            hd = ys / wave_sum_dpp_uniform(p_yy);
            fresh = true;
            if (STAGED) {
#pragma unroll
                for (int c = 0; c < LB_MAXC; ++c) {
                    const int i = lane + WAVE * c;
                    if (i < n) {
                        ldsY[(m - 1) * n + i] = y[c];
                        ldsS[(m - 1) * n + i] = sv[c];
                    }
                }
                if (lane == 0) ldsR[m - 1] = 1.0f / ys;
            }
        }
    }
    if (STAGED) {
        const int m_old = fresh ? m - 1 : m;                 // pairs that were in memory before this launch
        // the ring buffer is at most two contiguous runs of rows: slots [h0, H) then [0, ...); each is copied flat, sixteen
        // loads in flight per lane (the 64 problems' histories sit in L2: ~0.7 us a round trip, so depth is what counts)
        auto copy_flat = [&](const float* __restrict__ src, float* __restrict__ dst, int len) {
#pragma unroll 16
            for (int idx = lane; idx < len; idx += WAVE) dst[idx] = src[idx];
        };
        const int run0 = m_old < H - h0 ? m_old : H - h0;    // rows in the first run
        copy_flat(Y + h0 * n, ldsY, run0 * n);
        copy_flat(Sx + h0 * n, ldsS, run0 * n);
        copy_flat(R + h0, ldsR, run0);
        if (m_old > run0) {
            copy_flat(Y, ldsY + run0 * n, (m_old - run0) * n);
            copy_flat(Sx, ldsS + run0 * n, (m_old - run0) * n);
            copy_flat(R, ldsR + run0, m_old - run0);
        }
        __syncthreads();
    }
    // two-loop recursion over the m stored pairs
#pragma unroll
    for (int c = 0; c < LB_MAXC; ++c) q[c] = -gv[c];
    if constexpr (STAGED) {
        bool on[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) on[c] = lane + WAVE * c < n;
        float sc[NC], yc[NC], sn[NC], yn[NC], rc, rn, al0 = 0.0f, al1 = 0.0f;
        // rows are walked with stepped pointers (one add per array and step; indexing by k costs a v_mul_lo per step on a
        // lone wave that issues one instruction per 4 cycles); the prefetch of the row past the end reads padding
        auto fetch = [&](const float* sp_, const float* yp_, const float* rp_, float (&sr)[NC], float (&yr)[NC], float& rr) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float sv_ = sp_[WAVE * c], yv_ = yp_[WAVE * c];
                sr[c] = on[c] ? sv_ : 0.0f;
                yr[c] = on[c] ? yv_ : 0.0f;
            }
            rr = *rp_;
        };
        const float* sp = ldsS + (m - 1) * n + lane;
        const float* yp = ldsY + (m - 1) * n + lane;
        const float* rp = ldsR + (m - 1);
        if (m > 0) fetch(sp, yp, rp, sc, yc, rc);
        for (int k = m - 1; k >= 0; --k) {                   // newest -> oldest
            sp -= n;
            yp -= n;
            rp -= 1;
            fetch(sp, yp, rp, sn, yn, rn);                   // row k - 1 (row -1: the padding, never used)
            float part = 0.0f;
#pragma unroll
            for (int c = 0; c < NC; ++c) part = fmaf(sc[c], q[c], part);
            const float a = wave_sum_dpp_uniform(part) * rc;
            al0 = lane == k ? a : al0;
            al1 = lane == k - WAVE ? a : al1;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                q[c] = fmaf(-a, yc[c], q[c]);
                sc[c] = sn[c];
                yc[c] = yn[c];
            }
            rc = rn;
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) q[c] *= hd;
        sp = ldsS + lane;
        yp = ldsY + lane;
        rp = ldsR;
        if (m > 0) fetch(sp, yp, rp, sc, yc, rc);
        for (int k = 0; k < m; ++k) {                        // oldest -> newest
            sp += n;
            yp += n;
            rp += 1;
            fetch(sp, yp, rp, sn, yn, rn);                   // row k + 1 (row m: stale or padding, never used)
            float part = 0.0f;
#pragma unroll
            for (int c = 0; c < NC; ++c) part = fmaf(yc[c], q[c], part);
            const float alk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(k < WAVE ? al0 : al1), k & (WAVE - 1)));
            const float coef = alk - wave_sum_dpp_uniform(part) * rc;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                q[c] = fmaf(coef, sc[c], q[c]);
                sc[c] = sn[c];
                yc[c] = yn[c];
            }
            rc = rn;
        }
    } else {
        for (int k = m - 1; k >= 0; --k) {
            const int slot = (h0 + k) % H;
            float part = 0.0f;
#pragma unroll
            for (int c = 0; c < LB_MAXC; ++c) {
                const int i = lane + WAVE * c;
                if (i < n) part = fmaf(Sx[slot * n + i], q[c], part);
            }
            // (this wave's own writes above are visible to it: same lanes, same addresses for Y / Sx; R[pos] was written by
            // lane 0 and is re-read through the same lane + a broadcast)
            const float rk = __shfl(lane == 0 ? R[slot] : 0.0f, 0, WAVE);
            const float a = wave_sum_dpp_uniform(part) * rk;
            if (lane == 0) al[k] = a;
#pragma unroll
            for (int c = 0; c < LB_MAXC; ++c) {
                const int i = lane + WAVE * c;
                if (i < n) q[c] = fmaf(-a, Y[slot * n + i], q[c]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < LB_MAXC; ++c) q[c] *= hd;
        for (int k = 0; k < m; ++k) {
            const int slot = (h0 + k) % H;
            float part = 0.0f;
#pragma unroll
            for (int c = 0; c < LB_MAXC; ++c) {
                const int i = lane + WAVE * c;
                if (i < n) part = fmaf(Y[slot * n + i], q[c], part);
            }
            const float rk = __shfl(lane == 0 ? R[slot] : 0.0f, 0, WAVE);
            const float coef = al[k] - wave_sum_dpp_uniform(part) * rk;
#pragma unroll
            for (int c = 0; c < LB_MAXC; ++c) {
                const int i = lane + WAVE * c;
                if (i < n) q[c] = fmaf(coef, Sx[slot * n + i], q[c]);
            }
        }
    }
    // step length, directional derivative, move
    float p_abs = 0.0f, p_gtd = 0.0f;
#pragma unroll
    for (int c = 0; c < LB_MAXC; ++c) {
        p_abs += fabsf(gv[c]);
        p_gtd = fmaf(gv[c], q[c], p_gtd);
    }
    const float tn = first ? fminf(1.0f, 1.0f / wave_sum_dpp_uniform(p_abs)) * lr : lr;
    const float gtd = wave_sum_dpp_uniform(p_gtd);
    const bool live = !(gtd > -tol_change);
#pragma unroll
    for (int c = 0; c < LB_MAXC; ++c) {
        const int i = lane + WAVE * c;
        if (i < n) {
            d[s * n + i] = q[c];
            prev_g[s * n + i] = gv[c];
            if (live) params[s * n + i] = __fadd_rn(params[s * n + i], __fmul_rn(tn, q[c]));
        }
    }
    if (lane == 0) {
        n_iter[s] = ni;
        head[s] = h0;
        count[s] = m;
        h_diag[s] = hd;
        t[s] = tn;
        prev_loss[s] = loss_s;
        act[s] = live ? 1 : 0;
    }
}

// The part of the iteration after the closure has been re-evaluated: problems that moved take the new loss / gradient
// and run torch's three stopping tests (optimality, step size, loss change); act[s]: "moved" in, "still active" out.
//
// ``params`` != nullptr: the closure handed over the bare data term; the objective is  w_x * loss + w_reg * |params|_1
// (train.py:680-688: L1 over the raw parameters) and its gradient  w_x * g + w_reg * sign(params)  -- added here, in
// the arithmetic of the tensor-op form (products rounded, then one add), instead of seven more launches.
__global__ __launch_bounds__(WAVE) void lbfgs_accept_kernel(const float* __restrict__ new_loss, const float* __restrict__ new_g,
                                                            float* __restrict__ loss, float* __restrict__ g,
                                                            unsigned char* __restrict__ act, const float* __restrict__ d,
                                                            const float* __restrict__ t, const float* __restrict__ prev_loss,
                                                            int n, float tol_grad, float tol_change,
                                                            const float* __restrict__ params, float w_x, float w_reg) {
    const long s = blockIdx.x;
    const int lane = threadIdx.x;
    if (!act[s]) return;
    const float tt = t[s];
    float nl = new_loss[s];
    float gmax = 0.0f, dmax = 0.0f, p_l1 = 0.0f;
#pragma unroll
    for (int c = 0; c < LB_MAXC; ++c) {
        const int i = lane + WAVE * c;
        if (i < n) {
            float v = new_g[s * n + i];
            if (params != nullptr) {
                const float pv = params[s * n + i];
                const float sg = (float)(pv > 0.0f) - (float)(pv < 0.0f);      // torch.sign: 0 at 0 and at NaN
                v = __fadd_rn(__fmul_rn(w_x, v), __fmul_rn(w_reg, sg));
                p_l1 += fabsf(pv);
            }
            g[s * n + i] = v;
            const float av = fabsf(v), ad = fabsf(d[s * n + i] * tt);
            gmax = (av != av || gmax != gmax) ? __builtin_nanf("") : fmaxf(gmax, av);
            dmax = (ad != ad || dmax != dmax) ? __builtin_nanf("") : fmaxf(dmax, ad);
        }
    }
    gmax = wave_amax(gmax);
    dmax = wave_amax(dmax);
    if (params != nullptr) nl = __fadd_rn(__fmul_rn(w_x, nl), __fmul_rn(w_reg, wave_sum(p_l1)));
    if (lane == 0) {
        loss[s] = nl;
        const bool stop = (gmax <= tol_grad) || (dmax <= tol_change) || (fabsf(nl - prev_loss[s]) < tol_change);
        act[s] = stop ? 0 : 1;
    }
}

}  // namespace symode

extern "C" int symode_lbfgs_direction(const float* g, const float* old_dirs, const float* old_stps, const float* ro,
                                      const long* head, const long* count, const float* h_diag, long n_problems, int n,
                                      int history, float* d_out, void* stream) {
    using namespace symode;
    if (n_problems < 1 || n < 1 || n > WAVE * LB_MAXC || history < 1 || history > LB_MAXH) return SYMODE_E_BADSIZE;
    if (!g || !old_dirs || !old_stps || !ro || !head || !count || !h_diag || !d_out) return SYMODE_E_NULLPTR;
    lbfgs_direction_kernel<<<dim3((unsigned)n_problems), dim3(WAVE), 0, (hipStream_t)stream>>>(
        g, old_dirs, old_stps, ro, head, count, h_diag, n, history, d_out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SYMODE_OK : (int)e;
}

namespace symode {
__global__ __launch_bounds__(WAVE) void wave_sum_selftest_kernel(const float* __restrict__ in, float* __restrict__ butterfly,
                                                                 float* __restrict__ dpp) {
    const long i = (long)blockIdx.x * WAVE + threadIdx.x;
    const float v = in[i];
    butterfly[i] = wave_sum(v);
    dpp[i] = wave_sum_dpp(v);
}
}  // namespace symode

extern "C" int symode_selftest_wave_sum(const float* in, float* butterfly_out, float* dpp_out, long n_waves, void* stream) {
    using namespace symode;
    if (n_waves < 1) return SYMODE_E_BADSIZE;
    if (!in || !butterfly_out || !dpp_out) return SYMODE_E_NULLPTR;
    wave_sum_selftest_kernel<<<dim3((unsigned)n_waves), dim3(WAVE), 0, (hipStream_t)stream>>>(in, butterfly_out, dpp_out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SYMODE_OK : (int)e;
}

namespace symode {
inline int launch_lbfgs_update(bool accept, float* params, float* g, float* loss, unsigned char* act, long* n_iter, float* d, float* t,
                               float* old_dirs, float* old_stps, float* ro, long* head, long* count, float* h_diag, float* prev_g,
                               float* prev_loss, long n_problems, int n, int history, float lr, float tol_change, AcceptArgs acc,
                               void* stream) {
    if (n_problems < 1 || n < 1 || n > WAVE * LB_MAXC || history < 1 || history > LB_MAXH) return SYMODE_E_BADSIZE;
    if (!params || !g || !loss || !act || !n_iter || !d || !t || !old_dirs || !old_stps || !ro || !head || !count || !h_diag ||
        !prev_g || !prev_loss || (accept && (!acc.new_loss || !acc.new_g)))
        return SYMODE_E_NULLPTR;
    const size_t stage_bytes = (((size_t)2 * n + 1) * history + n + 256) * sizeof(float);
    const int nc = stage_bytes <= 60 * 1024 ? (n + WAVE - 1) / WAVE : 0;
#define SYMODE_LBFGS_UPDATE(NC_, BYTES_)                                                                                          \
    do {                                                                                                                          \
        if (accept)                                                                                                               \
            lbfgs_update_kernel<NC_, true><<<dim3((unsigned)n_problems), dim3(WAVE), BYTES_, (hipStream_t)stream>>>(              \
                params, g, loss, act, n_iter, d, t, old_dirs, old_stps, ro, head, count, h_diag, prev_g, prev_loss, n, history,   \
                lr, tol_change, acc);                                                                                             \
        else                                                                                                                      \
            lbfgs_update_kernel<NC_, false><<<dim3((unsigned)n_problems), dim3(WAVE), BYTES_, (hipStream_t)stream>>>(             \
                params, g, loss, act, n_iter, d, t, old_dirs, old_stps, ro, head, count, h_diag, prev_g, prev_loss, n, history,   \
                lr, tol_change, acc);                                                                                             \
    } while (0)
    switch (nc) {
        case 1: SYMODE_LBFGS_UPDATE(1, stage_bytes); break;
        case 2: SYMODE_LBFGS_UPDATE(2, stage_bytes); break;
        case 3: SYMODE_LBFGS_UPDATE(3, stage_bytes); break;
        case 4: SYMODE_LBFGS_UPDATE(4, stage_bytes); break;
        default: SYMODE_LBFGS_UPDATE(0, 0); break;
    }
#undef SYMODE_LBFGS_UPDATE
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SYMODE_OK : (int)e;
}
}  // namespace symode

extern "C" int symode_lbfgs_update(float* params, const float* g, const float* loss, unsigned char* act, long* n_iter, float* d,
                                   float* t, float* old_dirs, float* old_stps, float* ro, long* head, long* count,
                                   float* h_diag, float* prev_g, float* prev_loss, long n_problems, int n, int history,
                                   float lr, float tol_change, void* stream) {
    return symode::launch_lbfgs_update(false, params, const_cast<float*>(g), const_cast<float*>(loss), act, n_iter, d, t, old_dirs,
                                       old_stps, ro, head, count, h_diag, prev_g, prev_loss, n_problems, n, history, lr, tol_change,
                                       symode::AcceptArgs{nullptr, nullptr, 0.0f, 0, 1.0f, 0.0f}, stream);
}

extern "C" int symode_lbfgs_accept_update(const float* new_loss, const float* new_g, float tol_grad, int l1, float w_x, float w_reg,
                                          float* params, float* g, float* loss, unsigned char* act, long* n_iter, float* d,
                                          float* t, float* old_dirs, float* old_stps, float* ro, long* head, long* count,
                                          float* h_diag, float* prev_g, float* prev_loss, long n_problems, int n, int history,
                                          float lr, float tol_change, void* stream) {
    return symode::launch_lbfgs_update(true, params, g, loss, act, n_iter, d, t, old_dirs, old_stps, ro, head, count, h_diag, prev_g,
                                       prev_loss, n_problems, n, history, lr, tol_change,
                                       symode::AcceptArgs{new_loss, new_g, tol_grad, l1, w_x, w_reg}, stream);
}

extern "C" int symode_lbfgs_accept(const float* new_loss, const float* new_g, float* loss, float* g, unsigned char* act,
                                   const float* d, const float* t, const float* prev_loss, long n_problems, int n,
                                   float tol_grad, float tol_change, const float* params, float w_x, float w_reg,
                                   void* stream) {
    using namespace symode;
    if (n_problems < 1 || n < 1 || n > WAVE * LB_MAXC) return SYMODE_E_BADSIZE;
    if (!new_loss || !new_g || !loss || !g || !act || !d || !t || !prev_loss) return SYMODE_E_NULLPTR;
    lbfgs_accept_kernel<<<dim3((unsigned)n_problems), dim3(WAVE), 0, (hipStream_t)stream>>>(
        new_loss, new_g, loss, g, act, d, t, prev_loss, n, tol_grad, tol_change, params, w_x, w_reg);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SYMODE_OK : (int)e;
}
