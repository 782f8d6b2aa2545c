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
    const float* new_loss;   // (S)     closure value at the moved parameters [(S, 2) = (mse, regulariser) with w_pair != 0]
    const float* new_g;      // (S, n)  its gradient [(S, d p) = d/dXi under a coefficient map, see XiMap]
    float tol_grad;
    int l1;                  // != 0: new_loss / new_g are the bare data term, objective = w_x * data + w_reg * |params|_1
    float w_x, w_reg;
    int pair;                // != 0: new_loss holds (mse, regulariser) per problem, data term = mse + w_pair * regulariser
    float w_pair;
};

// The trainer's parametrisation of the coefficients (sindy.py:169-176): under the equivariance constraint the
// optimisation variables are [beta (r) | const (d)] and Xi (d, p) = reshape(Q beta) (+ const in column 0); q holds Q with
// its rows permuted into Xi's (d, p) row-major order (the host does the view / transpose of sindy.py:171-173 once).  The
// closure kernel reads Xi and returns d/dXi, so the optimiser launch converts on both sides: gradient in (Q^T g_xi, and
// g_xi[:, 0] for const), coefficients out after every move.  q == nullptr: Xi IS the parameter vector.
struct XiMap {
    const float* q;          // (d p, r) row-major, or nullptr
    float* xi;               // (S, d p): Xi at the current parameters, rewritten after every move
    int r, dp, p, allow_const;
};

// Trainer-only extras (nullptr in the sweep's tensor-op callers): problems the epoch logic has finished are skipped by
// the BEGIN launch; the L1 norm of the parameters the last closure was evaluated at is kept for the epoch's log record.
struct TrainerHook {
    const unsigned char* done;
    float* l1_last;
};

constexpr int LB_PLAIN = 0;    // update only: g / loss already hold this iteration's values
constexpr int LB_ACCEPT = 1;   // finish the previous iteration (take the re-evaluated loss / gradient, stopping tests), then update
constexpr int LB_BEGIN = 2;    // first iteration of an optimiser step: take the closure's loss / gradient, optimality test, then update

template <int NC, int MODE>
__global__ __launch_bounds__(WAVE) void lbfgs_update_kernel(float* __restrict__ params, float* __restrict__ g,
                                                            float* __restrict__ loss, unsigned char* __restrict__ act,
                                                            long* __restrict__ n_iter, float* __restrict__ d,
                                                            float* __restrict__ t, float* __restrict__ old_dirs,
                                                            float* __restrict__ old_stps, float* __restrict__ ro,
                                                            long* __restrict__ head, long* __restrict__ count,
                                                            float* __restrict__ h_diag, float* __restrict__ prev_g,
                                                            float* __restrict__ prev_loss, int n, int H, float lr,
                                                            float tol_change, AcceptArgs acc, XiMap map, TrainerHook hook) {
    constexpr bool STAGED = NC > 0;
    constexpr bool ACCEPT = MODE == LB_ACCEPT, BEGIN = MODE == LB_BEGIN;
    __shared__ float al[LB_MAXH];
    __shared__ float cvt[WAVE * LB_MAXC];                    // coefficient map: d/dXi on the way in, parameters on the way out
    extern __shared__ float staged[];                        // STAGED: pad [n] | Y [H][n] | S [H][n] | ro [H] | 256 floats of padding
    const long s = blockIdx.x;
    const int lane = threadIdx.x;
    // Every per-problem scalar and state vector of this launch is requested HERE, in one batch, before the first of them is
    // waited for: the state was written by other launches (the closure's last workgroup, the previous update -- other
    // XCDs' L2), so each dependent trip costs ~1 us on a lone wave, and read where they are used (behind the early
    // returns and the stores of g and loss) they queued up as four to five trips -- most of the launch's fixed cost.
    // Reading them for a problem that turns out to have stopped is harmless.
    const unsigned char act_in = BEGIN ? (unsigned char)1 : act[s];
    const float t_in = t[s], hd_in = h_diag[s], prev_loss_in = prev_loss[s];
    const long ni_in = n_iter[s], count_in = count[s], head_in = head[s];
    float d_in[LB_MAXC], pg_in[LB_MAXC];
#pragma unroll
    for (int c = 0; c < LB_MAXC; ++c) {
        const int i = lane + WAVE * c;
        d_in[c] = i < n ? d[s * n + i] : 0.0f;
        pg_in[c] = i < n ? prev_g[s * n + i] : 0.0f;
    }
    if constexpr (BEGIN) {
        if (hook.done != nullptr && __builtin_amdgcn_readfirstlane((int)hook.done[s])) {   // finished by the epoch logic
            if (lane == 0) act[s] = 0;
            return;
        }
    } else {
        if (!__builtin_amdgcn_readfirstlane((int)act_in)) return;   // wave-uniform: this problem stopped earlier
    }
    float* const ldsY = staged + n;                          // (a row of padding in front: the loops prefetch row -1)
    float* const ldsS = ldsY + H * n;
    float* const ldsR = ldsS + H * n;
    float gv[LB_MAXC], q[LB_MAXC];
    float loss_s;
    if constexpr (ACCEPT || BEGIN) {
        const float tt = BEGIN ? 0.0f : t_in;
        float nl = acc.pair ? fmaf(acc.w_pair, acc.new_loss[2 * s + 1], acc.new_loss[2 * s]) : acc.new_loss[s];
        float gmax = 0.0f, dmax = 0.0f, p_l1 = 0.0f;
        if (map.q != nullptr) {                              // d/dXi (d p) -> LDS; each lane then forms its parameter's gradient
#pragma unroll
            for (int c = 0; c < LB_MAXC; ++c) {
                const int j = lane + WAVE * c;
                if (j < map.dp) cvt[j] = acc.new_g[s * map.dp + j];
            }
            __syncthreads();
        }
#pragma unroll
        for (int c = 0; c < LB_MAXC; ++c) {
            const int i = lane + WAVE * c;
            gv[c] = 0.0f;
            if (i < n) {
                float v;
                if (map.q == nullptr) {
                    v = acc.new_g[s * n + i];
                } else if (i < map.r) {                      // (Q^T g_xi)_i, summed in row order
                    v = 0.0f;
                    for (int j = 0; j < map.dp; ++j) v = fmaf(map.q[j * map.r + i], cvt[j], v);
                } else {                                     // const_i: column 0 of equation i (zero when the model does not read it)
                    v = map.allow_const ? cvt[(i - map.r) * map.p] : 0.0f;
                }
                if (acc.l1) {
                    const float pv = params[s * n + i];
                    const float sg = (float)(pv > 0.0f) - (float)(pv < 0.0f);
                    v = __fadd_rn(__fmul_rn(acc.w_x, v), __fmul_rn(acc.w_reg, sg));
                    p_l1 += fabsf(pv);
                }
                g[s * n + i] = v;
                gv[c] = v;
                const float av = fabsf(v), ad = BEGIN ? 1.0f : fabsf(d_in[c] * tt);
                gmax = (av != av || gmax != gmax) ? __builtin_nanf("") : fmaxf(gmax, av);
                dmax = (ad != ad || dmax != dmax) ? __builtin_nanf("") : fmaxf(dmax, ad);
            }
        }
        gmax = wave_amax(gmax);
        dmax = wave_amax(dmax);
        if (acc.l1) {
            const float l1 = wave_sum(p_l1);
            nl = __fadd_rn(__fmul_rn(acc.w_x, nl), __fmul_rn(acc.w_reg, l1));
            if (hook.l1_last != nullptr && lane == 0) hook.l1_last[s] = l1;
        }
        // BEGIN: torch's "optimal condition" at the top of step() -- `if flat_grad.abs().max() <= tolerance_grad: return`,
        // i.e. a NaN gradient does NOT stop the step (the parameters go to NaN and the epoch logic's NaN guard ends the
        // run, train.py:697); ACCEPT: its three tests after the re-evaluation.
        const bool stop = BEGIN ? (gmax <= acc.tol_grad)
                                : ((gmax <= acc.tol_grad) || (dmax <= tol_change) || (fabsf(nl - prev_loss_in) < tol_change));
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
    const long ni = (long)__builtin_amdgcn_readfirstlane((int)ni_in) + 1;          // (scalars: uniform loop bounds below)
    const bool first = ni == 1;
    float* Y = old_dirs + s * (long)H * n;
    float* Sx = old_stps + s * (long)H * n;
    float* R = ro + s * (long)H;
    int m = first ? 0 : __builtin_amdgcn_readfirstlane((int)count_in);
    int h0 = first ? 0 : __builtin_amdgcn_readfirstlane((int)head_in);
    float hd = first ? 1.0f : __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(hd_in)));
    bool fresh = false;                                      // a pair was stored in this launch (logical slot m - 1)
    if (!first) {                                            // "do lbfgs update (update memory)"
        const float told = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(t_in)));
        float y[LB_MAXC], sv[LB_MAXC], p_ys = 0.0f, p_yy = 0.0f;
#pragma unroll
        for (int c = 0; c < LB_MAXC; ++c) {
            const int i = lane + WAVE * c;
            y[c] = i < n ? gv[c] - pg_in[c] : 0.0f;
            sv[c] = i < n ? d_in[c] * told : 0.0f;
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
    if (map.q != nullptr && live) __syncthreads();           // (cvt still holds d/dXi of this launch's first half)
#pragma unroll
    for (int c = 0; c < LB_MAXC; ++c) {
        const int i = lane + WAVE * c;
        if (i < n) {
            d[s * n + i] = q[c];
            prev_g[s * n + i] = gv[c];
            if (live) {
                const float moved = __fadd_rn(params[s * n + i], __fmul_rn(tn, q[c]));
                params[s * n + i] = moved;
                if (map.q != nullptr) cvt[i] = moved;
            }
        }
    }
    if (map.q != nullptr && live) {                          // Xi at the moved parameters for the next closure launch
        __syncthreads();
#pragma unroll
        for (int c = 0; c < LB_MAXC; ++c) {
            const int j = lane + WAVE * c;
            if (j < map.dp) {
                float v = 0.0f;
                for (int k = 0; k < map.r; ++k) v = fmaf(map.q[j * map.r + k], cvt[k], v);
                if (map.allow_const && j % map.p == 0) v += cvt[map.r + j / map.p];
                map.xi[s * map.dp + j] = v;
            }
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
inline int launch_lbfgs_update(int mode, float* params, float* g, float* loss, unsigned char* act, long* n_iter, float* d, float* t,
                               float* old_dirs, float* old_stps, float* ro, long* head, long* count, float* h_diag, float* prev_g,
                               float* prev_loss, long n_problems, int n, int history, float lr, float tol_change, AcceptArgs acc,
                               XiMap map, TrainerHook hook, void* stream) {
    if (n_problems < 1 || n < 1 || n > WAVE * LB_MAXC || history < 1 || history > LB_MAXH) return SYMODE_E_BADSIZE;
    if (!params || !g || !loss || !act || !n_iter || !d || !t || !old_dirs || !old_stps || !ro || !head || !count || !h_diag ||
        !prev_g || !prev_loss || (mode != LB_PLAIN && (!acc.new_loss || !acc.new_g)))
        return SYMODE_E_NULLPTR;
    if (map.q != nullptr && (!map.xi || map.dp < 1 || map.dp > WAVE * LB_MAXC || map.r < 0 || map.r > n || map.p < 1)) return SYMODE_E_BADSIZE;
    const size_t stage_bytes = (((size_t)2 * n + 1) * history + n + 256) * sizeof(float);
    const int nc = stage_bytes <= 60 * 1024 ? (n + WAVE - 1) / WAVE : 0;
#define SYMODE_LBFGS_LAUNCH(NC_, MODE_, BYTES_)                                                                                   \
    lbfgs_update_kernel<NC_, MODE_><<<dim3((unsigned)n_problems), dim3(WAVE), BYTES_, (hipStream_t)stream>>>(                     \
        params, g, loss, act, n_iter, d, t, old_dirs, old_stps, ro, head, count, h_diag, prev_g, prev_loss, n, history, lr,       \
        tol_change, acc, map, hook)
#define SYMODE_LBFGS_UPDATE(NC_, BYTES_)                                                                                          \
    do {                                                                                                                          \
        if (mode == LB_ACCEPT) SYMODE_LBFGS_LAUNCH(NC_, LB_ACCEPT, BYTES_);                                                       \
        else if (mode == LB_BEGIN) SYMODE_LBFGS_LAUNCH(NC_, LB_BEGIN, BYTES_);                                                    \
        else SYMODE_LBFGS_LAUNCH(NC_, LB_PLAIN, BYTES_);                                                                          \
    } while (0)
    switch (nc) {
        case 1: SYMODE_LBFGS_UPDATE(1, stage_bytes); break;
        case 2: SYMODE_LBFGS_UPDATE(2, stage_bytes); break;
        case 3: SYMODE_LBFGS_UPDATE(3, stage_bytes); break;
        case 4: SYMODE_LBFGS_UPDATE(4, stage_bytes); break;
        default: SYMODE_LBFGS_UPDATE(0, 0); break;
    }
#undef SYMODE_LBFGS_UPDATE
#undef SYMODE_LBFGS_LAUNCH
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SYMODE_OK : (int)e;
}
}  // namespace symode

extern "C" int symode_lbfgs_update(float* params, const float* g, const float* loss, unsigned char* act, long* n_iter, float* d,
                                   float* t, float* old_dirs, float* old_stps, float* ro, long* head, long* count,
                                   float* h_diag, float* prev_g, float* prev_loss, long n_problems, int n, int history,
                                   float lr, float tol_change, void* stream) {
    using namespace symode;
    return launch_lbfgs_update(LB_PLAIN, params, const_cast<float*>(g), const_cast<float*>(loss), act, n_iter, d, t, old_dirs,
                               old_stps, ro, head, count, h_diag, prev_g, prev_loss, n_problems, n, history, lr, tol_change,
                               AcceptArgs{nullptr, nullptr, 0.0f, 0, 1.0f, 0.0f, 0, 0.0f}, XiMap{nullptr, nullptr, 0, 0, 1, 0},
                               TrainerHook{nullptr, nullptr}, stream);
}

extern "C" int symode_lbfgs_accept_update(const float* new_loss, const float* new_g, float tol_grad, int l1, float w_x, float w_reg,
                                          float* params, float* g, float* loss, unsigned char* act, long* n_iter, float* d,
                                          float* t, float* old_dirs, float* old_stps, float* ro, long* head, long* count,
                                          float* h_diag, float* prev_g, float* prev_loss, long n_problems, int n, int history,
                                          float lr, float tol_change, void* stream) {
    using namespace symode;
    return launch_lbfgs_update(LB_ACCEPT, params, g, loss, act, n_iter, d, t, old_dirs, old_stps, ro, head, count, h_diag, prev_g,
                               prev_loss, n_problems, n, history, lr, tol_change,
                               AcceptArgs{new_loss, new_g, tol_grad, l1, w_x, w_reg, 0, 0.0f}, XiMap{nullptr, nullptr, 0, 0, 1, 0},
                               TrainerHook{nullptr, nullptr}, stream);
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

// =====================================================================================================================
// Device-resident trainer (include/symode.h, "Device-resident L-BFGS trainer")
// =====================================================================================================================
namespace symode {

enum TrainerField {
    TF_PARAMS, TF_XI, TF_MASK, TF_CL_LOSS, TF_CL_GRAD, TF_G, TF_LOSS, TF_ACT, TF_N_ITER, TF_D, TF_T, TF_OLD_DIRS, TF_OLD_STPS,
    TF_RO, TF_HEAD, TF_COUNT, TF_H_DIAG, TF_PREV_G, TF_PREV_LOSS, TF_PREV, TF_PPREV, TF_N_ITERS, TF_DONE, TF_NAN, TF_FINISHED,
    TF_EPOCHS, TF_NEAR, TF_L1_LAST, TF_TEST_GRAD, TF_COUNT_FIELDS
};
static_assert(TF_COUNT_FIELDS == SYMODE_TRAINER_FIELDS, "symode.h and the trainer disagree on the state layout");

inline size_t trainer_layout(long S, int n, int dp, int H, bool constrained, size_t* off) {
    const size_t f = sizeof(float);
    const size_t bytes[TF_COUNT_FIELDS] = {
        (size_t)S * n * f,              // params
        constrained ? (size_t)S * dp * f : 0,   // xi (unconstrained: the params array)
        (size_t)S * dp * f,             // mask
        (size_t)S * 2 * f,              // cl_loss
        (size_t)S * dp * f,             // cl_grad
        (size_t)S * n * f,              // g
        (size_t)S * f,                  // loss
        (size_t)S,                      // act
        (size_t)S * 8,                  // n_iter
        (size_t)S * n * f,              // d
        (size_t)S * f,                  // t
        (size_t)S * H * n * f,          // old_dirs
        (size_t)S * H * n * f,          // old_stps
        (size_t)S * H * f,              // ro
        (size_t)S * 8,                  // head
        (size_t)S * 8,                  // count
        (size_t)S * f,                  // h_diag
        (size_t)S * n * f,              // prev_g
        (size_t)S * f,                  // prev_loss
        (size_t)S * n * f,              // prev
        (size_t)S * n * f,              // pprev
        (size_t)S * 4,                  // n_iters
        (size_t)S, (size_t)S, (size_t)S,   // done, nan, finished
        (size_t)S * 4,                  // epochs
        (size_t)S * 4,                  // near
        (size_t)S * f,                  // l1_last
        (size_t)S * dp * f,             // test_grad
    };
    size_t at = 0;
    for (int k = 0; k < TF_COUNT_FIELDS; ++k) {
        size_t here = at;
        if (k == TF_XI && !constrained) here = 0;            // alias of params
        if (k == TF_CL_GRAD) here = at;                      // (cl_loss is S * 8 bytes: cl_grad follows it without a gap)
        if (off) off[k] = here;
        at += bytes[k];
        if (k != TF_CL_LOSS) at = (at + 255) & ~(size_t)255;
    }
    return at;
}

struct TrainerState {
    float *params, *xi, *mask, *cl_loss, *cl_grad, *g, *loss;
    unsigned char* act;
    long* n_iter;
    float *d, *t, *old_dirs, *old_stps, *ro;
    long *head, *count;
    float *h_diag, *prev_g, *prev_loss, *prev, *pprev;
    int* n_iters;
    unsigned char *done, *nan, *finished;
    int *epochs, *near;
    float *l1_last, *test_grad;
};

inline int trainer_state(const symode_trainer* T, TrainerState& st, int& dp) {
    if (!T) return SYMODE_E_NULLPTR;
    const int p = symode_lib_size(T->d, T->order, T->flags);
    if (p < 0) return SYMODE_E_UNSUPPORTED;
    dp = T->d * p;
    const bool con = T->q_eff != nullptr;
    if (T->n_problems < 1 || T->n_problems > 65535 || T->n_points < 1 || T->n_params != (con ? T->r + T->d : dp) || T->n_params > WAVE * LB_MAXC ||
        dp > WAVE * LB_MAXC || T->history < 1 || T->history > LB_MAXH || T->max_iter < 1 || T->log_epochs < 1 || (con && T->r < 1))
        return SYMODE_E_BADSIZE;
    if (!T->x || !T->dx || !T->state || !T->log || !T->log_test || !T->workspace || (T->gx && !T->jgx)) return SYMODE_E_NULLPTR;
    size_t off[TF_COUNT_FIELDS];
    if (T->state_bytes < trainer_layout(T->n_problems, T->n_params, dp, T->history, con, off)) return SYMODE_E_WORKSPACE;
    if (((uintptr_t)T->state % 256) != 0) return SYMODE_E_ALIGN;
    char* b = (char*)T->state;
    st.params = (float*)(b + off[TF_PARAMS]);       st.xi = (float*)(b + off[TF_XI]);           st.mask = (float*)(b + off[TF_MASK]);
    st.cl_loss = (float*)(b + off[TF_CL_LOSS]);     st.cl_grad = (float*)(b + off[TF_CL_GRAD]); st.g = (float*)(b + off[TF_G]);
    st.loss = (float*)(b + off[TF_LOSS]);           st.act = (unsigned char*)(b + off[TF_ACT]); st.n_iter = (long*)(b + off[TF_N_ITER]);
    st.d = (float*)(b + off[TF_D]);                 st.t = (float*)(b + off[TF_T]);             st.old_dirs = (float*)(b + off[TF_OLD_DIRS]);
    st.old_stps = (float*)(b + off[TF_OLD_STPS]);   st.ro = (float*)(b + off[TF_RO]);           st.head = (long*)(b + off[TF_HEAD]);
    st.count = (long*)(b + off[TF_COUNT]);          st.h_diag = (float*)(b + off[TF_H_DIAG]);   st.prev_g = (float*)(b + off[TF_PREV_G]);
    st.prev_loss = (float*)(b + off[TF_PREV_LOSS]); st.prev = (float*)(b + off[TF_PREV]);       st.pprev = (float*)(b + off[TF_PPREV]);
    st.n_iters = (int*)(b + off[TF_N_ITERS]);       st.done = (unsigned char*)(b + off[TF_DONE]); st.nan = (unsigned char*)(b + off[TF_NAN]);
    st.finished = (unsigned char*)(b + off[TF_FINISHED]); st.epochs = (int*)(b + off[TF_EPOCHS]); st.near = (int*)(b + off[TF_NEAR]);
    st.l1_last = (float*)(b + off[TF_L1_LAST]);     st.test_grad = (float*)(b + off[TF_TEST_GRAD]);
    return SYMODE_OK;
}

// h_diag = 1, mask = 1 (when no mask was handed over), Xi from the start parameters
__global__ __launch_bounds__(WAVE) void trainer_init_kernel(const float* __restrict__ params, float* __restrict__ h_diag,
                                                            float* __restrict__ mask, int fill_mask, XiMap map, int n) {
    __shared__ float cvt[WAVE * LB_MAXC];
    const long s = blockIdx.x;
    const int lane = threadIdx.x;
    if (lane == 0) h_diag[s] = 1.0f;
    if (fill_mask)
        for (int j = lane; j < map.dp; j += WAVE) mask[s * map.dp + j] = 1.0f;
    if (map.q == nullptr) return;
    for (int i = lane; i < n; i += WAVE) cvt[i] = params[s * n + i];
    __syncthreads();
    for (int j = lane; j < map.dp; j += WAVE) {
        float v = 0.0f;
        for (int k = 0; k < map.r; ++k) v = fmaf(map.q[j * map.r + k], cvt[k], v);
        if (map.allow_const && j % map.p == 0) v += cvt[map.r + j / map.p];
        map.xi[s * map.dp + j] = v;
    }
}

struct EpochArgs {
    float *params, *prev, *pprev, *xi, *mask;
    long *n_iter, *head, *count;
    float* h_diag;
    int* n_iters;
    unsigned char *done, *nan, *finished;
    int *epochs, *near;
    const float *cl_loss, *l1_last;
    float *log, *log_xi, *log_mask, *log_params;
    int n, r, dp, pair, st_freq, epoch, slot;
    float threshold, tol_update, near_band;
};

// The per-epoch logic of train.py:697-725 for one problem per wavefront (statement numbers of the reference in comments).
__global__ __launch_bounds__(WAVE) void trainer_epoch_kernel(EpochArgs a) {
    const long s = blockIdx.x;
    const long S = gridDim.x;
    const int lane = threadIdx.x;
    float* rec = a.log + ((long)a.slot * S + s) * 8;
    if (__builtin_amdgcn_readfirstlane((int)a.done[s])) {
        if (lane == 0) { rec[0] = -1.0f; rec[7] = (float)a.epoch; }
        return;
    }
    const int n = a.n, r_split = a.r > 0 ? a.r : n;           // constrained: two parameter tensors, beta (r) and const (d)
    float sq1a = 0.0f, sq1b = 0.0f, sq2a = 0.0f, sq2b = 0.0f;
    float pv[LB_MAXC];
    int bad = 0;
#pragma unroll
    for (int c = 0; c < LB_MAXC; ++c) {
        const int i = lane + WAVE * c;
        pv[c] = 0.0f;
        if (i < n) {
            const float v = a.params[s * n + i];
            pv[c] = v;
            bad |= (v != v);
            const float d1 = v - a.prev[s * n + i], d2 = v - a.pprev[s * n + i];
            if (i < r_split) { sq1a = fmaf(d1, d1, sq1a); sq2a = fmaf(d2, d2, sq2a); }
            else             { sq1b = fmaf(d1, d1, sq1b); sq2b = fmaf(d2, d2, sq2b); }
        }
    }
    const int n_it = a.n_iters[s] + 1;                                                   // :694
    const bool is_nan = __builtin_amdgcn_readfirstlane((int)(__ballot(bad) != 0ull));    // :697
    const float upd = sqrtf(wave_sum(sq1a)) + sqrtf(wave_sum(sq1b));                     // :702-704 (sum of per-tensor norms)
    const float upd2 = sqrtf(wave_sum(sq2a)) + sqrtf(wave_sum(sq2b));                    // :706-708
    const bool conv = !is_nan && upd < a.tol_update;                                     // :705
    const bool final = conv && upd2 < a.tol_update;                                      // :709
    const bool thr_conv = conv && !final;
    const bool thr_freq = !is_nan && !conv && a.st_freq > 0 && n_it % a.st_freq == 0;    // :720
    const bool ev = thr_conv || thr_freq;
    int near_here = 0;
    if (ev) {                                                                            // :716 / :722  set_threshold (sindy.py:192-195)
#pragma unroll
        for (int c = 0; c < LB_MAXC; ++c) {
            const int j = lane + WAVE * c;
            if (j < a.dp) {
                const float av = fabsf(a.xi[s * a.dp + j]), m = a.mask[s * a.dp + j];
                near_here += (fabsf(av - a.threshold) < a.near_band) && (m > 0.0f);
                a.mask[s * a.dp + j] = (av > a.threshold && m > 0.0f) ? 1.0f : 0.0f;     // strict >, monotone
            }
        }
        near_here = (int)wave_sum((float)near_here);
    }
#pragma unroll
    for (int c = 0; c < LB_MAXC; ++c) {
        const int i = lane + WAVE * c;
        if (i < n) {
            if (thr_conv) a.pprev[s * n + i] = pv[c];                                    // :718
            if (!is_nan && !final) a.prev[s * n + i] = pv[c];                            // :725
        }
    }
    if (a.log_xi != nullptr) {
        float* lx = a.log_xi + ((long)a.slot * S + s) * a.dp;
        float* lm = a.log_mask + ((long)a.slot * S + s) * a.dp;
        for (int j = lane; j < a.dp; j += WAVE) {
            lx[j] = a.xi[s * a.dp + j];
            lm[j] = a.mask[s * a.dp + j];
        }
        float* lp = a.log_params + ((long)a.slot * S + s) * n;
        for (int i = lane; i < n; i += WAVE) lp[i] = a.params[s * n + i];
    }
    if (lane == 0) {
        a.epochs[s] = a.epoch + 1;
        a.n_iters[s] = ev ? 0 : n_it;
        if (ev) {                                                                        // :717 / :723  a fresh optimiser
            a.n_iter[s] = 0;
            a.head[s] = 0;
            a.count[s] = 0;
            a.h_diag[s] = 1.0f;
            a.near[s] += near_here;
        }
        if (is_nan) { a.nan[s] = 1; a.done[s] = 1; }
        if (final) { a.finished[s] = 1; a.done[s] = 1; }
        rec[0] = is_nan ? 4.0f : final ? 3.0f : thr_conv ? 1.0f : thr_freq ? 2.0f : 0.0f;
        rec[1] = a.pair ? a.cl_loss[2 * s] : a.cl_loss[s];
        rec[2] = a.pair ? a.cl_loss[2 * s + 1] : 0.0f;
        rec[3] = a.l1_last[s];
        rec[4] = upd;
        rec[5] = upd2;
        rec[6] = (float)near_here;
        rec[7] = (float)a.epoch;
    }
}

inline XiMap trainer_map(const symode_trainer* T, const TrainerState& st, int dp) {
    return XiMap{T->q_eff, st.xi, T->r, dp, dp / T->d, T->allow_constant};
}

}  // namespace symode

extern "C" size_t symode_trainer_layout(long n_problems, int n_params, int dp, int history, int constrained, size_t* offsets_out) {
    if (n_problems < 1 || n_params < 1 || dp < 1 || history < 1) return 0;
    return symode::trainer_layout(n_problems, n_params, dp, history, constrained != 0, offsets_out);
}

extern "C" int symode_trainer_init(const symode_trainer* T, const float* params0, const float* mask0, void* stream) {
    using namespace symode;
    TrainerState st;
    int dp;
    if (int rc = trainer_state(T, st, dp)) return rc;
    if (!params0) return SYMODE_E_NULLPTR;
    hipStream_t hs = (hipStream_t)stream;
    const long S = T->n_problems;
    const size_t pb = (size_t)S * T->n_params * sizeof(float);
    hipError_t e = hipMemsetAsync(T->state, 0, T->state_bytes, hs);
    if (e == hipSuccess) e = hipMemcpyAsync(st.params, params0, pb, hipMemcpyDefault, hs);
    if (e == hipSuccess) e = hipMemcpyAsync(st.prev, params0, pb, hipMemcpyDefault, hs);
    if (e == hipSuccess) e = hipMemcpyAsync(st.pprev, params0, pb, hipMemcpyDefault, hs);
    if (e == hipSuccess && mask0) e = hipMemcpyAsync(st.mask, mask0, (size_t)S * dp * sizeof(float), hipMemcpyDefault, hs);
    if (e != hipSuccess) return (int)e;
    trainer_init_kernel<<<dim3((unsigned)S), dim3(WAVE), 0, hs>>>(st.params, st.h_diag, st.mask, mask0 ? 0 : 1, trainer_map(T, st, dp),
                                                                  T->n_params);
    e = hipGetLastError();
    return e == hipSuccess ? SYMODE_OK : (int)e;
}

extern "C" int symode_trainer_closure(const symode_trainer* T, float* loss_out, float* grad_out, void* stream) {
    using namespace symode;
    TrainerState st;
    int dp;
    if (int rc = trainer_state(T, st, dp)) return rc;
    float* lo = loss_out ? loss_out : st.cl_loss;
    float* go = grad_out ? grad_out : st.cl_grad;
    if (T->gx != nullptr)
        return symode_loss_grad_reversed(T->x, T->dx, T->gx, T->jgx, T->n_g, T->n_problems, T->n_points, T->d, T->order, T->flags,
                                         st.xi, st.mask, T->inv_count, T->w_sym, lo, go, T->workspace, T->workspace_bytes, stream);
    return symode_loss_grad(T->x, T->dx, T->n_problems, T->n_points, T->d, T->order, T->flags, st.xi, st.mask, T->inv_count, lo, go,
                            T->workspace, T->workspace_bytes, stream);
}

extern "C" int symode_trainer_update(const symode_trainer* T, int mode, void* stream) {
    using namespace symode;
    TrainerState st;
    int dp;
    if (int rc = trainer_state(T, st, dp)) return rc;
    if (mode != LB_ACCEPT && mode != LB_BEGIN) return SYMODE_E_BADSIZE;
    const int pair = T->gx != nullptr;
    return launch_lbfgs_update(mode, st.params, st.g, st.loss, st.act, st.n_iter, st.d, st.t, st.old_dirs, st.old_stps, st.ro, st.head,
                               st.count, st.h_diag, st.prev_g, st.prev_loss, T->n_problems, T->n_params, T->history, T->lr, T->tol_change,
                               AcceptArgs{st.cl_loss, st.cl_grad, T->tol_grad, 1, T->w_x, T->l1 ? T->w_reg : 0.0f, pair, T->w_sym},
                               trainer_map(T, st, dp), TrainerHook{st.done, st.l1_last}, stream);
}

extern "C" int symode_trainer_epoch_end(const symode_trainer* T, int epoch, void* stream) {
    using namespace symode;
    TrainerState st;
    int dp;
    if (int rc = trainer_state(T, st, dp)) return rc;
    if (epoch < 0) return SYMODE_E_BADSIZE;
    if ((T->log_xi == nullptr) != (T->log_mask == nullptr) || (T->log_xi == nullptr) != (T->log_params == nullptr)) return SYMODE_E_NULLPTR;
    EpochArgs a{st.params, st.prev, st.pprev, st.xi, st.mask, st.n_iter, st.head, st.count, st.h_diag, st.n_iters, st.done, st.nan,
                st.finished, st.epochs, st.near, st.cl_loss, st.l1_last, T->log, T->log_xi, T->log_mask, T->log_params, T->n_params,
                T->q_eff ? T->r : 0, dp, T->gx != nullptr, T->st_freq, epoch, epoch % T->log_epochs, T->threshold, T->tol_update,
                T->near_band};
    trainer_epoch_kernel<<<dim3((unsigned)T->n_problems), dim3(WAVE), 0, (hipStream_t)stream>>>(a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SYMODE_OK : (int)e;
}

extern "C" int symode_trainer_run(const symode_trainer* T, int epoch0, int n_epochs, int test_eval, void* stream) {
    using namespace symode;
    TrainerState st;
    int dp;
    if (int rc = trainer_state(T, st, dp)) return rc;
    if (epoch0 < 0 || n_epochs < 0) return SYMODE_E_BADSIZE;
    for (int e = epoch0; e < epoch0 + n_epochs; ++e) {
        for (int it = 0; it < T->max_iter; ++it) {           // torch.optim.LBFGS.step: max_iter moves, max_iter evaluations
            if (int rc = symode_trainer_closure(T, nullptr, nullptr, stream)) return rc;
            if (int rc = symode_trainer_update(T, it == 0 ? LB_BEGIN : LB_ACCEPT, stream)) return rc;
        }
        if (int rc = symode_trainer_epoch_end(T, e, stream)) return rc;
        if (test_eval) {
            const long slot = e % T->log_epochs;
            if (int rc = symode_trainer_closure(T, T->log_test + slot * T->n_problems * 2, st.test_grad, stream)) return rc;
        }
    }
    return SYMODE_OK;
}
