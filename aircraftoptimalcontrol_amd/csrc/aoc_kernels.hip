// aoc_kernels.hip — HIP kernels (gfx950) and the C-ABI of libaoc_hip.so (include/aoc.h).
//
// Execution model: one wavefront (64 lanes) per tile of 64 trajectories, one trajectory per lane,
// one workgroup = one wavefront.  Every pass walks the horizon sequentially inside the lane; the
// batch axis is the lane axis, so each (t, component) access of a wavefront is one coalesced 512-B
// segment and a wavefront streams through its own contiguous slab of every array.
// No MFMA (the blocks are 6x6/6x2), no LDS traffic in these first kernels: the per-lane state
// (P: 21, p: 6, lambda: 6 doubles, ...) lives in VGPRs.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/aoc.h"
#include "aoc_device.h"

using namespace aoc;

// State trajectories (tiled, 6 components) are stored either as fp64 or as float32.  Every
// propagated state of the reference is a float32 value (aircraft_simplified.py:300), so float32
// storage is lossless for samples t >= 1 and halves their HBM traffic; sample 0 is x0, an arbitrary
// fp64 value, and is always read from the separate fp64 x0 array ([ntiles][6][64]).
template <typename XT>
__device__ __forceinline__ void load_state(const XT* __restrict__ x, const double* __restrict__ x0, int tile, int T,
                                           int t, int lane, double xs[6]) {
    if (t == 0) {
#pragma unroll
        for (int c = 0; c < 6; c++) xs[c] = x0[((size_t)tile * 6 + c) * TILE + lane];
    } else {
#pragma unroll
        for (int c = 0; c < 6; c++) xs[c] = (double)x[tix<6>(tile, T, t, c, lane)];
    }
}

#define AOC_DISPATCH_BOOL(flag, NAME, ...)            \
    do {                                              \
        if (flag) { constexpr bool NAME = true; __VA_ARGS__; } \
        else { constexpr bool NAME = false; __VA_ARGS__; }      \
    } while (0)
#define AOC_DISPATCH_XT(f32, NAME, ...)               \
    do {                                              \
        if (f32) { using NAME = float; __VA_ARGS__; } \
        else { using NAME = double; __VA_ARGS__; }    \
    } while (0)

// ---------------------------------------------------------------------------------------------
// layout conversion
// ---------------------------------------------------------------------------------------------
template <typename ET>
__global__ void k_pack(int B, int T, int C, const double* __restrict__ src, ET* __restrict__ dst) {
    // one block per (tile, chunk of t); threads: lane fastest on the write side
    const int tile = blockIdx.x, lane = threadIdx.x;
    int b = tile * TILE + lane;
    if (b >= B) b = B - 1;
    for (int t = blockIdx.y; t < T; t += gridDim.y)
        for (int c = 0; c < C; c++)
            dst[(((size_t)tile * T + t) * C + c) * TILE + lane] = (ET)src[((size_t)b * C + c) * T + t];
}

template <typename ET>
__global__ void k_unpack(int B, int T, int C, const ET* __restrict__ src, double* __restrict__ dst) {
    const int tile = blockIdx.x, lane = threadIdx.x;
    const int b = tile * TILE + lane;
    if (b >= B) return;
    for (int t = blockIdx.y; t < T; t += gridDim.y)
        for (int c = 0; c < C; c++)
            dst[((size_t)b * C + c) * T + t] = (double)src[(((size_t)tile * T + t) * C + c) * TILE + lane];
}

// ---------------------------------------------------------------------------------------------
// unit-level kernels (AoS points, one per thread)
// ---------------------------------------------------------------------------------------------
__global__ void k_step_batch(KConst k, int n, const double* __restrict__ x, const double* __restrict__ u,
                             const double* __restrict__ lmbd, double* __restrict__ xp, double* __restrict__ fx,
                             double* __restrict__ fu, double* __restrict__ fxx, double* __restrict__ fuu,
                             double* __restrict__ fux) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double xs[6], xn[6];
    for (int c = 0; c < 6; c++) xs[c] = x[(size_t)i * 6 + c];
    const double u0 = u[(size_t)i * 2], u1 = u[(size_t)i * 2 + 1];
    const SC s = trig(xs[3], xs[5]);
    step_state(k, xs, u0, u1, s, xn);
    for (int c = 0; c < 6; c++) xp[(size_t)i * 6 + c] = xn[c];
    const Lin l = linearise(k, xs, u0, s);
    if (fx) {  // fx = A^T (aircraft_simplified.py:322)
        double A[36];
        for (int e = 0; e < 36; e++) A[e] = 0.0;
        A[0] = 1; A[7] = 1; A[21] = 1; A[28] = 1; A[3 * 6 + 4] = k.dt;
        A[0 * 6 + 2] = l.a02; A[0 * 6 + 5] = l.a05; A[1 * 6 + 2] = l.a12; A[1 * 6 + 5] = l.a15;
        A[2 * 6 + 2] = l.a22; A[2 * 6 + 3] = l.a23; A[2 * 6 + 5] = l.a25;
        A[5 * 6 + 2] = l.a52; A[5 * 6 + 3] = l.a53; A[5 * 6 + 5] = l.a55;
        for (int r = 0; r < 6; r++)
            for (int c = 0; c < 6; c++) fx[(size_t)i * 36 + r * 6 + c] = A[c * 6 + r];
    }
    if (fu) {
        for (int e = 0; e < 12; e++) fu[(size_t)i * 12 + e] = 0.0;
        fu[(size_t)i * 12 + 2] = l.b20; fu[(size_t)i * 12 + 5] = l.b50; fu[(size_t)i * 12 + 6 + 4] = k.b41;
    }
    if (lmbd) {
        double lam[6];
        for (int c = 0; c < 6; c++) lam[c] = lmbd[(size_t)i * 6 + c];
        const Hess h = hessian(k, xs, u0, s, lam);
        if (fxx) {
            double* F = fxx + (size_t)i * 36;
            for (int e = 0; e < 36; e++) F[e] = 0.0;
            F[2 * 6 + 2] = h.h22; F[2 * 6 + 3] = h.h23; F[3 * 6 + 2] = h.h23; F[2 * 6 + 5] = h.h25; F[5 * 6 + 2] = h.h25;
            F[3 * 6 + 3] = h.h33; F[3 * 6 + 5] = h.h35; F[5 * 6 + 3] = h.h35; F[5 * 6 + 5] = h.h55;
        }
        if (fux) {
            double* G = fux + (size_t)i * 12;
            for (int e = 0; e < 12; e++) G[e] = 0.0;
            G[2] = h.s02; G[3] = h.s03; G[5] = h.s05;
        }
        if (fuu) for (int e = 0; e < 4; e++) fuu[(size_t)i * 4 + e] = 0.0;
    }
}

template <bool DIAG>
__global__ void k_cost_batch(KConst k, int n, const double* __restrict__ x, const double* __restrict__ u,
                             const double* __restrict__ xr, const double* __restrict__ ur, double* __restrict__ ll,
                             double* __restrict__ lx, double* __restrict__ lu, double* __restrict__ llT,
                             double* __restrict__ lTx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double xs[6], ref[8], q[6], r[2];
    for (int c = 0; c < 6; c++) { xs[c] = x[(size_t)i * 6 + c]; ref[c] = xr[(size_t)i * 6 + c]; }
    const double u0 = u[(size_t)i * 2], u1 = u[(size_t)i * 2 + 1];
    ref[6] = ur[(size_t)i * 2]; ref[7] = ur[(size_t)i * 2 + 1];
    const double l = stage_cost<DIAG>(k, xs, u0, u1, ref, q, r);
    if (ll) ll[i] = l;
    if (lx) for (int c = 0; c < 6; c++) lx[(size_t)i * 6 + c] = q[c];
    if (lu) { lu[(size_t)i * 2] = r[0]; lu[(size_t)i * 2 + 1] = r[1]; }
    const double lT = term_cost<DIAG>(k, xs, ref, q);
    if (llT) llT[i] = lT;
    if (lTx) for (int c = 0; c < 6; c++) lTx[(size_t)i * 6 + c] = q[c];
}

// ---------------------------------------------------------------------------------------------
// pass-level kernels: one wavefront per tile
// ---------------------------------------------------------------------------------------------

// cost of a stored trajectory, t ascending then terminal (optcon.py:417-424)
template <bool DIAG, typename XT>
__global__ __launch_bounds__(TILE) void k_traj_cost(KConst k, const double* __restrict__ ref,
                                                    const XT* __restrict__ x, const double* __restrict__ u,
                                                    const double* __restrict__ x0, double* __restrict__ J) {
    const int tile = blockIdx.x, lane = threadIdx.x, T = k.T;
    double JJ = 0.0, xs[6], q[6], r[2];
    for (int t = 0; t < T - 1; t++) {
        load_state(x, x0, tile, T, t, lane, xs);
        const double u0 = u[tix<2>(tile, T, t, 0, lane)], u1 = u[tix<2>(tile, T, t, 1, lane)];
        JJ += stage_cost<DIAG>(k, xs, u0, u1, ref + (size_t)t * 8, q, r);
    }
    load_state(x, x0, tile, T, T - 1, lane, xs);
    JJ += term_cost<DIAG>(k, xs, ref + (size_t)(T - 1) * 8, q);
    J[tile * TILE + lane] = JJ;
}

// One nonlinear rollout with cost for per-lane step `a` (get_update + trial cost).
// WRITE: store x',u'.  wmask: lane writes only if true.
// The loop carries a strictly serial chain (x_t -> x_{t+1}), so the only way to keep HBM busy from
// one wavefront is to run the (u, du) loads far ahead: a register ring of ROLL_PF stages.
#ifndef AOC_ROLL_PF
#define AOC_ROLL_PF 8
#endif
constexpr int ROLL_PF = AOC_ROLL_PF;

template <bool DIAG, bool WRITE, typename XO>
__device__ __forceinline__ double rollout(const KConst& k, const double* __restrict__ ref, int tile, int lane,
                                          const double x0[6], const double* __restrict__ u,
                                          const double* __restrict__ du, double a, bool wmask,
                                          XO* __restrict__ x_out, double* __restrict__ u_out, int& flags) {
    const int T = k.T;
    double xs[6], xn[6], q[6], r[2];
    double JJ = 0.0;
#pragma unroll
    for (int c = 0; c < 6; c++) xs[c] = x0[c];
    if (WRITE && wmask) {
#pragma unroll
        for (int c = 0; c < 6; c++) x_out[tix<6>(tile, T, 0, c, lane)] = (XO)xs[c];
    }
    double ub[ROLL_PF][2], db[ROLL_PF][2];
#pragma unroll
    for (int i = 0; i < ROLL_PF; i++) {
        const int t = i < T - 1 ? i : T - 2;
        ub[i][0] = u[tix<2>(tile, T, t, 0, lane)];
        ub[i][1] = u[tix<2>(tile, T, t, 1, lane)];
        db[i][0] = du ? du[tix<2>(tile, T, t, 0, lane)] : 0.0;
        db[i][1] = du ? du[tix<2>(tile, T, t, 1, lane)] : 0.0;
    }
    for (int t0 = 0; t0 < T - 1; t0 += ROLL_PF) {
#pragma unroll
        for (int i = 0; i < ROLL_PF; i++) {
            const int t = t0 + i;
            if (t >= T - 1) break;
            const double uc0 = ub[i][0], uc1 = ub[i][1], dc0 = db[i][0], dc1 = db[i][1];
            {   // refill this slot with stage t + ROLL_PF (clamped: the tail re-reads the last stage)
                const int tn = t + ROLL_PF < T - 1 ? t + ROLL_PF : T - 2;
                ub[i][0] = u[tix<2>(tile, T, tn, 0, lane)];
                ub[i][1] = u[tix<2>(tile, T, tn, 1, lane)];
                if (du) {
                    db[i][0] = du[tix<2>(tile, T, tn, 0, lane)];
                    db[i][1] = du[tix<2>(tile, T, tn, 1, lane)];
                }
            }
            double u0, u1;
            {
#pragma clang fp contract(off)
                u0 = uc0 + a * dc0;  // optcon.py:197 / :253
                u1 = uc1 + a * dc1;
            }
            JJ += stage_cost<DIAG>(k, xs, u0, u1, ref + (size_t)t * 8, q, r);
            if (WRITE && !(xs[2] > 0.0)) flags |= AOC_ST_VNONPOS;  // trials are silent: only the update reports
            const SC s = trig(xs[3], xs[5]);
            step_state(k, xs, u0, u1, s, xn);
            if (WRITE && wmask) {
                u_out[tix<2>(tile, T, t, 0, lane)] = u0;
                u_out[tix<2>(tile, T, t, 1, lane)] = u1;
#pragma unroll
                for (int c = 0; c < 6; c++) x_out[tix<6>(tile, T, t + 1, c, lane)] = (XO)xn[c];
            }
#pragma unroll
            for (int c = 0; c < 6; c++) xs[c] = xn[c];
        }
    }
    JJ += term_cost<DIAG>(k, xs, ref + (size_t)(T - 1) * 8, q);
    if (WRITE && wmask) {
        u_out[tix<2>(tile, T, T - 1, 0, lane)] = 0.0;  // optcon.py:193: uu_temp[:, T-1] stays 0
        u_out[tix<2>(tile, T, T - 1, 1, lane)] = 0.0;
    }
    return JJ;
}

template <bool DIAG, bool WRITE, typename XO>
__global__ __launch_bounds__(TILE) void k_rollout_cost(KConst k, const double* __restrict__ ref,
                                                       const double* __restrict__ x0, const double* __restrict__ u,
                                                       const double* __restrict__ du, const double* __restrict__ alpha,
                                                       XO* __restrict__ x_out, double* __restrict__ u_out,
                                                       double* __restrict__ J_out, int* __restrict__ status) {
    const int tile = blockIdx.x, lane = threadIdx.x, b = tile * TILE + lane;
    double xs[6];
#pragma unroll
    for (int c = 0; c < 6; c++) xs[c] = x0[((size_t)tile * 6 + c) * TILE + lane];
    const double a = (du && alpha) ? alpha[b] : 0.0;
    int flags = 0;
    const double JJ = rollout<DIAG, WRITE, XO>(k, ref, tile, lane, xs, u, du, a, true, x_out, u_out, flags);
    if (JJ != JJ || JJ - JJ != 0.0) flags |= AOC_ST_NAN;
    J_out[b] = JJ;
    if (status && flags) status[b] |= flags;
}

// Dynamics.get_initial_trajectory (aircraft_simplified.py:126-148): P-controller rollout from x0.
template <typename XO>
__global__ __launch_bounds__(TILE) void k_initial_traj(KConst k, double kp, double kt, const double* __restrict__ ref,
                                                       const double* __restrict__ x0, XO* __restrict__ x_out,
                                                       double* __restrict__ u_out) {
    const int tile = blockIdx.x, lane = threadIdx.x, T = k.T;
    double xs[6], xn[6];
#pragma unroll
    for (int c = 0; c < 6; c++) {
        xs[c] = x0[((size_t)tile * 6 + c) * TILE + lane];
        x_out[tix<6>(tile, T, 0, c, lane)] = (XO)xs[c];
    }
    for (int i = 0; i < T - 1; i++) {
        const double* xr = ref + (size_t)(i + 1) * 8;
        double u0, u1;
        {
#pragma clang fp contract(off)
            u0 = kp * ((xs[0] - xr[0]) + (xs[1] - xr[1]));  // :143
            u1 = kt * ((xs[3] - xr[3]) + (xs[5] - xr[5]));  // :144
        }
        const SC s = trig(xs[3], xs[5]);
        step_state(k, xs, u0, u1, s, xn);
        u_out[tix<2>(tile, T, i, 0, lane)] = u0;
        u_out[tix<2>(tile, T, i, 1, lane)] = u1;
#pragma unroll
        for (int c = 0; c < 6; c++) { x_out[tix<6>(tile, T, i + 1, c, lane)] = (XO)xn[c]; xs[c] = xn[c]; }
    }
    u_out[tix<2>(tile, T, T - 1, 0, lane)] = 0.0;
    u_out[tix<2>(tile, T, T - 1, 1, lane)] = 0.0;
}

// Backward pass (see aoc_backward in include/aoc.h).
// The (x,u) loads of the next BW_PF stages are kept in flight in a register ring.
#ifndef AOC_BW_PF
#define AOC_BW_PF 1
#endif
constexpr int BW_PF = AOC_BW_PF;

template <bool DIAG, bool FULL, typename XT>
__global__ __launch_bounds__(TILE) void k_backward(KConst k, const double* __restrict__ ref,
                                                   const XT* __restrict__ x, const double* __restrict__ u,
                                                   const double* __restrict__ x0, double* __restrict__ Kt,
                                                   double* __restrict__ g,
                                                   double* __restrict__ lmbd0, int* __restrict__ status) {
    const int tile = blockIdx.x, lane = threadIdx.x, T = k.T;
    double P[21], p[6], lam[6], Qb[21], xs[6], q[6], r[2];
    int flags = 0;
    // terminal condition (optcon.py:429-432, :688-690, :716): P = Q_T, p = q_f/2, lambda = q_f
    load_state(x, x0, tile, T, T - 1, lane, xs);
    term_cost<DIAG>(k, xs, ref + (size_t)(T - 1) * 8, q);
#pragma unroll
    for (int i = 0; i < 6; i++) {
        lam[i] = q[i];
        p[i] = 0.5 * q[i];
#pragma unroll
        for (int j = i; j < 6; j++) {
            P[sidx(i, j)] = k.QT[i * 6 + j];
            Qb[sidx(i, j)] = k.Q[i * 6 + j];
        }
    }
    XT xb[BW_PF][6];  // kept in storage precision: a float32 ring costs half the registers
    double ubuf[BW_PF][2];
#pragma unroll
    for (int i = 0; i < BW_PF; i++) {
        const int tp = T - 2 - i >= 0 ? T - 2 - i : 0;
#pragma unroll
        for (int c = 0; c < 6; c++) xb[i][c] = x[tix<6>(tile, T, tp, c, lane)];
        ubuf[i][0] = u[tix<2>(tile, T, tp, 0, lane)];
        ubuf[i][1] = u[tix<2>(tile, T, tp, 1, lane)];
    }
    for (int tb = T - 2; tb >= 0; tb -= BW_PF) {
#pragma unroll
      for (int i = 0; i < BW_PF; i++) {
        const int t = tb - i;
        if (t < 0) break;
        if (t == 0) {  // sample 0 is x0 (fp64), see load_state
            load_state(x, x0, tile, T, 0, lane, xs);
        } else {
#pragma unroll
            for (int c = 0; c < 6; c++) xs[c] = (double)xb[i][c];
        }
        const double u0 = ubuf[i][0], u1 = ubuf[i][1];
        {   // refill this slot with stage t - BW_PF (clamped at 0: the head re-reads stage 0)
            const int tn = t - BW_PF >= 0 ? t - BW_PF : 0;
#pragma unroll
            for (int c = 0; c < 6; c++) xb[i][c] = x[tix<6>(tile, T, tn, c, lane)];
            ubuf[i][0] = u[tix<2>(tile, T, tn, 0, lane)];
            ubuf[i][1] = u[tix<2>(tile, T, tn, 1, lane)];
        }
        stage_cost<DIAG>(k, xs, u0, u1, ref + (size_t)t * 8, q, r);  // q = l_x, r = l_u (optcon.py:436)
        const SC s = trig(xs[3], xs[5]);
        const Lin l = linearise(k, xs, u0, s);
        // g = B^T lambda_{t+1} + r   (optcon.py:475)
        const double g0 = l.b20 * lam[2] + l.b50 * lam[5] + r[0];
        const double g1 = k.b41 * lam[4] + r[1];
        double Qs[21];
#pragma unroll
        for (int e = 0; e < 21; e++) Qs[e] = Qb[e];
        double s02 = 0.0, s03 = 0.0, s05 = 0.0;
        if (FULL) {  // kk > 8: Q += fxx.lambda, S = fux.lambda (optcon.py:443-446)
            const Hess h = hessian(k, xs, u0, s, lam);
            Qs[sidx(2, 2)] += h.h22; Qs[sidx(2, 3)] += h.h23; Qs[sidx(2, 5)] += h.h25;
            Qs[sidx(3, 3)] += h.h33; Qs[sidx(3, 5)] += h.h35; Qs[sidx(5, 5)] += h.h55;
            s02 = h.s02; s03 = h.s03; s05 = h.s05;
        }
        double hq[6], hr[2], Ks[14];
#pragma unroll
        for (int i = 0; i < 6; i++) hq[i] = 0.5 * q[i];   // optcon.py:673-674 (Q2)
        hr[0] = 0.5 * r[0]; hr[1] = 0.5 * r[1];           // optcon.py:679
        const StageFlags fl = lqr_stage(k, l, P, p, Qs, s02, s03, s05, hq, hr, Ks);
        if (fl.singular) flags |= AOC_ST_SINGULAR;
        if (fl.regularised) flags |= AOC_ST_REGULARISED;
        // costate: lambda_t = A^T lambda_{t+1} + l_x   (optcon.py:461)
        double al[6];
        At_vec(k, l, lam, al);
#pragma unroll
        for (int i = 0; i < 6; i++) lam[i] = al[i] + q[i];
#pragma unroll
        for (int c = 0; c < 14; c++) Kt[tix<14>(tile, T, t, c, lane)] = Ks[c];
        g[tix<2>(tile, T, t, 0, lane)] = g0;
        g[tix<2>(tile, T, t, 1, lane)] = g1;
      }
    }
    if (lmbd0) {
#pragma unroll
        for (int c = 0; c < 6; c++) lmbd0[((size_t)tile * 6 + c) * TILE + lane] = lam[c];
    }
    if (status && flags) status[tile * TILE + lane] |= flags;
}

// Forward pass (see aoc_forward in include/aoc.h).  NSPEC = number of Armijo candidate steps
// (alpha_0 .. alpha_{NSPEC-1}) whose trial rollouts ride along: the pass is bound by the K~ stream
// from HBM, so a second serial chain in the same lane is nearly free and saves a whole
// latency-bound trial round later.  Operands of the next FW_PF stages are in flight in a register ring.
#ifndef AOC_FW_PF
#define AOC_FW_PF 2
#endif
constexpr int FW_PF = AOC_FW_PF;

template <bool DIAG, int NSPEC, typename XT>
__global__ __launch_bounds__(TILE) void k_forward(KConst k, aoc_params prm, const double* __restrict__ ref,
                                                  const XT* __restrict__ x, const double* __restrict__ u,
                                                  const double* __restrict__ x0, const double* __restrict__ Kt,
                                                  const double* __restrict__ g, double* __restrict__ du_out,
                                                  double* __restrict__ descent, double* __restrict__ J_trial,
                                                  int* __restrict__ status) {
    const int tile = blockIdx.x, lane = threadIdx.x, b = tile * TILE + lane, T = k.T;
    const int Bp = k.ntiles * TILE;
    double dx[6], xp[NSPEC][6], q[6], r[2], JJ[NSPEC], alpha[NSPEC];
    int flags = 0;
    alpha[0] = prm.stepsize_0;
#pragma unroll
    for (int j = 1; j < NSPEC; j++) alpha[j] = prm.beta * alpha[j - 1];  // optcon.py:270
#pragma unroll
    for (int c = 0; c < 6; c++) {
        dx[c] = 0.0;  // ltv_LQR is called with x0 = 0 (optcon.py:470)
        const double v = x0[((size_t)tile * 6 + c) * TILE + lane];
#pragma unroll
        for (int j = 0; j < NSPEC; j++) xp[j][c] = v;
    }
#pragma unroll
    for (int j = 0; j < NSPEC; j++) JJ[j] = 0.0;
    double desc = 0.0;
    double Kb[FW_PF][14], gb[FW_PF][2], ub[FW_PF][2];
    XT xb[FW_PF][6];
#pragma unroll
    for (int i = 0; i < FW_PF; i++) {
        const int tp = i < T - 1 ? i : T - 2;
#pragma unroll
        for (int c = 0; c < 14; c++) Kb[i][c] = Kt[tix<14>(tile, T, tp, c, lane)];
        gb[i][0] = g[tix<2>(tile, T, tp, 0, lane)]; gb[i][1] = g[tix<2>(tile, T, tp, 1, lane)];
#pragma unroll
        for (int c = 0; c < 6; c++) xb[i][c] = x[tix<6>(tile, T, tp, c, lane)];
        ub[i][0] = u[tix<2>(tile, T, tp, 0, lane)]; ub[i][1] = u[tix<2>(tile, T, tp, 1, lane)];
    }
    for (int t0 = 0; t0 < T - 1; t0 += FW_PF) {
#pragma unroll
      for (int i = 0; i < FW_PF; i++) {
        const int t = t0 + i;
        if (t >= T - 1) break;
        double Kc[14], xs[6];
#pragma unroll
        for (int c = 0; c < 14; c++) Kc[c] = Kb[i][c];
        if (t == 0) {  // sample 0 is x0 (fp64), see load_state
            load_state(x, x0, tile, T, 0, lane, xs);
        } else {
#pragma unroll
            for (int c = 0; c < 6; c++) xs[c] = (double)xb[i][c];
        }
        const double g0 = gb[i][0], g1 = gb[i][1], uc0 = ub[i][0], uc1 = ub[i][1];
        {   // refill this slot with stage t + FW_PF (clamped: the tail re-reads the last stage)
            const int tn = t + FW_PF < T - 1 ? t + FW_PF : T - 2;
#pragma unroll
            for (int c = 0; c < 14; c++) Kb[i][c] = Kt[tix<14>(tile, T, tn, c, lane)];
            gb[i][0] = g[tix<2>(tile, T, tn, 0, lane)]; gb[i][1] = g[tix<2>(tile, T, tn, 1, lane)];
#pragma unroll
            for (int c = 0; c < 6; c++) xb[i][c] = x[tix<6>(tile, T, tn, c, lane)];
            ub[i][0] = u[tix<2>(tile, T, tn, 0, lane)]; ub[i][1] = u[tix<2>(tile, T, tn, 1, lane)];
        }
        // du_t = K~_t [1; dx_t]   (optcon.py:759)
        double d0 = Kc[0], d1 = Kc[7];
#pragma unroll
        for (int j = 0; j < 6; j++) { d0 += Kc[1 + j] * dx[j]; d1 += Kc[8 + j] * dx[j]; }
        desc += g0 * d0 + g1 * d1;  // optcon.py:475-477
        // dx_{t+1} = A dx_t + B du_t   (optcon.py:760), A,B re-linearised at the nominal (x_t,u_t)
        {
            const SC s = trig(xs[3], xs[5]);
            const Lin l = linearise(k, xs, uc0, s);
            double ax[6];
            A_vec(k, l, dx, ax);
            dx[0] = ax[0]; dx[1] = ax[1];
            dx[2] = ax[2] + l.b20 * d0;
            dx[3] = ax[3];
            dx[4] = ax[4] + k.b41 * d1;
            dx[5] = ax[5] + l.b50 * d0;
        }
        du_out[tix<2>(tile, T, t, 0, lane)] = d0;
        du_out[tix<2>(tile, T, t, 1, lane)] = d1;
        // Armijo trials alpha_0 .. alpha_{NSPEC-1} (optcon.py:250-264), independent chains
#pragma unroll
        for (int j = 0; j < NSPEC; j++) {
            double u0, u1, xpn[6];
            {
#pragma clang fp contract(off)
                u0 = uc0 + alpha[j] * d0;
                u1 = uc1 + alpha[j] * d1;
            }
            JJ[j] += stage_cost<DIAG>(k, xp[j], u0, u1, ref + (size_t)t * 8, q, r);
            const SC s2 = trig(xp[j][3], xp[j][5]);
            step_state(k, xp[j], u0, u1, s2, xpn);
#pragma unroll
            for (int c = 0; c < 6; c++) xp[j][c] = xpn[c];
        }
      }
    }
    du_out[tix<2>(tile, T, T - 1, 0, lane)] = 0.0;
    du_out[tix<2>(tile, T, T - 1, 1, lane)] = 0.0;
#pragma unroll
    for (int j = 0; j < NSPEC; j++) {
        JJ[j] += term_cost<DIAG>(k, xp[j], ref + (size_t)(T - 1) * 8, q);
        J_trial[(size_t)j * Bp + b] = JJ[j];
    }
    if (desc != desc || desc - desc != 0.0) flags |= AOC_ST_NAN;
    descent[b] = desc;
    if (status && flags) status[b] |= flags;
}

// ---------------------------------------------------------------------------------------------
// LQR tracking (lqr_tracking.py:245-283): linearise about (x_opt,u_opt), non-augmented Riccati with
// constant weights and S = 0, gains K (2x6), closed-loop nonlinear rollout.
// ---------------------------------------------------------------------------------------------
template <bool DIAG, typename XT>
__global__ __launch_bounds__(TILE) void k_track_gains(KConst k, const XT* __restrict__ x,
                                                      const double* __restrict__ u, const double* __restrict__ x_opt0,
                                                      double* __restrict__ Kout, int* __restrict__ status) {
    const int tile = blockIdx.x, lane = threadIdx.x, T = k.T;
    double P[21], p[6], Qb[21], xs[6], xn[6];
    int flags = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        p[i] = 0.0;
#pragma unroll
        for (int j = i; j < 6; j++) { P[sidx(i, j)] = k.QT[i * 6 + j]; Qb[sidx(i, j)] = k.Q[i * 6 + j]; }  // :716
    }
    load_state(x, x_opt0, tile, T, T - 2, lane, xn);
    double un0 = u[tix<2>(tile, T, T - 2, 0, lane)];
    for (int t = T - 2; t >= 0; t--) {
#pragma unroll
        for (int c = 0; c < 6; c++) xs[c] = xn[c];
        const double u0 = un0;
        if (t > 0) {
            load_state(x, x_opt0, tile, T, t - 1, lane, xn);
            un0 = u[tix<2>(tile, T, t - 1, 0, lane)];
        }
        const SC s = trig(xs[3], xs[5]);
        const Lin l = linearise(k, xs, u0, s);
        const double z6[6] = {0, 0, 0, 0, 0, 0}, z2[2] = {0, 0};
        double Ks[14];
        const StageFlags fl = lqr_stage(k, l, P, p, Qb, 0.0, 0.0, 0.0, z6, z2, Ks);
        if (fl.singular) flags |= AOC_ST_SINGULAR;
        if (fl.regularised) flags |= AOC_ST_REGULARISED;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            Kout[tix<12>(tile, T, t, j, lane)] = Ks[1 + j];
            Kout[tix<12>(tile, T, t, 6 + j, lane)] = Ks[8 + j];
        }
    }
#pragma unroll
    for (int j = 0; j < 12; j++) Kout[tix<12>(tile, T, T - 1, j, lane)] = 0.0;  // KK[:,:,T-1] stays 0 (:700)
    if (status && flags) status[tile * TILE + lane] |= flags;
}

template <typename XT, typename XO>
__global__ __launch_bounds__(TILE) void k_track_rollout(KConst k, const XT* __restrict__ x,
                                                        const double* __restrict__ u, const double* __restrict__ x_opt0,
                                                        const double* __restrict__ Kin, const double* __restrict__ x0,
                                                        XO* __restrict__ x_reg, double* __restrict__ u_reg,
                                                        int* __restrict__ status) {
    const int tile = blockIdx.x, lane = threadIdx.x, T = k.T;
    double xs[6], xn[6], xo[6];
    int flags = 0;
#pragma unroll
    for (int c = 0; c < 6; c++) {
        xs[c] = x0[((size_t)tile * 6 + c) * TILE + lane];
        x_reg[tix<6>(tile, T, 0, c, lane)] = (XO)xs[c];
    }
    for (int t = 0; t < T - 1; t++) {
        double d[6], u0, u1;
        load_state(x, x_opt0, tile, T, t, lane, xo);
        {
#pragma clang fp contract(off)
            // uu_reg = uu_opt + KK @ (xx_reg - xx_opt)   (lqr_tracking.py:280)
#pragma unroll
            for (int c = 0; c < 6; c++) d[c] = xs[c] - xo[c];
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int c = 0; c < 6; c++) {
                a0 += Kin[tix<12>(tile, T, t, c, lane)] * d[c];
                a1 += Kin[tix<12>(tile, T, t, 6 + c, lane)] * d[c];
            }
            u0 = u[tix<2>(tile, T, t, 0, lane)] + a0;
            u1 = u[tix<2>(tile, T, t, 1, lane)] + a1;
        }
        if (!(xs[2] > 0.0)) flags |= AOC_ST_VNONPOS;
        const SC s = trig(xs[3], xs[5]);
        step_state(k, xs, u0, u1, s, xn);
        u_reg[tix<2>(tile, T, t, 0, lane)] = u0;
        u_reg[tix<2>(tile, T, t, 1, lane)] = u1;
#pragma unroll
        for (int c = 0; c < 6; c++) { x_reg[tix<6>(tile, T, t + 1, c, lane)] = (XO)xn[c]; xs[c] = xn[c]; }
    }
    u_reg[tix<2>(tile, T, T - 1, 0, lane)] = 0.0;
    u_reg[tix<2>(tile, T, T - 1, 1, lane)] = 0.0;
    if (status && flags) status[tile * TILE + lane] |= flags;
}

// ---------------------------------------------------------------------------------------------
// Generic ltv_LQR (optcon.py:533-771; copy lqr_tracking.py:6-242) with caller-supplied A,B,Q,R,S per
// stage: dense N x N recursion (N = 6, or 7 for the augmented affine form), one problem per lane,
// in the reference's order of operations (Riccati loop, gain loop with the PD test, rollout loop).
// Not a throughput path: it backs the drop-in optcon.ltv_LQR and pins L1/L2 of SURVEY 8a.
// Layouts are time-major per problem b: A [b][t][36], Bm [b][t][12] (6x2), Q [b][t][36], R [b][t][4],
// S [b][t][12] (2x6), Qf [b][36], x0 [b][6], q [b][t][6], r [b][t][2], qf [b][6];
// outputs KK [b][t][2*N], PP [b][t][N*N] (workspace, always needed), xx [b][t][6], uu [b][t][2].
// ---------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
template <int N>
__global__ void k_ltv_lqr(int nb, int T, const double* __restrict__ Ain, const double* __restrict__ Bin,
                          const double* __restrict__ Qin, const double* __restrict__ Rin,
                          const double* __restrict__ Sin, const double* __restrict__ Qfin,
                          const double* __restrict__ x0, const double* __restrict__ qq, const double* __restrict__ rr,
                          const double* __restrict__ qqf, double* __restrict__ KK, double* __restrict__ PP,
                          double* __restrict__ xxo, double* __restrict__ uuo, int* __restrict__ nreg,
                          int* __restrict__ nsing) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    constexpr bool AUG = (N == 7);
    constexpr int O = AUG ? 1 : 0;
    const double* A_ = Ain + (size_t)b * T * 36;
    const double* B_ = Bin + (size_t)b * T * 12;
    const double* Q_ = Qin + (size_t)b * T * 36;
    const double* R_ = Rin + (size_t)b * T * 4;
    const double* S_ = Sin + (size_t)b * T * 12;
    double* P_ = PP + (size_t)b * T * N * N;
    double* K_ = KK + (size_t)b * T * 2 * N;
    int reg = 0, sing = 0;
    auto Aa = [&](int t, int i, int j) -> double {  // augmented A~ = blkdiag(1, A)  (:684-685)
        if (!AUG) return A_[(size_t)t * 36 + i * 6 + j];
        if (i == 0 || j == 0) return (i == 0 && j == 0) ? 1.0 : 0.0;
        return A_[(size_t)t * 36 + (i - 1) * 6 + (j - 1)];
    };
    auto Ba = [&](int t, int i, int j) -> double {  // B~ = [0; B]  (:686)
        if (AUG && i == 0) return 0.0;
        return B_[(size_t)t * 12 + (i - O) * 2 + j];
    };
    auto Qa = [&](int t, int i, int j) -> double {  // Q~ = [[0, q/2],[q/2, Q]]  (:673-675)
        if (!AUG) return Q_[(size_t)t * 36 + i * 6 + j];
        if (i == 0 && j == 0) return 0.0;
        if (i == 0) return 0.5 * (qq ? qq[((size_t)b * T + t) * 6 + j - 1] : 0.0);
        if (j == 0) return 0.5 * (qq ? qq[((size_t)b * T + t) * 6 + i - 1] : 0.0);
        return Q_[(size_t)t * 36 + (i - 1) * 6 + (j - 1)];
    };
    auto Sa = [&](int t, int i, int j) -> double {  // S~ = [r/2, S]  (:679-680)
        if (!AUG) return S_[(size_t)t * 12 + i * 6 + j];
        if (j == 0) return 0.5 * (rr ? rr[((size_t)b * T + t) * 2 + i] : 0.0);
        return S_[(size_t)t * 12 + i * 6 + j - 1];
    };
    // P_{T-1} = Q~f  (:688-690, :716)
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) {
            double v;
            if (!AUG) v = Qfin[(size_t)b * 36 + i * 6 + j];
            else if (i == 0 && j == 0) v = 0.0;
            else if (i == 0) v = 0.5 * (qqf ? qqf[(size_t)b * 6 + j - 1] : 0.0);
            else if (j == 0) v = 0.5 * (qqf ? qqf[(size_t)b * 6 + i - 1] : 0.0);
            else v = Qfin[(size_t)b * 36 + (i - 1) * 6 + (j - 1)];
            P_[(size_t)(T - 1) * N * N + i * N + j] = v;
        }
    auto stage = [&](int t, bool gains) {
        const double* Pn = P_ + (size_t)(t + 1) * N * N;
        double AtP[N][N], BtP[2][N], G[2][N], M[4];
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < N; j++) {
                double a = 0.0;
                for (int l = 0; l < N; l++) a += Ba(t, l, i) * Pn[l * N + j];
                BtP[i][j] = a;
            }
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < N; j++) {
                double a = 0.0;
                for (int l = 0; l < N; l++) a += BtP[i][l] * Aa(t, l, j);
                G[i][j] = a + Sa(t, i, j);
            }
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++) {
                double a = 0.0;
                for (int l = 0; l < N; l++) a += BtP[i][l] * Ba(t, l, j);
                M[i * 2 + j] = R_[(size_t)t * 4 + i * 2 + j] + a;
            }
        if (gains) {
            const double tr = M[0] + M[3], det0 = M[0] * M[3] - M[1] * M[2];
            const double disc = 0.25 * (M[0] - M[3]) * (M[0] - M[3]) + M[1] * M[2];
            const bool pd = (disc >= 0.0) ? (tr > 0.0 && det0 > 0.0) : (tr > 0.0);  // all(eigvals > 0), :745
            if (!pd) { M[0] += 0.5; M[3] += 0.5; reg++; }
        }
        const double det = M[0] * M[3] - M[1] * M[2];
        if (det == 0.0) sing++;
        const double Mi[4] = {M[3] / det, -M[1] / det, -M[2] / det, M[0] / det};
        if (gains) {  // K = (-inv(M)) @ G   (:751)
            for (int i = 0; i < 2; i++)
                for (int j = 0; j < N; j++)
                    K_[(size_t)t * 2 * N + i * N + j] = (-Mi[i * 2 + 0]) * G[0][j] + (-Mi[i * 2 + 1]) * G[1][j];
            return;
        }
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) {
                double a = 0.0;
                for (int l = 0; l < N; l++) a += Aa(t, l, i) * Pn[l * N + j];
                AtP[i][j] = a;
            }
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) {
                double apa = 0.0;
                for (int l = 0; l < N; l++) apa += AtP[i][l] * Aa(t, l, j);
                const double gm0 = G[0][i] * Mi[0] + G[1][i] * Mi[2], gm1 = G[0][i] * Mi[1] + G[1][i] * Mi[3];
                P_[(size_t)t * N * N + i * N + j] = (Qa(t, i, j) + apa) - (gm0 * G[0][j] + gm1 * G[1][j]);  // :727-728
            }
    };
    for (int t = T - 2; t >= 0; t--) stage(t, false);   // Riccati (:719-728), never regularised (Q3)
    for (int t = 0; t < T - 1; t++) stage(t, true);     // gains (:732-751)
    for (int j = 0; j < 2 * N; j++) K_[(size_t)(T - 1) * 2 * N + j] = 0.0;
    // closed-loop linear rollout (:756-762)
    double xt[N], xn[N];
    if (AUG) xt[0] = 1.0;
    for (int i = 0; i < 6; i++) xt[O + i] = x0[(size_t)b * 6 + i];
    for (int t = 0; t < T; t++) {
        for (int i = 0; i < 6; i++) xxo[((size_t)b * T + t) * 6 + i] = xt[O + i];
        if (t == T - 1) { uuo[((size_t)b * T + t) * 2] = 0.0; uuo[((size_t)b * T + t) * 2 + 1] = 0.0; break; }
        double uu[2];
        for (int i = 0; i < 2; i++) {
            double a = 0.0;
            for (int l = 0; l < N; l++) a += K_[(size_t)t * 2 * N + i * N + l] * xt[l];
            uu[i] = a;
            uuo[((size_t)b * T + t) * 2 + i] = a;
        }
        for (int i = 0; i < N; i++) {
            double a = 0.0, c = 0.0;
            for (int l = 0; l < N; l++) a += Aa(t, i, l) * xt[l];
            for (int l = 0; l < 2; l++) c += Ba(t, i, l) * uu[l];
            xn[i] = a + c;
        }
        for (int i = 0; i < N; i++) xt[i] = xn[i];
    }
    if (nreg) nreg[b] = reg;
    if (nsing) nsing[b] = sing;
}
#pragma clang fp contract(fast)

#pragma clang fp contract(off)
__device__ __forceinline__ bool armijo_reject(double Jt, double JP, double cc, double a, double descent) {
    return Jt > JP + cc * a * descent;  // optcon.py:268
}
#pragma clang fp contract(fast)

// ---------------------------------------------------------------------------------------------
// Armijo back-tracking (optcon.py:243-273) + final update (optcon.py:488-491), batched.
//
// A rollout is a strictly serial chain over T stages and a wavefront issues in order, so one trial
// costs ~T * (a few hundred instructions) of wall time however few trajectories still search, and
// back-tracking inside a tile would cost max-over-64-lanes trials per wavefront (mean trials are
// 1.2-2.2, the max over a tile is 5-8).  So the search runs in rounds over the still-searching
// trajectories only, compacted in tile order over as few wavefronts as possible, and when those are
// fewer than the chip has SIMDs the round evaluates the next K candidate steps of every trajectory
// at once, each (trajectory, step) pair on its own lane: the reference tries steps one by one, the
// first accepted index is the same.
//   k_ls_init  : trial-0 verdicts (J' of aoc_forward) -> bit mask per tile; step table a_r = s0*beta^r
//   k_ls_plan  : (one workgroup) resolve the previous round, prefix-sum the masks, choose K
//   k_ls_trial : cost-only rollouts of (trajectory, a_{r+k}); accept -> atomicMin(first_ok[b], r+k)
//   k_ls_final : resolve the last round; exhausted searches take the never-evaluated
//                s0*beta^maxiters (Q5); every trajectory is rolled out with its step into x_new/u_new
//                (full-tile coalesced stores) and J_new
// A compacted lane gathers its (u, du) elements from its home tile; the work list is in tile order,
// so a wavefront touches about the cache lines a coalesced pass over the same tiles would.
// ---------------------------------------------------------------------------------------------
constexpr int LS_MAX_STEPS = 64;   // armijo_maxiters <= 63
constexpr int LS_NOT_FOUND = 0x7fffffff;

struct LsState {
    int r_next;   // first step index not yet evaluated
    int r_start;  // this round evaluates indices r_start .. r_start+K-1
    int K;
    int count;    // trajectories searching in this round
    int nw;       // wavefronts per step index = ceil(count/64)
    int round;    // rounds planned so far
    int pad[2];
    double alpha[LS_MAX_STEPS];
};

struct LsScratch {
    unsigned long long* mask;  // [ntiles]  bit l: lane l still searching
    int* prefix;               // [ntiles+1]
    int* first_ok;             // [ntiles*64]
    LsState* st;
};

__global__ __launch_bounds__(TILE) void k_ls_init(aoc_params prm, int nspec, int Bp, const double* __restrict__ J_cur,
                                                  const double* __restrict__ descent,
                                                  const double* __restrict__ J_trial, double* __restrict__ stepsize,
                                                  int* __restrict__ ntrials, LsScratch sc) {
    const int tile = blockIdx.x, lane = threadIdx.x, b = tile * TILE + lane;
    // verdicts of the nspec trials evaluated by aoc_forward, in the reference's order
    const double JP = J_cur[b], d = descent[b];
    double a = prm.stepsize_0;
    bool searching = true;
    int ntr = 0;
    double acc = a;
    for (int j = 0; j < nspec && j < prm.armijo_maxiters; j++) {
        if (searching) {
            ntr = j + 1;
            if (!armijo_reject(J_trial[(size_t)j * Bp + b], JP, prm.cc, a, d)) { searching = false; acc = a; }
        }
        a = prm.beta * a;
    }
    const unsigned long long m = __ballot(searching);
    stepsize[b] = acc;
    ntrials[b] = ntr;
    sc.first_ok[b] = LS_NOT_FOUND;
    if (lane == 0) sc.mask[tile] = m;
    if (tile == 0 && lane == 0) {
        double al = prm.stepsize_0;
        sc.st->alpha[0] = al;
        for (int i = 1; i <= prm.armijo_maxiters; i++) { al = prm.beta * al; sc.st->alpha[i] = al; }  // optcon.py:270
        const int done = nspec < prm.armijo_maxiters ? nspec : prm.armijo_maxiters;
        sc.st->r_next = done;
        sc.st->r_start = done;
        sc.st->K = 0;
        sc.st->count = 0;
        sc.st->nw = 0;
        sc.st->round = 0;
    }
}

// resolve one tile's searching lanes against first_ok; returns the updated mask (lane-parallel)
__device__ __forceinline__ unsigned long long ls_resolve(unsigned long long m, int b, int r_done,
                                                         const LsState* st, int* __restrict__ first_ok,
                                                         double* __restrict__ stepsize, int* __restrict__ ntrials,
                                                         int lane) {
    const bool searching = (m >> lane) & 1ull;
    bool still = false;
    if (searching) {
        const int f = first_ok[b];
        if (f != LS_NOT_FOUND) { stepsize[b] = st->alpha[f]; ntrials[b] = f + 1; }
        else { ntrials[b] = r_done; still = true; }
    }
    return __ballot(still);
}

// resolve the previous round tile by tile (one wavefront per tile)
__global__ __launch_bounds__(TILE) void k_ls_resolve(LsScratch sc, double* __restrict__ stepsize,
                                                     int* __restrict__ ntrials) {
    const int tile = blockIdx.x, lane = threadIdx.x;
    const LsState* st = sc.st;
    if (st->K <= 0) return;
    const unsigned long long m = sc.mask[tile];
    if (m == 0ull) return;
    const unsigned long long m2 = ls_resolve(m, tile * TILE + lane, st->r_next, st, sc.first_ok, stepsize, ntrials, lane);
    if (lane == 0) sc.mask[tile] = m2;
}

// One workgroup: exclusive prefix sum of the per-tile popcounts and the plan of the next round:
// K candidate steps per searching trajectory, as many as keep the round within `wcap` wavefronts,
// but not more than round+1 (rejection thins out geometrically: deep speculation only pays late).
__global__ __launch_bounds__(1024) void k_ls_plan(int ntiles, int maxiters, int wcap, int kgrow, LsScratch sc) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    LsState* st = sc.st;
    const int r_done = st->r_next;  // indices < r_next have been evaluated for every searching lane
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < ntiles; base += 1024) {
        const int i = base + tid;
        const int c = (i < ntiles) ? __popcll(sc.mask[i]) : 0;
        int incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wv; w++) woff += wsum[w];
        const int carry = carry_s;
        if (i < ntiles) sc.prefix[i] = carry + woff + incl - c;
        __syncthreads();
        if (tid == 1023) carry_s = carry + woff + incl;
        __syncthreads();
    }
    if (tid == 0) {
        const int count = carry_s;
        sc.prefix[ntiles] = count;
        const int nw = (count + TILE - 1) / TILE;
        const int round = st->round + 1;
        int K = 0;
        if (count > 0 && r_done < maxiters) {
            K = wcap / nw;
            if (K > kgrow * round + 1) K = kgrow * round + 1;
            if (K < 1) K = 1;
            if (K > maxiters - r_done) K = maxiters - r_done;
        }
        st->round = round;
        st->r_start = r_done;
        st->K = K;
        st->count = count;
        st->nw = nw;
        st->r_next = r_done + K;
    }
}

__device__ __forceinline__ int nth_set_bit(unsigned long long m, int rank) {
    int pos = 0;
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) {
        const unsigned long long low = m & ((1ull << w) - 1ull);
        const int c = __popcll(low);
        if (rank >= c) { rank -= c; m >>= w; pos += w; }
        else m = low;
    }
    return pos;
}

template <bool DIAG>
__global__ __launch_bounds__(TILE) void k_ls_trial(KConst k, aoc_params prm, const double* __restrict__ ref,
                                                   const double* __restrict__ u, const double* __restrict__ x0,
                                                   const double* __restrict__ du, const double* __restrict__ J_cur,
                                                   const double* __restrict__ descent, LsScratch sc) {
    const LsState* st = sc.st;
    const int K = st->K, nw = st->nw;
    const int w = blockIdx.x;
    if (w >= K * nw) return;
    const int kidx = w / nw, slot = w - kidx * nw;
    const int r = st->r_start + kidx;
    const double a_r = st->alpha[r];
    const int count = st->count;
    const int lane = threadIdx.x;
    const int first = slot * TILE;
    const bool valid = first + lane < count;
    const int j = valid ? first + lane : count - 1;  // idle lanes shadow the last item (loads stay in range)
    const int* __restrict__ prefix = sc.prefix;
    int lo = 0, hi = k.ntiles;  // tile with prefix[tile] <= j < prefix[tile+1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (prefix[mid] <= j) lo = mid; else hi = mid;
    }
    const int tile = lo;
    const int hl = nth_set_bit(sc.mask[tile], j - prefix[tile]);  // home lane inside the tile
    const int b = tile * TILE + hl;
    double xs[6];
#pragma unroll
    for (int c = 0; c < 6; c++) xs[c] = x0[((size_t)tile * 6 + c) * TILE + hl];
    int f2 = 0;
    const double Jt = rollout<DIAG, false, double>(k, ref, tile, hl, xs, u, du, a_r, false, nullptr, nullptr, f2);
    if (valid && !armijo_reject(Jt, J_cur[b], prm.cc, a_r, descent[b])) atomicMin(&sc.first_ok[b], r);
}

template <bool DIAG, typename XO>
__global__ __launch_bounds__(TILE) void k_ls_final(KConst k, int maxiters, const double* __restrict__ ref,
                                                   const double* __restrict__ u, const double* __restrict__ x0,
                                                   const double* __restrict__ du, XO* __restrict__ x_new,
                                                   double* __restrict__ u_new, double* __restrict__ J_new,
                                                   double* __restrict__ stepsize, int* __restrict__ ntrials,
                                                   int* __restrict__ status, LsScratch sc) {
    const int tile = blockIdx.x, lane = threadIdx.x, b = tile * TILE + lane;
    const LsState* st = sc.st;
    int flags = 0;
    const unsigned long long m = sc.mask[tile];
    if (m != 0ull) {  // resolve the last round; what is still searching has exhausted the line search
        const unsigned long long left = (st->K > 0) ? ls_resolve(m, b, st->r_next, st, sc.first_ok, stepsize, ntrials, lane) : m;
        if ((left >> lane) & 1ull) {
            stepsize[b] = st->alpha[maxiters];  // never evaluated (Q5, optcon.py:327)
            ntrials[b] = maxiters;
            flags |= AOC_ST_ARMIJO_EXH;
        }
    }
    const double a = stepsize[b];
    double xs[6];
#pragma unroll
    for (int c = 0; c < 6; c++) xs[c] = x0[((size_t)tile * 6 + c) * TILE + lane];
    const double Jf = rollout<DIAG, true, XO>(k, ref, tile, lane, xs, u, du, a, true, x_new, u_new, flags);
    if (Jf != Jf || Jf - Jf != 0.0) flags |= AOC_ST_NAN;
    J_new[b] = Jf;
    if (status && flags) status[b] |= flags;
}

// ---------------------------------------------------------------------------------------------
// host side of the C-ABI
// ---------------------------------------------------------------------------------------------
static thread_local char g_hip_err[256] = "";

static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_hip_err, sizeof g_hip_err, "%s: %s", what, hipGetErrorString(e));
        return AOC_ELAUNCH;
    }
    return AOC_OK;
}

static bool is_diag(const double* M, int n) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++)
            if (i != j && M[i * n + j] != 0.0) return false;
    return true;
}

static KConst make_const(const aoc_model& md, const double* Q, const double* R, const double* QT, int B, int T) {
    KConst k;
    memset(&k, 0, sizeof k);
    k.cd0 = md.cd0; k.cda = md.cda; k.cla = md.cla; k.m = md.m; k.g = md.g; k.S = md.S; k.rho = md.rho;
    k.J = md.J; k.dt = md.dt;
    k.dtm = md.dt / md.m;
    k.mg = md.m * md.g;
    k.hrho = 0.5 * md.rho;
    k.krs = md.rho * md.S;
    k.b41 = md.dt / md.J;
    if (Q) memcpy(k.Q, Q, sizeof k.Q);
    if (R) memcpy(k.R, R, sizeof k.R);
    if (QT) memcpy(k.QT, QT, sizeof k.QT);
    k.B = B; k.T = T; k.ntiles = (B + TILE - 1) / TILE;
    k.diag = (Q && R && QT) ? (is_diag(Q, 6) && is_diag(R, 2) && is_diag(QT, 6)) : 1;
    return k;
}

static KConst make_const(const aoc_problem* p) {
    return make_const(p->model, p->QQt, p->RRt, p->QQT, p->B, p->T);
}

static int check_problem(const aoc_problem* p) {
    if (!p || !p->ref) return AOC_EINVAL;
    if (p->B < 1 || p->T < 3) return AOC_EINVAL;
    // R must be symmetric for the 2x2 closed forms used by the gain solve
    if (p->RRt[1] != p->RRt[2]) return AOC_EINVAL;
    return AOC_OK;
}

extern "C" {

const char* aoc_version(void) { return "aoc-hip 0.1 (gfx950)"; }

const char* aoc_strerror(int code) {
    switch (code) {
        case AOC_OK: return "ok";
        case AOC_EINVAL: return "invalid argument";
        case AOC_ELAUNCH: return "HIP launch/runtime error";
        case AOC_ENODEV: return "no usable device";
        default: return "unknown error";
    }
}

const char* aoc_last_hip_error(void) { return g_hip_err; }

int32_t aoc_ntiles(int32_t B) { return (B + TILE - 1) / TILE; }

size_t aoc_tiled_elems(int32_t B, int32_t T, int32_t C) { return (size_t)aoc_ntiles(B) * T * C * TILE; }

int aoc_pack(int32_t B, int32_t T, int32_t C, const double* src, double* dst, void* stream) {
    if (!src || !dst || B < 1 || T < 1 || C < 1) return AOC_EINVAL;
    dim3 grid(aoc_ntiles(B), T < 64 ? T : 64);
    hipLaunchKernelGGL(k_pack<double>, grid, dim3(TILE), 0, (hipStream_t)stream, B, T, C, src, dst);
    return check_launch("k_pack");
}

int aoc_unpack(int32_t B, int32_t T, int32_t C, const double* src, double* dst, void* stream) {
    if (!src || !dst || B < 1 || T < 1 || C < 1) return AOC_EINVAL;
    dim3 grid(aoc_ntiles(B), T < 64 ? T : 64);
    hipLaunchKernelGGL(k_unpack<double>, grid, dim3(TILE), 0, (hipStream_t)stream, B, T, C, src, dst);
    return check_launch("k_unpack");
}

int aoc_pack_f32(int32_t B, int32_t T, int32_t C, const double* src, float* dst, void* stream) {
    if (!src || !dst || B < 1 || T < 1 || C < 1) return AOC_EINVAL;
    dim3 grid(aoc_ntiles(B), T < 64 ? T : 64);
    hipLaunchKernelGGL(k_pack<float>, grid, dim3(TILE), 0, (hipStream_t)stream, B, T, C, src, dst);
    return check_launch("k_pack");
}

int aoc_unpack_f32(int32_t B, int32_t T, int32_t C, const float* src, double* dst, void* stream) {
    if (!src || !dst || B < 1 || T < 1 || C < 1) return AOC_EINVAL;
    dim3 grid(aoc_ntiles(B), T < 64 ? T : 64);
    hipLaunchKernelGGL(k_unpack<float>, grid, dim3(TILE), 0, (hipStream_t)stream, B, T, C, src, dst);
    return check_launch("k_unpack");
}

int aoc_step_batch(const aoc_model* model, int32_t n, const double* x, const double* u, const double* lmbd,
                   double* xp, double* fx, double* fu, double* fxx, double* fuu, double* fux, void* stream) {
    if (!model || !x || !u || !xp || n < 1) return AOC_EINVAL;
    KConst k = make_const(*model, nullptr, nullptr, nullptr, n, 3);
    hipLaunchKernelGGL(k_step_batch, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, k, n, x, u, lmbd, xp,
                       fx, fu, fxx, fuu, fux);
    return check_launch("k_step_batch");
}

int aoc_cost_batch(const aoc_problem* prob, int32_t n, const double* x, const double* u, const double* xr,
                   const double* ur, double* ll, double* lx, double* lu, double* llT, double* lTx, void* stream) {
    if (!prob || !x || !u || !xr || !ur || n < 1) return AOC_EINVAL;
    KConst k = make_const(prob->model, prob->QQt, prob->RRt, prob->QQT, n, 3);
    if (k.diag)
        hipLaunchKernelGGL(k_cost_batch<true>, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, k, n, x, u, xr,
                           ur, ll, lx, lu, llT, lTx);
    else
        hipLaunchKernelGGL(k_cost_batch<false>, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, k, n, x, u, xr,
                           ur, ll, lx, lu, llT, lTx);
    return check_launch("k_cost_batch");
}

int aoc_traj_cost(const aoc_problem* p, const void* x, const double* u, const double* x0, double* J) {
    int rc = check_problem(p);
    if (rc) return rc;
    if (!x || !u || !x0 || !J) return AOC_EINVAL;
    KConst k = make_const(p);
    hipStream_t st = (hipStream_t)p->stream;
    AOC_DISPATCH_BOOL(k.diag, D, AOC_DISPATCH_XT(p->x_in_f32, XT,
        hipLaunchKernelGGL((k_traj_cost<D, XT>), dim3(k.ntiles), dim3(TILE), 0, st, k, p->ref, (const XT*)x, u, x0, J)));
    return check_launch("k_traj_cost");
}

int aoc_initial_trajectory(const aoc_problem* p, double kp, double kt, const double* x0, void* x, double* u) {
    int rc = check_problem(p);
    if (rc) return rc;
    if (!x0 || !x || !u) return AOC_EINVAL;
    KConst k = make_const(p);
    AOC_DISPATCH_XT(p->x_out_f32, XO,
        hipLaunchKernelGGL((k_initial_traj<XO>), dim3(k.ntiles), dim3(TILE), 0, (hipStream_t)p->stream, k, kp, kt,
                           p->ref, x0, (XO*)x, u));
    return check_launch("k_initial_traj");
}

int aoc_rollout_cost(const aoc_problem* p, const double* x0, const double* u, const double* du,
                     const double* alpha, void* x_out, double* u_out, double* J_out, int32_t* status) {
    int rc = check_problem(p);
    if (rc) return rc;
    if (!x0 || !u || !J_out) return AOC_EINVAL;
    if ((x_out == nullptr) != (u_out == nullptr)) return AOC_EINVAL;
    if (du && !alpha) return AOC_EINVAL;
    KConst k = make_const(p);
    hipStream_t st = (hipStream_t)p->stream;
    AOC_DISPATCH_BOOL(k.diag, D, AOC_DISPATCH_BOOL(x_out != nullptr, W, AOC_DISPATCH_XT(p->x_out_f32, XO,
        hipLaunchKernelGGL((k_rollout_cost<D, W, XO>), dim3(k.ntiles), dim3(TILE), 0, st, k, p->ref, x0, u, du, alpha,
                           (XO*)x_out, u_out, J_out, status))));
    return check_launch("k_rollout_cost");
}

int aoc_backward(const aoc_problem* p, int32_t full_hessian, const void* x, const double* u, const double* x0,
                 double* Kt, double* g, double* lmbd0, int32_t* status) {
    int rc = check_problem(p);
    if (rc) return rc;
    if (!x || !u || !x0 || !Kt || !g) return AOC_EINVAL;
    KConst k = make_const(p);
    hipStream_t st = (hipStream_t)p->stream;
    AOC_DISPATCH_BOOL(k.diag, D, AOC_DISPATCH_BOOL(full_hessian != 0, F, AOC_DISPATCH_XT(p->x_in_f32, XT,
        hipLaunchKernelGGL((k_backward<D, F, XT>), dim3(k.ntiles), dim3(TILE), 0, st, k, p->ref, (const XT*)x, u, x0, Kt,
                           g, lmbd0, status))));
    return check_launch("k_backward");
}

int aoc_forward(const aoc_problem* p, const aoc_params* prm, int32_t n_spec, const void* x, const double* u,
                const double* x0, const double* Kt, const double* g, double* du, double* descent, double* J_trial,
                int32_t* status) {
    int rc = check_problem(p);
    if (rc) return rc;
    if (!prm || !x || !u || !x0 || !Kt || !g || !du || !descent || !J_trial) return AOC_EINVAL;
    if (n_spec < 1 || n_spec > 3) return AOC_EINVAL;
    KConst k = make_const(p);
    hipStream_t st = (hipStream_t)p->stream;
#define LAUNCH_FW(N)                                                                                                \
    AOC_DISPATCH_BOOL(k.diag, D, AOC_DISPATCH_XT(p->x_in_f32, XT,                                                   \
        hipLaunchKernelGGL((k_forward<D, N, XT>), dim3(k.ntiles), dim3(TILE), 0, st, k, *prm, p->ref, (const XT*)x, \
                           u, x0, Kt, g, du, descent, J_trial, status)))
    if (n_spec == 1) LAUNCH_FW(1);
    else if (n_spec == 2) LAUNCH_FW(2);
    else LAUNCH_FW(3);
#undef LAUNCH_FW
    return check_launch("k_forward");
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

size_t aoc_linesearch_scratch_bytes(int32_t B) {
    const size_t nt = (size_t)aoc_ntiles(B);
    return align_up(nt * sizeof(unsigned long long), 16) + align_up((nt + 1) * sizeof(int), 16) +
           align_up(nt * TILE * sizeof(int), 16) + align_up(sizeof(LsState), 16);
}

int aoc_linesearch(const aoc_problem* p, const aoc_params* prm, int32_t n_spec, const double* u, const double* x0,
                   const double* du, const double* J_cur, const double* descent, const double* J_trial0,
                   void* x_new, double* u_new, double* J_new, double* stepsize, int32_t* ntrials, int32_t* status,
                   void* scratch) {
    int rc = check_problem(p);
    if (rc) return rc;
    if (!prm || !u || !x0 || !du || !J_cur || !descent || !J_trial0 || !x_new || !u_new || !J_new || !stepsize ||
        !ntrials || !scratch)
        return AOC_EINVAL;
    if (prm->armijo_maxiters < 1 || prm->armijo_maxiters >= LS_MAX_STEPS) return AOC_EINVAL;
    if (n_spec < 1 || n_spec > 3) return AOC_EINVAL;
    KConst k = make_const(p);
    hipStream_t st = (hipStream_t)p->stream;
    const size_t nt = (size_t)k.ntiles;
    LsScratch sc;
    char* base = (char*)scratch;
    sc.mask = (unsigned long long*)base;  base += align_up(nt * sizeof(unsigned long long), 16);
    sc.prefix = (int*)base;               base += align_up((nt + 1) * sizeof(int), 16);
    sc.first_ok = (int*)base;             base += align_up(nt * TILE * sizeof(int), 16);
    sc.st = (LsState*)base;
    // wavefronts a trial round may occupy: about one per SIMD (256 CUs x 4 SIMDs), at least one per tile
    static const int wcap_env = getenv("AOC_LS_WCAP") ? atoi(getenv("AOC_LS_WCAP")) : 1024;
    static const int kgrow_env = getenv("AOC_LS_KGROW") ? atoi(getenv("AOC_LS_KGROW")) : 1;
    const int wcap = k.ntiles > wcap_env ? k.ntiles : wcap_env;
    hipLaunchKernelGGL(k_ls_init, dim3(k.ntiles), dim3(TILE), 0, st, *prm, n_spec, k.ntiles * TILE, J_cur, descent,
                       J_trial0, stepsize, ntrials, sc);
    for (int r = n_spec; r < prm->armijo_maxiters; r++) {
        if (r > n_spec) hipLaunchKernelGGL(k_ls_resolve, dim3(k.ntiles), dim3(TILE), 0, st, sc, stepsize, ntrials);
        hipLaunchKernelGGL(k_ls_plan, dim3(1), dim3(1024), 0, st, k.ntiles, prm->armijo_maxiters, wcap_env, kgrow_env, sc);
        if (k.diag)
            hipLaunchKernelGGL(k_ls_trial<true>, dim3(wcap), dim3(TILE), 0, st, k, *prm, p->ref, u, x0, du, J_cur,
                               descent, sc);
        else
            hipLaunchKernelGGL(k_ls_trial<false>, dim3(wcap), dim3(TILE), 0, st, k, *prm, p->ref, u, x0, du, J_cur,
                               descent, sc);
    }
    AOC_DISPATCH_BOOL(k.diag, D, AOC_DISPATCH_XT(p->x_out_f32, XO,
        hipLaunchKernelGGL((k_ls_final<D, XO>), dim3(k.ntiles), dim3(TILE), 0, st, k, prm->armijo_maxiters, p->ref, u, x0,
                           du, (XO*)x_new, u_new, J_new, stepsize, ntrials, status, sc)));
    return check_launch("aoc_linesearch");
}

int aoc_lqr_tracking(const aoc_problem* p, const void* x_opt, const double* u_opt, const double* x_opt0,
                     const double* x0_reg, double* Kgain, void* x_reg, double* u_reg, int32_t* status) {
    int rc = check_problem(p);
    if (rc) return rc;
    if (!x_opt || !u_opt || !x_opt0 || !Kgain) return AOC_EINVAL;
    if ((x_reg == nullptr) != (u_reg == nullptr) || (x_reg && !x0_reg)) return AOC_EINVAL;
    KConst k = make_const(p);
    hipStream_t st = (hipStream_t)p->stream;
    AOC_DISPATCH_BOOL(k.diag, D, AOC_DISPATCH_XT(p->x_in_f32, XT,
        hipLaunchKernelGGL((k_track_gains<D, XT>), dim3(k.ntiles), dim3(TILE), 0, st, k, (const XT*)x_opt, u_opt, x_opt0,
                           Kgain, status)));
    if (x_reg)
        AOC_DISPATCH_XT(p->x_in_f32, XT, AOC_DISPATCH_XT(p->x_out_f32, XO,
            hipLaunchKernelGGL((k_track_rollout<XT, XO>), dim3(k.ntiles), dim3(TILE), 0, st, k, (const XT*)x_opt, u_opt,
                               x_opt0, Kgain, x0_reg, (XO*)x_reg, u_reg, status)));
    return check_launch("aoc_lqr_tracking");
}

int aoc_ltv_lqr(int32_t nb, int32_t T, int32_t augmented, const double* A, const double* Bm, const double* Q,
                const double* R, const double* S, const double* Qf, const double* x0, const double* q,
                const double* r, const double* qf, double* KK, double* PP, double* xx, double* uu, int32_t* nreg,
                int32_t* nsing, void* stream) {
    if (nb < 1 || T < 2 || !A || !Bm || !Q || !R || !S || !Qf || !x0 || !KK || !PP || !xx || !uu) return AOC_EINVAL;
    if (!augmented && (q || r || qf)) return AOC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((nb + 63) / 64), blk(64);
    if (augmented)
        hipLaunchKernelGGL(k_ltv_lqr<7>, grid, blk, 0, st, nb, T, A, Bm, Q, R, S, Qf, x0, q, r, qf, KK, PP, xx, uu, nreg, nsing);
    else
        hipLaunchKernelGGL(k_ltv_lqr<6>, grid, blk, 0, st, nb, T, A, Bm, Q, R, S, Qf, x0, q, r, qf, KK, PP, xx, uu, nreg, nsing);
    return check_launch("k_ltv_lqr");
}

size_t aoc_workspace_bytes(int32_t B, int32_t T) {
    // K~ (14) + g (2) + du (2) components, tiled; trial costs [3][ntiles*64]; then the line-search scratch
    return aoc_tiled_elems(B, T, 18) * sizeof(double) + 3 * (size_t)aoc_ntiles(B) * TILE * sizeof(double) +
           aoc_linesearch_scratch_bytes(B);
}

int aoc_newton_iterate(const aoc_problem* p, const aoc_params* prm, int32_t kk, const void* x, const double* u,
                       const double* x0, const double* J_cur, void* workspace, void* x_new, double* u_new,
                       double* J_new, double* descent, double* stepsize, int32_t* ntrials, int32_t* status) {
    int rc = check_problem(p);
    if (rc) return rc;
    if (!prm || !workspace) return AOC_EINVAL;
    double* Kt = (double*)workspace;
    double* g = Kt + aoc_tiled_elems(p->B, p->T, 14);
    double* du = g + aoc_tiled_elems(p->B, p->T, 2);
    double* J_trial = du + aoc_tiled_elems(p->B, p->T, 2);
    void* scratch = (void*)(J_trial + 3 * (size_t)aoc_ntiles(p->B) * TILE);
    static const int nspec = getenv("AOC_NSPEC") ? atoi(getenv("AOC_NSPEC")) : 2;
    rc = aoc_backward(p, kk > prm->hessian_switch, x, u, x0, Kt, g, nullptr, status);
    if (rc) return rc;
    rc = aoc_forward(p, prm, nspec, x, u, x0, Kt, g, du, descent, J_trial, status);
    if (rc) return rc;
    return aoc_linesearch(p, prm, nspec, u, x0, du, J_cur, descent, J_trial, x_new, u_new, J_new, stepsize, ntrials,
                          status, scratch);
}

}  // extern "C"
