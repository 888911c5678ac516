// aoc_kernels.hip — HIP kernels (gfx950) and the C-ABI of libaoc_hip.so (include/aoc.h).
//
// Execution model: one wavefront (64 lanes) per tile of 64 trajectories, one trajectory per lane,
// one workgroup = one wavefront.  Every pass walks the horizon sequentially inside the lane; the
// batch axis is the lane axis, so each (t, component) access of a wavefront is one coalesced 512-B
// segment and a wavefront streams through its own contiguous slab of every array.
// No MFMA (the blocks are 6x6/6x2); the per-lane state (P: 21, p: 6, lambda: 6 doubles, ...) lives in VGPRs.
// The kernels themselves are in aoc_passes.inc (which includes passes/*.inc, one file per pass: layout, unit, cost_rollout,
// backward, forward, tracking, hcut, ltv_lqr, linesearch, mpc; then the launch functions api.inc and the solve loop
// solve.inc) + aoc_device.h, compiled twice: fp64 (aoc64, the parity path) and float32 (aoc32, BASELINE config 3); this file
// holds what is common and the C-ABI.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/aoc.h"

namespace aoc_common {
constexpr int TILE = 64;
}
#define AOC_DEVICE_COMMON
using aoc_common::TILE;

#define AOC_DISPATCH_BOOL(flag, NAME, ...)            \
    do {                                              \
        if (flag) { constexpr bool NAME = true; __VA_ARGS__; } \
        else { constexpr bool NAME = false; __VA_ARGS__; }      \
    } while (0)
// reference curves shared (false) or per trajectory (true); the float32 build (sizeof(real) == 4, entry points
// aoc_*_f32) only has the shared form, so that its kernels are not instantiated twice
#define AOC_DISPATCH_RPT(flag, NAME, ...)                                         \
    do {                                                                          \
        if (flag) { constexpr bool NAME = sizeof(real) == 8; __VA_ARGS__; }       \
        else { constexpr bool NAME = false; __VA_ARGS__; }                        \
    } while (0)
#define AOC_DISPATCH_DR(kc, NAME_D, NAME_R, ...) \
    AOC_DISPATCH_BOOL((kc).diag, NAME_D, AOC_DISPATCH_RPT((kc).rpt, NAME_R, __VA_ARGS__))
#define AOC_DISPATCH_WPE(one, NAME, ...)              \
    do {                                              \
        if (one) { constexpr int NAME = 1; __VA_ARGS__; } \
        else { constexpr int NAME = 2; __VA_ARGS__; }     \
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
// Scalar summary of a shard (aoc_summary): what the path's one collective reduces over the ranks.  One workgroup of
// 1024 threads, a fixed reduction tree: the sums are the same bits from run to run (no atomics).
// out[0] += sum of the finite costs, out[1] += sum of the descents of those trajectories, out[2] += sum of the trial
// counts, out[3] += B, out[4] += number of non-finite costs.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_summary(int B, const double* __restrict__ cost, const double* __restrict__ descent,
                                                   const int* __restrict__ ntrials, double* __restrict__ out, int accumulate) {
    __shared__ double sh[3][16];
    __shared__ long long shi[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double sc = 0.0, sd = 0.0;
    long long st = 0, nn = 0;
    for (int i = tid; i < B; i += 1024) {
        const double c = cost[i];
        const bool ok = __builtin_isfinite(c);
        if (ok) { sc += c; sd += descent[i]; } else nn++;
        st += ntrials[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sc += __shfl_down(sc, off);
        sd += __shfl_down(sd, off);
        st += __shfl_down(st, off);
        nn += __shfl_down(nn, off);
    }
    if (lane == 0) { sh[0][wv] = sc; sh[1][wv] = sd; shi[0][wv] = st; shi[1][wv] = nn; }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, b = 0.0;
        long long c = 0, d = 0;
        for (int w = 0; w < 16; w++) { a += sh[0][w]; b += sh[1][w]; c += shi[0][w]; d += shi[1][w]; }
        if (accumulate) { out[0] += a; out[1] += b; out[2] += (double)c; out[3] += (double)B; out[4] += (double)d; }
        else { out[0] = a; out[1] = b; out[2] = (double)c; out[3] = (double)B; out[4] = (double)d; }
    }
}

// ---------------------------------------------------------------------------------------------
// Do two HIP streams run concurrently?  (aoc_streams_concurrent)  The runtime maps streams onto a few hardware queues
// (four by default) and two streams on ONE queue take turns: a solver that cuts its batch in two halves on two such
// streams runs at one-stream speed (measured: 5.7 instead of 4.6-4.95 ms per iteration; of eight streams created one after
// the other, {2, 3, 7}, {0, 5} and {1, 4} shared a queue).  One wavefront per stream waits on the constant-rate clock.
// ---------------------------------------------------------------------------------------------
__global__ void k_spin(long long ticks, int* sink) {
    const long long t0 = wall_clock64();
    int n = 0;
    while (wall_clock64() - t0 < ticks && n < (1 << 24)) n++;   // bounded: every wave reaches the end
    if (sink && n < 0) *sink = n;
}

// ---------------------------------------------------------------------------------------------
// host side common to both arithmetic types
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

// argument errors leave their reason where aoc_last_hip_error() finds it
static int einval(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_hip_err, sizeof g_hip_err, fmt, ap);
    va_end(ap);
    return AOC_EINVAL;
}

// ---------------------------------------------------------------------------------------------
// scheduling knobs (aoc_tuning, include/aoc.h): environment read once, replaceable through aoc_set_tuning
// ---------------------------------------------------------------------------------------------
static aoc_tuning g_tuning;
static std::once_flag g_tuning_once;

static void tuning_defaults(aoc_tuning* t) {
    auto env = [](const char* name, int dflt) { const char* e = getenv(name); return e && *e ? atoi(e) : dflt; };
    memset(t, 0, sizeof *t);
    t->nspec = env("AOC_NSPEC", 0);
    t->split_tiles = env("AOC_SPLIT_TILES", 512);
    t->split_bw_tiles = env("AOC_SPLIT_BW_TILES", 512);
    t->fw_lin = env("AOC_FW_LIN", -1);
    t->ls_wcap = env("AOC_LS_WCAP", 0);
    t->ls_kgrow = env("AOC_LS_KGROW", 0);
    t->trial_split = env("AOC_TRIAL_SPLIT", 1);
    t->solve_norepack = env("AOC_SOLVE_NOREPACK", 0);
    t->ls_worklist = env("AOC_LS_WORKLIST", -1);
    t->ls_cpl = env("AOC_LS_CPL", 1);
    t->ls_depth_min = env("AOC_LS_DEPTH_MIN", 2);
    t->fw_recompute = env("AOC_FW_RECOMPUTE", 1);
    t->store_candidates = env("AOC_STORE_CANDIDATES", 1);
    t->bw4_tiles = env("AOC_BW4_TILES", 256);
    t->bw5 = env("AOC_BW5", 1);
    t->solve_repack_pct = env("AOC_SOLVE_REPACK_PCT", 70);
    t->solve_sync_fast = env("AOC_SOLVE_SYNC_FAST", 2);
    t->solve_split_tiles = env("AOC_SOLVE_SPLIT_TILES", 2048);
    t->track_hcut = env("AOC_TRACK_HCUT", -1);
    t->bw_hcut = env("AOC_BW_HCUT", -1);
    t->fw_wpe1 = env("AOC_FW_WPE1", 1);
    t->hcut_chain6 = env("AOC_HCUT_CHAIN6", 0);
    t->bw_hcut_full = env("AOC_BW_HCUT_FULL", 2);
    t->fw_duo = env("AOC_FW_DUO", 1);
    t->hcut_waves = env("AOC_HCUT_WAVES", 1);
    t->hcut_pairs = env("AOC_HCUT_PAIRS", 1);
}

static const aoc_tuning& tuning() {
    std::call_once(g_tuning_once, [] { tuning_defaults(&g_tuning); });
    return g_tuning;
}

static bool is_diag(const double* M, int n) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++)
            if (i != j && M[i * n + j] != 0.0) return false;
    return true;
}

static int check_problem(const aoc_problem* p) {
    if (!p) return einval("aoc_problem is NULL");
    if (!p->ref) return einval("aoc_problem.ref (reference curves) is NULL");
    if (p->B < 1 || p->T < 3) return einval("aoc_problem: B = %d, T = %d (need B >= 1, T >= 3)", p->B, p->T);
    if (p->ref_per_traj && p->ref_T != 0 && p->ref_T < p->T)
        return einval("aoc_problem: ref_T = %d < T = %d (samples per trajectory of the per-trajectory reference array)", p->ref_T, p->T);
    // R must be symmetric for the 2x2 closed forms used by the gain solve (every driver of the reference uses a
    // diagonal R; the reference itself would accept any R)
    if (p->RRt[1] != p->RRt[2])
        return einval("aoc_problem.RRt is not symmetric (R01 = %g, R10 = %g): the 2x2 gain solve needs R = R^T", p->RRt[1],
                      p->RRt[2]);
    return AOC_OK;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Diagnostic timeline of aoc_newton_solve (aoc_solve_trace, include/aoc.h): one row per iteration launched.
struct SolveTrace { double* rows; int cap; int n; };
constexpr int SOLVE_TRACE_COLS = 6;   // part, kk, batch in flight, tiles in flight, still iterating as the host last read it (-1: not read), ms since the start
static SolveTrace g_solve_trace = {nullptr, 0, 0};

// forward declarations used by the launch functions
extern "C" int32_t aoc_ntiles(int32_t B);
extern "C" size_t aoc_tiled_elems(int32_t B, int32_t T, int32_t C);
extern "C" size_t aoc_linesearch_scratch_bytes(int32_t B, int32_t T);
extern "C" int32_t aoc_default_nspec(int32_t B, int32_t armijo_maxiters);
constexpr int AOC_SPEC_MAX = 15;   // Armijo candidates that may ride along in the forward pass (5 workgroups per tile)

#define AOC_ARITH_NS aoc64
#define AOC_REAL double
#include "aoc_device.h"
#include "aoc_passes.inc"
#undef AOC_ARITH_NS
#undef AOC_REAL
#undef R

#define AOC_ARITH_NS aoc32
#define AOC_REAL float
#include "aoc_device.h"
#include "aoc_passes.inc"
#undef AOC_ARITH_NS
#undef AOC_REAL
#undef R

extern "C" {

const char* aoc_version(void) { return "aoc-hip 0.5 (gfx950)"; }
int32_t aoc_abi_version(void) { return AOC_ABI_VERSION; }

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

void aoc_get_tuning(aoc_tuning* out) {
    if (out) *out = tuning();
}

void aoc_set_tuning(const aoc_tuning* t) {
    tuning();   // make sure the one-time initialisation cannot overwrite what is set here
    if (t) g_tuning = *t;
    else tuning_defaults(&g_tuning);
}

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


int32_t aoc_spec_max(void) { return AOC_SPEC_MAX; }

// Number of Armijo candidates aoc_newton_iterate lets ride along in the forward pass.  2 in general (the pass is
// bound by its K~ stream, two extra chains per lane are nearly free); for batches so small that one wavefront per
// candidate still leaves SIMDs idle, all of them: the line search then needs no trial round.  aoc_tuning.nspec overrides.
int32_t aoc_default_nspec(int32_t B, int32_t armijo_maxiters) {
    const int forced = tuning().nspec;
    if (forced > 0) return forced < AOC_SPEC_MAX ? forced : AOC_SPEC_MAX;
    const long nt = aoc_ntiles(B);
    // one workgroup (four wavefronts, a CU's four SIMDs) per three candidates and tile; pays up to two workgroups
    // per CU, i.e. tiles x groups <= 512.  Measured, ms per iteration with 2 candidates -> with this rule:
    // 4096 trajectories 1.85 -> 1.44 (all 10), 8192 1.89 -> 1.58 (10), 10 240 1.81 -> 1.47 (9), 16 384 1.86 -> 1.78 (6);
    // one group more than the rule allows: 12 288 with 9 candidates 1.92, 16 384 with 10 2.21.
    // When every candidate rides along, so does the step an EXHAUSTED search applies — stepsize_0 * beta^armijo_maxiters,
    // never judged (Q5, optcon.py:327), candidate index armijo_maxiters: its trajectory is then stored like the others
    // and the update of a tile with exhausted searches is a copy too, not one more serial rollout (the tail of a
    // converge-mode solve and the storms of near-converged batches exhaust half of their searches).
    if (armijo_maxiters >= 1 && armijo_maxiters <= AOC_SPEC_MAX) {
        const long m1 = armijo_maxiters + 1 <= AOC_SPEC_MAX ? armijo_maxiters + 1 : armijo_maxiters;
        long g = (m1 + 2) / 3;
        if (512 / nt < g) g = 512 / nt;
        if (g >= 2 && 3 * g >= m1) return (int32_t)m1;
        g = (armijo_maxiters + 2) / 3;
        if (512 / nt < g) g = 512 / nt;
        if (g >= 2) return 3 * g < armijo_maxiters ? (int32_t)(3 * g) : armijo_maxiters;
    }
    return armijo_maxiters < 2 ? 1 : 2;
}

size_t aoc_linesearch_scratch_bytes(int32_t B, int32_t T) {
    (void)T;
    const size_t nt = (size_t)aoc_ntiles(B);
    return align_up(nt * sizeof(unsigned long long), 16) + align_up((nt + 1) * sizeof(int), 16) +
           align_up(nt * TILE * sizeof(int), 16) + align_up(sizeof(aoc64::LsState), 16) +
           align_up(nt * TILE * aoc64::LS_WL_IPL * sizeof(int2), 16) + align_up(nt * TILE * sizeof(int), 16) +
           align_up(2 * aoc64::LS_WL_ROUNDS * sizeof(int), 16);
}


// ---- fp64 entry points (aoc64) -------------------------------------------------------------------
int aoc_step_batch(const aoc_model* model, int32_t n, const double* x, const double* u, const double* lmbd,
                   double* xp, double* fx, double* fu, double* fxx, double* fuu, double* fux, void* stream) {
    return aoc64::api_step_batch(model, n, x, u, lmbd, xp, fx, fu, fxx, fuu, fux, stream);
}
int aoc_cost_batch(const aoc_problem* prob, int32_t n, const double* x, const double* u, const double* xr,
                   const double* ur, double* ll, double* lx, double* lu, double* llT, double* lTx, void* stream) {
    return aoc64::api_cost_batch(prob, n, x, u, xr, ur, ll, lx, lu, llT, lTx, stream);
}
int aoc_traj_cost(const aoc_problem* p, const void* x, const double* u, const double* x0, double* J) {
    return aoc64::api_traj_cost(p, x, u, x0, J);
}
int aoc_initial_trajectory(const aoc_problem* p, double kp, double kt, const double* x0, void* x, double* u) {
    return aoc64::api_initial_trajectory(p, kp, kt, x0, x, u);
}
int aoc_rollout_cost(const aoc_problem* p, const double* x0, const double* u, const double* du, const double* alpha,
                     void* x_out, double* u_out, double* J_out, int32_t* status) {
    return aoc64::api_rollout_cost(p, x0, u, du, alpha, x_out, u_out, J_out, status);
}
int aoc_backward(const aoc_problem* p, int32_t full_hessian, const void* x, const double* u, const double* x0,
                 double* Kt, double* lmbd0, int32_t* status, void* scratch, size_t scratch_bytes) {
    return aoc64::api_backward(p, full_hessian, x, u, x0, Kt, lmbd0, status, scratch, scratch_bytes);
}
size_t aoc_backward_scratch_bytes(int32_t B, int32_t T) {
    (void)T;
    const int S = B >= 1 ? aoc64::hcut_segments(tuning().bw_hcut, aoc_ntiles(B)) : 0;
    return S >= 2 ? aoc64::hcut_full_scratch_bytes(aoc_ntiles(B), S) : 0;   // (what a full-Hessian pass takes; Gauss-Newton: a part of it)
}
int aoc_gradient(const aoc_problem* p, const void* x, const double* u, const double* x0, double* du, double* slope,
                 int32_t* status) {
    return aoc64::api_gradient(p, x, u, x0, du, slope, status);
}
int aoc_forward(const aoc_problem* p, const aoc_params* prm, int32_t n_spec, const void* x, const double* u,
                const double* x0, const double* Kt, double* du, double* descent, double* J_trial, int32_t* status,
                void* cand, size_t cand_bytes, const int32_t* ntrials_hint) {
    return aoc64::api_forward(p, prm, n_spec, x, u, x0, Kt, du, descent, J_trial, status, cand, cand_bytes, ntrials_hint);
}
size_t aoc_candidate_bytes(int32_t B, int32_t T, int32_t n_spec) {
    return B >= 1 && T >= 1 && n_spec >= 1 ? aoc64::cand_bytes(B, T, n_spec) : 0;
}
int32_t aoc_default_ncand(int32_t B, int32_t n_spec, int32_t armijo_maxiters) {
    return B >= 1 && n_spec >= 1 && tuning().store_candidates ? aoc64::cand_count(B, n_spec, armijo_maxiters) : 0;
}
int aoc_linesearch(const aoc_problem* p, const aoc_params* prm, int32_t n_spec, const double* u, const double* x0,
                   const double* du, const double* J_cur, const double* descent, const double* J_trial, void* x_new,
                   double* u_new, double* J_new, double* stepsize, int32_t* ntrials, int32_t* status, void* scratch,
                   size_t scratch_bytes, const void* cand, size_t cand_bytes) {
    return aoc64::api_linesearch(p, prm, n_spec, u, x0, du, J_cur, descent, J_trial, x_new, u_new, J_new, stepsize,
                                 ntrials, status, scratch, scratch_bytes, aoc64::LsFrozen{nullptr, nullptr}, cand, cand_bytes);
}
int aoc_linesearch_search(const aoc_problem* p, const aoc_params* prm, int32_t n_spec, const double* u, const double* x0,
                          const double* du, const double* J_cur, const double* descent, const double* J_trial,
                          double* stepsize, int32_t* ntrials, void* scratch, size_t scratch_bytes) {
    return aoc64::api_ls_search(p, prm, n_spec, u, x0, du, J_cur, descent, J_trial, stepsize, ntrials, scratch, scratch_bytes);
}
int aoc_linesearch_update(const aoc_problem* p, const aoc_params* prm, const double* u, const double* x0, const double* du,
                          void* x_new, double* u_new, double* J_new, double* stepsize, int32_t* ntrials, int32_t* status,
                          void* scratch, size_t scratch_bytes, int32_t n_spec, const double* J_trial, const void* cand,
                          size_t cand_bytes) {
    return aoc64::api_ls_update(p, prm, u, x0, du, x_new, u_new, J_new, stepsize, ntrials, status, scratch, scratch_bytes,
                                n_spec, J_trial, cand, cand_bytes);
}
int aoc_lqr_tracking(const aoc_problem* p, const void* x_opt, const double* u_opt, const double* x_opt0,
                     const double* x0_reg, double* Kgain, void* x_reg, double* u_reg, int32_t* status) {
    return aoc64::api_lqr_tracking(p, x_opt, u_opt, x_opt0, x0_reg, Kgain, x_reg, u_reg, status);
}
int aoc_ltv_lqr(int32_t nb, int32_t T, int32_t augmented, const double* A, const double* Bm, const double* Q,
                const double* R, const double* S, const double* Qf, const double* x0, const double* q,
                const double* r, const double* qf, double* KK, double* PP, double* xx, double* uu, int32_t* nreg,
                int32_t* nsing, void* stream) {
    return aoc64::api_ltv_lqr(nb, T, augmented, A, Bm, Q, R, S, Qf, x0, q, r, qf, KK, PP, xx, uu, nreg, nsing, stream);
}
size_t aoc_workspace_bytes(int32_t B, int32_t T) { return aoc64::api_workspace_bytes(B, T); }
int aoc_newton_iterate(const aoc_problem* p, const aoc_params* prm, int32_t kk, const void* x, const double* u,
                       const double* x0, const double* J_cur, void* workspace, size_t workspace_bytes, void* x_new,
                       double* u_new, double* J_new, double* descent, double* stepsize, int32_t* ntrials, int32_t* status) {
    return aoc64::api_newton_iterate(p, prm, kk, x, u, x0, J_cur, workspace, workspace_bytes, x_new, u_new, J_new, descent,
                                     stepsize, ntrials, status);
}

size_t aoc_solve_workspace_bytes(int32_t B, int32_t T) { return aoc64::api_solve_workspace_bytes(B, T); }
int aoc_newton_solve(const aoc_problem* p, const aoc_params* prm, const void* x_init, const double* u_init,
                     const double* x0, void* workspace, size_t workspace_bytes, int32_t sync_every, void* x_star,
                     double* u_star, int32_t* iters, int32_t* ret_index, int32_t* status, double* hist_cost,
                     double* hist_descent, double* hist_stepsize, int32_t* hist_ntrials, int32_t* n_run) {
    return aoc64::api_newton_solve(p, prm, x_init, u_init, x0, workspace, workspace_bytes, sync_every, x_star, u_star, iters, ret_index,
                                   status, hist_cost, hist_descent, hist_stepsize, hist_ntrials, n_run, nullptr);
}
int aoc_newton_solve2(const aoc_problem* p, const aoc_params* prm, const void* x_init, const double* u_init,
                      const double* x0, void* workspace, size_t workspace_bytes, int32_t sync_every, void* x_star,
                      double* u_star, int32_t* iters, int32_t* ret_index, int32_t* status, double* hist_cost,
                      double* hist_descent, double* hist_stepsize, int32_t* hist_ntrials, int32_t* n_run, void* stream2) {
    return aoc64::api_newton_solve(p, prm, x_init, u_init, x0, workspace, workspace_bytes, sync_every, x_star, u_star, iters, ret_index,
                                   status, hist_cost, hist_descent, hist_stepsize, hist_ntrials, n_run, stream2);
}

int aoc_summary(int32_t B, const double* cost, const double* descent, const int32_t* ntrials, double* out5,
                int32_t accumulate, void* stream) {
    if (B < 1 || !cost || !descent || !ntrials || !out5) return einval("aoc_summary: NULL argument or B < 1");
    hipLaunchKernelGGL(k_summary, dim3(1), dim3(1024), 0, (hipStream_t)stream, B, cost, descent, ntrials, out5, accumulate);
    return check_launch("k_summary");
}

int aoc_streams_concurrent(void* stream_a, void* stream_b) {
    hipStream_t a = (hipStream_t)stream_a, b = (hipStream_t)stream_b;
    if (a == b) return 0;
    (void)hipGetLastError();   // an older sticky error is not this call's result
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    auto destroy = [&] { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); };
    for (int i = 0; i < 3; i++)
        if (hipEventCreate(&ev[i]) != hipSuccess) {
            destroy();
            (void)check_launch("aoc_streams_concurrent");   // leaves the reason in aoc_last_hip_error()
            return AOC_ELAUNCH;
        }
    hipEvent_t e0 = ev[0], e1 = ev[1], e2 = ev[2];
    int verdict = AOC_ELAUNCH;     // until a repetition has run to its end
    const long long ticks = 20000;   // 100 MHz constant clock: ~0.2 ms per kernel (the verdict is relative to ONE kernel's time)
    // (the events live on the CURRENT device: the caller selects the device the streams belong to — batch.concurrent_streams
    // does — and a record on a stream of another device fails here instead of leaving the elapsed times at zero)
    for (int rep = 0; rep < 2; rep++) {   // the first repetition also pays for loading the code object
        if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) break;
        float alone = 0.f, ta = 0.f, tb = 0.f;
        if (hipEventRecord(e0, a) != hipSuccess) break;        // one kernel by itself
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, ticks, (int*)nullptr);
        if (hipEventRecord(e1, a) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) break;
        if (hipEventElapsedTime(&alone, e0, e1) != hipSuccess) break;
        if (hipEventRecord(e0, a) != hipSuccess) break;        // one on each stream
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, ticks, (int*)nullptr);
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, ticks, (int*)nullptr);
        if (hipEventRecord(e1, a) != hipSuccess || hipEventRecord(e2, b) != hipSuccess) break;
        if (hipEventSynchronize(e1) != hipSuccess || hipEventSynchronize(e2) != hipSuccess) break;
        if (hipEventElapsedTime(&ta, e0, e1) != hipSuccess || hipEventElapsedTime(&tb, e0, e2) != hipSuccess) break;
        const float span = ta > tb ? ta : tb;      // both kernels done, from the start of the first
        verdict = span < 1.6f * alone ? 1 : 0;     // side by side: ~1x, one after the other: ~2x
    }
    destroy();
    const int rc = check_launch("aoc_streams_concurrent");
    if (rc) return rc;
    if (verdict < 0) snprintf(g_hip_err, sizeof g_hip_err, "aoc_streams_concurrent: an event record / elapsed-time query failed (streams of another device than the current one?)");
    return verdict;
}

int aoc_solve_trace(double* rows, int32_t cap_rows) {
    g_solve_trace.rows = rows;
    g_solve_trace.cap = rows ? cap_rows : 0;
    g_solve_trace.n = 0;
    return AOC_OK;
}
int32_t aoc_solve_trace_rows(void) { return g_solve_trace.n; }

int aoc_mpc_step(const aoc_problem* p_track, const aoc_problem* p_next, const aoc_params* prm, int32_t n_newton,
                 const void* x_cur, const double* u_cur, double* x0, double* x_true, const double* disturbance,
                 void* workspace, size_t workspace_bytes, double* Kgain, void* x_a, double* u_a, void* x_b, double* u_b,
                 double* J_a, double* J_b, double* descent, double* stepsize, int32_t* ntrials, int32_t* status, double* K0,
                 double* u_applied, int32_t* final_slot, const aoc_mpc_noise* noise, double* disturbance_out) {
    return aoc64::api_mpc_step(p_track, p_next, prm, n_newton, x_cur, u_cur, x0, x_true, disturbance, workspace, workspace_bytes, Kgain, x_a,
                               u_a, x_b, u_b, J_a, J_b, descent, stepsize, ntrials, status, K0, u_applied, final_slot, noise,
                               disturbance_out);
}

// ---- float32 arithmetic (aoc32): every array, the reference curves and the workspace are float32 ------
int aoc_traj_cost_f32(const aoc_problem* p, const float* x, const float* u, const float* x0, float* J) {
    if (p && p->ref_per_traj) return einval("aoc_traj_cost_f32: per-trajectory reference curves exist in the fp64 build only");
    return aoc32::api_traj_cost(p, x, u, x0, J);
}
int aoc_initial_trajectory_f32(const aoc_problem* p, double kp, double kt, const float* x0, float* x, float* u) {
    if (p && p->ref_per_traj) return einval("aoc_initial_trajectory_f32: per-trajectory reference curves exist in the fp64 build only");
    return aoc32::api_initial_trajectory(p, (float)kp, (float)kt, x0, x, u);
}
int aoc_rollout_cost_f32(const aoc_problem* p, const float* x0, const float* u, const float* du, const float* alpha,
                         float* x_out, float* u_out, float* J_out, int32_t* status) {
    if (p && p->ref_per_traj) return einval("aoc_rollout_cost_f32: per-trajectory reference curves exist in the fp64 build only");
    return aoc32::api_rollout_cost(p, x0, u, du, alpha, x_out, u_out, J_out, status);
}
size_t aoc_workspace_bytes_f32(int32_t B, int32_t T) { return aoc32::api_workspace_bytes(B, T); }
int aoc_newton_iterate_f32(const aoc_problem* p, const aoc_params* prm, int32_t kk, const float* x, const float* u,
                           const float* x0, const float* J_cur, void* workspace, size_t workspace_bytes, float* x_new,
                           float* u_new, float* J_new, float* descent, float* stepsize, int32_t* ntrials, int32_t* status) {
    if (p && p->ref_per_traj) return einval("aoc_newton_iterate_f32: per-trajectory reference curves exist in the fp64 build only");
    return aoc32::api_newton_iterate(p, prm, kk, x, u, x0, J_cur, workspace, workspace_bytes, x_new, u_new, J_new, descent,
                                     stepsize, ntrials, status);
}

}  // extern "C"
