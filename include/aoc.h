/*
 * aoc.h — C-ABI of libaoc_hip.so: batched Newton/LQR trajectory optimiser for the 6-state /
 * 2-input planar aircraft, hand-written HIP for MI355X (gfx950).
 *
 * The reference (MohamedAtwan/AirCraftOptimalControl) is pure Python and has no FFI; the boundary
 * this library replaces is its Python call surface, cited per entry point below as file:line of
 * the reference.  Every entry point takes plain pointers and sizes only.
 *
 * Conventions
 *   - All array pointers are DEVICE pointers (HBM), allocated and owned by the caller; the
 *     library keeps no pointer after a call returns and allocates nothing.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls enqueue work
 *     on that stream and return; they do not synchronise (the one exception: aoc_newton_solve with
 *     sync_every > 0 reads a 4-byte counter back every sync_every iterations).
 *   - Return value: AOC_OK (0) or a negative AOC_E* code; aoc_strerror() names it.  Numerical
 *     trouble of individual trajectories is reported in the per-trajectory `status` words, never by
 *     the return value.
 *
 * Device ("tiled") trajectory layout
 *   A batch of B trajectories is cut into tiles of AOC_TILE = 64 consecutive trajectories (one
 *   wavefront each); ntiles = ceil(B/64).  A trajectory array with C components and T samples is
 *       elem(b, t, c)  at  (((b/64)*T + t)*C + c)*64 + (b%64)            [fp64]
 *   i.e. [tile][t][component][lane], lane fastest, so that one wavefront reads/writes 512
 *   contiguous bytes per (t, component) and walks its own contiguous slab over the horizon.
 *   aoc_pack()/aoc_unpack() convert from/to the reference's per-trajectory (C,T) C-order arrays
 *   stacked as (B,C,T).  Lanes of the last tile beyond B replicate trajectory B-1.
 *   Per-trajectory scalars are plain arrays of length ntiles*64; per-trajectory 6-vectors (x0) are
 *   [ntiles][6][64] fp64.
 *
 * State storage
 *   Every state the reference propagates is a float32 value widened to fp64: Dynamics.step writes
 *   its result into a float32 array (aircraft_simplified.py:300).  State trajectories (the tiled C=6
 *   arrays `x`, `x_new`, `x_opt`, `x_reg`, declared `void*`) may therefore be stored as float32 without
 *   loss for samples t >= 1, which halves their HBM traffic.  Sample 0 is x0 — any fp64 value — and is
 *   ALWAYS taken from the separate fp64 `x0` argument, never from the array.  aoc_problem.x_in_f32 /
 *   x_out_f32 select the element type of the state arrays a call reads / writes.  A caller-supplied
 *   initial iterate with arbitrary fp64 samples must be passed as fp64 (x_in_f32 = 0).
 *   Inputs u, du, gains and costs are always fp64.
 */
#ifndef AOC_H
#define AOC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Revision of this ABI: bumped whenever a struct layout, an argument list or the meaning of a size query changes.
 * aoc_abi_version() returns the value the library was built with; a binding compares it with the header (or the
 * constant) it was written against and refuses to run on a mismatch instead of passing shifted arguments.
 *   3: round 3 (scratch_bytes / cand_bytes on the pass-level entries; aoc_workspace_bytes(B,T) covers exactly B)
 *   4: aoc_tuning + solve_repack_pct / solve_sync_fast / solve_split_tiles / track_hcut / bw_hcut; aoc_newton_solve2,
 *      aoc_summary, aoc_solve_trace, aoc_abi_version, aoc_backward_scratch_bytes, aoc_streams_concurrent; aoc_backward takes a scratch region;
 *      aoc_solve_workspace_bytes includes a fourth iterate buffer; aoc_default_nspec counts the step of an exhausted
 *      search (armijo_maxiters + 1 where every candidate rides along)
 *   5: aoc_mpc_step takes aoc_mpc_noise (the disturbance drawn on the device) and disturbance_out; the horizon cut is
 *      decided once per aoc_newton_solve from the caller's batch (a trajectory's bits no longer depend on the generation or
 *      half it is solved in); aoc_tuning.fw_wpe1, hcut_chain6, bw_hcut_full, fw_duo, hcut_waves, hcut_pairs */
#define AOC_ABI_VERSION 5

#define AOC_TILE 64
#define AOC_NS 6
#define AOC_NI 2

enum {
    AOC_OK = 0,
    AOC_EINVAL = -1,   /* bad argument (NULL pointer, B/T out of range, ...) */
    AOC_ELAUNCH = -2,  /* HIP launch / runtime error; see aoc_last_hip_error() */
    AOC_ENODEV = -3    /* no usable gfx950 device */
};

/* per-trajectory status bit flags (SURVEY 8b "error conventions") */
enum {
    AOC_ST_NAN = 1,          /* NaN/Inf met in cost or descent */
    AOC_ST_VNONPOS = 2,      /* V <= 0 met in a rollout (division by V, aircraft_simplified.py:310) */
    AOC_ST_SINGULAR = 4,     /* det(M) == 0 (np.linalg.inv would raise, optcon.py:728/:751) */
    AOC_ST_REGULARISED = 8,  /* M + 0.5 I applied at some stage (optcon.py:745-749) */
    AOC_ST_ARMIJO_EXH = 16,  /* line search exhausted; untested step applied (optcon.py:243-273) */
    AOC_ST_CONVERGED = 32    /* descent >= term_cond reached (optcon.py:499) */
};

/* aircraft_simplified.py:108-118 (Dynamics.__init__) */
typedef struct aoc_model {
    double cd0, cda, cla, m, g, S, rho, J, dt;
} aoc_model;

/* Problem = Dynamics constants + Cost(QQt,RRt,QQT) (aircraft_simplified.py:20-23) + reference
 * curves (NewtonMethod.__init__ xx_ref/uu_ref, optcon.py:335-339) + batch geometry. */
typedef struct aoc_problem {
    aoc_model model;
    double QQt[36];   /* row-major 6x6 */
    double RRt[4];    /* row-major 2x2 */
    double QQT[36];   /* row-major 6x6 */
    int32_t B;        /* trajectories */
    int32_t T;        /* samples per trajectory = int(tf/dt), optcon.py:378 (T-1 stages) */
    int32_t x_in_f32; /* element type of the tiled STATE arrays a call reads: 0 = fp64, 1 = float32 */
    int32_t x_out_f32;/* ... and of those it writes (see "State storage" above) */
    int32_t ref_per_traj; /* 0: `ref` is ONE curve shared by the batch, time-major [T][8]: xx_ref[0..5,t], uu_ref[0..1,t]
                             (what NewtonMethod.__init__ captures, optcon.py:335-339, for every trajectory);
                             1: one curve PER TRAJECTORY (B NewtonMethod instances with their own xx_ref/uu_ref): `ref` is
                             a tiled array with C = 8 components, elem(b,t,c) at (((b/64)*ref_T + t)*8 + c)*64 + b%64.
                             fp64 entry points only. */
    int32_t ref_T;    /* samples per trajectory in a per-trajectory `ref` array (0 = T); > T lets a caller keep a long
                         curve on the device and pass windows of it by offsetting `ref` by s*8*64 elements */
    int32_t x_is_rollout; /* 1: the state array a call READS is the rollout of its input array `u` from `x0` as this library
                             computes it — x_new of aoc_linesearch / aoc_newton_iterate, x_out of aoc_rollout_cost, x of
                             aoc_initial_trajectory — so a pass that walks forward in time may re-compute the states
                             instead of reading them (aoc_forward does: 24 B per stage less).  0: arbitrary states (a
                             caller-supplied initial iterate, optcon.py:395): they are read.  Same results either way. */
    int32_t reserved;
    const void *ref;  /* DEVICE; fp64 (float32 for the *_f32 entry points) */
    void *stream;     /* hipStream_t */
} aoc_problem;

/* Solver parameters = NewtonMethod constructor arguments (optcon.py:335-339) plus the two constants
 * the reference hard-codes. */
typedef struct aoc_params {
    int32_t max_iters;       /* optcon.py:367 */
    int32_t armijo_maxiters; /* optcon.py:230 */
    double stepsize_0;       /* optcon.py:224 */
    double cc;               /* optcon.py:227 */
    double beta;             /* optcon.py:229 */
    double term_cond;        /* -1e-6 hard-coded at optcon.py:368 (constructor value is ignored) */
    int32_t hessian_switch;  /* 8: full Hessian when kk > 8, optcon.py:443 */
    int32_t reserved;
} aoc_params;

/* Scheduling knobs.  They select kernel variants and launch shapes only — results never depend on them
 * (tests/test_gpu_parity.py checks bit-identity across settings) — with ONE exception, the two *_hcut knobs at the end:
 * the horizon cut evaluates the Riccati recursion in another order (gains equal to ~1e-14 of their scale; Armijo steps,
 * trial counts and float32 states identical in every test, tests/test_gpu_hcut.py).  The defaults are read ONCE per process, at
 * the first call that needs them, from the environment variables named below; aoc_set_tuning() replaces them
 * (test hook / tuning tools), aoc_set_tuning(NULL) restores the defaults.  The settings are one plain process-wide
 * struct: aoc_set_tuning() must not run concurrently with any other call of this library.  (It MAY run between
 * aoc_linesearch_search and aoc_linesearch_update: the search records the scheme it used in `scratch`, and the update
 * resolves by that record, not by the settings of the moment.) */
typedef struct aoc_tuning {
    int32_t nspec;          /* AOC_NSPEC        Armijo candidates riding along in the forward pass; 0 = by batch size */
    int32_t split_tiles;    /* AOC_SPLIT_TILES  several wavefronts per tile in forward/final/rollout/gains up to this many tiles (512) */
    int32_t split_bw_tiles; /* AOC_SPLIT_BW_TILES  ... in the backward pass (512) */
    int32_t fw_lin;         /* AOC_FW_LIN       small-batch forward pass with the nominal-point work (cost gradients, sin/cos, Jacobians) on a
                               wavefront of its own beside the LQR wavefront (k_forward_lin): 1 = always, 0 = never, -1 = where it
                               wins: at most two candidates riding along (-1) */
    int32_t ls_wcap;        /* AOC_LS_WCAP      wavefronts a trial round may occupy (0 = 1024) / the first work list may take with every remaining candidate (0 = 2048) */
    int32_t ls_kgrow;       /* AOC_LS_KGROW     round-based line search: growth of the candidates per round; 0 = by batch size */
    int32_t trial_split;    /* AOC_TRIAL_SPLIT  two-wavefront trial kernels for latency-bound rounds (1) */
    int32_t solve_norepack; /* AOC_SOLVE_NOREPACK  aoc_newton_solve never re-packs (0) */
    int32_t ls_worklist;    /* AOC_LS_WORKLIST  work-list line search: -1 = above split_tiles tiles, 0 = never, 1 = always */
    int32_t ls_cpl;         /* AOC_LS_CPL       work-list line search: candidate steps per work item / lane (1, 2 or 4; default 1) */
    int32_t ls_depth_min;   /* AOC_LS_DEPTH_MIN work-list line search: candidates tried in the first round at least (2) */
    int32_t fw_recompute;   /* AOC_FW_RECOMPUTE aoc_forward re-computes the states when aoc_problem.x_is_rollout allows it (1) */
    int32_t store_candidates; /* AOC_STORE_CANDIDATES aoc_newton_iterate: small batches keep the trajectories of the Armijo candidates
                               rolled out in the forward pass, the update then copies the accepted one (1) */
    int32_t bw4_tiles;      /* AOC_BW4_TILES    Gauss-Newton backward pass on four wavefronts per tile (one producer, Riccati columns over three) up to this many tiles (256) */
    int32_t bw5;            /* AOC_BW5          ... with the producer itself on two wavefronts (five per tile, k_backward5) (1) */
    int32_t solve_repack_pct;  /* AOC_SOLVE_REPACK_PCT  aoc_newton_solve re-packs when at most this per cent of the batch in flight still iterates (70) */
    int32_t solve_sync_fast;   /* AOC_SOLVE_SYNC_FAST   ... and reads the count every this many iterations once trajectories have begun to stop (2) */
    int32_t solve_split_tiles; /* AOC_SOLVE_SPLIT_TILES aoc_newton_solve2 cuts batches of at least this many tiles in two halves on its two streams (2048) */
    int32_t track_hcut;        /* AOC_TRACK_HCUT        aoc_mpc_step: tracking gains with the HORIZON cut in this many segments that run in parallel
                                  (k_track_hcut_*; 0 = never, -1 = 16 for batches of at most 64 tiles, 8 up to 128 tiles (-1)).  Unlike every other knob this one
                                  changes the order of the arithmetic: gains agree with the sequential kernels to ~1e-14 of their scale */
    int32_t bw_hcut;           /* AOC_BW_HCUT           the same for the Gauss-Newton backward pass of aoc_newton_iterate (k_bw_hcut; full-Hessian passes, see bw_hcut_full: at most
                                  64 tiles) (-1) */
    int32_t fw_wpe1;           /* AOC_FW_WPE1           small-batch forward pass: launches of at most one workgroup per CU run the build
                                  compiled for one wavefront per SIMD (512 registers, nothing spilt) (1) */
    int32_t hcut_chain6;       /* AOC_HCUT_CHAIN6       horizon cut: the serial chain of boundary hops on four wavefronts per tile (three take two
                                  columns of the dense 6x6 algebra each, one stages the next map through LDS: k_hcut_chain6; bit-identical
                                  to the one-wavefront chain).  Measured SLOWER than one wavefront (102 vs 73 us), hence off (0) */
    int32_t bw_hcut_full;      /* AOC_BW_HCUT_FULL      the horizon cut also for full-Hessian backward passes (costate maps first; tiles with an
                                  indefinite or singular M anywhere are recomputed by the sequential kernel, bit-identical to an uncut pass; 2: only
                                  those with a singular / ill-conditioned M or a value function that is not a number) (2) */
    int32_t fw_duo;            /* AOC_FW_DUO            forward pass of tiny batches (every candidate rides along, tiles x ceil(candidates / 2) <= 256)
                                  with the stage on seven roles, two candidates per workgroup (k_forward_duo); 0 = never, 1 = up to 256 workgroups,
                                  n > 1 = up to n workgroups (1) */
    int32_t hcut_waves;        /* AOC_HCUT_WAVES        segment kernels of the horizon cut on three / two wavefronts per (tile, segment) (linearisation |
                                  H half | Phi half of a map stage; linearisation | recursion and gains: k_*_hcut_map3, k_*_hcut_gains2), bit-identical
                                  to the one-wavefront kernels; 0 = never, 1 = where every workgroup gets a CU of its own (tiles x segments <= 256),
                                  2 = always (1) */
    int32_t hcut_pairs;        /* AOC_HCUT_PAIRS        the maps of neighbouring segments of the horizon cut composed pairwise before the chain of
                                  boundary hops (k_hcut_pair, k_hcut_odd): about S/2 + 2 hop-times in series instead of S - 2.  Like the cut
                                  itself another order of the same arithmetic (gains to 1e-14 of their scale); 0 = never, 1 = for cuts in
                                  at least 12 segments (i.e. where the cut itself is decided: by the caller's tile count), 2 = whenever
                                  there is a pair (1) */
} aoc_tuning;
void aoc_get_tuning(aoc_tuning *out);
void aoc_set_tuning(const aoc_tuning *t);

const char *aoc_version(void);
int32_t aoc_abi_version(void);   /* AOC_ABI_VERSION of the library's build */
const char *aoc_strerror(int code);
const char *aoc_last_hip_error(void);
/* number of doubles in a tiled array of C components: ntiles*T*C*64 */
size_t aoc_tiled_elems(int32_t B, int32_t T, int32_t C);
int32_t aoc_ntiles(int32_t B);

/* (B,C,T) C-order  <->  tiled.  Replaces nothing in the reference: it is the price of the batch
 * layout.  src/dst are device pointers. */
int aoc_pack(int32_t B, int32_t T, int32_t C, const double *src_bct, double *dst_tiled, void *stream);
int aoc_unpack(int32_t B, int32_t T, int32_t C, const double *src_tiled, double *dst_bct, void *stream);
/* same with a float32 tiled array (state storage) */
int aoc_pack_f32(int32_t B, int32_t T, int32_t C, const double *src_bct, float *dst_tiled, void *stream);
int aoc_unpack_f32(int32_t B, int32_t T, int32_t C, const float *src_tiled, double *dst_bct, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Unit level
 * --------------------------------------------------------------------------------------------- */

/* Dynamics.step(xx,uu[,lmbd])  (aircraft_simplified.py:263-393), n independent points.
 * x (n,6), u (n,2), lmbd (n,6) or NULL; outputs (any may be NULL except xp): xp (n,6) [fp64 values
 * rounded to float32, :300], fx (n,6,6) = A^T, fu (n,2,6) = B^T, and when lmbd != NULL the
 * contracted fxx (n,6,6), fuu (n,2,2), fux (n,2,6)  (:384-388). */
int aoc_step_batch(const aoc_model *model, int32_t n, const double *x, const double *u, const double *lmbd,
                   double *xp, double *fx, double *fu, double *fxx, double *fuu, double *fux, void *stream);

/* Cost.stagecost / Cost.termcost (aircraft_simplified.py:25-97), n independent points.
 * x,xr (n,6); u,ur (n,2).  ll (n), lx (n,6), lu (n,2); llT (n), lTx (n,6).  Any output may be NULL. */
int aoc_cost_batch(const aoc_problem *prob, int32_t n, const double *x, const double *u, const double *xr,
                   const double *ur, double *ll, double *lx, double *lu, double *llT, double *lTx, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Pass level (tiled layout).  One call = one pass over the horizon for the whole batch.
 * --------------------------------------------------------------------------------------------- */

/* Cost of a stored trajectory: the loop at optcon.py:417-424.  J[ntiles*64]. */
int aoc_traj_cost(const aoc_problem *prob, const void *x, const double *u, const double *x0, double *J);

/* Dynamics.get_initial_trajectory (aircraft_simplified.py:126-148): P-controller rollout
 *   u_i = [kp((X-Xr)+(Z-Zr)), kt((th-thr)+(ga-gar))] against xx_ref[:, i+1], from x0 ([ntiles][6][64]).
 * The reference uses kp = 5, kt = 2.5.  Evaluated in fp64 with the float32 state rounding of step();
 * the reference's own call runs mostly in float32 (it feeds step() its float32 output), so results
 * agree to ~1e-4 only — this produces an initial GUESS.  x, u tiled outputs. */
int aoc_initial_trajectory(const aoc_problem *prob, double kp, double kt, const double *x0, void *x, double *u);

/* get_update (optcon.py:176-200) fused with the cost loop of one Armijo trial (optcon.py:250-264):
 * u' = u + alpha[b]*du, x' rolled out from x0 with the float32 state rounding, J' accumulated.
 * x0 [ntiles][6][64]; alpha, J_out [ntiles*64]; du may be NULL (alpha ignored: plain rollout of u).
 * x_out/u_out may be NULL (cost only).  status is OR-ed. */
int aoc_rollout_cost(const aoc_problem *prob, const double *x0, const double *u, const double *du,
                     const double *alpha, void *x_out, double *u_out, double *J_out, int32_t *status);

/* Backward pass of one Newton iteration: terminal condition, quadratisation (Gauss-Newton, or full
 * Hessian with the costate sweep lambda_t = A^T lambda_{t+1} + l_x, optcon.py:461) and the affine
 * Riccati/gain recursion of ltv_LQR, fused (optcon.py:429-464 + :655-751).  Writes per stage the gain
 * K~ (2x7: column 0 feed-forward sigma, columns 1..6 feedback K), stored column by column with the two rows of a
 * column side by side (one 16-byte access per column and lane):
 * Kt: [ntiles][T][7][64][2] doubles, elem(b,t,p,row) at ((((b/64)*T + t)*7 + p)*64 + b%64)*2 + row — the size of a
 * tiled C=14 array, aoc_tiled_elems(B,T,14); sample T-1 unused.
 * lmbd0 (optional, [ntiles][6][64]) receives lambda_0 (forces the costate sweep).
 * scratch (may be NULL): device memory of scratch_bytes bytes that lets a Gauss-Newton pass of a batch of at most 128 tiles
 * run with the horizon cut in parallel segments (aoc_tuning.bw_hcut; aoc_backward_scratch_bytes(B, T) says how much it
 * takes; aoc_newton_iterate lends the part of its workspace behind K~).  Without it the sequential kernels run. */
size_t aoc_backward_scratch_bytes(int32_t B, int32_t T);
int aoc_backward(const aoc_problem *prob, int32_t full_hessian, const void *x, const double *u,
                 const double *x0, double *Kt, double *lmbd0, int32_t *status, void *scratch, size_t scratch_bytes);

/* Descent direction of GradientMethod.optimize (optcon.py:101-123): the costate sweep lambda_t = A^T lambda_{t+1} + l_x
 * from lambda_{T-1} = grad l_T and du_t = -(B_t^T lambda_{t+1} + l_u); du tiled C=2 (sample T-1 zero).
 * slope[b] = -sum_t |du_t|^2, the directional derivative of the cost along du — the `descent` argument aoc_linesearch
 * expects (the reference accumulates +sum |du_t|^2 under that name, :123, prints it and stops when it is <= 1e-6).
 * One steepest-descent iteration = aoc_gradient + aoc_linesearch with n_spec = 0 (J_trial = NULL).  The reference's
 * own GradientMethod.optimize cannot run (it calls armijo_stepsize with 8 of its 9 arguments, optcon.py:125 vs :204 —
 * TypeError): this is that loop with the missing JP = JJ[kk] supplied, parity unpinned (oracle restatement only). */
int aoc_gradient(const aoc_problem *prob, const void *x, const double *u, const double *x0, double *du,
                 double *slope, int32_t *status);

/* Forward pass: closed-loop linear rollout of ltv_LQR (optcon.py:756-762) giving du; the descent
 *   sum_t (B_t^T lambda_{t+1} + r_t)^T du_t   (optcon.py:474-477)
 * evaluated through the adjoint identity  sum_t (q_t^T dx_t + r_t^T du_t) + q_f^T dx_{T-1}  (same number,
 * 1e-13 relative on the golden cases; the costate then never leaves the backward pass); fused with
 * the first n_spec (1..aoc_spec_max()) Armijo trials: for step alpha_j = stepsize_0*beta^j, u' = u + alpha_j*du,
 * nonlinear rollout x' from x0, cost J'_j (optcon.py:250-264).  The trials ride along with the K~
 * stream; the reference evaluates them one after the other, the verdict order is kept by
 * aoc_linesearch.
 * Outputs: du (tiled C=2), descent[ntiles*64], J_trial[n_spec][ntiles*64].
 * cand (may be NULL; used when n_spec > 3 or the batch runs its passes on several wavefronts per tile): device memory of
 * cand_bytes >= aoc_candidate_bytes(B, T, n_spec) bytes (checked: a smaller region is AOC_EINVAL, not a memory fault) in
 * which every trial keeps the trajectory it rolls out (x' as float32, u',
 * flags); handed to aoc_linesearch / aoc_linesearch_update together with n_spec and J_trial, the update of a tile whose
 * trajectories all accepted one of these candidates is a copy, parallel over the horizon, instead of one more serial
 * rollout — the same values either way.  ntrials_hint (may be NULL): the trial counts of the previous iteration (the
 * `ntrials` array aoc_linesearch filled; zeros = none): a tile then stores only as many candidates as its trajectories
 * needed last time plus two, and the update falls back to the rollout for a tile that needed more.
 * aoc_default_ncand(B, n_spec, armijo_maxiters): the number of candidates aoc_newton_iterate keeps for such a batch
 * (n_spec, or 0 = none: the batch is too large, or not every candidate rides along). */
size_t aoc_candidate_bytes(int32_t B, int32_t T, int32_t n_spec);
int32_t aoc_default_ncand(int32_t B, int32_t n_spec, int32_t armijo_maxiters);
int aoc_forward(const aoc_problem *prob, const aoc_params *prm, int32_t n_spec, const void *x, const double *u,
                const double *x0, const double *Kt, double *du, double *descent, double *J_trial,
                int32_t *status, void *cand, size_t cand_bytes, const int32_t *ntrials_hint);

/* Armijo back-tracking (optcon.py:243-273) and the final update (optcon.py:488-491).
 * Trial ii uses alpha_ii = stepsize_0*beta^ii and is accepted iff J'(alpha_ii) <= J_cur +
 * cc*alpha_ii*descent; trials 0..n_spec-1 are judged from J_trial (written by aoc_forward; n_spec = 0: none,
 * J_trial may be NULL).
 * Trajectories still rejected search on: small batches in rounds over a compacted list, several
 * candidate steps of each at once when few remain; large batches through a work list of (trajectory,
 * candidate step) items, one cost-only rollout per item and lane, the items of one trajectory neighbours on
 * the list (aoc_tuning.ls_cpl = 2 or 4 lets an item carry that many consecutive candidates from one read of
 * (u, du): measured slower, not the default).  The accepted index is the first one that passes, as in the reference's
 * sequential loop, whatever the schedule (aoc_tuning).  On exhaustion the untested
 * stepsize_0*beta^armijo_maxiters is used (Q5).  Finally EVERY trajectory is rolled out with its
 * step into x_new/u_new and J_new.  stepsize[b], ntrials[b] report the result.
 * armijo_maxiters <= 63.  scratch: device memory of scratch_bytes >= aoc_linesearch_scratch_bytes(B, T) bytes.  cand: NULL,
 * or the candidate store aoc_forward filled for the same n_spec (see there), of cand_bytes >= aoc_candidate_bytes(B, T,
 * n_spec) bytes.  Both sizes are checked: too small a region is AOC_EINVAL, not a memory fault. */
size_t aoc_linesearch_scratch_bytes(int32_t B, int32_t T);
/* Largest n_spec aoc_forward / aoc_linesearch take, and the n_spec aoc_newton_iterate uses for a batch of B
 * trajectories: 2 in general, armijo_maxiters for batches small enough to give every candidate step its own
 * wavefront (environment variable AOC_NSPEC overrides). */
int32_t aoc_spec_max(void);
int32_t aoc_default_nspec(int32_t B, int32_t armijo_maxiters);
int aoc_linesearch(const aoc_problem *prob, const aoc_params *prm, int32_t n_spec, const double *u,
                   const double *x0, const double *du, const double *J_cur, const double *descent,
                   const double *J_trial, void *x_new, double *u_new, double *J_new, double *stepsize,
                   int32_t *ntrials, int32_t *status, void *scratch, size_t scratch_bytes, const void *cand,
                   size_t cand_bytes);

/* The two halves of aoc_linesearch, which is exactly _search followed by _update on the same arguments:
 *   aoc_linesearch_search = armijo_stepsize for every trajectory (optcon.py:204-327): stepsize[b], ntrials[b]; the
 *     verdicts of the last round stay in `scratch` until _update resolves them.  On entry ntrials[b] is read as a
 *     scheduling hint — the trial count of the previous iteration, or anything — which decides how many candidate
 *     steps a trajectory tries at once, never what it accepts;
 *   aoc_linesearch_update = get_update with that step (optcon.py:488-491, :176-200): x_new, u_new, J_new, status.
 * (Separate entry points so that a caller can time or overlap the latency-bound search and the streaming update.) */
int aoc_linesearch_search(const aoc_problem *prob, const aoc_params *prm, int32_t n_spec, const double *u,
                          const double *x0, const double *du, const double *J_cur, const double *descent,
                          const double *J_trial, double *stepsize, int32_t *ntrials, void *scratch,
                          size_t scratch_bytes);
int aoc_linesearch_update(const aoc_problem *prob, const aoc_params *prm, const double *u, const double *x0,
                          const double *du, void *x_new, double *u_new, double *J_new, double *stepsize,
                          int32_t *ntrials, int32_t *status, void *scratch, size_t scratch_bytes, int32_t n_spec,
                          const double *J_trial, const void *cand, size_t cand_bytes);

/* lqr_tracking.lqr_tracking (lqr_tracking.py:245-283): linearise about (x_opt,u_opt), non-augmented
 * Riccati/gain recursion with the constant weights QQt,RRt,QQT of `prob` and S = 0
 * (lqr_tracking.py:276), then the closed-loop nonlinear rollout u = u_opt + K (x - x_opt) from
 * x0_reg (= x_opt[:,0] + perturbation).  x_opt0 = sample 0 of x_opt, x0_reg: [ntiles][6][64] fp64.
 * Kgain: tiled C=12 (row 0 then row 1 of the 2x6 gain; sample T-1 zero).  x_reg/u_reg/x0_reg may be
 * NULL (gains only).  No scratch and no workspace: every array of this call is a tiled trajectory array
 * (aoc_tiled_elems) or a per-trajectory vector. */
int aoc_lqr_tracking(const aoc_problem *prob, const void *x_opt, const double *u_opt, const double *x_opt0,
                     const double *x0_reg, double *Kgain, void *x_reg, double *u_reg, int32_t *status);

/* optcon.ltv_LQR(AAin,BBin,QQin,RRin,SSin,QQfin,TT,x0,qq,rr,qqf)  (optcon.py:533-771; identical copy
 * lqr_tracking.py:6-242) for nb independent problems with caller-supplied per-stage matrices, in
 * the reference's order of operations; augmented != 0 selects the affine (7x7) form (q, r, qf may
 * then be NULL = zeros).  Time-major per problem: A [nb][T][36], Bm [nb][T][12] (6x2), Q [nb][T][36],
 * R [nb][T][4], S [nb][T][12] (2x6), Qf [nb][36], x0 [nb][6], q [nb][T][6], r [nb][T][2], qf [nb][6].
 * Outputs: KK [nb][T][2n], PP [nb][T][n*n], xx [nb][T][6], uu [nb][T][2] with n = 6 or 7;
 * nreg[nb] = stages regularised (optcon.py:745-749), nsing[nb] = singular M met.  Generic, not a
 * throughput path (the Newton iteration uses aoc_backward/aoc_forward). */
int aoc_ltv_lqr(int32_t nb, int32_t T, int32_t augmented, const double *A, const double *Bm, const double *Q,
                const double *R, const double *S, const double *Qf, const double *x0, const double *q,
                const double *r, const double *qf, double *KK, double *PP, double *xx, double *uu,
                int32_t *nreg, int32_t *nsing, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Iteration level.  Workspace: caller-allocated device memory of aoc_workspace_bytes(B,T) bytes — what a batch of
 * exactly B trajectories needs (K~, du, trial costs, line-search scratch, and the candidate store where a batch of that
 * size keeps one) — its size is passed along (`workspace_bytes`) and checked: too small a workspace is an AOC_EINVAL,
 * not a memory fault; one without room for the candidate store makes the update roll out instead of copy (same
 * results).  aoc_solve_workspace_bytes() includes the largest candidate store of any batch up to B, because
 * aoc_newton_solve runs its re-packed, smaller generations in the workspace of the first.
 * --------------------------------------------------------------------------------------------- */
size_t aoc_workspace_bytes(int32_t B, int32_t T);

/* One outer iteration kk of NewtonMethod.optimize for every trajectory (optcon.py:415-491, steps
 * A-G of SURVEY 3.2): backward, forward, line search.  (x,u) current iterate, J_cur its cost
 * (from aoc_traj_cost for kk = 0, J_new of the previous iteration afterwards — the reference
 * recomputes the same number, optcon.py:417-424).  Results: x_new,u_new,J_new and the
 * per-trajectory scalars descent, stepsize, ntrials.  No trajectory is skipped (fixed-iteration
 * mode); convergence bookkeeping is the caller's (aoc_newton_solve does it). */
int aoc_newton_iterate(const aoc_problem *prob, const aoc_params *prm, int32_t kk, const void *x,
                       const double *u, const double *x0, const double *J_cur, void *workspace,
                       size_t workspace_bytes, void *x_new, double *u_new, double *J_new, double *descent,
                       double *stepsize, int32_t *ntrials, int32_t *status);

/* ---------------------------------------------------------------------------------------------
 * Solve level: NewtonMethod.optimize (optcon.py:341-529) for every trajectory, termination included.
 * --------------------------------------------------------------------------------------------- */
size_t aoc_solve_workspace_bytes(int32_t B, int32_t T);

/* The reference's outer loop kk = 0 .. max_iters-2 with its per-trajectory stopping rule, run on the
 * device without returning to the host between iterations:
 *   - cost of the initial iterate (optcon.py:417-424), then aoc_newton_iterate per kk for the batch in flight: a stopped
 *     trajectory keeps riding along in its tile with its results frozen until (sync_every > 0) the still-iterating ones
 *     are re-packed into a dense smaller batch (aoc_tuning.solve_repack_pct; never with per-trajectory reference curves);
 *   - a trajectory stops at the first kk with descent[kk] >= prm->term_cond (optcon.py:499, Q6; a NaN
 *     descent never stops) and returns iterate kk-1 (optcon.py:500-504, Q7): ret_index[b] = kk-1, where -1
 *     means the reference's all-zero last history slot and 0 the initial iterate;
 *   - a trajectory that never stops returns the last computed iterate, ret_index[b] = max_iters-1.  A DIVERGED trajectory
 *     never stops either (NaN >= term_cond is False, as in the reference): once every input sample of its iterate is NaN
 *     nothing changes any more, and after one more iteration has run on that iterate it is retired with exactly what the
 *     remaining iterations would produce (iterate, ret_index = iters = max_iters-1, status, history rows);
 *   - u_star[:, T-1] = u_star[:, T-2] (optcon.py:505, Q8).
 * (x_init, u_init): initial iterate, x_init of element type prob->x_in_f32, never written.  x0: fp64
 * [ntiles][6][64], xx_init[:,0] of every trajectory (optcon.py:398).
 * x_star (element type prob->x_out_f32), u_star: the returned iterates, tiled.  Sample 0 of x_star is the
 *   stored copy of x0 (rounded if float32): take it from x0.  For ret_index = 0 x_star/u_star hold the
 *   caller's initial iterate converted to the output type.  (The states of an iterate of index >= 1 are written at the END
 *   of the call, as the rollout of its inputs — which they are, bit for bit: only the inputs are copied when a
 *   trajectory stops.)
 * iters[b]: iterations the trajectory took part in; status[b]: flags raised up to its stopping iteration,
 *   AOC_ST_CONVERGED if it stopped by the descent test.
 * hist_* (each may be NULL): per-iteration scalars [max_iters-1][ntiles*64] — cost of the iterate the iteration
 *   started from, descent, accepted step, Armijo trials — NaN / -1 where the trajectory no longer iterated
 *   (rows beyond *n_run are not written).
 * sync_every: 0 = run all max_iters-1 iterations without any host synchronisation (and without re-packing); n > 0 = every n
 *   iterations — every aoc_tuning.solve_sync_fast once trajectories have begun to stop — read back the number of
 *   trajectories still iterating (one blocking 8-byte copy), stop launching when it is 0, re-pack when it is small.
 *   *n_run (host, may be NULL): iterations the histories cover = iterations launched, or max_iters-1 when a diverged
 *   trajectory was retired (its history rows run to the end).
 * workspace: aoc_solve_workspace_bytes(B,T) bytes of device memory. */
int aoc_newton_solve(const aoc_problem *prob, const aoc_params *prm, const void *x_init, const double *u_init,
                     const double *x0, void *workspace, size_t workspace_bytes, int32_t sync_every, void *x_star, double *u_star,
                     int32_t *iters, int32_t *ret_index, int32_t *status, double *hist_cost,
                     double *hist_descent, double *hist_stepsize, int32_t *hist_ntrials, int32_t *n_run);

/* Scalar summary of a shard — the five numbers the path's only collective reduces over the GPUs of a node (SURVEY 8e;
 * no reference counterpart: the reference prints one trajectory's cost and descent per iteration, optcon.py:497-498):
 *   out5[0] = sum of the finite costs, [1] = sum of the descents of those trajectories, [2] = sum of the Armijo trial
 *   counts, [3] = B, [4] = number of trajectories whose cost is not finite;   accumulate != 0 adds to what out5 holds
 * (a shard kept as several batches: one call per batch, in a fixed order).  cost, descent, ntrials: per-trajectory
 * device arrays as aoc_newton_iterate fills them (the first B entries are read); out5: DEVICE, five doubles.  One
 * workgroup with a fixed reduction order: the same bits from run to run. */
int aoc_summary(int32_t B, const double *cost, const double *descent, const int32_t *ntrials, double *out5,
                int32_t accumulate, void *stream);

/* aoc_newton_solve with a second stream: batches of at least aoc_tuning.solve_split_tiles tiles are cut in two halves
 * (by tiles), each solved as aoc_newton_solve would solve it — generations, re-packing, histories — the first on
 * prob->stream, the second on stream2, driven by this one host thread: while it waits for one half's counter the other
 * half's launches are queued.  stream2 first waits for what the caller queued on prob->stream, and prob->stream
 * continues after both halves, so the call orders like aoc_newton_solve.  Same results (per-trajectory results never
 * depend on the batch a trajectory is solved in).  stream2 = NULL, or a smaller batch: exactly aoc_newton_solve. */
int aoc_newton_solve2(const aoc_problem *prob, const aoc_params *prm, const void *x_init, const double *u_init,
                      const double *x0, void *workspace, size_t workspace_bytes, int32_t sync_every, void *x_star, double *u_star,
                      int32_t *iters, int32_t *ret_index, int32_t *status, double *hist_cost,
                      double *hist_descent, double *hist_stepsize, int32_t *hist_ntrials, int32_t *n_run, void *stream2);

/* Do two HIP streams run side by side?  The runtime maps streams onto a few hardware queues (four by default) and two
 * streams on one queue take turns — a caller who cuts a batch in two halves on two such streams (aoc_newton_solve2, or two
 * solvers of its own) gets one-stream speed.  Runs one 0.2 ms single-wavefront kernel on each and compares the span:
 * 1 = concurrent, 0 = they share a queue (or are the same stream), < 0 = AOC_E*.  Blocking (synchronises both streams). */
int aoc_streams_concurrent(void *stream_a, void *stream_b);

/* Diagnostic (no reference counterpart): timeline of the aoc_newton_solve calls that follow.  rows: HOST memory for
 * cap_rows rows of 6 doubles — part (0, or 1 = the half on stream2), iteration kk, trajectories in flight in that
 * iteration (the generation's batch), its tiles, the count of still-iterating trajectories if the host read it after
 * that iteration (else -1), milliseconds from the start of the solve to the end of that iteration's launches (HIP
 * events on the part's stream; one more row per part closes the final bookkeeping).  NULL switches the trace off.  aoc_solve_trace_rows(): rows the last solve wrote.
 * Process-wide like aoc_set_tuning; tracing adds one event record per iteration and one synchronisation at the end. */
int aoc_solve_trace(double *rows, int32_t cap_rows);
int32_t aoc_solve_trace_rows(void);

/* ---------------------------------------------------------------------------------------------
 * Receding horizon (BASELINE.json configs[4]; the reference has no MPC, the loop is SURVEY 8d's "Config 5"
 * built from the reference's pieces).  One step for every instance, entirely on the device:
 *   1. gains K about the current optimum (x_cur,u_cur; sample 0 of x_cur is x0), weights of prob_track
 *      (lqr_tracking.py:268-276, :324-328) -> Kgain (tiled C=12, as aoc_lqr_tracking); K0 = K[:,:,0]
 *   2. plant: u_applied = u_cur[:,0] + K0 (x_true - x0)  (lqr_tracking.py:280),
 *             x_true <- Dynamics.step(x_true, u_applied)[0] + disturbance  (lqr_tracking.py:281), x0 <- x_true
 *   3. inputs shifted by one sample (the last one repeats), 4. warm start = their rollout from the new x0
 *      (get_update with step 0, optcon.py:176-200) against prob_next, whose `ref` is the reference window of the
 *      NEXT step
 *   5. n_newton iterations of aoc_newton_iterate with kk restarting at 0 (Gauss-Newton Hessian only).
 * State arrays are float32 (prob->x_in_f32 = x_out_f32 = 1).  (x_a,u_a,J_a) and (x_b,u_b,J_b): two iterate
 * buffers distinct from (x_cur,u_cur); *final_slot (host) = 0 / 1: which of them holds the new optimum and
 * its cost.  x0, x_true, disturbance (may be NULL), K0 ([ntiles][12][64], may be NULL), u_applied
 * ([ntiles][2][64]): per-instance vectors laid out like x0.  workspace: aoc_workspace_bytes(B,T).
 * The disturbance of step 2 is  disturbance[b] (caller's array, may be NULL)  +  the draw of the device's own model when
 * `noise` (HOST pointer, may be NULL) is given: component c of instance b gets sigma[c] * z, z standard normal from the
 * counter-based generator Philox4x32-10 with key = seed and counter = (first + b, step, c / 2, 0), two normals per counter by
 * Box-Muller (csrc/aoc_device.h: mpc_noise_draw) — a function of (seed, global instance index, step) alone: no state
 * between calls, the same draws whatever the sharding, no host round trip.  disturbance_out (DEVICE, [ntiles][6][64], may
 * be NULL) receives what was added, for a checker. */
typedef struct aoc_mpc_noise {
    uint64_t seed;
    uint32_t step;      /* closed-loop step index of this call */
    uint32_t first;     /* global index of instance 0 of this batch (shard offset) */
    double sigma[6];    /* standard deviation per state component */
} aoc_mpc_noise;
int aoc_mpc_step(const aoc_problem *prob_track, const aoc_problem *prob_next, const aoc_params *prm,
                 int32_t n_newton, const void *x_cur, const double *u_cur, double *x0, double *x_true,
                 const double *disturbance, void *workspace, size_t workspace_bytes, double *Kgain, void *x_a, double *u_a, void *x_b,
                 double *u_b, double *J_a, double *J_b, double *descent, double *stepsize, int32_t *ntrials,
                 int32_t *status, double *K0, double *u_applied, int32_t *final_slot, const aoc_mpc_noise *noise,
                 double *disturbance_out);

/* ---------------------------------------------------------------------------------------------
 * float32 arithmetic (BASELINE.json configs[2]: "fp32 with tolerance sweep").
 * The same kernels compiled with float as the arithmetic type: EVERY array (states, inputs, gains,
 * costs, per-trajectory scalars, x0, the reference curves `prob->ref`, the workspace) is float32 and
 * every operation is a float32 operation; aoc_problem/aoc_params keep their fp64 fields and are
 * converted once on the host.  prob->x_in_f32 / x_out_f32 are ignored.  Semantics otherwise as the
 * fp64 entry point of the same name.  This variant is NOT the parity path: the reference computes in
 * fp64; tests/test_gpu_f32.py reports how far the float32 iterates drift from the fp64 ones.
 * --------------------------------------------------------------------------------------------- */
int aoc_traj_cost_f32(const aoc_problem *prob, const float *x, const float *u, const float *x0, float *J);
int aoc_initial_trajectory_f32(const aoc_problem *prob, double kp, double kt, const float *x0, float *x, float *u);
int aoc_rollout_cost_f32(const aoc_problem *prob, const float *x0, const float *u, const float *du,
                         const float *alpha, float *x_out, float *u_out, float *J_out, int32_t *status);
size_t aoc_workspace_bytes_f32(int32_t B, int32_t T);
int aoc_newton_iterate_f32(const aoc_problem *prob, const aoc_params *prm, int32_t kk, const float *x,
                           const float *u, const float *x0, const float *J_cur, void *workspace,
                           size_t workspace_bytes, float *x_new, float *u_new, float *J_new, float *descent,
                           float *stepsize, int32_t *ntrials, int32_t *status);

#ifdef __cplusplus
}
#endif
#endif /* AOC_H */
