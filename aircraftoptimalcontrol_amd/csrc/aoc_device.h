// aoc_device.h — per-trajectory (per-lane) device math of the Newton/LQR path for gfx950.
//
// One lane owns one trajectory; everything here is straight-line fp64 register arithmetic on that
// lane's state.  The 6x6 dynamics Jacobian is  A = I + N  with 11 structural non-zeros in N and the
// input Jacobian B has 3, so the Riccati products are written out sparsely instead of as dense 6x6
// (or, as the reference does, 7x7 augmented) matrix products.
//
// Reference behaviour reproduced here (file:line of the reference):
//   Dynamics.step            aircraft_simplified.py:263-393  (x+ rounded to float32: :300)
//   Cost.stagecost/termcost  aircraft_simplified.py:25-97
//   ltv_LQR (augmented)      optcon.py:655-751  in the equivalent 6-dim affine form, see lqr_stage()
// This header is included once per arithmetic type: AOC_ARITH_NS names the namespace, AOC_REAL the type
// (double: the parity path; float: BASELINE config 3, "fp32 arithmetic everywhere").
#include <hip/hip_runtime.h>

#if !defined(AOC_ARITH_NS) || !defined(AOC_REAL)
#error "define AOC_ARITH_NS and AOC_REAL before including aoc_device.h"
#endif

#ifndef AOC_DEVICE_COMMON
#define AOC_DEVICE_COMMON
namespace aoc_common {
constexpr int TILE = 64;
}
#endif

// Contraction: products and sums are fused only where one source expression contains both ("on"), never across
// statements at the back end's discretion ("fast", the HIP default).  The roundings of a function are then the
// same in every kernel it is inlined into, which is what makes the multi-wavefront variants of a pass (aoc_passes.inc)
// bit-identical to the one-wavefront ones.  State propagation, cost and Armijo test switch contraction off.
#pragma clang fp contract(on)

namespace AOC_ARITH_NS {
using aoc_common::TILE;
typedef AOC_REAL real;
#define R(x) ((real)(x))

// Uniform (wave-invariant) constants: live in SGPRs / scalar loads from the kernarg segment.
struct KConst {
    // model (aircraft_simplified.py:108-118) and derived uniform values
    real cd0, cda, cla, m, g, S, rho, J, dt;
    real dtm;    // dt/m
    real mg;     // m*g
    real hrho;   // 0.5*rho
    real krs;    // rho*S
    real b41;    // dt/J
    real rJ;     // 1/J rounded to nearest (div_by_const)
    real Q[36], R[4], QT[36];
    int T, ntiles, B, diag;  // diag: Q,R,QT all diagonal (every driver of the reference)
    int rpt, refT;           // reference curves per trajectory (tiled, refT samples each) instead of one shared curve
};

// ---------------------------------------------------------------------------------------------
// tiled addressing: elem(tile, t, c, lane) = (((tile*T + t)*C + c)*64 + lane)
// ---------------------------------------------------------------------------------------------
template <int C>
__device__ __forceinline__ size_t tix(int tile, int T, int t, int c, int lane) {
    return (((size_t)tile * T + t) * C + c) * TILE + lane;
}

// ---------------------------------------------------------------------------------------------
// Nonlinear step x+ = f(x,u) with the reference's float32 rounding of the result, and the stage
// cost.  Written with contraction OFF and in the reference's association order so that a rollout
// is bit-identical to NumPy's given identical sin/cos values.
// ---------------------------------------------------------------------------------------------
struct SC { real sg, cg, sa, ca; };

// sin and cos of one argument.  fp64: branch-free for |x| < 2^20 (flight-path and attack angles are
// O(1)): Cody-Waite reduction by pi/2 with three fused steps, then the classic minimax kernels on
// [-pi/4, pi/4] (degree 13 / 14, < 1 ulp).  The library sincos() spends ~150 instructions per call,
// mostly on a Payne-Hanek path these arguments never take; this one is ~45 and has no control flow,
// so the two calls per stage interleave.  float32 arithmetic (config 3) uses the library's sincosf.
// The constants of the fp64 evaluation as a value: by default they fold back into literals (the compiler keeps them in
// SGPRs or re-materialises them with s_mov pairs, which is free beside other wavefronts' vector work); pin() turns them
// into VGPR residents for the small-batch kernels, see pin_consts().
template <typename T> struct TrigK {};
template <> struct TrigK<double> {
    double t2p = 6.36619772367581382433e-01;                                     // 2/pi
    double p1 = 1.57079632679489655800e+00, p2 = 6.12323399573676603587e-17, p3 = -1.49738490485916983689e-33;
    double s5 = 1.58969099521155010221e-10, s4 = -2.50507602534068634195e-08, s3 = 2.75573137070700676789e-06,
           s2 = -1.98412698298579493134e-04, s1 = 8.33333333332248946124e-03, s0 = -1.66666666666666324348e-01;
    double c5 = -1.13596475577881948265e-11, c4 = 2.08757232129817482790e-09, c3 = -2.75573143513906633035e-07,
           c2 = 2.48015872894767294178e-05, c1 = -1.38888888888741095749e-03, c0 = 4.16666666666666019037e-02;
};
__device__ __forceinline__ void sincos_fast(double x, double* sp, double* cp, const TrigK<double>& K = TrigK<double>()) {
    const double n = __builtin_rint(x * K.t2p);                                 // x * 2/pi
    double r = __builtin_fma(-n, K.p1, x);                                      // pi/2, exact step
    r = __builtin_fma(-n, K.p2, r);
    r = __builtin_fma(-n, K.p3, r);
    const int q = (int)n;
    const double z = r * r;
    // sin(r) = r + r z (S1 + z (S2 + ... ))
    double ps = __builtin_fma(z, K.s5, K.s4);
    ps = __builtin_fma(z, ps, K.s3);
    ps = __builtin_fma(z, ps, K.s2);
    ps = __builtin_fma(z, ps, K.s1);
    ps = __builtin_fma(z, ps, K.s0);
    const double sr = __builtin_fma(r * z, ps, r);
    // cos(r) = 1 - z/2 + z^2 (C1 + z (C2 + ... )), summed so that the leading 1 - z/2 stays exact
    double pc = __builtin_fma(z, K.c5, K.c4);
    pc = __builtin_fma(z, pc, K.c3);
    pc = __builtin_fma(z, pc, K.c2);
    pc = __builtin_fma(z, pc, K.c1);
    pc = __builtin_fma(z, pc, K.c0);
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double cr = w + (((1.0 - w) - hz) + (z * z) * pc);
    // quadrant
    const bool swap = q & 1;
    const double s = swap ? cr : sr, c = swap ? sr : cr;
    // sin changes sign in quadrants 2, 3 (bit 1 of q), cos in quadrants 1, 2 (bit 1 of q + 1): move that bit
    // onto the sign bit with integer operations (two per value instead of and/compare/negate/select)
    const unsigned qs = (unsigned)q << 30;
    const unsigned long long fs = (unsigned long long)(qs & 0x80000000u) << 32;
    const unsigned long long fc = (unsigned long long)((qs + 0x40000000u) & 0x80000000u) << 32;
    *sp = __longlong_as_double(__double_as_longlong(s) ^ (long long)fs);
    *cp = __longlong_as_double(__double_as_longlong(c) ^ (long long)fc);
}
// float32 arithmetic (config 3): the same scheme in single precision — three-step Cody-Waite by pi/2, the
// classic degree-7 / degree-8 kernels on [-pi/4, pi/4] (~1 ulp), branch-free; the library's sincosf (with its
// large-argument path) only for finite |x| >= 2^17, see trig().
__device__ __forceinline__ void sincos_fast(float x, float* sp, float* cp, const TrigK<float>& = TrigK<float>()) {
    const float n = __builtin_rintf(x * 6.36619772367581382433e-01f);
    float r = __builtin_fmaf(-n, 1.57079625129699707031e+00f, x);
    r = __builtin_fmaf(-n, 7.54978941586159635335e-08f, r);
    r = __builtin_fmaf(-n, 5.39030285815811905290e-15f, r);
    const int q = (int)n;
    const float z = r * r;
    float ps = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = __builtin_fmaf(z, ps, -1.6666654611e-1f);
    const float sr = __builtin_fmaf(r * z, ps, r);
    float pc = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = __builtin_fmaf(z, pc, 4.166664568298827e-2f);
    const float cr = __builtin_fmaf(z * z, pc, __builtin_fmaf(z, -0.5f, 1.0f));
    const bool swap = q & 1;
    const float s = swap ? cr : sr, c = swap ? sr : cr;
    const unsigned qs = (unsigned)q << 30;
    *sp = __uint_as_float(__float_as_uint(s) ^ (qs & 0x80000000u));
    *cp = __uint_as_float(__float_as_uint(c) ^ ((qs + 0x40000000u) & 0x80000000u));
}
__device__ __forceinline__ void sincos_lib(double x, double* sp, double* cp) { sincos(x, sp, cp); }
__device__ __forceinline__ void sincos_lib(float x, float* sp, float* cp) { sincosf(x, sp, cp); }

// Both angle pairs of a stage.  The two fast evaluations are independent straight-line code (they
// interleave); ONE rarely-taken branch afterwards redoes them with the library for huge finite
// arguments.  (A wave-uniform shortcut for |angle| <= pi/4, where the reduction is the
// identity, was measured and bought nothing: the extra branch costs what the dozen instructions save.)
// trig() = trig_fast() + trig_fix() under trig_huge(); a kernel that evaluates several chains per stage (k_forward: the
// nominal point and the Armijo trials) calls the fast parts of all of them first and guards ONE fix-up region with the OR
// of the tests, so that the common path of the stage is one basic block the scheduler can interleave freely — with a
// branch after every sin/cos pair the six polynomial evaluations of a stage sat in three blocks, one chain after the other.
__device__ __forceinline__ SC trig_fast(real th, real ga, const TrigK<real>& K = TrigK<real>()) {
    SC s;
    sincos_fast(ga, &s.sg, &s.cg, K);
    sincos_fast(th - ga, &s.sa, &s.ca, K);
    return s;
}
// finite and huge only: for NaN and +-inf the fast path already returns NaN, as sin/cos do.  (A diverged
// trajectory is NaN from some stage on; sending it through the library made its wavefront the straggler
// of the launch: 13 NaN trajectories in 65 536 cost the forward pass +55 %.)
__device__ __forceinline__ bool trig_huge(real th, real ga) {
    const real aga = __builtin_fabs(ga), aal = __builtin_fabs(th - ga), inf = __builtin_inf();
    const real big = sizeof(real) == 8 ? R(1048576.0) : R(131072.0);  // where the Cody-Waite reduction stops being exact
    return (aga >= big && aga < inf) || (aal >= big && aal < inf);
}
__device__ __forceinline__ void trig_fix(real th, real ga, SC& s) {
    if (trig_huge(th, ga)) {
        sincos_lib(ga, &s.sg, &s.cg);
        sincos_lib(th - ga, &s.sa, &s.ca);
    }
}
__device__ __forceinline__ SC trig(real th, real ga, const TrigK<real>& K = TrigK<real>()) {
    SC s = trig_fast(th, ga, K);
    trig_fix(th, ga, s);
    return s;
}

// Small batches run ONE wavefront per SIMD.  That wavefront pays an issue slot for every instruction, scalar ones
// included, and waits out every scalar load itself; with the ~50 uniform doubles of a rollout stage (model, weights,
// polynomial coefficients) competing for ~100 SGPRs the compiler re-materialises literals with s_mov pairs and
// re-loads kernel arguments inside the stage loop (60 of the 274 instructions of a rollout stage, ten scalar-load
// waits).  pin_consts() moves the uniform doubles into VGPRs, which those kernels have to spare, behind an empty asm
// the optimiser cannot see through: loaded once per kernel, same values, same arithmetic.
__device__ __forceinline__ void vpin(double& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void vpin(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin_trig(TrigK<double>& K) {
    vpin(K.t2p); vpin(K.p1); vpin(K.p2); vpin(K.p3);
    vpin(K.s5); vpin(K.s4); vpin(K.s3); vpin(K.s2); vpin(K.s1); vpin(K.s0);
    vpin(K.c5); vpin(K.c4); vpin(K.c3); vpin(K.c2); vpin(K.c1); vpin(K.c0);
}
__device__ __forceinline__ void pin_trig(TrigK<float>&) {}
// the two halves of pin_consts, for kernels whose wavefronts play different roles (k_forward_duo): a role pins what IT uses
__device__ __forceinline__ void pin_model(KConst& k) {
    vpin(k.cd0); vpin(k.cda); vpin(k.cla); vpin(k.m); vpin(k.g); vpin(k.S); vpin(k.rho); vpin(k.J); vpin(k.dt);
    vpin(k.dtm); vpin(k.mg); vpin(k.hrho); vpin(k.krs); vpin(k.b41); vpin(k.rJ);
}
template <bool DIAG>
__device__ __forceinline__ void pin_weights(KConst& k) {
#pragma unroll
    for (int i = 0; i < 36; i++)
        if (!DIAG || i % 7 == 0) { vpin(k.Q[i]); vpin(k.QT[i]); }
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (!DIAG || i % 3 == 0) vpin(k.R[i]);
}
template <bool DIAG>
__device__ __forceinline__ void pin_consts(KConst& k) {
    pin_model(k);
    pin_weights<DIAG>(k);
}

// one fused multiply-add in the arithmetic type (a bare __builtin_fma on floats is a DOUBLE fma between two conversions)
__device__ __forceinline__ double fma_r(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_r(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// a / b for a wave-uniform b with rb = RN(1/b): q0 = RN(a rb), r = a - q0 b (exact, one fma), RN(q0 + r rb).
// Markstein's correction: the result IS the correctly rounded quotient (b finite, not of all-ones
// significand; a, a/b in the normal range), i.e. bit-identical to the reference's `uu[1] / J`, in three
// instructions instead of the ~15 (one quarter-rate) of the IEEE division expansion.  a = +-inf gives NaN.
__device__ __forceinline__ real div_by_const(real a, real b, real rb) {
    const real q0 = a * rb;
    const real r = fma_r(-q0, b, a);
    return fma_r(r, rb, q0);
}

// 1/b to ~1 ulp for the linearisation and the 2x2 gain solve (results compared at 1e-8, not bit for bit):
// v_rcp_f64 and two Newton steps, without the range scaling / fix-up of the IEEE expansion.
__device__ __forceinline__ double rcp_fast(double b) {
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    return __builtin_fma(y, e, y);
}
// float32 arithmetic is not a parity path (config 3: tolerance sweep): v_rcp_f32 (1 ulp) as it is, instead of the ten
// instructions of the IEEE division expansion
__device__ __forceinline__ float rcp_fast(float b) { return __builtin_amdgcn_rcpf(b); }

// a / b where the reference divides (aircraft_simplified.py:310, dt / (m V)): the parity path divides exactly, the
// float32 build multiplies by the reciprocal
__device__ __forceinline__ double div_r(double a, double b) { return a / b; }
__device__ __forceinline__ float div_r(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }

#pragma clang fp contract(off)
__device__ __forceinline__ void step_state(const KConst& k, const real x[6], real u0, real u1,
                                           const SC& s, real xp[6]) {
    const real V = x[2], al = x[3] - x[5];
    const real V2 = V * V;
    const real D = k.hrho * V2 * k.S * (k.cd0 + k.cda * (al * al));   // aircraft_simplified.py:228
    const real L = k.hrho * V2 * k.S * k.cla * al;                    // :253
    xp[0] = (real)(float)(x[0] + k.dt * V * s.cg);                    // :303
    xp[1] = (real)(float)(x[1] - k.dt * V * s.sg);                    // :304
    xp[2] = (real)(float)(V + k.dtm * (-D - k.mg * s.sg + u0 * s.ca)); // :306
    xp[3] = (real)(float)(x[3] + k.dt * x[4]);                        // :307
    xp[4] = (real)(float)(x[4] + k.dt * div_by_const(u1, k.J, k.rJ)); // :309  u1 / J
    xp[5] = (real)(float)(x[5] + div_r(k.dt, k.m * V) * (L - k.mg * s.cg + u0 * s.sa)); // :310
}

// Stage and terminal cost.  The reference evaluates l = (0.5 dx)^T (Q dx) + (0.5 du)^T (R du)
// (aircraft_simplified.py:61) and sums l over the horizon (optcon.py:417-424).  Multiplying by 0.5 is
// exact and commutes with every rounding of the products and sums involved (no under/overflow at
// these magnitudes), so the functions below return 2l, the callers accumulate 2J in the same order
// and halve ONCE at the end: bit-identical to the reference's J, eight multiplications fewer per stage.
// q = Q dx, r = R du are lx, lu (:63-64).
template <bool DIAG>
__device__ __forceinline__ real stage_cost2(const KConst& k, const real x[6], real u0, real u1,
                                            const real* __restrict__ ref, real q[6], real r[2]) {
    real dx[6], du[2];
#pragma unroll
    for (int i = 0; i < 6; i++) dx[i] = x[i] - ref[i];
    du[0] = u0 - ref[6];
    du[1] = u1 - ref[7];
    if (DIAG) {
#pragma unroll
        for (int i = 0; i < 6; i++) q[i] = k.Q[i * 6 + i] * dx[i];
        r[0] = k.R[0] * du[0];
        r[1] = k.R[3] * du[1];
    } else {
#pragma unroll
        for (int i = 0; i < 6; i++) {
            real a = R(0.0);
#pragma unroll
            for (int j = 0; j < 6; j++) a += k.Q[i * 6 + j] * dx[j];
            q[i] = a;
        }
        r[0] = (R(0.0) + k.R[0] * du[0]) + k.R[1] * du[1];
        r[1] = (R(0.0) + k.R[2] * du[0]) + k.R[3] * du[1];
    }
    real a = R(0.0), b = R(0.0);
#pragma unroll
    for (int i = 0; i < 6; i++) a += dx[i] * q[i];
    b += du[0] * r[0];
    b += du[1] * r[1];
    return a + b;
}

// 2 l_T(x) = (dx^T Q_T) dx, q_f = Q_T dx   (aircraft_simplified.py:92-94)
template <bool DIAG>
__device__ __forceinline__ real term_cost2(const KConst& k, const real x[6],
                                           const real* __restrict__ ref, real qf[6]) {
    real dx[6], v[6];
#pragma unroll
    for (int i = 0; i < 6; i++) dx[i] = x[i] - ref[i];
    if (DIAG) {
#pragma unroll
        for (int i = 0; i < 6; i++) { v[i] = dx[i] * k.QT[i * 6 + i]; qf[i] = k.QT[i * 6 + i] * dx[i]; }
    } else {
#pragma unroll
        for (int j = 0; j < 6; j++) {
            real a = R(0.0), c = R(0.0);
#pragma unroll
            for (int i = 0; i < 6; i++) { a += dx[i] * k.QT[i * 6 + j]; c += k.QT[j * 6 + i] * dx[i]; }
            v[j] = a; qf[j] = c;
        }
    }
    real ll = R(0.0);
#pragma unroll
    for (int j = 0; j < 6; j++) ll += v[j] * dx[j];
    return ll;
}
#pragma clang fp contract(on)

// ---------------------------------------------------------------------------------------------
// Linearisation  A = df/dx, B = df/du  at (x,u)   (aircraft_simplified.py:316-325)
// Non-constant entries only; A00=A11=A33=A44=1, A34=dt, B41=dt/J.
// ---------------------------------------------------------------------------------------------
struct Lin {
    real a02, a05, a12, a15, a22, a23, a25, a52, a53, a55;
    real b20, b50;
};

// (V, al = theta - gamma given: what the five-wavefront backward kernel hands from one producer to the other)
__device__ __forceinline__ Lin linearise_va(const KConst& k, real V, real al, real u0, const SC& s) {
    Lin l;
    const real V2 = V * V;
    const real iV = rcp_fast(V);
    const real dtmV = k.dtm * iV;                     // dt/(m V)
    l.a02 = k.dt * s.cg;
    l.a05 = -k.dt * V * s.sg;
    l.a12 = -k.dt * s.sg;
    l.a15 = -k.dt * V * s.cg;
    const real cdt = k.cd0 + k.cda * al * al;
    l.a22 = R(1.0) - k.dtm * (k.krs * V * cdt);
    const real dA = k.cda * k.krs * al * V2 + u0 * s.sa;     // (Cda S rho 2 al V^2)/2 + u0 sin(al)
    l.a23 = -k.dtm * dA;
    l.a25 = k.dtm * (dA - k.mg * s.cg);
    const real hl = R(0.5) * k.cla * k.krs * V2;                // Cla S rho V^2 / 2
    const real lA = hl * al + u0 * s.sa - k.mg * s.cg;
    l.a52 = k.dtm * (k.cla * k.krs * al) - dtmV * iV * lA;
    const real lB = hl + u0 * s.ca;
    l.a53 = dtmV * lB;
    l.a55 = R(1.0) - dtmV * (lB - k.mg * s.sg);
    l.b20 = k.dtm * s.ca;
    l.b50 = dtmV * s.sa;
    return l;
}
__device__ __forceinline__ Lin linearise(const KConst& k, const real x[6], real u0, const SC& s) {
    return linearise_va(k, x[2], x[3] - x[5], u0, s);
}

// Second-order terms contracted with the costate (aircraft_simplified.py:339-388): the symmetric
// 6x6  sum_k lam_k d2f_k/dx2  has six distinct non-zeros, the 2x6  sum_k lam_k d2f_k/dudx  three
// (row 0 only); d2f/du2 == 0.
struct Hess {
    real h22, h23, h25, h33, h35, h55;  // fxx
    real s02, s03, s05;                 // fux row 0
};

__device__ __forceinline__ Hess hessian(const KConst& k, const real x[6], real u0, const SC& s,
                                        const real lam[6]) {
    Hess h;
    const real V = x[2], al = x[3] - x[5], V2 = V * V;
    const real iV = rcp_fast(V), iV2 = iV * iV;
    const real dtmV = k.dtm * iV, dtmV2 = k.dtm * iV2;
    const real l0 = lam[0], l1 = lam[1], l2 = lam[2], l5 = lam[5];
    // k = 0, 1  (:339-352)
    const real f0_25 = -k.dt * s.sg, f0_55 = -k.dt * V * s.cg;
    const real f1_25 = -k.dt * s.cg, f1_55 = k.dt * V * s.sg;
    // k = 2  (:354-359)
    const real f2_22 = -k.dtm * (k.krs * (k.cd0 + k.cda * al * al));
    const real f2_23 = -k.dtm * (k.cda * k.krs * V * (R(2.0) * al));
    const real e2 = k.cda * k.krs * V2 + u0 * s.ca;
    const real f2_33 = -k.dtm * e2;
    const real f2_55 = -k.dtm * (e2 - k.mg * s.sg);
    // k = 5  (:361-366)
    const real hl = R(0.5) * k.cla * k.krs * V2;
    const real lA = hl * al + u0 * s.sa - k.mg * s.cg;
    const real cl = k.cla * k.krs * k.dtm;                    // Cla S dt rho / m
    const real f5_22 = R(2.0) * dtmV2 * iV * lA - cl * al * iV;
    const real lB = hl + u0 * s.ca;
    const real f5_23 = cl - dtmV2 * lB;
    const real f5_25 = dtmV2 * (lB - k.mg * s.sg) - cl;
    const real f5_33 = -dtmV * (u0 * s.sa);
    const real f5_55 = -dtmV * (u0 * s.sa - k.mg * s.cg);
    h.h22 = l2 * f2_22 + l5 * f5_22;
    h.h23 = l2 * f2_23 + l5 * f5_23;
    h.h25 = l0 * f0_25 + l1 * f1_25 - l2 * f2_23 + l5 * f5_25;
    h.h33 = l2 * f2_33 + l5 * f5_33;
    h.h35 = -h.h33;   // = -l2 * f2_33 - l5 * f5_33, which rounds to exactly that (aircraft_simplified.py:347)
    h.h55 = l0 * f0_55 + l1 * f1_55 + l2 * f2_55 + l5 * f5_55;
    // fux (:375-379)
    const real g2_03 = -k.dtm * s.sa;
    const real g5_02 = -dtmV2 * s.sa, g5_03 = dtmV * s.ca;
    h.s02 = l5 * g5_02;
    h.s03 = l2 * g2_03 + l5 * g5_03;
    h.s05 = -h.s03;   // = -l2 * g2_03 - l5 * g5_03 likewise (:379)
    return h;
}

// y = A^T v  (sparse)
__device__ __forceinline__ void At_vec(const KConst& k, const Lin& l, const real v[6], real y[6]) {
    y[0] = v[0];
    y[1] = v[1];
    y[2] = l.a02 * v[0] + l.a12 * v[1] + l.a22 * v[2] + l.a52 * v[5];
    y[3] = l.a23 * v[2] + v[3] + l.a53 * v[5];
    y[4] = k.dt * v[3] + v[4];
    y[5] = l.a05 * v[0] + l.a15 * v[1] + l.a25 * v[2] + l.a55 * v[5];
}

// y = A v  (sparse)
__device__ __forceinline__ void A_vec(const KConst& k, const Lin& l, const real v[6], real y[6]) {
    y[0] = v[0] + l.a02 * v[2] + l.a05 * v[5];
    y[1] = v[1] + l.a12 * v[2] + l.a15 * v[5];
    y[2] = l.a22 * v[2] + l.a23 * v[3] + l.a25 * v[5];
    y[3] = v[3] + k.dt * v[4];
    y[4] = v[4];
    y[5] = l.a52 * v[2] + l.a53 * v[3] + l.a55 * v[5];
}

// symmetric 6x6 stored as upper triangle, index of (i,j), i<=j
__host__ __device__ constexpr int sidx(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }
#define SYM(P, i, j) ((i) <= (j) ? P[sidx((i), (j))] : P[sidx((j), (i))])

// ---------------------------------------------------------------------------------------------
// One stage of the affine LTV-LQR backward recursion in 6-dim form.
//
// The reference augments the state with a constant 1 (optcon.py:655-690):
//   P~ = [[c, p^T],[p, P]],  A~ = blkdiag(1,A),  B~ = [0;B],  Q~ = [[0,q^T/2],[q/2,Q]],  S~ = [r/2, S]
// so that its 7x7 recursion (optcon.py:719-728) and gain formula (:732-751) read, block-wise,
//   G  = B^T P A + S                 (2x6)        h = B^T p + r/2        (2)
//   M  = R + B^T P B                 (2x2)
//   P_t = Q + A^T P A - G^T M^-1 G                p_t = q/2 + A^T p - G^T M^-1 h
//   K~_t = -Mreg^-1 [h, G]   with Mreg = M, or M + 0.5 I when M is not positive definite (:745-749;
//                            the Riccati update itself always uses the unregularised M — Q3)
// P is carried as a symmetric matrix (the reference's P is symmetric to rounding, |P-P^T|<=7e-11).
// Inputs: P,p at t+1; Q (sym upper, incl. Hessian terms), S row 0 entries, q/2, r/2.  Outputs: P,p
// at t (in place), the seven columns of K~_t through emit(c, K~[0][c], K~[1][c]) and flags.
// ---------------------------------------------------------------------------------------------
struct StageFlags { bool singular, regularised; };

// Columns COLS (bit j: column j of P_t and of the feedback gains K) of one stage, and with AFFINE the affine terms
// (sigma, and p_t in place).  P is read only; Pn receives the entries (i,j), i <= j, of the columns in COLS; every
// column of K~ this part owns is handed to emit(c, K~[0][c], K~[1][c]) — c = 0 the feed-forward sigma, c = 1 + j the
// feedback column j — the moment it exists, so that a caller which stores it at once never holds the gains of a stage
// in registers (14 doubles, which the full-Hessian backward kernel does not have to spare).
// lqr_stage = all six columns + affine terms.  The four-wavefront backward pass of small batches
// (k_backward4) gives three wavefronts the columns {0,1,2}, {3,4} and {5} + affine: every entry is computed by the
// expressions below whichever wavefront owns it, so the split changes no rounding.  A column j needs the rows
// G[:,i] = (B^T P A + S)[:,i] of all i <= j (three entries of column i of P A each), the affine terms need all six.
template <int COLS, bool AFFINE, typename Emit>
__device__ __forceinline__ StageFlags lqr_stage_part(const KConst& k, const Lin& l, const real P[21], real p[6],
                                                     const real Qs[21], real s02, real s03, real s05,
                                                     const real hq[6], const real hr[2], real Pn[21], Emit emit) {
    constexpr int JMAX = AFFINE ? 5 : (COLS >= 32 ? 5 : COLS >= 16 ? 4 : COLS >= 8 ? 3 : COLS >= 4 ? 2 : COLS >= 2 ? 1 : 0);
    // M = R + B^T P B  (2x2, symmetric), from the six entries of P that B touches
    const real P22 = P[sidx(2, 2)], P24 = P[sidx(2, 4)], P25 = P[sidx(2, 5)], P44 = P[sidx(4, 4)],
                 P45 = P[sidx(4, 5)], P55 = P[sidx(5, 5)];
    const real bp2 = l.b20 * P22 + l.b50 * P25;   // (B^T P)[0,2]
    const real bp5 = l.b20 * P25 + l.b50 * P55;   // (B^T P)[0,5]
    const real bp4 = l.b20 * P24 + l.b50 * P45;   // (B^T P)[0,4]
    const real M00 = k.R[0] + (bp2 * l.b20 + bp5 * l.b50);
    const real M01 = k.R[1] + bp4 * k.b41;
    const real M11 = k.R[3] + (k.b41 * P44) * k.b41;
    StageFlags fl;
    const real det = M00 * M11 - M01 * M01;
    fl.singular = (det == R(0.0));
    const real idet = rcp_fast(det);
    const real i00 = M11 * idet, i01 = -M01 * idet, i11 = M00 * idet;
    // M positive definite <=> tr > 0 and det > 0 (equivalent to all(eigvals(M) > 0), optcon.py:745)
    const bool pd = (M00 + M11 > R(0.0)) && (det > R(0.0));
    fl.regularised = !pd;
    // The gains are K~ = -Mreg^-1 [h, G] with Mreg = M, or M + 0.5 I where M is not positive definite (optcon.py:745-751),
    // while the Riccati update below always uses the unregularised M (Q3).  n = -Mreg^-1 is settled HERE, before the
    // column loop (a short, rarely taken branch while little is alive), so that the loop is one straight run of code that
    // can hand out each column of K~ as it goes: with the regularised gains patched in a branch AFTER the loop the
    // compiler kept M^-1 G, its negated copy and G alive side by side across that branch.
    real n00 = -i00, n01 = -i01, n11 = -i11;
    if (!pd) {   // (a branch-free form — always evaluated, selected — was measured: no difference, 1.72 vs 1.72 ms)
        const real r00 = M00 + R(0.5), r11 = M11 + R(0.5);
        const real rdet = r00 * r11 - M01 * M01;
        if (rdet == R(0.0)) fl.singular = true;
        const real ird = rcp_fast(rdet);
        const real j00 = r11 * ird, j01 = -M01 * ird, j11 = r00 * ird;
        n00 = -j00; n01 = -j01; n11 = -j11;
    }
    // affine column: h = B^T p + r/2, M^-1 h; and the part of p_t = q/2 + A^T p - G^T (M^-1 h) that does not wait for
    // G: tq = q/2 + A^T p, after which neither p nor q/2 is needed (the subtraction at the end completes the same
    // expression: same roundings)
    real mh0 = R(0.0), mh1 = R(0.0), tq[6];
    if (AFFINE) {
        const real h0 = l.b20 * p[2] + l.b50 * p[5] + hr[0];
        const real h1 = k.b41 * p[4] + hr[1];
        mh0 = i00 * h0 + i01 * h1; mh1 = i01 * h0 + i11 * h1;
        emit(0, n00 * h0 + n01 * h1, n01 * h0 + n11 * h1);
        real ap[6];
        At_vec(k, l, p, ap);
#pragma unroll
        for (int i = 0; i < 6; i++) tq[i] = hq[i] + ap[i];
    }
    // Column by column: W[:,j] = (P A)[:,j], G[:,j] = (B^T W)[:,j] + S[:,j], M^-1 G[:,j],
    // z = (A^T W)[:,j], P_t[i,j] = Q[i,j] + z[i] - G[:,i]^T M^-1 G[:,j] for i <= j.  Only one column of W
    // is alive at a time; the new P is built beside the old one.
    real G0[6], G1[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const bool own = (COLS >> j) & 1;
        if (!own && j > JMAX) continue;
        real w[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            if (!own && i != 2 && i != 4 && i != 5) continue;   // G needs rows 2, 4, 5 of the column only
            const real pi0 = SYM(P, i, 0), pi1 = SYM(P, i, 1), pi2 = SYM(P, i, 2), pi3 = SYM(P, i, 3),
                         pi4 = SYM(P, i, 4), pi5 = SYM(P, i, 5);
            w[i] = j == 0 ? pi0
                 : j == 1 ? pi1
                 : j == 2 ? pi0 * l.a02 + pi1 * l.a12 + pi2 * l.a22 + pi5 * l.a52
                 : j == 3 ? pi2 * l.a23 + pi3 + pi5 * l.a53
                 : j == 4 ? pi3 * k.dt + pi4
                          : pi0 * l.a05 + pi1 * l.a15 + pi2 * l.a25 + pi5 * l.a55;
        }
        real g0 = l.b20 * w[2] + l.b50 * w[5];
        const real g1 = k.b41 * w[4];
        if (j == 2) g0 += s02;
        if (j == 3) g0 += s03;
        if (j == 5) g0 += s05;
        G0[j] = g0; G1[j] = g1;
        if (!own) continue;
        emit(1 + j, n00 * g0 + n01 * g1, n01 * g0 + n11 * g1);
        const real mg0 = i00 * g0 + i01 * g1, mg1 = i01 * g0 + i11 * g1;
        const real z[6] = {w[0], w[1],
                             l.a02 * w[0] + l.a12 * w[1] + l.a22 * w[2] + l.a52 * w[5],
                             l.a23 * w[2] + w[3] + l.a53 * w[5],
                             k.dt * w[3] + w[4],
                             l.a05 * w[0] + l.a15 * w[1] + l.a25 * w[2] + l.a55 * w[5]};
#pragma unroll
        for (int i = 0; i <= j; i++) Pn[sidx(i, j)] = Qs[sidx(i, j)] + z[i] - (G0[i] * mg0 + G1[i] * mg1);
    }
    if (AFFINE) {  // p_t = q/2 + A^T p - G^T (M^-1 h)
#pragma unroll
        for (int i = 0; i < 6; i++) p[i] = tq[i] - (G0[i] * mh0 + G1[i] * mh1);
    }
    return fl;
}

template <typename Emit>
__device__ __forceinline__ StageFlags lqr_stage(const KConst& k, const Lin& l, real P[21], real p[6],
                                                const real Qs[21], real s02, real s03, real s05,
                                                const real hq[6], const real hr[2], Emit emit) {
    real Pn[21];
    const StageFlags fl = lqr_stage_part<63, true>(k, l, P, p, Qs, s02, s03, s05, hq, hr, Pn, emit);
#pragma unroll
    for (int e = 0; e < 21; e++) P[e] = Pn[e];
    return fl;
}

#undef SYM

// ---------------------------------------------------------------------------------------------
// Disturbance model of the receding-horizon plant step, drawn on the device (the reference's closed loop,
// lqr_tracking.py:279-281, has none: this is the build's own, BASELINE configs[4] "small seeded disturbance").
// Counter-based: Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11) with key = seed and
// counter = (global instance index, closed-loop step, pair index j, 0) — no state, so the draw of (instance, step) does
// not depend on the shard an instance lives in nor on what was drawn before.  One counter gives four 32-bit words = two
// 53-bit uniforms u1, u2 in (0, 1] = two standard normals by Box-Muller, sqrt(-2 ln u1) (cos, sin)(2 pi u2); three
// counters per instance and step give the six components, component c scaled by sigma[c].
// Always evaluated in fp64 (once per instance and step: no throughput matters here).
// ---------------------------------------------------------------------------------------------
struct MpcNoise {
    unsigned key0, key1;     // seed (low, high word)
    unsigned step, first;    // closed-loop step; global index of instance 0 of this batch
    double sigma[6];
    int on;
};

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned out[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0, hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// the six disturbance components of global instance `inst` at the step of `nz`
__device__ __forceinline__ void mpc_noise_draw(const MpcNoise& nz, unsigned inst, double d[6]) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
        unsigned w[4];
        philox4x32_10(inst, nz.step, (unsigned)j, 0u, nz.key0, nz.key1, w);
        // 53-bit uniforms in (0, 1]: (27 high bits of one word, 26 of the next) + 1, times 2^-53
        const double u1 = ((double)(w[0] >> 5) * 67108864.0 + (double)(w[1] >> 6) + 1.0) * 0x1.0p-53;
        const double u2 = ((double)(w[2] >> 5) * 67108864.0 + (double)(w[3] >> 6) + 1.0) * 0x1.0p-53;
        const double rad = sqrt(-2.0 * log(u1));
        double sn, cs;
        sincos(6.283185307179586476925 * u2, &sn, &cs);
        d[2 * j] = nz.sigma[2 * j] * (rad * cs);
        d[2 * j + 1] = nz.sigma[2 * j + 1] * (rad * sn);
    }
}
}  // namespace AOC_ARITH_NS
