/*
 * aoc_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, fp64, one-trajectory-at-a-time restatement of the reference's Newton/LQR hot path
 * (MohamedAtwan/AirCraftOptimalControl: aircraft_simplified.py, optcon.py, lqr_tracking.py).
 * It follows the reference statement by statement (augmented 7x7 LQR, explicit inverse, separate
 * Riccati / gain / rollout loops, full-history indexing quirks) so that it can be checked against
 * golden vectors produced by running the reference itself (tests/golden/make_golden.py), and then
 * serve as the checker for the HIP path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (aircraftoptimalcontrol_amd/) never links, imports or calls it.
 *
 * Parity status: PINNED by the .npz fixtures in tests/golden/ (generated from the reference in the build container).
 *
 * Array conventions here are the reference's single-trajectory ones: xx is (6,T) C-order
 * (xx[c*T+t]), uu is (2,T); small matrices are row-major.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NS 6
#define NI 2

/* Scratch arrays of one call come from a per-thread arena (chunks are kept and reused): with
 * calloc/free per call the batched driver spends its time in the allocator and in page faults once
 * many threads run, which would understate the CPU baseline. */
#define ARENA_CHUNKS 64
typedef struct { char *base[ARENA_CHUNKS]; size_t cap[ARENA_CHUNKS]; int n, cur; size_t top; } orc_arena;
static _Thread_local orc_arena g_arena;
typedef struct { int cur; size_t top; } orc_mark;

static orc_mark arena_mark(void) { orc_mark m = {g_arena.cur, g_arena.top}; return m; }
static void arena_release(orc_mark m) { g_arena.cur = m.cur; g_arena.top = m.top; }
static void *arena_calloc(size_t n, size_t sz) {
    orc_arena *a = &g_arena;
    size_t bytes = (n * sz + 63) & ~(size_t)63;
    for (;;) {
        if (a->cur < a->n && a->top + bytes <= a->cap[a->cur]) {
            void *p = a->base[a->cur] + a->top;
            a->top += bytes;
            memset(p, 0, bytes);
            return p;
        }
        if (a->cur + 1 < a->n) { a->cur++; a->top = 0; continue; }
        if (a->n == ARENA_CHUNKS) return calloc(n, sz); /* never in practice; leaks by design */
        size_t cap = bytes > ((size_t)4 << 20) ? bytes : ((size_t)4 << 20);
        a->base[a->n] = (char *)malloc(cap);
        a->cap[a->n] = cap;
        a->cur = a->n++;
        a->top = 0;
    }
}
#define NA 7 /* augmented state dimension ns+1 (optcon.py:657-666) */

typedef struct {
    double cd0, cda, cla, m, g, S, rho, J, dt; /* aircraft_simplified.py:108-118 */
} orc_model;

typedef struct {
    orc_model mdl;
    double QQt[36], RRt[4], QQT[36]; /* Cost(QQt,RRt,QQT)  aircraft_simplified.py:20-23 */
    int32_t T;                       /* TT = int(tf/dt)    optcon.py:378 */
    int32_t pad_;
    const double *xx_ref;            /* (6,T) */
    const double *uu_ref;            /* (2,T) */
} orc_problem;

typedef struct {
    int32_t max_iters;       /* optcon.py:367 */
    int32_t armijo_maxiters; /* optcon.py:230 */
    double stepsize_0, cc, beta; /* optcon.py:224-229 */
    double term_cond;        /* hard-coded -1e-6 in the reference, optcon.py:368 */
    int32_t hessian_switch;  /* full Hessian when kk > hessian_switch (8), optcon.py:443 */
    int32_t pad_;
} orc_params;

void orc_default_model(orc_model *m) {
    m->cd0 = 0.1716; m->cda = 2.395; m->cla = 3.256; m->m = 12; m->g = 9.81;
    m->S = 0.61; m->rho = 1.2; m->J = 0.24; m->dt = 1e-3;
}

/* ------------------------------------------------------------------------------------------
 * Dynamics.step  (aircraft_simplified.py:263-393), with dragForce (:212-236), liftForce
 * (:238-261) and tensorCont (:397-404).
 *   xp  : next state, computed in fp64 and ROUNDED TO FLOAT32 (the container is float32, :300)
 *   fx  : (6,6) as the reference returns it, i.e. A^T   (:316-322)
 *   fu  : (2,6) = B^T                                   (:324-325)
 *   fxx (6,6), fuu (2,2), fux (2,6): second-order tensors contracted with lmbd (:384-388).
 *   If lmbd == NULL the contracted outputs are skipped; fxx3/fux3 (uncontracted (6,6,6)/(2,6,6),
 *   index [i][j][k] = d2 f_k / dx_i dx_j) are filled when non-NULL.
 * ------------------------------------------------------------------------------------------ */
void orc_step(const orc_model *md, const double *xx, const double *uu, const double *lmbd,
              double *xp, double *fx, double *fu, double *fxx, double *fuu, double *fux,
              double *fxx3, double *fux3) {
    const double m = md->m, J = md->J, rho = md->rho, Cla = md->cla, S = md->S, Cd0 = md->cd0,
                 Cda = md->cda, g = md->g, dt = md->dt;
    const double x2 = xx[2], x3 = xx[3], x4 = xx[4], x5 = xx[5];
    const double u0 = uu[0], u1 = uu[1];
    const double alpha = x3 - x5;                                          /* :295 */
    const double V2 = pow(x2, 2.0);                                        /* xx[2,0]**2 */
    const double D = 0.5 * rho * V2 * S * (Cd0 + Cda * pow(alpha, 2.0));   /* :228 */
    const double L = 0.5 * rho * V2 * S * Cla * alpha;                     /* :253 */
    const double s5 = sin(x5), c5 = cos(x5), sa = sin(alpha), ca = cos(alpha);

    if (xp) {                                                              /* :303-310 */
        xp[0] = (double)(float)(xx[0] + dt * x2 * c5);
        xp[1] = (double)(float)(xx[1] - dt * x2 * s5);
        xp[2] = (double)(float)(x2 + (dt / m) * (-D - m * g * s5 + u0 * ca));
        xp[3] = (double)(float)(x3 + dt * x4);
        xp[4] = (double)(float)(x4 + dt * (u1 / J));
        xp[5] = (double)(float)(x5 + (dt / (m * x2)) * (L - m * g * c5 + u0 * sa));
    }
    if (!fx) return;

    /* A = df/dx as written row by row at :316-321; the reference returns fx = A^T (:322) */
    double A[36];
    memset(A, 0, sizeof A);
    A[0 * 6 + 0] = 1; A[0 * 6 + 2] = dt * c5;  A[0 * 6 + 5] = -dt * x2 * s5;
    A[1 * 6 + 1] = 1; A[1 * 6 + 2] = -dt * s5; A[1 * 6 + 5] = -dt * x2 * c5;
    A[2 * 6 + 2] = 1 - (S * dt * rho * x2 * (Cd0 + Cda * pow(x3 - x5, 2.0))) / m;
    A[2 * 6 + 3] = -(dt * ((Cda * S * rho * (2 * x3 - 2 * x5) * V2) / 2 + u0 * sa)) / m;
    A[2 * 6 + 5] = (dt * ((Cda * S * rho * (2 * x3 - 2 * x5) * V2) / 2 + u0 * sa - g * m * c5)) / m;
    A[3 * 6 + 3] = 1; A[3 * 6 + 4] = dt;
    A[4 * 6 + 4] = 1;
    A[5 * 6 + 2] = (Cla * S * dt * rho * (x3 - x5)) / m
                   - (dt * ((Cla * S * rho * (x3 - x5) * V2) / 2 + u0 * sa - g * m * c5)) / (m * V2);
    A[5 * 6 + 3] = (dt * ((Cla * S * rho * V2) / 2 + u0 * ca)) / (m * x2);
    A[5 * 6 + 5] = 1 - (dt * ((Cla * S * rho * V2) / 2 + u0 * ca - g * m * s5)) / (m * x2);
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) fx[i * 6 + j] = A[j * 6 + i];

    /* B = df/du (6,2) at :324; fu = B^T (2,6) */
    memset(fu, 0, 12 * sizeof(double));
    fu[0 * 6 + 2] = (dt * ca) / m;
    fu[1 * 6 + 4] = dt / J;
    fu[0 * 6 + 5] = (dt * sa) / (m * x2);

    if (!lmbd && !fxx3) return;

    /* second-order tensors: fxx[:,:,k] for k in {0,1,2,5} (:339-371), fux[:,:,k] for k in {2,5}
       (:375-379), fuu == 0 (:382) */
    double F0[36], F1[36], F2[36], F5[36], G2[12], G5[12];
    memset(F0, 0, sizeof F0); memset(F1, 0, sizeof F1); memset(F2, 0, sizeof F2);
    memset(F5, 0, sizeof F5); memset(G2, 0, sizeof G2); memset(G5, 0, sizeof G5);
    F0[2 * 6 + 5] = -dt * s5; F0[5 * 6 + 2] = -dt * s5; F0[5 * 6 + 5] = -dt * x2 * c5;
    F1[2 * 6 + 5] = -dt * c5; F1[5 * 6 + 2] = -dt * c5; F1[5 * 6 + 5] = dt * x2 * s5;

    F2[2 * 6 + 2] = -(S * dt * rho * (Cd0 + Cda * pow(x3 - x5, 2.0))) / m;
    F2[2 * 6 + 3] = -(Cda * S * dt * rho * x2 * (2 * x3 - 2 * x5)) / m;
    F2[2 * 6 + 5] = (Cda * S * dt * rho * x2 * (2 * x3 - 2 * x5)) / m;
    F2[3 * 6 + 2] = F2[2 * 6 + 3];
    F2[3 * 6 + 3] = -(dt * (Cda * S * rho * V2 + u0 * ca)) / m;
    F2[3 * 6 + 5] = (dt * (Cda * S * rho * V2 + u0 * ca)) / m;
    F2[5 * 6 + 2] = F2[2 * 6 + 5];
    F2[5 * 6 + 3] = F2[3 * 6 + 5];
    F2[5 * 6 + 5] = -(dt * (Cda * S * rho * V2 + u0 * ca - g * m * s5)) / m;

    const double V3 = pow(x2, 3.0);
    F5[2 * 6 + 2] = (2 * dt * ((Cla * S * rho * (x3 - x5) * V2) / 2 + u0 * sa - g * m * c5)) / (m * V3)
                    - (Cla * S * dt * rho * (x3 - x5)) / (m * x2);
    F5[2 * 6 + 3] = (Cla * S * dt * rho) / m - (dt * ((Cla * S * rho * V2) / 2 + u0 * ca)) / (m * V2);
    F5[2 * 6 + 5] = (dt * ((Cla * S * rho * V2) / 2 + u0 * ca - g * m * s5)) / (m * V2) - (Cla * S * dt * rho) / m;
    F5[3 * 6 + 2] = F5[2 * 6 + 3];
    F5[3 * 6 + 3] = -(dt * u0 * sa) / (m * x2);
    F5[3 * 6 + 5] = (dt * u0 * sa) / (m * x2);
    F5[5 * 6 + 2] = F5[2 * 6 + 5];
    F5[5 * 6 + 3] = F5[3 * 6 + 5];
    F5[5 * 6 + 5] = -(dt * (u0 * sa - g * m * c5)) / (m * x2);

    G2[0 * 6 + 3] = -(dt * sa) / m;  G2[0 * 6 + 5] = (dt * sa) / m;
    G5[0 * 6 + 2] = -(dt * sa) / (m * V2);
    G5[0 * 6 + 3] = (dt * ca) / (m * x2);
    G5[0 * 6 + 5] = -(dt * ca) / (m * x2);

    if (fxx3) {
        memset(fxx3, 0, 216 * sizeof(double));
        for (int e = 0; e < 36; e++) {
            fxx3[e * 6 + 0] = F0[e]; fxx3[e * 6 + 1] = F1[e]; fxx3[e * 6 + 2] = F2[e]; fxx3[e * 6 + 5] = F5[e];
        }
    }
    if (fux3) {
        memset(fux3, 0, 72 * sizeof(double));
        for (int e = 0; e < 12; e++) { fux3[e * 6 + 2] = G2[e]; fux3[e * 6 + 5] = G5[e]; }
    }
    if (lmbd) {
        /* tensorCont: T = 0; for i ascending: T += P[:,:,i]*a[i]  (:397-404) */
        for (int e = 0; e < 36; e++) {
            double t = 0.0;
            t += F0[e] * lmbd[0]; t += F1[e] * lmbd[1]; t += F2[e] * lmbd[2];
            t += 0.0 * lmbd[3];   t += 0.0 * lmbd[4];   t += F5[e] * lmbd[5];
            fxx[e] = t;
        }
        for (int e = 0; e < 12; e++) {
            double t = 0.0;
            t += 0.0 * lmbd[0]; t += 0.0 * lmbd[1]; t += G2[e] * lmbd[2];
            t += 0.0 * lmbd[3]; t += 0.0 * lmbd[4]; t += G5[e] * lmbd[5];
            fux[e] = t;
        }
        for (int e = 0; e < 4; e++) fuu[e] = 0.0;
    }
}

/* Cost.stagecost (aircraft_simplified.py:25-69): ll, lx=Q(x-xr), lu=R(u-ur). lxx=Q, luu=R, lxu=lux=0. */
double orc_stagecost(const orc_problem *p, const double *x, const double *u, const double *xr,
                     const double *ur, double *lx, double *lu) {
    double dx[6], du[2], qx[6], ru[2];
    for (int i = 0; i < 6; i++) dx[i] = x[i] - xr[i];
    for (int i = 0; i < 2; i++) du[i] = u[i] - ur[i];
    for (int i = 0; i < 6; i++) {
        double s = 0.0;
        for (int k = 0; k < 6; k++) s += p->QQt[i * 6 + k] * dx[k];
        qx[i] = s;
    }
    for (int i = 0; i < 2; i++) {
        double s = 0.0;
        for (int k = 0; k < 2; k++) s += p->RRt[i * 2 + k] * du[k];
        ru[i] = s;
    }
    /* ll = (0.5*dx^T)@(Q@dx) + (0.5*du^T)@(R@du)   (:61; '*' and '@' associate left to right) */
    double a = 0.0, b = 0.0;
    for (int i = 0; i < 6; i++) a += (0.5 * dx[i]) * qx[i];
    for (int i = 0; i < 2; i++) b += (0.5 * du[i]) * ru[i];
    if (lx) for (int i = 0; i < 6; i++) lx[i] = qx[i];
    if (lu) for (int i = 0; i < 2; i++) lu[i] = ru[i];
    return a + b;
}

/* Cost.termcost (aircraft_simplified.py:71-97): llT = ((0.5*dx^T)@QQT)@dx, lTx = QQT@dx */
double orc_termcost(const orc_problem *p, const double *x, const double *xr, double *lTx) {
    double dx[6], v[6];
    for (int i = 0; i < 6; i++) dx[i] = x[i] - xr[i];
    for (int j = 0; j < 6; j++) {
        double s = 0.0;
        for (int k = 0; k < 6; k++) s += (0.5 * dx[k]) * p->QQT[k * 6 + j];
        v[j] = s;
    }
    double ll = 0.0;
    for (int j = 0; j < 6; j++) ll += v[j] * dx[j];
    if (lTx)
        for (int i = 0; i < 6; i++) {
            double s = 0.0;
            for (int k = 0; k < 6; k++) s += p->QQT[i * 6 + k] * dx[k];
            lTx[i] = s;
        }
    return ll;
}

/* ------------------------------------------------------------------------------------------
 * small dense helpers (row-major, ascending-k accumulation, no contraction)
 * ------------------------------------------------------------------------------------------ */
static void mm(int n, int k, int m, const double *A, const double *B, double *C) { /* C(n,m)=A(n,k)B(k,m) */
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) {
            double s = 0.0;
            for (int l = 0; l < k; l++) s += A[i * k + l] * B[l * m + j];
            C[i * m + j] = s;
        }
}
static void tr(int n, int m, const double *A, double *At) { /* At(m,n) = A(n,m)^T */
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) At[j * n + i] = A[i * m + j];
}

/* np.linalg.inv for the 2x2 M (LAPACK dgesv on the identity: LU with partial pivoting, then two
   triangular solves).  Returns 1 if exactly singular (NumPy would raise LinAlgError). */
static int inv2(const double *M, double *Mi) {
    double a[4] = {M[0], M[1], M[2], M[3]};
    int piv = fabs(a[2]) > fabs(a[0]);
    if (piv) { double t0 = a[0], t1 = a[1]; a[0] = a[2]; a[1] = a[3]; a[2] = t0; a[3] = t1; }
    if (a[0] == 0.0) return 1;
    double l = a[2] * (1.0 / a[0]);
    double u22 = a[3] - l * a[1];
    if (u22 == 0.0) return 1;
    for (int c = 0; c < 2; c++) {
        double b0 = (c == 0) ? 1.0 : 0.0, b1 = (c == 1) ? 1.0 : 0.0;
        if (piv) { double t = b0; b0 = b1; b1 = t; }
        double y1 = b1 - l * b0;
        double x1 = y1 / u22;
        double x0 = (b0 - a[1] * x1) / a[0];
        Mi[0 * 2 + c] = x0; Mi[1 * 2 + c] = x1;
    }
    return 0;
}

/* `np.all(np.linalg.eigvals(MM) > 0)` for a real 2x2 (optcon.py:745).  Real spectrum: both roots
   of the characteristic polynomial positive.  Complex pair: NumPy orders complex numbers
   lexicographically, so the test reduces to Re > 0 (or Re == 0 and Im > 0, false for one of the two). */
static int all_eig_positive2(const double *M) {
    double tr2 = 0.5 * (M[0] + M[3]);
    double disc = 0.25 * (M[0] - M[3]) * (M[0] - M[3]) + M[1] * M[2];
    if (disc >= 0.0) {
        double s = sqrt(disc);
        /* stable small root: det / large root */
        double det = M[0] * M[3] - M[1] * M[2];
        double big = tr2 >= 0 ? tr2 + s : tr2 - s;
        double small = (big != 0.0) ? det / big : tr2 - s;
        return (big > 0.0) && (small > 0.0);
    }
    if (disc != disc) return 0; /* NaN */
    return tr2 > 0.0;
}

/* ------------------------------------------------------------------------------------------
 * ltv_LQR  (optcon.py:533-771; identical copy lqr_tracking.py:6-242).
 * Inputs are per-stage, time-major for the oracle's convenience: AA[t*36..], BB[t*12..] (6x2),
 * QQ[t*36..], RR[t*4..], SS[t*12..] (2x6); qq[t*6..], rr[t*2..] and qqf[6] are the affine terms
 * (all three NULL => non-augmented path, n = 6; otherwise augmented, n = 7).
 * Outputs: KK[t*(2*n)..], PP[t*n*n..] (may be NULL), xxout[t*6..], uuout[t*2..].
 * nreg (may be NULL) counts stages whose M was regularised (optcon.py:745-749).
 * Returns number of singular-M events (NumPy would raise).
 * ------------------------------------------------------------------------------------------ */
int orc_ltv_lqr(int T, const double *AAin, const double *BBin, const double *QQin, const double *RRin,
                const double *SSin, const double *QQfin, const double *x0, const double *qq,
                const double *rr, const double *qqf, double *KK, double *PPout, double *xxout,
                double *uuout, int *nreg) {
    const orc_mark mark_ = arena_mark();
    const int aug = (qq != NULL) || (rr != NULL) || (qqf != NULL); /* :614 */
    const int n = aug ? NA : NS;
    int nsing = 0, reg = 0;
    double *PP = (double *)arena_calloc((size_t)T * n * n, sizeof(double));
    double *AA = (double *)arena_calloc((size_t)T * n * n, sizeof(double));
    double *BB = (double *)arena_calloc((size_t)T * n * 2, sizeof(double));
    double *QQ = (double *)arena_calloc((size_t)T * n * n, sizeof(double));
    double *SS = (double *)arena_calloc((size_t)T * 2 * n, sizeof(double));
    double *xx = (double *)arena_calloc((size_t)T * n, sizeof(double));
    double QQf[NA * NA];
    memset(QQf, 0, sizeof QQf);
    memset(KK, 0, (size_t)T * 2 * n * sizeof(double));
    memset(uuout, 0, (size_t)T * 2 * sizeof(double));

    for (int t = 0; t < T; t++) {
        double *Q = QQ + (size_t)t * n * n, *A = AA + (size_t)t * n * n, *Bm = BB + (size_t)t * n * 2,
               *Sm = SS + (size_t)t * 2 * n;
        if (aug) { /* :669-686 */
            for (int i = 0; i < 6; i++) {
                double h = 0.5 * (qq ? qq[t * 6 + i] : 0.0);
                Q[(i + 1) * n + 0] = h; Q[0 * n + (i + 1)] = h;
                for (int j = 0; j < 6; j++) {
                    Q[(i + 1) * n + (j + 1)] = QQin[t * 36 + i * 6 + j];
                    A[(i + 1) * n + (j + 1)] = AAin[t * 36 + i * 6 + j];
                }
                for (int j = 0; j < 2; j++) Bm[(i + 1) * 2 + j] = BBin[t * 12 + i * 2 + j];
            }
            A[0] = 1.0;
            for (int i = 0; i < 2; i++) {
                Sm[i * n + 0] = 0.5 * (rr ? rr[t * 2 + i] : 0.0);
                for (int j = 0; j < 6; j++) Sm[i * n + (j + 1)] = SSin[t * 12 + i * 6 + j];
            }
        } else {
            memcpy(Q, QQin + t * 36, 36 * sizeof(double));
            memcpy(A, AAin + t * 36, 36 * sizeof(double));
            memcpy(Bm, BBin + t * 12, 12 * sizeof(double));
            memcpy(Sm, SSin + t * 12, 12 * sizeof(double));
        }
    }
    if (aug) { /* :688-697 */
        for (int i = 0; i < 6; i++) {
            double h = 0.5 * (qqf ? qqf[i] : 0.0);
            QQf[(i + 1) * n + 0] = h; QQf[0 * n + (i + 1)] = h;
            for (int j = 0; j < 6; j++) QQf[(i + 1) * n + (j + 1)] = QQfin[i * 6 + j];
        }
        for (int t = 0; t < T; t++) xx[t * n + 0] = 1.0;
        for (int i = 0; i < 6; i++) xx[0 * n + 1 + i] = x0[i];
    } else {
        memcpy(QQf, QQfin, 36 * sizeof(double));
        for (int i = 0; i < 6; i++) xx[i] = x0[i];
    }
    memcpy(PP + (size_t)(T - 1) * n * n, QQf, (size_t)n * n * sizeof(double)); /* :716 */

    double At[NA * NA], Bt[2 * NA], T1[NA * NA], APA[NA * NA], BtP[2 * NA], G[2 * NA], M[4], Mi[4], Gt[NA * 2],
        T2[NA * 2], T3[NA * NA];
    /* Riccati (:719-728): P_t = Q + A^T P A - (B^T P A + S)^T inv(R + B^T P B) (B^T P A + S) */
    for (int t = T - 2; t >= 0; t--) {
        const double *Q = QQ + (size_t)t * n * n, *A = AA + (size_t)t * n * n, *Bm = BB + (size_t)t * n * 2,
                     *Sm = SS + (size_t)t * 2 * n, *R = RRin + t * 4, *Pn = PP + (size_t)(t + 1) * n * n;
        double *P = PP + (size_t)t * n * n;
        tr(n, n, A, At); tr(n, 2, Bm, Bt);
        mm(n, n, n, At, Pn, T1); mm(n, n, n, T1, A, APA);       /* (A^T P) A */
        mm(2, n, n, Bt, Pn, BtP); mm(2, n, n, BtP, A, G);       /* (B^T P) A */
        for (int e = 0; e < 2 * n; e++) G[e] = G[e] + Sm[e];
        mm(2, n, 2, BtP, Bm, M);                                /* (B^T P) B */
        for (int e = 0; e < 4; e++) M[e] = R[e] + M[e];
        if (inv2(M, Mi)) nsing++;
        tr(2, n, G, Gt);
        mm(n, 2, 2, Gt, Mi, T2); mm(n, 2, n, T2, G, T3);        /* (G^T Minv) G */
        for (int e = 0; e < n * n; e++) P[e] = (Q[e] + APA[e]) - T3[e];
    }
    /* gains (:732-751) */
    for (int t = 0; t < T - 1; t++) {
        const double *A = AA + (size_t)t * n * n, *Bm = BB + (size_t)t * n * 2, *Sm = SS + (size_t)t * 2 * n,
                     *R = RRin + t * 4, *Pn = PP + (size_t)(t + 1) * n * n;
        tr(n, 2, Bm, Bt);
        mm(2, n, n, Bt, Pn, BtP);
        mm(2, n, 2, BtP, Bm, M);
        for (int e = 0; e < 4; e++) M[e] = R[e] + M[e];
        if (!all_eig_positive2(M)) { M[0] += 0.5; M[3] += 0.5; reg++; } /* MM += 0.5*eye(ni) */
        mm(2, n, n, BtP, A, G);
        for (int e = 0; e < 2 * n; e++) G[e] = G[e] + Sm[e];
        if (inv2(M, Mi)) nsing++;
        for (int e = 0; e < 4; e++) Mi[e] = -Mi[e];             /* (-inv(M)) @ G */
        mm(2, 2, n, Mi, G, KK + (size_t)t * 2 * n);
    }
    /* closed-loop linear rollout (:756-762) */
    for (int t = 0; t < T - 1; t++) {
        const double *A = AA + (size_t)t * n * n, *Bm = BB + (size_t)t * n * 2, *K = KK + (size_t)t * 2 * n;
        double u[2], ax[NA], bu[NA];
        mm(2, n, 1, K, xx + (size_t)t * n, u);
        uuout[t * 2 + 0] = u[0]; uuout[t * 2 + 1] = u[1];
        mm(n, n, 1, A, xx + (size_t)t * n, ax);
        mm(n, 2, 1, Bm, u, bu);
        for (int i = 0; i < n; i++) xx[(size_t)(t + 1) * n + i] = ax[i] + bu[i];
    }
    for (int t = 0; t < T; t++)
        for (int i = 0; i < 6; i++) xxout[t * 6 + i] = xx[(size_t)t * n + (aug ? 1 : 0) + i];
    if (PPout) memcpy(PPout, PP, (size_t)T * n * n * sizeof(double));
    if (nreg) *nreg = reg;
    arena_release(mark_);
    return nsing;
}

/* ------------------------------------------------------------------------------------------
 * get_update (optcon.py:176-200) and the rollout+cost of one Armijo trial (optcon.py:247-264).
 * xx_t (6,T), uu_t (2,T); uu_t[:,T-1] = 0.
 * ------------------------------------------------------------------------------------------ */
void orc_get_update(const orc_problem *p, double stepsize, const double *uu, const double *du,
                    const double *x0, double *xx_t, double *uu_t) {
    const int T = p->T;
    double x[6], u[2], xn[6];
    for (int c = 0; c < 6; c++) { x[c] = x0[c]; xx_t[c * T + 0] = x0[c]; }
    for (int c = 0; c < 2; c++) uu_t[c * T + (T - 1)] = 0.0;
    for (int t = 0; t < T - 1; t++) {
        for (int c = 0; c < 2; c++) { u[c] = uu[c * T + t] + stepsize * du[c * T + t]; uu_t[c * T + t] = u[c]; }
        orc_step(&p->mdl, x, u, NULL, xn, NULL, NULL, NULL, NULL, NULL, NULL, NULL);
        for (int c = 0; c < 6; c++) { x[c] = xn[c]; xx_t[c * T + t + 1] = xn[c]; }
    }
}

double orc_traj_cost(const orc_problem *p, const double *xx, const double *uu) { /* optcon.py:417-424 / :257-264 */
    const int T = p->T;
    double JJ = 0.0, x[6], u[2], xr[6], ur[2];
    for (int t = 0; t < T - 1; t++) {
        for (int c = 0; c < 6; c++) { x[c] = xx[c * T + t]; xr[c] = p->xx_ref[c * T + t]; }
        for (int c = 0; c < 2; c++) { u[c] = uu[c * T + t]; ur[c] = p->uu_ref[c * T + t]; }
        JJ += orc_stagecost(p, x, u, xr, ur, NULL, NULL);
    }
    for (int c = 0; c < 6; c++) { x[c] = xx[c * T + T - 1]; xr[c] = p->xx_ref[c * T + T - 1]; }
    JJ += orc_termcost(p, x, xr, NULL);
    return JJ;
}

/* armijo_stepsize (optcon.py:204-327).  Returns the step; *ntrials = rollouts evaluated.
   On exhaustion the returned step was never evaluated (Q5). */
double orc_armijo(const orc_problem *p, const orc_params *prm, const double *uu, const double *du,
                  const double *x0, double descent, double JP, int *ntrials, double *wx, double *wu) {
    double stepsize = prm->stepsize_0;
    int n = 0;
    for (int ii = 0; ii < prm->armijo_maxiters; ii++) {
        orc_get_update(p, stepsize, uu, du, x0, wx, wu);
        double JJt = orc_traj_cost(p, wx, wu);
        n++;
        if (JJt > JP + prm->cc * stepsize * descent) stepsize = prm->beta * stepsize;
        else break;
    }
    if (ntrials) *ntrials = n;
    return stepsize;
}

/* ------------------------------------------------------------------------------------------
 * One outer iteration kk of NewtonMethod.optimize (optcon.py:415-491), steps A-G of SURVEY 3.2.
 * in : xx (6,T), uu (2,T) current iterate; x0 (6) = xx_init[:,0] (optcon.py:398)
 * out: xx_new, uu_new; scalars JJ, descent, stepsize, ntrials, nreg;
 *      optional KK (T,2,7), du (2,T), dx (6,T), lmbd (6,T).
 * ------------------------------------------------------------------------------------------ */
int orc_newton_iterate(const orc_problem *p, const orc_params *prm, int kk, const double *xx,
                       const double *uu, const double *x0, double *xx_new, double *uu_new,
                       double *JJ_out, double *descent_out, double *stepsize_out, int *ntrials_out,
                       int *nreg_out, double *KK_out, double *du_out, double *dx_out, double *lmbd_out) {
    const orc_mark mark_ = arena_mark();
    const int T = p->T;
    double *AA = (double *)arena_calloc((size_t)T * 36, sizeof(double));
    double *BB = (double *)arena_calloc((size_t)T * 12, sizeof(double));
    double *QQ = (double *)arena_calloc((size_t)T * 36, sizeof(double));
    double *RR = (double *)arena_calloc((size_t)T * 4, sizeof(double));
    double *SS = (double *)arena_calloc((size_t)T * 12, sizeof(double));
    double *qq = (double *)arena_calloc((size_t)T * 6, sizeof(double));
    double *rr = (double *)arena_calloc((size_t)T * 2, sizeof(double));
    double *lm = (double *)arena_calloc((size_t)T * 6, sizeof(double));
    double *KK = (double *)arena_calloc((size_t)T * 14, sizeof(double));
    double *dxl = (double *)arena_calloc((size_t)T * 6, sizeof(double));
    double *dul = (double *)arena_calloc((size_t)T * 2, sizeof(double));
    double *du = (double *)arena_calloc((size_t)T * 2, sizeof(double));
    double *wx = (double *)arena_calloc((size_t)T * 6, sizeof(double));
    double *wu = (double *)arena_calloc((size_t)T * 2, sizeof(double));
    double x[6], u[2], xr[6], ur[2], a[6], b[2], fx[36], fu[12], fxx[36], fuu[4], fux[12];

    double JJ = orc_traj_cost(p, xx, uu);                                  /* A  :417-424 */

    for (int c = 0; c < 6; c++) { x[c] = xx[c * T + T - 1]; xr[c] = p->xx_ref[c * T + T - 1]; }
    orc_termcost(p, x, xr, a);                                             /* B  :429-432 */
    for (int c = 0; c < 6; c++) { lm[(T - 1) * 6 + c] = a[c]; qq[(T - 1) * 6 + c] = a[c]; }
    memcpy(QQ + (size_t)(T - 1) * 36, p->QQT, 36 * sizeof(double));

    const int full = kk > prm->hessian_switch;                             /* :443 */
    for (int t = T - 2; t >= 0; t--) {                                     /* C  :434-464 */
        for (int c = 0; c < 6; c++) { x[c] = xx[c * T + t]; xr[c] = p->xx_ref[c * T + t]; }
        for (int c = 0; c < 2; c++) { u[c] = uu[c * T + t]; ur[c] = p->uu_ref[c * T + t]; }
        orc_stagecost(p, x, u, xr, ur, a, b);
        orc_step(&p->mdl, x, u, lm + (size_t)(t + 1) * 6, NULL, fx, fu, fxx, fuu, fux, NULL, NULL);
        /* AA = fx.T, BB = fu.T */
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) AA[t * 36 + i * 6 + j] = fx[j * 6 + i];
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 2; j++) BB[t * 12 + i * 2 + j] = fu[j * 6 + i];
        for (int e = 0; e < 36; e++) QQ[t * 36 + e] = full ? p->QQt[e] + fxx[e] : p->QQt[e];
        for (int e = 0; e < 4; e++) RR[t * 4 + e] = full ? p->RRt[e] + fuu[e] : p->RRt[e];
        for (int e = 0; e < 12; e++) SS[t * 12 + e] = full ? 0.0 + fux[e] : 0.0;
        for (int c = 0; c < 6; c++) qq[t * 6 + c] = a[c];
        for (int c = 0; c < 2; c++) rr[t * 2 + c] = b[c];
        /* lmbd_t = AA.T @ lmbd_{t+1} + aa   (:461); AA.T = fx */
        for (int i = 0; i < 6; i++) {
            double s = 0.0;
            for (int k = 0; k < 6; k++) s += fx[i * 6 + k] * lm[(t + 1) * 6 + k];
            lm[t * 6 + i] = s + a[i];
        }
    }
    double zero6[6] = {0, 0, 0, 0, 0, 0};
    int nreg = 0;                                                          /* D  :468-470 */
    int nsing = orc_ltv_lqr(T, AA, BB, QQ, RR, SS, QQ + (size_t)(T - 1) * 36, zero6, qq, rr,
                            qq + (size_t)(T - 1) * 6, KK, NULL, dxl, dul, &nreg);
    for (int t = 0; t < T; t++) { du[0 * T + t] = dul[t * 2 + 0]; du[1 * T + t] = dul[t * 2 + 1]; }

    double descent = 0.0;                                                  /* E  :474-477 */
    for (int t = T - 2; t >= 0; t--) {
        double g0 = 0.0, g1 = 0.0;
        for (int k = 0; k < 6; k++) { g0 += BB[t * 12 + k * 2 + 0] * lm[(t + 1) * 6 + k]; g1 += BB[t * 12 + k * 2 + 1] * lm[(t + 1) * 6 + k]; }
        g0 = g0 + rr[t * 2 + 0]; g1 = g1 + rr[t * 2 + 1];
        double tmp = 0.0;
        tmp += g0 * du[0 * T + t]; tmp += g1 * du[1 * T + t];
        descent += tmp;
    }
    int ntr = 0;                                                           /* F  :482 */
    double stepsize = orc_armijo(p, prm, uu, du, x0, descent, JJ, &ntr, wx, wu);
    orc_get_update(p, stepsize, uu, du, x0, xx_new, uu_new);               /* G  :488-491 */

    if (JJ_out) *JJ_out = JJ;
    if (descent_out) *descent_out = descent;
    if (stepsize_out) *stepsize_out = stepsize;
    if (ntrials_out) *ntrials_out = ntr;
    if (nreg_out) *nreg_out = nreg;
    if (KK_out) memcpy(KK_out, KK, (size_t)T * 14 * sizeof(double));
    if (du_out) memcpy(du_out, du, (size_t)T * 2 * sizeof(double));
    if (dx_out) for (int t = 0; t < T; t++) for (int c = 0; c < 6; c++) dx_out[c * T + t] = dxl[t * 6 + c];
    if (lmbd_out) for (int t = 0; t < T; t++) for (int c = 0; c < 6; c++) lmbd_out[c * T + t] = lm[t * 6 + c];
    arena_release(mark_);
    return nsing;
}

/* ------------------------------------------------------------------------------------------
 * NewtonMethod.optimize (optcon.py:341-529) with its termination and return-index behaviour:
 *   loop kk = 0 .. max_iters-2; stop when descent[kk] >= term_cond, setting max_iters = kk;
 *   return iterate index max_iters-1 (python negative index -1 == last history slot, all zeros,
 *   when kk == 0), then uu_star[:, -1] = uu_star[:, -2].
 * hist_* arrays have max_iters entries (may be NULL).  iters_out = number of iterations executed
 * (printed 'Iter' lines).  ret_index_out = history index returned (may be -1).
 * ------------------------------------------------------------------------------------------ */
int orc_newton_optimize(const orc_problem *p, const orc_params *prm, const double *xx_init,
                        const double *uu_init, double *xx_star, double *uu_star, double *hist_J,
                        double *hist_descent, double *hist_step, int32_t *hist_ntrials,
                        int32_t *iters_out, int32_t *ret_index_out) {
    const int T = p->T;
    const size_t nx = (size_t)6 * T, nu = (size_t)2 * T;
    /* ring of three iterates: enough to serve index kk-1 after computing kk+1 */
    double *X[3], *U[3];
    for (int i = 0; i < 3; i++) { X[i] = (double *)calloc(nx, sizeof(double)); U[i] = (double *)calloc(nu, sizeof(double)); }
    memcpy(X[0], xx_init, nx * sizeof(double)); memcpy(U[0], uu_init, nu * sizeof(double));
    double x0[6];
    for (int c = 0; c < 6; c++) x0[c] = xx_init[c * T + 0];
    int max_iters = prm->max_iters, executed = 0, nsing = 0;
    int kk;
    for (kk = 0; kk < prm->max_iters - 1; kk++) {
        double J, d, s; int ntr;
        nsing += orc_newton_iterate(p, prm, kk, X[kk % 3], U[kk % 3], x0, X[(kk + 1) % 3], U[(kk + 1) % 3],
                                    &J, &d, &s, &ntr, NULL, NULL, NULL, NULL, NULL);
        if (hist_J) hist_J[kk] = J;
        if (hist_descent) hist_descent[kk] = d;
        if (hist_step) hist_step[kk] = s;
        if (hist_ntrials) hist_ntrials[kk] = ntr;
        executed++;
        if (d >= prm->term_cond) { max_iters = kk; break; }
    }
    int ret = max_iters - 1;
    if (ret < 0) { /* xx[:,:,-1]: the untouched last slot of the history arrays */
        memset(xx_star, 0, nx * sizeof(double)); memset(uu_star, 0, nu * sizeof(double));
    } else {
        memcpy(xx_star, X[ret % 3], nx * sizeof(double)); memcpy(uu_star, U[ret % 3], nu * sizeof(double));
    }
    for (int c = 0; c < 2; c++) uu_star[c * T + T - 1] = uu_star[c * T + T - 2]; /* :505 */
    if (iters_out) *iters_out = executed;
    if (ret_index_out) *ret_index_out = ret;
    for (int i = 0; i < 3; i++) { free(X[i]); free(U[i]); }
    return nsing;
}

/* ------------------------------------------------------------------------------------------
 * One outer iteration of GradientMethod.optimize (optcon.py:86-136), steepest descent.
 * PARITY UNPINNED: the reference's own method raises TypeError at optcon.py:125 (it calls armijo_stepsize with 8 of
 * the 9 arguments of optcon.py:204), so no golden vector exists.  This restates the loop with the one repair that
 * lets it run: JP = JJ[kk] is passed, and the slope handed to armijo_stepsize is the directional derivative
 * -descent[kk] (the reference accumulates descent[kk] = +sum |deltau_t|^2, :123; armijo_stepsize's test
 * J' > JP + cc*stepsize*descent, :268, needs the negative quantity, as NewtonMethod passes it).
 * out: xx_new, uu_new; JJ, descent (= sum |deltau|^2, what the reference prints), stepsize, ntrials; du (2,T) optional.
 * ------------------------------------------------------------------------------------------ */
void orc_gradient_iterate(const orc_problem *p, const orc_params *prm, const double *xx, const double *uu,
                          const double *x0, double *xx_new, double *uu_new, double *JJ_out, double *descent_out,
                          double *stepsize_out, int *ntrials_out, double *du_out) {
    const orc_mark mark_ = arena_mark();
    const int T = p->T;
    double *lm = (double *)arena_calloc((size_t)T * 6, sizeof(double));
    double *du = (double *)arena_calloc((size_t)T * 2, sizeof(double));
    double *wx = (double *)arena_calloc((size_t)T * 6, sizeof(double));
    double *wu = (double *)arena_calloc((size_t)T * 2, sizeof(double));
    double x[6], u[2], xr[6], ur[2], a[6], b[2], fx[36], fu[12];
    const double JJ = orc_traj_cost(p, xx, uu);                             /* :88-96 */
    for (int c = 0; c < 6; c++) { x[c] = xx[c * T + T - 1]; xr[c] = p->xx_ref[c * T + T - 1]; }
    orc_termcost(p, x, xr, a);                                              /* :101-102 */
    for (int c = 0; c < 6; c++) lm[(T - 1) * 6 + c] = a[c];
    double descent = 0.0;
    for (int t = T - 2; t >= 0; t--) {                                      /* :104-123 */
        for (int c = 0; c < 6; c++) { x[c] = xx[c * T + t]; xr[c] = p->xx_ref[c * T + t]; }
        for (int c = 0; c < 2; c++) { u[c] = uu[c * T + t]; ur[c] = p->uu_ref[c * T + t]; }
        orc_stagecost(p, x, u, xr, ur, a, b);
        orc_step(&p->mdl, x, u, NULL, NULL, fx, fu, NULL, NULL, NULL, NULL, NULL);
        /* lmbd_t = AA.T @ lmbd_{t+1} + aa with AA.T = fx (:114); deltau_t = -BB.T @ lmbd_{t+1} - bb with BB.T = fu (:115) */
        for (int i = 0; i < 6; i++) {
            double s = 0.0;
            for (int k = 0; k < 6; k++) s += fx[i * 6 + k] * lm[(t + 1) * 6 + k];
            lm[t * 6 + i] = s + a[i];
        }
        double d[2];
        for (int i = 0; i < 2; i++) {
            double s = 0.0;
            for (int k = 0; k < 6; k++) s += (-fu[i * 6 + k]) * lm[(t + 1) * 6 + k];
            d[i] = s - b[i];
            du[i * T + t] = d[i];
        }
        descent += d[0] * d[0] + d[1] * d[1];                              /* :123 */
    }
    int ntr = 0;
    const double stepsize = orc_armijo(p, prm, uu, du, x0, -descent, JJ, &ntr, wx, wu);   /* :125, repaired */
    orc_get_update(p, stepsize, uu, du, x0, xx_new, uu_new);                /* :131-134 */
    if (JJ_out) *JJ_out = JJ;
    if (descent_out) *descent_out = descent;
    if (stepsize_out) *stepsize_out = stepsize;
    if (ntrials_out) *ntrials_out = ntr;
    if (du_out) memcpy(du_out, du, (size_t)T * 2 * sizeof(double));
    arena_release(mark_);
}

/* Dynamics.get_initial_trajectory (aircraft_simplified.py:126-148): P-controller rollout.
   xx_ref_b (6,T): reference whose column 0 is the start state. */
void orc_initial_trajectory(const orc_model *md, int T, const double *xx_ref, double *xx, double *uu) {
    const double kp = 5, kt = 2.5;
    double x[6], u[2], xn[6];
    memset(uu, 0, (size_t)2 * T * sizeof(double));
    for (int c = 0; c < 6; c++) { x[c] = xx_ref[c * T]; xx[c * T] = x[c]; }
    for (int i = 0; i < T - 1; i++) {
        u[0] = kp * ((x[0] - xx_ref[0 * T + i + 1]) + (x[1] - xx_ref[1 * T + i + 1]));
        u[1] = kt * ((x[3] - xx_ref[3 * T + i + 1]) + (x[5] - xx_ref[5 * T + i + 1]));
        orc_step(md, x, u, NULL, xn, NULL, NULL, NULL, NULL, NULL, NULL, NULL);
        for (int c = 0; c < 6; c++) { x[c] = xn[c]; xx[c * T + i + 1] = xn[c]; }
        uu[0 * T + i] = u[0]; uu[1 * T + i] = u[1];
    }
}

/* lqr_tracking.lqr_tracking (lqr_tracking.py:245-283): linearise about (xx_opt,uu_opt), non-augmented
   ltv_LQR with constant weights and S = 0, closed-loop nonlinear rollout from xx_opt[:,0] + delta.
   KK_out (T,2,6) may be NULL. */
int orc_lqr_tracking(const orc_model *md, int T, const double *QQt, const double *RRt, const double *QQT,
                     const double *xx_opt, const double *uu_opt, const double *delta, double *xx_reg,
                     double *uu_reg, double *KK_out) {
    const orc_mark mark_ = arena_mark();
    double *AA = (double *)arena_calloc((size_t)T * 36, sizeof(double));
    double *BB = (double *)arena_calloc((size_t)T * 12, sizeof(double));
    double *QQ = (double *)arena_calloc((size_t)T * 36, sizeof(double));
    double *RR = (double *)arena_calloc((size_t)T * 4, sizeof(double));
    double *SS = (double *)arena_calloc((size_t)T * 12, sizeof(double));
    double *KK = (double *)arena_calloc((size_t)T * 12, sizeof(double));
    double *lx = (double *)arena_calloc((size_t)T * 6, sizeof(double));
    double *lu = (double *)arena_calloc((size_t)T * 2, sizeof(double));
    double x[6], u[2], xn[6], fx[36], fu[12];
    for (int t = 0; t < T; t++) { /* :268-273 */
        for (int c = 0; c < 6; c++) x[c] = xx_opt[c * T + t];
        for (int c = 0; c < 2; c++) u[c] = uu_opt[c * T + t];
        orc_step(md, x, u, NULL, xn, fx, fu, NULL, NULL, NULL, NULL, NULL);
        for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) AA[t * 36 + i * 6 + j] = fx[j * 6 + i];
        for (int i = 0; i < 6; i++) for (int j = 0; j < 2; j++) BB[t * 12 + i * 2 + j] = fu[j * 6 + i];
        memcpy(QQ + t * 36, QQt, 36 * sizeof(double));
        memcpy(RR + t * 4, RRt, 4 * sizeof(double));
    }
    int nsing = orc_ltv_lqr(T, AA, BB, QQ, RR, SS, QQT, delta, NULL, NULL, NULL, KK, NULL, lx, lu, NULL); /* :276 */
    memset(uu_reg, 0, (size_t)2 * T * sizeof(double));
    for (int c = 0; c < 6; c++) { x[c] = xx_opt[c * T] + delta[c]; xx_reg[c * T] = x[c]; } /* :265 */
    for (int t = 0; t < T - 1; t++) { /* :279-281 */
        double d[6];
        for (int c = 0; c < 6; c++) d[c] = x[c] - xx_opt[c * T + t];
        for (int i = 0; i < 2; i++) {
            double s = 0.0;
            for (int k = 0; k < 6; k++) s += KK[t * 12 + i * 6 + k] * d[k];
            u[i] = uu_opt[i * T + t] + s;
            uu_reg[i * T + t] = u[i];
        }
        orc_step(md, x, u, NULL, xn, NULL, NULL, NULL, NULL, NULL, NULL, NULL);
        for (int c = 0; c < 6; c++) { x[c] = xn[c]; xx_reg[c * T + t + 1] = xn[c]; }
    }
    if (KK_out) memcpy(KK_out, KK, (size_t)T * 12 * sizeof(double));
    arena_release(mark_);
    return nsing;
}

/* ------------------------------------------------------------------------------------------
 * Batched drivers (OpenMP over independent trajectories) — used by the parity tests and by
 * bench.py's cpu_baseline leg.  Batched arrays are (B,6,T)/(B,2,T) C-order.
 * n_iters fixed iterations starting at iteration index kk0 (no early exit), like the bench.
 * ------------------------------------------------------------------------------------------ */
int orc_newton_iterate_batch(const orc_problem *p, const orc_params *prm, int B, int kk0, int n_iters,
                             double *xx, double *uu, const double *x0, double *JJ, double *descent,
                             double *stepsize, int32_t *ntrials, int32_t *nreg, int nthreads) {
    const int T = p->T;
    int nsing = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) reduction(+ : nsing)
#endif
    for (int b = 0; b < B; b++) {
        double *xn = (double *)malloc((size_t)6 * T * sizeof(double));
        double *un = (double *)malloc((size_t)2 * T * sizeof(double));
        double *xb = xx + (size_t)b * 6 * T, *ub = uu + (size_t)b * 2 * T;
        for (int it = 0; it < n_iters; it++) {
            double J, d, s; int ntr, nr = 0;
            nsing += orc_newton_iterate(p, prm, kk0 + it, xb, ub, x0 + (size_t)b * 6, xn, un, &J, &d, &s, &ntr,
                                        &nr, NULL, NULL, NULL, NULL);
            if (nreg) nreg[(size_t)b * n_iters + it] = nr;   /* stages whose M was regularised (optcon.py:745-749) */
            memcpy(xb, xn, (size_t)6 * T * sizeof(double)); memcpy(ub, un, (size_t)2 * T * sizeof(double));
            if (JJ) JJ[(size_t)b * n_iters + it] = J;
            if (descent) descent[(size_t)b * n_iters + it] = d;
            if (stepsize) stepsize[(size_t)b * n_iters + it] = s;
            if (ntrials) ntrials[(size_t)b * n_iters + it] = ntr;
        }
        free(xn); free(un);
    }
    return nsing;
}

/* Initial guesses for a batch: the P-controller rollout from every x0[b] against the shared reference
   (get_initial_trajectory with xx_ref[:,0] replaced by x0[b]). */
void orc_initial_trajectory_batch(const orc_model *md, int B, int T, const double *xx_ref, const double *x0,
                                  double *xx, double *uu, int nthreads) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
    for (int b = 0; b < B; b++) {
        double *xr = (double *)malloc((size_t)6 * T * sizeof(double));
        memcpy(xr, xx_ref, (size_t)6 * T * sizeof(double));
        for (int c = 0; c < 6; c++) xr[c * T] = x0[(size_t)b * 6 + c];
        orc_initial_trajectory(md, T, xr, xx + (size_t)b * 6 * T, uu + (size_t)b * 2 * T);
        free(xr);
    }
}

int orc_max_threads(void) {
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
