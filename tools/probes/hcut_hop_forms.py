"""A boundary hop of the horizon cut, P_s = H + Phi^T P (I + Gam P)^-1 Phi, in three forms (CPU, numpy; no GPU, no library):
  lu        I + Gam P by LU with partial pivoting (what k_track_hcut_chain does; 1/3 of a hop's instructions are the swaps)
  woodbury  Gam = G G^T (Cholesky, pivots floored at 8 eps of the diagonal), P (I + Gam P)^-1 = P - P G (I + G^T P G)^-1 G^T P
  cholP     P = L L^T,                                                      P (I + Gam P)^-1 = L (I + L^T Gam L)^-1 L^T
both pivot-free (I + G^T P G and I + L^T Gam L are symmetric with eigenvalues >= 1).  Reference: the pivoted elimination in
long double.  Printed: median and worst error of the new P relative to its largest entry over random PSD (P, Gam) of given
ranks and row scalings 10^(+-s).  Result (EXPERIMENTS.md, round 5): the symmetric forms lose up to ten more digits than the
LU on badly scaled or rank-deficient inputs — Woodbury by cancellation, cholP because an unpivoted Cholesky of a singular
matrix is unstable — so the chain keeps its pivoted LU."""
import numpy as np

LD = np.longdouble


def chol_floor(A):
    n = 6
    L = np.zeros((n, n))
    for j in range(n):
        d = A[j, j]
        fl = 1.8e-15 * d
        d -= sum(L[j, k] ** 2 for k in range(j))
        d = max(d, fl)
        s = np.sqrt(d) if d > 0 else 0.0
        r = 1 / s if d > 0 else 0.0
        L[j, j] = s
        for i in range(j + 1, n):
            L[i, j] = (A[i, j] - sum(L[i, k] * L[j, k] for k in range(j))) * r
    return L


def ldl_solve_factory(W):
    """W = V^T D V (V unit upper): returns f(N) = V^-T N and 1/D"""
    Wc = W.copy(); V = np.eye(6); inv = np.zeros(6)
    for i in range(6):
        inv[i] = 1 / Wc[i, i]
        for j in range(i + 1, 6):
            v = Wc[i, j] * inv[i]; V[i, j] = v
            Wc[j, j:] -= v * Wc[i, j:]

    def fwd(N):
        E = np.zeros_like(N)
        for k in range(6):
            E[k] = N[k] - sum(V[i, k] * E[i] for i in range(k))
        return E
    return fwd, inv


def hop_lu(H, Phi, Gam, P):
    return H + Phi.T @ (P @ np.linalg.solve(np.eye(6) + Gam @ P, Phi))


def hop_woodbury(H, Phi, Gam, P):
    G = chol_floor(Gam)
    U = P @ G
    fwd, inv = ldl_solve_factory(np.eye(6) + G.T @ U)
    E = fwd(U.T)
    return H + Phi.T @ ((P - E.T @ (inv[:, None] * E)) @ Phi)


def hop_cholP(H, Phi, Gam, P):
    L = chol_floor(P)
    fwd, inv = ldl_solve_factory(np.eye(6) + L.T @ (Gam @ L))
    E = fwd(L.T @ Phi)
    return H + E.T @ (inv[:, None] * E)


def hop_exact(H, Phi, Gam, P):
    A = np.eye(6, dtype=LD) + Gam.astype(LD) @ P.astype(LD)
    B = Phi.astype(LD).copy()
    for c in range(6):
        pv = c + np.argmax(abs(A[c:, c])); A[[c, pv]] = A[[pv, c]]; B[[c, pv]] = B[[pv, c]]
        for r in range(c + 1, 6):
            m = A[r, c] / A[c, c]; A[r] -= m * A[c]; B[r] -= m * B[c]
    for c in range(5, -1, -1):
        B[c] = (B[c] - A[c, c + 1:] @ B[c + 1:]) / A[c, c]
    return H.astype(LD) + Phi.T.astype(LD) @ (P.astype(LD) @ B)


if __name__ == "__main__":
    rng = np.random.default_rng(2)
    print("%-28s %-22s %-22s %-22s" % ("rank P / rank Gam / sP / sG", "lu  med  max", "woodbury  med  max", "cholP  med  max"))
    for rP, rG, sP, sG in ((6, 6, 0, 0), (6, 6, 3, 0), (6, 6, 0, 4), (6, 6, 3, 4), (3, 6, 0, 0), (6, 3, 0, 0), (3, 3, 3, 4)):
        e = []
        for _ in range(400):
            d1 = np.diag(10.0 ** rng.uniform(-sG, sG, 6)); C = rng.normal(size=(6, rG)); Gam = d1 @ C @ C.T @ d1
            d2 = np.diag(10.0 ** rng.uniform(-sP, sP, 6)); D = rng.normal(size=(6, rP)); P = d2 @ D @ D.T @ d2
            Phi = rng.normal(size=(6, 6)); H = np.eye(6)
            X = hop_exact(H, Phi, Gam, P); sc = float(np.abs(X).max())
            e.append([float(np.abs(f(H, Phi, Gam, P) - X).max()) / sc for f in (hop_lu, hop_woodbury, hop_cholP)])
        e = np.array(e)
        print("%-28s " % ("%d / %d / %g / %g" % (rP, rG, sP, sG)) + " ".join("%9.1e %9.1e  " % (np.median(e[:, i]), e[:, i].max()) for i in range(3)))
