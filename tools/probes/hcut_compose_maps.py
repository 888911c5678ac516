"""Composition of two segment maps of the horizon cut into one (numpy check of the formulas a k_hcut_pair kernel would use).
A map (H, Phi, Gam, eta, beta) of a segment [s, e):   P_s = H + Phi^T P_e (I + Gam P_e)^-1 Phi,
                                                       p_s = eta + Phi^T (I + P_e Gam)^-1 (p_e + P_e beta).
Map 1 = the earlier segment [s, m), map 2 = the later one [m, e).  With W = (I + Gam1 H2)^-1:
    Phi = Phi2 W Phi1,   Gam = Gam2 + Phi2 W Gam1 Phi2^T,   H = H1 + Phi1^T H2 W Phi1,
    beta = beta2 + Phi2 W (beta1 - Gam1 eta2),   eta = eta1 + Phi1^T W^T (eta2 + H2 beta1)
(the combination rule of Sarkka & Garcia-Fernandez' parallel-in-time LQR in this code's notation)."""
import numpy as np


def apply(m, P, p):
    H, Phi, Gam, eta, beta = m
    Y = P @ np.linalg.inv(np.eye(6) + Gam @ P)
    return H + Phi.T @ Y @ Phi, eta + Phi.T @ np.linalg.solve(np.eye(6) + P @ Gam, p + P @ beta)


def compose(m1, m2):
    H1, F1, G1, e1, b1 = m1
    H2, F2, G2, e2, b2 = m2
    W = np.linalg.inv(np.eye(6) + G1 @ H2)
    return (H1 + F1.T @ H2 @ W @ F1, F2 @ W @ F1, G2 + F2 @ W @ G1 @ F2.T, e1 + F1.T @ W.T @ (e2 + H2 @ b1), b2 + F2 @ W @ (b1 - G1 @ e2))


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    worst = 0.0
    for _ in range(200):
        def rmap():
            A = rng.normal(size=(6, 6)); B = rng.normal(size=(6, 4))
            return (A @ A.T, rng.normal(size=(6, 6)), B @ B.T, rng.normal(size=6), rng.normal(size=6))
        m1, m2 = rmap(), rmap()
        C = rng.normal(size=(6, 6)); P = C @ C.T; p = rng.normal(size=6)
        Pa, pa = apply(m1, *apply(m2, P, p))
        Pb, pb = apply(compose(m1, m2), P, p)
        worst = max(worst, np.abs(Pa - Pb).max() / np.abs(Pa).max(), np.abs(pa - pb).max() / np.abs(pa).max())
        Hc, Fc, Gc, _, _ = compose(m1, m2)
        assert np.abs(Hc - Hc.T).max() < 1e-9 * np.abs(Hc).max() and np.abs(Gc - Gc.T).max() < 1e-9 * np.abs(Gc).max()
    print("two maps applied in turn against their composition: worst relative difference %.1e" % worst)
