"""The reference algorithm's OWN sensitivity to a last-bit change of its input — oracle only, no GPU.

For every iteration kk of the 4096-trajectory sample the parity sweep uses (tests/parity_sweep.py: random x0, step
maneuver, T = 500, seed 4242) the oracle runs ONE iteration twice from the same iterate: once as it is, once with every
input sample of `uu` moved by one ulp (random direction per element).  The difference of the two new input trajectories
is what an implementation that differs from the reference in the last bit of ONE operation upstream cannot avoid; it is
reported in exactly the measures the sweep reports for HIP-vs-oracle (channel-scaled and SURVEY 8c's elementwise figure
with its 1e-3 floor), so that the two can be read side by side:

    python tools/oracle_noise_floor.py [B=4096] [iters=20] [sweep.json] [ulps=1] > profiles/r05_oracle_noise_floor.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from aircraftoptimalcontrol_amd import problems   # problem builders only (host data); no GPU is touched
from oracle import oracle as orc


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    sweep = json.load(open(sys.argv[3])) if len(sys.argv) > 3 and sys.argv[3] != "-" else None
    ulps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    pr = problems.step_maneuver(1.0, 2e-3)
    op = orc.OracleProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    prm = orc.params(stepsize_0=1.0, armijo_maxiters=10)
    x0 = problems.random_x0(B, seed=4242)
    X, U = orc.initial_guess_batch(orc.default_model(pr.dt), pr.xx_ref, x0)
    rng = np.random.default_rng(99)
    rows = []
    for kk in range(n_it):
        Xa, Ua = X.copy(), U.copy()
        Xb = X.copy()
        way = np.where(rng.random(U.shape) < 0.5, -np.inf, np.inf)
        Ub = U
        for _ in range(ulps):
            Ub = np.nextafter(Ub, way)                                                  # `ulps` ulps, either way
        Ub[:, :, -1] = U[:, :, -1]
        ha = orc.newton_iterate_batch(op, prm, Xa, Ua, X[:, :, 0].copy(), kk, 1)
        hb = orc.newton_iterate_batch(op, prm, Xb, Ub, X[:, :, 0].copy(), kk, 1)
        fin = np.isfinite(Ua).all((1, 2)) & np.isfinite(Ub).all((1, 2)) & np.isfinite(ha["cost"][:, 0]) & np.isfinite(hb["cost"][:, 0])
        same = fin & (ha["stepsize"][:, 0] == hb["stepsize"][:, 0]) & (ha["ntrials"][:, 0] == hb["ntrials"][:, 0])
        reg = (ha["nreg"][:, 0] > 0) | (hb["nreg"][:, 0] > 0)
        d = np.where(np.isfinite(Ua) & np.isfinite(Ub), np.abs(Ua - Ub), 0.0)
        Uf = np.where(np.isfinite(Ua), Ua, 0.0)
        elem = (d / np.maximum(np.abs(Uf), 1e-3)).max((1, 2))
        chan = (d.max(2) / np.maximum(np.abs(Uf).max(2), 1e-3)).max(1)
        xsame = np.array([np.array_equal(Xa[b], Xb[b], equal_nan=True) for b in range(B)])
        ok = same & ~reg
        q = lambda a, m: {"median": float(np.median(a[m])), "p99.9": float(np.percentile(a[m], 99.9)), "max": float(a[m].max())} if m.any() else None
        row = dict(kk=kk, finite=int(fin.sum()), verdict_flips=int((fin & ~same).sum()), regularised=int((reg & fin).sum()),
                   u_channel_rel=q(chan, ok), u_elementwise_rel=q(elem, ok), n_elementwise_over_1e8=int((elem[ok] > 1e-8).sum()),
                   u_channel_rel_regularised=q(chan, same & reg), state_rounding_flips=int((ok & ~xsame).sum()))
        if sweep:
            g = sweep["per_iteration"][kk]
            row["hip_vs_oracle"] = {k: g[k] for k in ("u_channel_rel_max_unflagged", "u_channel_rel_median", "u_elementwise_rel_max_unflagged",
                                                       "u_elementwise_rel_p999", "u_elementwise_rel_median", "n_elementwise_over_1e8",
                                                       "step_mismatch", "u_channel_rel_max")}
        rows.append(row)
        print(json.dumps(row), file=sys.stderr, flush=True)
        X, U = Xa, Ua            # the unperturbed run continues (free-running oracle)
    print(json.dumps({"what": "oracle vs oracle under a perturbation of every input sample by `ulps` ulps, one iteration at a time", "ulps": ulps,
                      "B": B, "T": pr.T, "iterations": n_it, "per_iteration": rows}, indent=1))


if __name__ == "__main__":
    main()
