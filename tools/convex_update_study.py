"""CPU study for the rank-k treatment of the slack box (round 5, verdict item 1).

For instances of BASELINE configs[1] (and configs[3] with --cfg4) run the primal-dual active-set iteration of the reduced
system twice -- (a) re-factoring K = G + lam D(act) in every iteration (what the kernels did until round 4), (b) keeping the
factor of the EMPTY active set and treating the switched components S as a rank-k diagonal modification
    K(act) = K0 - d E E',  d = lam / lamb_sigma,   beta = L^-T ( y' + W S^-1 W' y' ),  W = L^-1 E,  S = I / d - W' W,  y' = y + bound W s
-- and report the distribution of k = |S| per iteration, whether both take the same iterations / end in the same active set,
and the distance between their solutions.  Test / study infrastructure only.
"""
import argparse
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from direct_data_driven_mpc_amd import harness          # noqa: E402
from oracle import ddmpc_oracle as orc                    # noqa: E402
from oracle.reduced_form import component_tables         # noqa: E402
from oracle.ddmpc_oracle import hankel_matrix            # noqa: E402
import scipy.linalg as sla                                # noqa: E402


def study(spec, N, seeds):
    n, m, p, L, Ln = spec.n, spec.m, spec.p, spec.L, spec.Ln
    d = harness.generate_batch(seeds, N=N)
    lam = spec.lamb_alpha * spec.eps_max
    dd = lam / spec.lamb_sigma
    bound = spec.c * spec.eps_max
    w_pred = slice(Ln * m + n * p, Ln * (m + p))
    ks, its, worst, mism = [], [], 0.0, 0
    for b in range(len(seeds)):
        u_d, y_d = d["u_d"][b], d["y_d"][b]
        up, yp = u_d[-n:].reshape(-1), y_d[-n:].reshape(-1)
        H = np.vstack([hankel_matrix(u_d, Ln), hankel_matrix(y_d, Ln)])
        G = H @ H.T
        # (a) re-factor
        act = np.zeros(L * p, dtype=int)
        it_a = 0
        while True:
            it_a += 1
            D, t = component_tables(spec, up, yp, act)
            Lc = np.linalg.cholesky(G + lam * np.diag(D))
            beta_a = sla.solve_triangular(Lc.T, sla.solve_triangular(Lc, t, lower=True), lower=False)
            sh = -lam * beta_a[w_pred] / spec.lamb_sigma
            new = np.where(sh > bound, 1, np.where(sh < -bound, -1, 0))
            if np.array_equal(new, act):
                break
            act = new
        act_a = act
        # (b) keep the first factor
        act = np.zeros(L * p, dtype=int)
        D0, t0 = component_tables(spec, up, yp, act)
        L0 = np.linalg.cholesky(G + lam * np.diag(D0))
        y0 = sla.solve_triangular(L0, t0, lower=True)
        it_b = 0
        while True:
            it_b += 1
            S_idx = np.nonzero(act)[0]
            k = len(S_idx)
            if it_b > 1:
                ks.append(k)
            if k == 0:
                v = y0
            else:
                rows = np.arange(Ln * m + n * p, Ln * (m + p))[S_idx]
                E = np.zeros((G.shape[0], k)); E[rows, np.arange(k)] = 1.0
                W = sla.solve_triangular(L0, E, lower=True)
                yq = y0 + bound * (W @ act[S_idx])
                Sm = np.eye(k) / dd - W.T @ W
                Ls = np.linalg.cholesky(Sm)
                cvec = sla.cho_solve((Ls, True), W.T @ yq)
                v = yq + W @ cvec
            beta_b = sla.solve_triangular(L0.T, v, lower=False)
            sh = -lam * beta_b[w_pred] / spec.lamb_sigma
            new = np.where(sh > bound, 1, np.where(sh < -bound, -1, 0))
            if np.array_equal(new, act):
                break
            act = new
        its.append(it_a)
        if it_a != it_b or not np.array_equal(act, act_a):
            mism += 1
        worst = max(worst, np.max(np.abs(beta_a - beta_b)) / np.max(np.abs(beta_a)))
    ks = np.array(ks)
    print("instances %d  iterations mean %.3f max %d  |  k per update: n=%d mean %.2f  median %d  p90 %d  p99 %d  max %d  (k>15: %d)"
          % (len(seeds), np.mean(its), np.max(its), len(ks), ks.mean() if len(ks) else 0, np.median(ks) if len(ks) else 0,
             np.percentile(ks, 90) if len(ks) else 0, np.percentile(ks, 99) if len(ks) else 0, ks.max() if len(ks) else 0, int(np.sum(ks > 15))))
    print("iteration-count / active-set mismatches between re-factoring and the kept factor: %d; max |beta_a - beta_b| / |beta| = %.2e" % (mism, worst))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg4", action="store_true")
    ap.add_argument("--count", type=int, default=256)
    a = ap.parse_args()
    if a.cfg4:
        spec = orc.spec_from_params(slack_var_constraint_type=1, L=60)
        study(spec, 1000, range(a.count))
    else:
        spec = orc.spec_from_params(slack_var_constraint_type=1)
        study(spec, 400, range(a.count))
