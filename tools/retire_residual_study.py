"""Dev tool (CPU, numpy): the running residual diagonal of dependent rows in the rank-revealing Cholesky of the permuted Gram
matrix -- what the early retirement of dependent row tiles (Rr2Chol::res, ddmpc_rr2.hpp) rests on.

    python tools/retire_residual_study.py

For BASELINE configs[4] (m = p = 8, exact data) and for the SISO plant of tools/nominal_fuzz.py case 35: after every 64
columns, how many of the rows below sit at a residue under 1e-8 / 1e-10 / 1e-12 / 1e-14 of the largest diagonal entry.  The counts
under 1e-12 and 1e-14 agree everywhere (rows at their rounding floor); the SISO plant has a few rows between 1e-12 and 1e-8
for some panels -- genuinely small but not yet complete, which is why the retirement threshold is 1e-12 and not the pivot
tolerance 1e-8."""
import sys
import numpy as np
sys.path.insert(0, ".")
from direct_data_driven_mpc_amd.harness import generate_batch
from oracle import ddmpc_oracle as orc


def study(tag, plant, m, p, n, Lh, N, inst):
    d = generate_batch(range(inst, inst + 1), N=N, plant=plant)
    Ln = Lh + n
    Hu, Hy = orc.hankel_matrix(d["u_d"][0], Ln), orc.hankel_matrix(d["y_d"][0], Ln)
    fix = list(range(n)) + list(range(Ln - n, Ln)); free = list(range(n, Ln - n))
    ru = lambda steps: [t * m + a for t in steps for a in range(m)]
    ry = lambda steps: [t * p + a for t in steps for a in range(p)]
    H = np.vstack([Hu[ru(fix)], Hy[ry(fix)], Hu[ru(free)], Hy[ry(free)]])          # fixed first, inputs before outputs
    G = H @ H.T; r = G.shape[0]; dmax = G.diagonal().max(); tol = 1e-8 * dmax
    Lm = np.zeros_like(G); skip = np.zeros(r, bool)
    print("%s: %d rows" % (tag, r))
    for j in range(r):
        v = G[j:, j] - Lm[j:, :j] @ Lm[j, :j]
        if v[0] <= tol: skip[j] = True
        else: Lm[j:, j] = v / np.sqrt(v[0])
        if (j + 1) % 64 == 0:
            dr = (G.diagonal() - (Lm[:, :j + 1] ** 2).sum(1)) / dmax
            below = np.arange(r) >= j + 1
            print("  after column %3d: rows below with residue < 1e-8: %3d, < 1e-10: %3d, < 1e-12: %3d, < 1e-14: %3d" % (
                j + 1, (dr[below] < 1e-8).sum(), (dr[below] < 1e-10).sum(), (dr[below] < 1e-12).sum(), (dr[below] < 1e-14).sum()))
    piv = G.diagonal() - (np.tril(Lm, -1) ** 2).sum(1)
    print("  rank %d; smallest accepted pivot %.1e, largest residue of a skipped one %.1e (of the largest diagonal entry)" % (
        (~skip).sum(), piv[~skip].min() / dmax, np.abs(piv[skip]).max() / dmax))


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    A = rng.normal(size=(8, 8)); A *= 0.9 / max(abs(np.linalg.eigvals(A)))
    study("BASELINE configs[4], instance 0", dict(A=A, B=rng.normal(size=(8, 8)), C=rng.normal(size=(8, 8)), D=np.zeros((8, 8)), eps_max=0.0),
          8, 8, 8, 30, 2000, 0)
    case = 35
    rng = np.random.default_rng(9000 + case)
    ns = n = int(rng.integers(2, 5))
    rows = int(rng.integers(280, 640))
    Lh = max(2 * n, rows // 2 - n); N = 2 * (Lh + 2 * n) + int(rng.integers(100, 300))
    A = rng.normal(size=(ns, ns)); A *= rng.uniform(0.5, 0.9) / max(abs(np.linalg.eigvals(A)))
    study("nominal_fuzz case 35 (SISO), instance 350", dict(A=A, B=rng.normal(size=(ns, 1)), C=rng.normal(size=(1, ns)), D=np.zeros((1, 1)), eps_max=0.0),
          1, 1, n, Lh, N, 350)
