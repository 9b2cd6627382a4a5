"""Dev tool: per-phase breakdown of ddmpc_nominal_rr_kernel at the cfg-5 size (in-kernel s_memrealtime stamps, 100 MHz,
median over the instances of the batch) at several batch sizes: 64 / 256 = at most one workgroup per CU, 512 = two.
A cold solve at this size is two launches (MODE 1: the factors; MODE 2: the solve on them); a warm step is the second alone.

    python tools/rr_stamps.py [batch ...]
"""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
from direct_data_driven_mpc_amd.harness import generate_batch

batches = [int(x) for x in sys.argv[1:]] or [64, 256, 512]
rng = np.random.default_rng(0)
ns = n = 8; m = p = 8; Lh = 30; N = 2000
A = rng.normal(size=(ns, ns)); A *= 0.9 / max(abs(np.linalg.eigvals(A)))
plant = dict(A=A, B=rng.normal(size=(ns, m)), C=rng.normal(size=(p, ns)), D=np.zeros((p, m)), eps_max=0.0)
u_s = 0.1 * np.ones(m); y_s = (plant["C"] @ np.linalg.inv(np.eye(ns) - A) @ plant["B"]) @ u_s


def show(tag, st, seq, names):
    print("   %s" % tag)
    for i, nm in enumerate(names):
        dt = (st[:, seq[i + 1]] - st[:, seq[i]]) / 100.0
        print("      %-44s median %8.1f us   max %8.1f" % (nm, np.median(dt), dt.max()))
    tot = (st[:, seq[-1]] - st[:, seq[0]]) / 100.0
    print("      total median %.1f us, slowest instance %.1f us; last end %.1f us after the first start" % (
        np.median(tot), tot.max(), (st[:, seq[-1]].max() - st[:, seq[0]].min()) / 100.0))


for B in batches:
    d = generate_batch(range(B), N=N, plant=plant)
    up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
    eng = BatchedDDMPC(n=n, m=m, p=p, L_=Lh, N=N, Q=3.0, R=1e-4, u_s=u_s, y_s=y_s, batch=B, controller_type=L.NOMINAL)
    eng.set_data(d["u_d"], d["y_d"])
    print("B = %d" % B)
    # the factor launch alone (ddmpc_prepare), stamps 0..5
    eng.prepare()
    eng._lib.ddmpc_set_data  # (prepare again needs a new data registration)
    eng.set_data(d["u_d"], d["y_d"])
    eng.debug_stamps(True)
    eng.prepare()
    st = eng.debug_stamps(False, fetch=True).astype(np.int64).reshape(-1, 16)[:B]
    show("MODE 1 (ddmpc_prepare): factors", st, [0, 1, 2, 4, 5], ["Gram", "Cholesky of G", "C'WC", "its Cholesky"])
    # a solve on those factors (ddmpc_step), stamps 0..15
    eng.step(up, yp)
    eng.debug_stamps(True)
    eng.step(up, yp)
    sw = eng.debug_stamps(False, fetch=True).astype(np.int64).reshape(-1, 16)[:B]
    show("MODE 2 (ddmpc_step): solve on the factors", sw, [0, 3, 4, 6, 8, 9, 10, 11, 12, 13, 14, 15, 7],
         ["pattern + L_FF w = f, residual, z0 = L_RF w", "rhs C'W(zs - z0)", "T v = rhs (two substitutions)",
          "pass 1: x = L^-T w", "pass 1: H(H'x)", "pass 1: multipliers (rows, cols, L_FF' mu)", "pass 1: H(H'v)",
          "pass 1: L_I^-1 .", "pass 1: dw1 = L_FF^-1 .", "pass 1: rows, cols, T forward", "pass 1: T backward",
          "later passes + outputs"])
    eng.close()
