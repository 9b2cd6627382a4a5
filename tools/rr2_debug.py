import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "tools")
from direct_data_driven_mpc_amd import _lib as L
from test_gpu_round4 import _exact_plant_case, _spec_engine
from rr2_check import workspace, unpack, pk_row
m, p, n, Lh, N = 5, 4, 5, 40, 1200
B = 5
spec, plant, d, up, yp = _exact_plant_case(100 + m * 10 + p, m, p, n, Lh, N, B)
r = (m + p) * (Lh + n); rv = (r + 1) & ~1; n16 = (r + 15) & ~15
W = {}
for mode in ("one_workgroup", "phases"):
    with _spec_engine(spec, N, B) as eng:
        eng.set_large_pipeline(mode)
        eng.set_refinement("auto", max_passes=1)
        eng.set_data(d["u_d"], d["y_d"])
        eng.solve(up, yp)
        W[mode] = workspace(eng, 3)
(w0, m0), (w1, m1) = W["one_workgroup"], W["phases"]
print("nlive", m0[2 * rv], m1[2 * rv], "nRl", m0[2 * rv + 1], m1[2 * rv + 1])
s0, s1 = m0[:r], m1[:r]
print("skip differs at", np.nonzero(s0 != s1)[0].tolist())
L0, L1 = unpack(w0, r), unpack(w1, r)
d0, d1 = np.diag(L0), np.diag(L1)
dm = max(np.max(d0), np.max(d1)) ** 2
print("smallest accepted pivots (relative to the largest): old", np.sort((d0[s0 == 0] ** 2) / dm)[:6], "new", np.sort((d1[s1 == 0] ** 2) / dm)[:6])
dif = np.abs(L0 - L1)
i, j = np.unravel_index(np.argmax(dif), dif.shape)
print("factor of G: max abs diff %.3e at (%d, %d); first row with diff > 1e-6: %s" % (dif[i, j], i, j, np.nonzero(np.max(dif, axis=1) > 1e-6)[0][:5].tolist()))
nRl = int(min(m0[2 * rv + 1], m1[2 * rv + 1])); toff = pk_row(n16)
T0, T1 = unpack(w0, nRl, toff), unpack(w1, nRl, toff)
st0, st1 = m0[rv:rv + nRl], m1[rv:rv + nRl]
print("skipT differs at", np.nonzero(st0 != st1)[0].tolist())
dt = np.abs(T0 - T1); i, j = np.unravel_index(np.argmax(dt), dt.shape)
print("factor of T: max abs diff %.3e at (%d, %d), max |T| %.3e; diag ratio min old %.3e new %.3e" % (dt[i, j], i, j, np.max(np.abs(T0)), np.min(np.diag(T0)[st0 == 0]) / np.max(np.diag(T0)), np.min(np.diag(T1)[st1 == 0]) / np.max(np.diag(T1))))
