"""Dev tool (GPU): the pivot candidates of G's rank-revealing factorisation (phase kernels) for exact-data NOMINAL problems --
cfg 5, the 9-channel plant whose noise residue sits above the 1e-8 tolerance (seed 154), a few plants of the fuzz set -- sorted,
relative to the largest diagonal entry: where is the gap between rounding residues and genuine pivots, and how wide is it."""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC


def candidates(eng, b, n16):
    lib = L.load()
    out = np.zeros(n16)
    L.check(lib.ddmpc_debug_workspace(eng._h, b, C.c_void_p(out.ctypes.data), -1, None, 0, None, None))
    return out


def show(tag, spec, N, d, up, yp, B):
    from test_gpu_round4 import _spec_engine
    r = (spec.m + spec.p) * (spec.L + spec.n)
    n16 = (r + 15) & ~15
    with _spec_engine(spec, N, B) as eng:
        eng.set_data(d["u_d"], d["y_d"])
        u, cost, status, _ = eng.solve(up, yp)
        for b in range(B):
            c = candidates(eng, b, n16)[:r]
            rel = c / c.max()
            pos = np.sort(rel[rel > 0])
            small = pos[pos < 1e-3]
            ratios = small[1:] / small[:-1]
            j = int(np.argmax(ratios)) if len(ratios) else 0
            acc = rel[rel > 1e-8]
            bound = spec.m * (spec.L + spec.n) + spec.n              # rank H <= m (L + n) + n for exact data (Willems' lemma)
            desc = np.sort(rel)[::-1]
            note = ""
            if len(acc) > bound:                                      # what rr2_rank_margin_kernel does with this instance
                note = "  -> MORE than the rank bound %d: tolerance moved to %.2e (between %.2e and %.2e), factored again" % (
                    bound, np.sqrt(desc[bound - 1] * desc[bound]), desc[bound], desc[bound - 1])
            print("%-22s b=%d status %d: candidates > 0: %d, above the fixed 1e-8: %d (rank bound %d); smallest of them %.2e, largest below %.2e; "
                  "largest gap below 1e-3: %.2e .. %.2e (x%.0f)%s" % (tag, b, status[b], len(pos), len(acc), bound, acc.min(),
                                                                     rel[(rel > 0) & (rel <= 1e-8)].max() if np.any((rel > 0) & (rel <= 1e-8)) else 0.0,
                                                                     small[j] if len(small) else 0.0, small[j + 1] if len(small) > j + 1 else 0.0,
                                                                     ratios[j] if len(ratios) else 0.0, note), flush=True)


if __name__ == "__main__":
    from test_gpu_round3 import _config5
    from test_gpu_round4 import _exact_plant_case
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    spec, plant, N, d, up, yp = _config5(nb)
    show("cfg5", spec, N, d, up, yp, nb)
    for seed, (m, p, n, Lh, N) in ((154, (5, 4, 5, 40, 1200)), (4, (5, 4, 5, 40, 1200)), (125, (2, 3, 3, 60, 900)), (142, (2, 2, 4, 70, 700)), (7, (3, 3, 4, 50, 900)), (11, (1, 1, 6, 150, 900))):
        spec, plant, d, up, yp = _exact_plant_case(seed, m, p, n, Lh, N, nb)
        show("seed %d (%dx%d)" % (seed, m, p), spec, N, d, up, yp, nb)
