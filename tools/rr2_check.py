"""Dev tool (GPU): the phase-kernel pipeline of ddmpc_rr2.hpp against the one-workgroup kernels and the CPU checkers.

    python tools/rr2_check.py [--dump] [--time]

NOMINAL controllers beyond the register-resident kernels: BASELINE configs[4] (m = p = 8, exact data), a five-channel plant
whose row count is no multiple of 16, the four-tank plant with a long horizon, and noisy data (full rank).  For each: both
pipelines (DDMPC_OPT_LARGE_PIPELINE) against each other and, on exact data, against the model-based solution of the same QP;
--dump compares the workspace (Gram factor, pivot record) of instance 0 between the two pipelines and against numpy."""
import argparse, sys, time
import numpy as np
sys.path.insert(0, ".")
import torch            # (before libddmpc.so: torch brings its own HIP runtime and must initialise first)
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
from direct_data_driven_mpc_amd.harness import generate_batch
from oracle import ddmpc_oracle as orc
from oracle.nominal_exact import solve_nominal_model_based
import ctypes as C


def pk_row(i):
    t = i >> 4
    return 128 * t * (t + 1) + (i & 15) * 16 * (t + 1)


def unpack(ws, n, off=0):
    M = np.zeros((n, n))
    for i in range(n):
        M[i, :i + 1] = ws[off + pk_row(i): off + pk_row(i) + i + 1]
    return M


def workspace(eng, b=0):
    lib = L.load()
    na, nm = C.c_int64(), C.c_int64()
    L.check(lib.ddmpc_debug_workspace(eng._h, b, None, 0, None, 0, C.byref(na), C.byref(nm)))
    ws = np.empty(na.value); meta = np.empty(nm.value, dtype=np.int32)
    L.check(lib.ddmpc_debug_workspace(eng._h, b, C.c_void_p(ws.ctypes.data), na.value, C.c_void_p(meta.ctypes.data), nm.value, None, None))
    return ws, meta


def case(tag, m, p, n, Lh, N, B, eps, seed, dump=False, timeit=False):
    rng = np.random.default_rng(seed)
    ns = n
    A = rng.normal(size=(ns, ns)); A *= 0.9 / max(abs(np.linalg.eigvals(A)))
    plant = dict(A=A, B=rng.normal(size=(ns, m)), C=rng.normal(size=(p, ns)), D=np.zeros((p, m)), eps_max=eps)
    u_s = 0.1 * np.ones(m); y_s = (plant["C"] @ np.linalg.inv(np.eye(ns) - A) @ plant["B"]) @ u_s
    spec = orc.QPSpec(n=n, m=m, p=p, L=Lh, Q=3.0 * np.eye(p * Lh), R=1e-4 * np.eye(m * Lh), u_s=u_s, y_s=y_s, robust=False,
                      eps_max=0.0, lamb_alpha=0.0, lamb_sigma=0.0, c=0.0, slack="none", tec=True)
    d = generate_batch(range(B), N=N, plant=plant)
    up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
    res, wsd = {}, {}
    for mode in ("one_workgroup", "phases"):
        with BatchedDDMPC(n=n, m=m, p=p, L_=Lh, N=N, Q=3.0, R=1e-4, u_s=u_s, y_s=y_s, batch=B, controller_type=L.NOMINAL) as eng:
            eng.set_large_pipeline(mode)
            eng.set_data(d["u_d"], d["y_d"])
            res[mode] = tuple(x.copy() for x in eng.solve(up, yp))
            if dump:
                wsd[mode] = workspace(eng, B - 1)
            if timeit:
                import torch
                dev = torch.device("cuda", 0)
                t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
                ud, yd, upt, ypt = t(d["u_d"]), t(d["y_d"]), t(up), t(yp)
                eng.set_data(ud, yd)
                out = eng.solve(upt, ypt)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(5):
                    eng.solve(upt, ypt, *out)
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
                torch.cuda.synchronize(); t0 = time.perf_counter()
                eng.prepare()
                torch.cuda.synchronize(); tp = time.perf_counter() - t0
                w = eng.step(upt, ypt)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(5):
                    eng.step(upt, ypt, *w)
                torch.cuda.synchronize(); ds = (time.perf_counter() - t0) / 5
                print("  %-14s B=%d: cold %.3f ms (%.3e solves/s), prepare %.3f ms, step %.3f ms" % (mode, B, dt * 1e3, B / dt, tp * 1e3, ds * 1e3), flush=True)
                if mode == "phases":
                    eng.set_large_affine_law(True)
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    eng.prepare()
                    torch.cuda.synchronize(); tp = time.perf_counter() - t0
                    wl = eng.step(upt, ypt)
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for _ in range(20):
                        eng.step(upt, ypt, *wl)
                    torch.cuda.synchronize(); ds = (time.perf_counter() - t0) / 20
                    eu = float((wl[0] - out[0]).abs().max() / out[0].abs().max())
                    print("  %-14s B=%d: affine law: prepare %.2f ms, step %.4f ms (%.3e steps/s), max rel diff vs cold u %.2e" % (mode, B, tp * 1e3, ds * 1e3, B / ds, eu), flush=True)
    a, bq = res["one_workgroup"], res["phases"]
    sc = np.max(np.abs(a[0]), axis=1)
    du = float(np.max(np.max(np.abs(a[0] - bq[0]), axis=1) / sc))
    dc = float(np.max(np.abs(a[1] - bq[1]) / np.maximum(np.abs(a[1]), 1e-12)))
    line = "%-40s r=%d status old %s new %s | old vs new: u %.2e cost %.2e" % (tag, (m + p) * (Lh + n), np.bincount(a[2], minlength=5).tolist(),
                                                                       np.bincount(bq[2], minlength=5).tolist(), du, dc)
    if eps == 0.0:
        wu = {"one_workgroup": 0.0, "phases": 0.0}; wc = dict(wu)
        for b in range(min(B, 8)):
            mod = solve_nominal_model_based(spec, plant, up[b], yp[b])
            for mode in wu:
                wu[mode] = max(wu[mode], np.max(np.abs(res[mode][0][b] - mod["optimal_u"])) / np.max(np.abs(mod["optimal_u"])))
                wc[mode] = max(wc[mode], abs(res[mode][1][b] - mod["cost"]) / abs(mod["cost"]))
        line += " | vs model-based: old u %.2e c %.2e, new u %.2e c %.2e" % (wu["one_workgroup"], wc["one_workgroup"], wu["phases"], wc["phases"])
    else:       # noisy data: H has full row rank, so the optimum is u = u_s exactly (cost 0)
        us = np.tile(u_s, Lh)
        line += " | max |u - u_s|: old %.2e new %.2e; cost old %.2e new %.2e" % (np.max(np.abs(a[0] - us)), np.max(np.abs(bq[0] - us)), np.max(np.abs(a[1])), np.max(np.abs(bq[1])))
    print(line, flush=True)
    if dump:
        r = (m + p) * (Lh + n); n16 = (r + 15) & ~15
        (w0, m0), (w1, m1) = wsd["one_workgroup"], wsd["phases"]
        rv = (r + 1) & ~1
        print("  meta: nlive old %d new %d, nRl old %d new %d; skip equal %s, skipT equal %s" % (
            m0[2 * rv], m1[2 * rv], m0[2 * rv + 1], m1[2 * rv + 1], np.array_equal(m0[:r], m1[:r]), np.array_equal(m0[rv:rv + m0[2 * rv + 1]], m1[rv:rv + m1[2 * rv + 1]])))
        L0, L1 = unpack(w0, r), unpack(w1, r)
        print("  factor of G: max |old - new| / max|old| = %.3e (first differing row %s)" % (
            np.max(np.abs(L0 - L1)) / np.max(np.abs(L0)), (np.argwhere(np.max(np.abs(L0 - L1), axis=1) > 1e-6 * np.max(np.abs(L0)))[:1].ravel().tolist())))
        nRl = int(m1[2 * rv + 1]); nR = None
        toff = pk_row(n16)
        T0, T1 = unpack(w0, nRl, toff), unpack(w1, nRl, toff)
        print("  factor of T: max |old - new| / max|old| = %.3e" % (np.max(np.abs(T0 - T1)) / max(np.max(np.abs(T0)), 1e-300)))
        # the permuted Gram matrix and its rank-revealing factor in numpy
        Hu, Hy = orc.hankel_matrix(d["u_d"][0], Lh + n), orc.hankel_matrix(d["y_d"][0], Lh + n)


if __name__ == "__main__":
    ap = argparse.ArgumentParser(); ap.add_argument("--dump", action="store_true"); ap.add_argument("--time", action="store_true")
    a = ap.parse_args()
    case("cfg5 shape, exact data", 8, 8, 8, 30, 2000, 8, 0.0, 0, dump=a.dump)
    case("m=2 p=3 n=3 L=60 (315 rows), exact", 2, 3, 3, 60, 900, 4, 0.0, 1, dump=a.dump)
    case("m=2 p=2 n=4 L=70 (296 rows), exact", 2, 2, 4, 70, 700, 4, 0.0, 2, dump=a.dump)
    case("m=3 p=3 n=4 L=50 (324 rows), noisy", 3, 3, 4, 50, 900, 4, 0.002, 3, dump=a.dump)
    case("m=5 p=4 n=5 L=40 (405 rows), exact", 5, 4, 5, 40, 1200, 4, 0.0, 4, dump=a.dump)
    if a.time:
        case("cfg5, timing", 8, 8, 8, 30, 2000, 512, 0.0, 0, timeit=True)
