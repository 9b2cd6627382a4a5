"""Timing only (no oracle, no worker processes: safe under rocprofv3) of the BASELINE configs[4] problem on
ddmpc_nominal_rr_kernel, or with --robust of the same size with the ROBUST scheme + slack box on ddmpc_large_solve_kernel.

    python tools/cfg5_time.py [--robust] [--steps 5] [--warm]

Prints ms per batch, solves/s and the algorithmic TFLOP/s against the fp64-MFMA peak (flop model below).  Parity at this
size is tools/config5_check.py and the GPU tests."""
import argparse, sys
import numpy as np
sys.path.insert(0, ".")
import torch
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
from direct_data_driven_mpc_amd.harness import generate_batch

PEAK_TF = 78.6


def large_flops(m, p, n, Lh, N, robust, iters=1.0, passes=1):
    """Algorithmic flops of one solve on the global-workspace kernels (what the mathematics needs, not what the tiles
    execute): Hankel-structured Gram (lag sums + window walk), Cholesky, reduced system, refinement with exact products."""
    nch, Ln = m + p, Lh + n
    r, c = nch * Ln, N - Ln + 1
    gram = 2.0 * nch * nch * Ln * c + 2.0 * r * r
    refine = passes * (8.0 * r * c + 8.0 * r * r)            # two H(H'x) products and four triangular solves per pass
    if robust:
        nB = p * Lh                                          # boxed components: re-factored per active-set iteration
        return gram + r ** 3 / 3.0 + (iters - 1.0) * nB ** 3 / 3.0 + 4.0 * r * r * iters + refine / 2.0
    nF = 2 * n * nch                                         # fixed components (past window + terminal steps)
    nR = r - nF
    return gram + r ** 3 / 3.0 + 2.0 * nR ** 3 / 3.0 + 6.0 * r * r + refine


if __name__ == "__main__":
    ap = argparse.ArgumentParser(); ap.add_argument("--robust", action="store_true"); ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--warm", action="store_true", help="also time ddmpc_prepare and ddmpc_step (solves on what ddmpc_prepare kept)")
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    ns = n = 8; m = p = 8; Lh = 30; N = 2000; B = a.batch
    A = rng.normal(size=(ns, ns)); A *= 0.9 / max(abs(np.linalg.eigvals(A)))
    plant = dict(A=A, B=rng.normal(size=(ns, m)), C=rng.normal(size=(p, ns)), D=np.zeros((p, m)), eps_max=0.002 if a.robust else 0.0)
    u_s = 0.1 * np.ones(m); y_s = (plant["C"] @ np.linalg.inv(np.eye(ns) - A) @ plant["B"]) @ u_s
    d = generate_batch(range(B), N=N, plant=plant)
    dev = torch.device("cuda", 0)
    ud, yd = torch.from_numpy(d["u_d"]).to(dev), torch.from_numpy(d["y_d"]).to(dev)
    up = torch.from_numpy(d["u_d"][:, -n:, :].reshape(B, -1).copy()).to(dev)
    yp = torch.from_numpy(d["y_d"][:, -n:, :].reshape(B, -1).copy()).to(dev)
    if a.robust:
        eng = BatchedDDMPC(n=n, m=m, p=p, L_=Lh, N=N, Q=3.0, R=1e-4, u_s=u_s, y_s=y_s, batch=B, controller_type=L.ROBUST,
                           slack_type=L.SLACK_CONVEX, eps_max=0.002, lamb_alpha=50.0, lamb_sigma=1000.0, c=1.0)
    else:
        eng = BatchedDDMPC(n=n, m=m, p=p, L_=Lh, N=N, Q=3.0, R=1e-4, u_s=u_s, y_s=y_s, batch=B, controller_type=L.NOMINAL)
    eng.set_data(ud, yd)
    out = eng.solve(up, yp)
    eng.solve(up, yp, *out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.steps):
        eng.solve(up, yp, *out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.steps
    its = float(out[3].double().mean())
    fl = large_flops(m, p, n, Lh, N, a.robust, iters=its)
    rate = B / ms * 1e3
    print("%s B=%d: %.2f ms per batch, %.3e solves/s; status ok %s; %.1f MFLOP per solve (algorithmic) -> %.2f TFLOP/s = %.3f of the "
          "%.1f TF fp64-MFMA peak" % (eng.kernel_name(), B, ms, rate, bool((out[2] == 0).all()), fl / 1e6, rate * fl / 1e12,
                                      rate * fl / 1e12 / PEAK_TF, PEAK_TF))
    if a.warm:
        torch.cuda.synchronize()
        e0.record()
        eng.prepare()
        e1.record(); torch.cuda.synchronize()
        prep_ms = e0.elapsed_time(e1)
        w = eng.step(up, yp)
        same = bool((w[0] == out[0]).all() and (w[1] == out[1]).all())
        eng.step(up, yp, *w)
        e0.record()
        for _ in range(a.steps):
            eng.step(up, yp, *w)
        e1.record(); torch.cuda.synchronize()
        wms = e0.elapsed_time(e1) / a.steps
        print("warm: ddmpc_prepare %.2f ms once per data set; ddmpc_step %.2f ms per batch, %.3e steps/s (bit-equal to ddmpc_solve: %s)"
              % (prep_ms, wms, B / wms * 1e3, same))
        if not a.robust:
            eng.set_large_affine_law(True)             # DDMPC_OPT_LARGE_AFFINE_LAW: the law z(past), one HBM-bound launch per step
            torch.cuda.synchronize()
            e0.record()
            eng.prepare()
            e1.record(); torch.cuda.synchronize()
            prep_ms = e0.elapsed_time(e1)
            w = eng.step(up, yp)
            eng.step(up, yp, *w)
            e0.record()
            for _ in range(20):
                eng.step(up, yp, *w)
            e1.record(); torch.cuda.synchronize()
            wms = e0.elapsed_time(e1) / 20
            nf, r, nFp = n * (m + p), (m + p) * (Lh + n), 2 * n * (m + p)
            gbytes = 8.0 * (nf + 1) * (r + nFp)
            print("affine law: ddmpc_prepare %.1f ms once per data set; ddmpc_step %.4f ms per batch, %.3e steps/s, %.0f GB/s of law = %.2f of "
                  "the 8 TB/s HBM peak; max rel diff vs ddmpc_solve u %.2e" % (prep_ms, wms, B / wms * 1e3, gbytes * B / wms / 1e6,
                                                                          gbytes * B / wms / 1e6 / 8000.0, float((w[0] - out[0]).abs().max() / out[0].abs().max())))
