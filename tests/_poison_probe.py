"""Helper of test_gpu_round4.py::test_results_do_not_depend_on_what_the_allocator_hands_back.  digest(fill) pre-fills every fresh
device buffer of the library with the byte `fill` (ddmpc_debug_poison_allocations; 0: off), solves a few controllers whose
shapes leave parts of the work buffers unused (a short trajectory: fewer column groups than partial-sum slots; row counts off
the tile / block sizes) and returns a digest of the raw output bits."""
import hashlib
import sys

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: F401,E402  (initialises the HIP runtime before libddmpc.so)
from direct_data_driven_mpc_amd import _lib as L                       # noqa: E402
from direct_data_driven_mpc_amd.engine import BatchedDDMPC            # noqa: E402
from direct_data_driven_mpc_amd.harness import generate_batch         # noqa: E402

CASES = [  # (m, p, n, L, N, eps, controller, pipeline)
    (1, 1, 4, 257, 642, 0.0, "NOMINAL", "phases"),          # 522 rows, 382 Hankel columns: 5 column groups of 8
    (1, 1, 4, 257, 642, 0.0, "NOMINAL", "one_workgroup"),
    (2, 3, 3, 60, 900, 0.0, "NOMINAL", "phases"),           # 315 rows: not a multiple of 16
    (2, 2, 4, 30, 400, 0.002, "ROBUST", None),              # the register-resident kernels, structured Gram in the kernel
    (3, 2, 3, 24, 400, 0.002, "ROBUST", None),              # ... structured Gram from the launch ahead of it
    (2, 2, 4, 60, 1000, 0.002, "ROBUST", None),             # the global-workspace kernel
]


def digest(fill):
    lib = L.load()
    lib.ddmpc_debug_poison_allocations(int(fill))
    h = hashlib.sha256()
    try:
        for (m, p, n, Lh, N, eps, ctl, pipe) in CASES:
            rng = np.random.default_rng(7)
            A = rng.normal(size=(n, n)); A *= 0.8 / max(abs(np.linalg.eigvals(A)))
            plant = dict(A=A, B=rng.normal(size=(n, m)), C=rng.normal(size=(p, n)), D=np.zeros((p, m)), eps_max=eps)
            u_s = 0.1 * np.ones(m); y_s = (plant["C"] @ np.linalg.inv(np.eye(n) - A) @ plant["B"]) @ u_s
            B = 3
            d = generate_batch(range(B), N=N, plant=plant)
            up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
            kw = dict(controller_type=L.NOMINAL) if ctl == "NOMINAL" else dict(controller_type=L.ROBUST, slack_type=L.SLACK_CONVEX,
                                                                              eps_max=eps, lamb_alpha=50.0, lamb_sigma=1000.0, c=1.0)
            with BatchedDDMPC(n=n, m=m, p=p, L_=Lh, N=N, Q=3.0, R=1e-2, u_s=u_s, y_s=y_s, batch=B, **kw) as eng:
                if pipe:
                    eng.set_large_pipeline(pipe)
                eng.set_data(d["u_d"], d["y_d"])
                u, cost, status, it = eng.solve(up, yp)
                eng.prepare()
                w = eng.step(up, yp)
            for a in (u, cost, status, it, w[0], w[1]):
                h.update(np.ascontiguousarray(a).tobytes())
            assert np.all(np.isfinite(u)) and np.all(status == 0), (m, p, n, Lh, ctl, pipe, status)
    finally:
        lib.ddmpc_debug_poison_allocations(0)
    return h.hexdigest()


if __name__ == "__main__":
    for f in (0, 63, 255):
        print("DIGEST", f, digest(f))
