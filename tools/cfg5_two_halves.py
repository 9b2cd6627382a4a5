"""Dev tool (GPU): what two half batches of BASELINE configs[4] on two streams gain over one batch on one stream.

    python tools/cfg5_two_halves.py [--parts 2]

Each part is its own handle (own stream); the solves are enqueued back to back and timed together."""
import argparse, sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
from direct_data_driven_mpc_amd.harness import generate_batch

ap = argparse.ArgumentParser(); ap.add_argument("--parts", type=int, default=2); ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
rng = np.random.default_rng(0)
ns = n = 8; m = p = 8; Lh = 30; N = 2000; B = 512
A = rng.normal(size=(ns, ns)); A *= 0.9 / max(abs(np.linalg.eigvals(A)))
plant = dict(A=A, B=rng.normal(size=(ns, m)), C=rng.normal(size=(p, ns)), D=np.zeros((p, m)), eps_max=0.0)
u_s = 0.1 * np.ones(m); y_s = (plant["C"] @ np.linalg.inv(np.eye(ns) - A) @ plant["B"]) @ u_s
d = generate_batch(range(B), N=N, plant=plant)
dev = torch.device("cuda", 0)
for parts in (1, a.parts):
    nb = B // parts
    engs, args, outs = [], [], []
    for k in range(parts):
        sl = slice(k * nb, (k + 1) * nb)
        eng = BatchedDDMPC(n=n, m=m, p=p, L_=Lh, N=N, Q=3.0, R=1e-4, u_s=u_s, y_s=y_s, batch=nb, controller_type=L.NOMINAL)
        ud, yd = torch.from_numpy(d["u_d"][sl]).to(dev), torch.from_numpy(d["y_d"][sl]).to(dev)
        up = torch.from_numpy(d["u_d"][sl, -n:, :].reshape(nb, -1).copy()).to(dev)
        yp = torch.from_numpy(d["y_d"][sl, -n:, :].reshape(nb, -1).copy()).to(dev)
        eng.set_data(ud, yd)
        out = eng.solve(up, yp)
        engs.append(eng); args.append((up, yp, ud, yd)); outs.append(out)
    for e in engs: e.synchronize()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps):
        for e, (up, yp, _, _), out in zip(engs, args, outs):
            e.solve(up, yp, *out)
    for e in engs: e.synchronize()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
    print("%d part(s) of %d instances: %.3f ms per %d solves (%.3e solves/s)" % (parts, nb, dt * 1e3, B, B / dt), flush=True)
    for e in engs: e.close()
