"""Dev tool: the two stages of the AUTO refinement trigger -- q = eps max G_kk |beta|_inf / |t|_inf (the free a-priori bound)
and res = |t - (H (H' beta) + lam D beta)|_inf / |t|_inf (the exact-Hankel residual), both read from the kernel's
diagnostic stamp -- and the effect of refinement, on the benchmark data and on the seeded random-plant sweep of
tests/test_gpu_parity.py: per case q, res, and the errors with refinement off / auto (default threshold) / always.

    python tools/refine_calib.py [ncases]
"""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_parity as T
from direct_data_driven_mpc_amd import _lib as L
from oracle import ddmpc_oracle as orc
DEFAULT = -10.7

def run(eng, up, yp, mode, res=None):
    eng.set_refinement(mode, res_log10=res)
    eng.debug_stamps(True)
    u, c, s, it = eng.solve(up, yp)
    st = eng.debug_stamps(False, fetch=True)
    est = st[:, 12].copy().view(np.float64)
    return u.copy(), c.copy(), s.copy(), est

def errs(spec, d, up, yp, u, c):
    eu = ec = 0.0
    for b in range(u.shape[0]):
        sol = orc.solve_fullspace(spec, d["u_d"][b], d["y_d"][b], up[b], yp[b])
        eu = max(eu, np.max(np.abs(u[b] - sol.optimal_u)) / max(np.max(np.abs(sol.optimal_u)), 1e-3))
        ec = max(ec, abs(c[b] - sol.cost) / max(abs(sol.cost), 1e-6))
    return eu, ec

# benchmark data
for kw in (dict(), dict(slack_var_constraint_type=1), dict(tec=False)):
    spec = orc.spec_from_params(**kw)
    B = 4096 if not kw else 64                 # the benchmark configuration: the whole batch
    u_d, y_d, up, yp = T._instances(B)
    d = dict(u_d=u_d, y_d=y_d)
    with T._engine(spec, 400, B) as eng:
        eng.set_data(u_d, y_d)
        u0, c0, s0, q = run(eng, up, yp, "auto", 0.0)              # threshold 1: never checked exactly, stamp = q, nothing flagged
        _, _, _, est = run(eng, up, yp, "auto", -300.0)            # threshold 1e-300: always checked exactly, stamp = res
        u2, c2, s2, _ = run(eng, up, yp, "always")
    print("four-tank %-36s q %.2e..%.2e  res %.2e..%.2e  res/q %.2f..%.2f   err off %.1e/%.1e   always %.1e/%.1e" % (
        kw, q.min(), q.max(), est.min(), est.max(), (est / q).min(), (est / q).max(),
        *errs(spec, d, up, yp, u0[:8], c0[:8]), *errs(spec, d, up, yp, u2[:8], c2[:8])), flush=True)

# the random-plant sweep: rebuild each case's inputs the way the test does
import inspect
src = inspect.getsource(T.test_random_systems_against_oracle)
body = src.split("    B = 3\n")[0].split("(gpu, case, refine):\n", 1)[1]          # everything up to the engine part
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 48
worst_off = worst_on = 0.0
for case in range(ncases):
    env = dict(T.__dict__); env["case"] = case
    exec("if True:\n" + body, env)
    spec, plant, N, n = env["spec"], env["plant"], env["N"], env["n"]
    B = 3
    d = T.generate_batch(range(case * 10, case * 10 + B), N=N, plant=plant)
    up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
    try:
        with T._engine(spec, N, B) as eng:
            eng.set_data(d["u_d"], d["y_d"])
            u0, c0, s0, q = run(eng, up, yp, "auto", 0.0)
            _, _, _, est = run(eng, up, yp, "auto", -300.0)
            u1, c1, s1, _ = run(eng, up, yp, "auto", DEFAULT)
            u2, c2, s2, _ = run(eng, up, yp, "always")
            name = eng.kernel_name()
    except L.DDMPCError as e:
        print("case %3d rejected: %s" % (case, str(e)[:80])); continue
    e0, e1, e2 = errs(spec, d, up, yp, u0, c0), errs(spec, d, up, yp, u1, c1), errs(spec, d, up, yp, u2, c2)
    worst_off = max(worst_off, e0[0]); worst_on = max(worst_on, e1[0])
    if not spec.robust:
        q = est = np.ones(1)
    print("case %3d %-28s r=%3d robust=%d slack=%-6s q %.1e..%.1e res %.1e..%.1e res/q %.2f..%.2f  off %.1e/%.1e  auto %.1e/%.1e  always %.1e/%.1e%s" % (
        case, name, (spec.m + spec.p) * (spec.L + spec.n), spec.robust, spec.slack, q.min(), q.max(), est.min(), est.max(),
        (est / q).min(), (est / q).max(), *e0, *e1, *e2,
        "   <-- over the bars" if spec.robust and (e1[0] > 1e-8 or e1[1] > 1e-9) else ""), flush=True)
print("worst u error: off %.2e, auto %.2e" % (worst_off, worst_on))
