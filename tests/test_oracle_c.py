"""CPU tests: the compiled C restatement (oracle/ddmpc_oracle_c.c, the bench's cpu_baseline) against the
golden vectors and the numpy full-space oracle.  No GPU."""
import numpy as np
import pytest

from oracle import ddmpc_oracle as orc
from oracle import oracle_c

N, L, n, m, p = 400, 30, 4, 2, 2


def _batch(golden):
    u_d = np.stack([golden[f"s{s}_u_d"] for s in range(5)])
    y_d = np.stack([golden[f"s{s}_y_d"] for s in range(5)])
    return u_d, y_d, u_d[:, -n:, :].reshape(5, -1).copy(), y_d[:, -n:, :].reshape(5, -1).copy()


@pytest.mark.parametrize("tag,kw", [("none", {}), ("convex", dict(slack_var_constraint_type=1)), ("ucon", dict(tec=False))])
@pytest.mark.parametrize("structured", [True, False])
def test_c_restatement_matches_golden_solutions(golden, tag, kw, structured):
    u_d, y_d, up, yp = _batch(golden)
    spec = orc.spec_from_params(**kw)
    u, c, st, it = oracle_c.solve_batch(spec, N, u_d, y_d, up, yp, threads=2, structured=structured)
    assert np.all(st == 0)
    for s in range(5):
        ur, cr = golden[f"s{s}_{tag}_u"], float(golden[f"s{s}_{tag}_cost"][0])
        assert np.max(np.abs(u[s] - ur)) / np.max(np.abs(ur)) < 1e-9
        assert abs(c[s] - cr) / abs(cr) < 1e-10
    if tag == "convex":
        assert np.all(it >= 2)          # the box is active on the example data: at least one re-factorisation


def test_c_restatement_other_sizes_and_diagonal_weights():
    rng = np.random.default_rng(3)
    Ls, Ns = 10, 120
    q = rng.uniform(1.0, 4.0, p * Ls); r = rng.uniform(1e-4, 1e-2, m * Ls)
    spec = orc.spec_from_params(L=Ls, N=Ns, slack_var_constraint_type=1)
    spec.Q, spec.R = np.diag(q), np.diag(r)
    insts = [orc.generate_instance(s, N=Ns) for s in range(3)]
    u_d = np.stack([i["u_d"] for i in insts]); y_d = np.stack([i["y_d"] for i in insts])
    up = u_d[:, -n:, :].reshape(3, -1).copy(); yp = y_d[:, -n:, :].reshape(3, -1).copy()
    u, c, st, it = oracle_c.solve_batch(spec, Ns, u_d, y_d, up, yp)
    for b in range(3):
        sol = orc.solve_fullspace(spec, u_d[b], y_d[b], up[b], yp[b])
        assert st[b] == 0 and it[b] == max(sol.iters, 1)
        assert np.max(np.abs(u[b] - sol.optimal_u)) / np.max(np.abs(sol.optimal_u)) < 1e-9
        assert abs(c[b] - sol.cost) / abs(sol.cost) < 1e-10


def test_c_restatement_thread_count_does_not_change_results(golden):
    u_d, y_d, up, yp = _batch(golden)
    spec = orc.spec_from_params()
    a = oracle_c.solve_batch(spec, N, u_d, y_d, up, yp, threads=1)
    b = oracle_c.solve_batch(spec, N, u_d, y_d, up, yp, threads=4)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


# ------------------------------------------------------------------ sanitizer build of the CPU-side C
def _sanitized(src, out, extra=()):
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = ["gcc", "-std=c11", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           "-I" + os.path.join(root, "include")] + list(extra) + [src, "-o", out, "-lm"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-3000:]
    return out


@pytest.mark.parametrize("tag,kw", [("none", {}), ("convex", dict(slack_var_constraint_type=1)), ("ucon", dict(tec=False))])
def test_c_restatement_under_address_and_undefined_behaviour_sanitizers(golden, tmp_path, tag, kw):
    """oracle/ddmpc_oracle_c.c compiled with -fsanitize=address,undefined (no recovery: any finding aborts the run) into a small
    driver, on the five golden four-tank instances of BASELINE configs[1] -- slack NONE, the CONVEX box (active-set iterations)
    and the unconstrained-terminal scheme, structured and dense Gram: no finding, and the results equal the optimised shared
    library's to rounding (the instrumented build is -O1 without FMA contraction) and the golden solutions at their bars."""
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    exe = _sanitized(os.path.join(here, "c", "oracle_sanitizer_driver.c"), str(tmp_path / "drv"))
    u_d, y_d, up, yp = _batch(golden)
    spec = orc.spec_from_params(**kw)
    B = u_d.shape[0]
    for structured in (1, 0):
        with open(tmp_path / "in.bin", "wb") as f:
            np.array([B, N, m, p, n, L, int(spec.slack == "convex"), int(spec.tec), structured], dtype=np.int32).tofile(f)
            np.array([spec.eps_max, spec.lamb_alpha, spec.lamb_sigma, spec.c], dtype=np.float64).tofile(f)
            for a in (np.diag(spec.Q), np.diag(spec.R), spec.u_s, spec.y_s, u_d, y_d, up, yp):
                np.ascontiguousarray(a, dtype=np.float64).tofile(f)
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
        res = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, env=env, timeout=600)
        assert res.returncode == 0 and "runtime error" not in res.stderr and "AddressSanitizer" not in res.stderr, res.stderr[-3000:]
        raw = np.fromfile(tmp_path / "out.bin", dtype=np.uint8)
        nu = B * L * m
        u = raw[:8 * nu].view(np.float64).reshape(B, L * m); c = raw[8 * nu:8 * (nu + B)].view(np.float64)
        st = raw[8 * (nu + B):8 * (nu + B) + 4 * B].view(np.int32); it = raw[8 * (nu + B) + 4 * B:].view(np.int32)
        ur, cr, sr, ir = oracle_c.solve_batch(spec, N, u_d, y_d, up, yp, threads=1, structured=bool(structured))
        assert np.array_equal(st, sr) and np.array_equal(it, ir) and np.all(st == 0)
        assert np.max(np.abs(u - ur)) <= 1e-10 * np.max(np.abs(ur)) and np.max(np.abs(c - cr) / np.abs(cr)) < 1e-11
        for s in range(B):
            g_u, g_c = golden[f"s{s}_{tag}_u"], float(golden[f"s{s}_{tag}_cost"][0])
            assert np.max(np.abs(u[s] - g_u)) / np.max(np.abs(g_u)) < 1e-9 and abs(c[s] - g_c) / abs(g_c) < 1e-10


def test_plain_c_caller_builds_under_sanitizers_and_fails_cleanly_without_a_device(tmp_path):
    """tests/c/capi_caller.c (the C99 host program of the GPU suite) built with -fsanitize=address,undefined against
    include/ddmpc.h and the in-tree libddmpc.so.  Without a HIP device its own guard ends it with exit code 5 before any
    compute call -- the sanitizers watch its file parsing and the library's load / version / device-count entry points."""
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    libdir = os.path.join(root, "direct_data_driven_mpc_amd")
    if not os.path.exists(os.path.join(libdir, "libddmpc.so")):
        pytest.skip("libddmpc.so not built")
    exe = _sanitized(os.path.join(here, "c", "capi_caller.c"), str(tmp_path / "capi"),
                     extra=["-L" + libdir, "-Wl,-rpath," + libdir, "-Wl,--no-as-needed", "-lddmpc"])
    exe = str(tmp_path / "capi")
    B, Nn = 2, 400
    insts = [orc.generate_instance(s, N=Nn) for s in range(B)]
    u_d = np.stack([i["u_d"] for i in insts]); y_d = np.stack([i["y_d"] for i in insts])
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([B, Nn, m, p, n, L, 2], dtype=np.int32).tofile(f)
        for a in (u_d, y_d, u_d[:, -n:, :].reshape(B, -1), y_d[:, -n:, :].reshape(B, -1)):
            np.ascontiguousarray(a, dtype=np.float64).tofile(f)
    from direct_data_driven_mpc_amd import _lib
    have_gpu = _lib.load().ddmpc_device_count() > 0
    if have_gpu:
        pytest.skip("sanitizer runs are for the CPU build only (the GPU suite runs the same program uninstrumented)")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", LD_LIBRARY_PATH=libdir + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    res = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, env=env, timeout=600)
    assert "runtime error" not in res.stderr and "ERROR: AddressSanitizer" not in res.stderr, res.stderr[-3000:]
    assert res.returncode == 5, (res.returncode, res.stderr[-2000:])
