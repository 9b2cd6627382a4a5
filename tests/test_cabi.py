"""CPU tests of the C-ABI library: it loads, exports every symbol include/ddmpc.h
declares, validates parameters like the reference constructor, and fails loudly
(no CPU fallback) when there is no HIP device.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from direct_data_driven_mpc_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "ddmpc.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ddmpc_[a-z_]+)\s*\(", src)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = L.load()
    names = _declared_functions()
    assert len(names) >= 14
    assert sorted(L.EXPORTS) == names
    for name in names:
        assert getattr(lib, name) is not None
    assert lib.ddmpc_version() == L.ABI_VERSION


def _params(**over):
    keep = {}
    q = np.array([3.0]); r = np.array([1e-4]); us = np.array([1.0, 1.0]); ys = np.array([0.65, 0.77])
    prm = L.Params()
    prm.struct_size = C.sizeof(L.Params)
    prm.m, prm.p, prm.n, prm.L, prm.N = 2, 2, 4, 30, 400
    prm.controller_type, prm.slack_type, prm.use_terminal_constraint = L.ROBUST, L.SLACK_NONE, 1
    prm.weight_kind = L.WEIGHT_SCALAR
    prm.Q = q.ctypes.data_as(L.c_double_p); prm.R = r.ctypes.data_as(L.c_double_p)
    prm.eps_max, prm.lamb_alpha, prm.lamb_sigma, prm.c = 0.002, 50.0, 1000.0, 1.0
    prm.u_s = us.ctypes.data_as(L.c_double_p); prm.y_s = ys.ctypes.data_as(L.c_double_p)
    for k, v in over.items():
        setattr(prm, k, v)
    keep["arrays"] = (q, r, us, ys)
    return prm, keep


@pytest.mark.parametrize("over,code,msg", [
    (dict(controller_type=7), L.ERR_INVALID, "Unsupported controller type."),
    (dict(slack_type=9), L.ERR_INVALID, "Unsupported slack variable constraint type."),
    (dict(slack_type=L.SLACK_NON_CONVEX), L.ERR_UNSUPPORTED, "Non-Convex slack variable"),
    (dict(L=7), L.ERR_INVALID, "two times the estimated"),
    (dict(controller_type=L.NOMINAL, L=3), L.ERR_INVALID, "greater than or equal to the estimated system order"),
    (dict(N=20), L.ERR_INVALID, "N must be greater than or equal to L."),
    (dict(struct_size=8), L.ERR_INVALID, "struct_size"),
    (dict(eps_max=0.0), L.ERR_INVALID, "eps_max"),
    (dict(weight_kind=5), L.ERR_UNSUPPORTED, "weight_kind"),
])
def test_create_validates_like_the_reference_constructor(over, code, msg):
    lib = L.load()
    prm, _keep = _params(**over)
    h = C.c_void_p()
    rc = lib.ddmpc_create(C.byref(prm), 4, 0, C.byref(h))
    assert rc == code
    assert msg in L.last_error()
    assert not h


def test_no_device_means_loud_failure_not_a_cpu_fallback():
    lib = L.load()
    if lib.ddmpc_device_count() > 0:
        pytest.skip("a HIP device is visible here")
    prm, _keep = _params()
    h = C.c_void_p()
    rc = lib.ddmpc_create(C.byref(prm), 4, 0, C.byref(h))
    assert rc == L.ERR_NO_DEVICE and "no CPU fallback" in L.last_error()
    X = np.zeros((1, 8, 2)); H = np.zeros((1, 4, 7))
    rc = lib.ddmpc_hankel(C.c_void_p(X.ctypes.data), 1, 8, 2, 2, C.c_void_p(H.ctypes.data), L.MEM_HOST, 0)
    assert rc == L.ERR_NO_DEVICE
    with pytest.raises(L.DDMPCError):
        from direct_data_driven_mpc_amd.engine import BatchedDDMPC
        BatchedDDMPC(n=4, m=2, p=2, L_=30, N=400, Q=3.0, R=1e-4, u_s=[1, 1], y_s=[0.65, 0.77], batch=2,
                     eps_max=0.002, lamb_alpha=50.0, lamb_sigma=1000.0, c=1.0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "direct_data_driven_mpc_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".inc")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("no oracle", ""), os.path.join(dirpath, f)


def test_header_is_plain_c_and_matches_the_ctypes_struct(tmp_path):
    # include/ddmpc.h must compile as C (the boundary is a C ABI: plain pointers and sizes), and the ctypes
    # mirror of ddmpc_params / ddmpc_plant must have the same size and field offsets as the C structs
    import os, shutil, subprocess
    import ctypes as C
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "probe.c"
    fields = [n for n, _ in L.Params._fields_]
    pfields = [n for n, _ in L.Plant._fields_]
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "ddmpc.h"\nint main(void) {\n'
        '  printf("%zu\\n", sizeof(ddmpc_params));\n'
        + "".join('  printf("%%zu\\n", offsetof(ddmpc_params, %s));\n' % f for f in fields)
        + '  printf("%zu\\n", sizeof(ddmpc_plant));\n'
        + "".join('  printf("%%zu\\n", offsetof(ddmpc_plant, %s));\n' % f for f in pfields)
        + '  printf("%d\\n", DDMPC_ABI_VERSION);\n  return 0;\n}\n')
    exe = tmp_path / "probe"
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)],
                   check=True, capture_output=True)
    out = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    n = len(fields)
    assert out[0] == C.sizeof(L.Params)
    assert out[1:1 + n] == [getattr(L.Params, f).offset for f in fields]
    assert out[1 + n] == C.sizeof(L.Plant)
    assert out[2 + n:2 + n + len(pfields)] == [getattr(L.Plant, f).offset for f in pfields]
    assert out[-1] == L.ABI_VERSION
