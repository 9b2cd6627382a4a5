"""ctypes binding of libddmpc.so (C ABI: include/ddmpc.h).

The HIP extension is the only compute path: if the shared object is missing or
cannot be loaded this module raises -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libddmpc.so")

# ---- constants mirrored from include/ddmpc.h ---------------------------------
ABI_VERSION = 2
OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_HIP, ERR_NOT_READY = 0, -1, -2, -3, -4, -5
NOMINAL, ROBUST = 0, 1
SLACK_NON_CONVEX, SLACK_CONVEX, SLACK_NONE = 0, 1, 2
STATUS_STRINGS = {0: "optimal", 1: "optimal_inaccurate", 2: "infeasible", 3: "unbounded", 4: "solver_error"}
WEIGHT_SCALAR, WEIGHT_DIAG, WEIGHT_DENSE = 0, 1, 2
MEM_HOST, MEM_DEVICE = 0, 1
GRAM_AUTO, GRAM_DENSE, GRAM_STRUCTURED = 0, 1, 2
OPT_CLOSED_LOOP_PATH = 1
OPT_CLOSED_LOOP_GRAPH = 2
OPT_REFINE, OPT_REFINE_MAX, OPT_REFINE_RES_LOG10 = 3, 4, 5
OPT_LARGE_PIPELINE = 6
OPT_LARGE_AFFINE_LAW = 7
OPT_CONVEX_UPDATE = 8
OPT_GRAM_LAUNCH = 9
PIPELINE_ONE_WORKGROUP, PIPELINE_PHASES = 0, 1
REFINE_OFF, REFINE_AUTO, REFINE_ALWAYS = 0, 1, 2
PATH_AUTO, PATH_COLD, PATH_WARM = 0, 1, 2
SOL_ALPHA, SOL_UBAR, SOL_YBAR, SOL_SIGMA = 0, 1, 2, 3

EXPORTS = (
    "ddmpc_version", "ddmpc_last_error", "ddmpc_device_count", "ddmpc_create", "ddmpc_destroy",
    "ddmpc_set_stream", "ddmpc_synchronize", "ddmpc_set_data", "ddmpc_solve", "ddmpc_set_setpoints",
    "ddmpc_get_solution", "ddmpc_hankel", "ddmpc_cost_model", "ddmpc_kernel_name", "ddmpc_debug_stamps",
    "ddmpc_closed_loop", "ddmpc_prepare", "ddmpc_step", "ddmpc_get_gain", "ddmpc_set_option",
    "ddmpc_pe_guard", "ddmpc_solve_from_host", "ddmpc_debug_workspace", "ddmpc_debug_poison_allocations",
)

c_double_p = C.POINTER(C.c_double)


class Params(C.Structure):
    """struct ddmpc_params (include/ddmpc.h)."""
    _fields_ = [
        ("struct_size", C.c_int32),
        ("m", C.c_int32), ("p", C.c_int32), ("n", C.c_int32), ("L", C.c_int32), ("N", C.c_int32),
        ("controller_type", C.c_int32), ("slack_type", C.c_int32),
        ("use_terminal_constraint", C.c_int32), ("weight_kind", C.c_int32),
        ("Q", c_double_p), ("R", c_double_p),
        ("eps_max", C.c_double), ("lamb_alpha", C.c_double), ("lamb_sigma", C.c_double), ("c", C.c_double),
        ("u_s", c_double_p), ("y_s", c_double_p),
        ("max_iter", C.c_int32), ("gram_mode", C.c_int32),
    ]


class Plant(C.Structure):
    """struct ddmpc_plant (include/ddmpc.h)."""
    _fields_ = [("ns", C.c_int32), ("A", c_double_p), ("B", c_double_p), ("C", c_double_p), ("D", c_double_p)]


class DDMPCError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__("libddmpc error %d: %s" % (code, message))
        self.code = code
        self.message = message


_lib = None


def load() -> C.CDLL:
    """Load libddmpc.so (building is a separate, explicit step: build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libddmpc.so is missing (%s). Build the HIP extension first: "
            "`python -m direct_data_driven_mpc_amd.build`. There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, i32p = C.c_void_p, C.c_void_p
    lib.ddmpc_version.restype = C.c_int
    lib.ddmpc_last_error.restype = C.c_char_p
    lib.ddmpc_device_count.restype = C.c_int
    lib.ddmpc_create.argtypes = [C.POINTER(Params), C.c_int64, C.c_int, C.POINTER(vp)]
    lib.ddmpc_destroy.argtypes = [vp]
    lib.ddmpc_set_stream.argtypes = [vp, vp]
    lib.ddmpc_synchronize.argtypes = [vp]
    lib.ddmpc_set_data.argtypes = [vp, vp, vp, C.c_int]
    lib.ddmpc_solve.argtypes = [vp, vp, vp, vp, vp, i32p, i32p, C.c_int]
    lib.ddmpc_set_setpoints.argtypes = [vp, vp, vp]
    lib.ddmpc_get_solution.argtypes = [vp, C.c_int, vp, C.c_int]
    lib.ddmpc_hankel.argtypes = [vp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, vp, C.c_int, C.c_int]
    lib.ddmpc_cost_model.argtypes = [vp, c_double_p, c_double_p]
    lib.ddmpc_kernel_name.argtypes = [vp]
    lib.ddmpc_kernel_name.restype = C.c_char_p
    lib.ddmpc_debug_stamps.argtypes = [vp, C.c_int, vp]
    lib.ddmpc_debug_stamps.restype = C.c_int
    lib.ddmpc_closed_loop.argtypes = [vp, C.POINTER(Plant), C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp, C.c_int]
    lib.ddmpc_closed_loop.restype = C.c_int
    lib.ddmpc_prepare.argtypes = [vp]
    lib.ddmpc_step.argtypes = [vp, vp, vp, vp, vp, i32p, i32p, C.c_int]
    lib.ddmpc_get_gain.argtypes = [vp, vp, C.c_int]
    lib.ddmpc_set_option.argtypes = [vp, C.c_int, C.c_int]
    lib.ddmpc_pe_guard.argtypes = [vp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, vp, C.c_int, C.c_int]
    lib.ddmpc_solve_from_host.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32p, i32p]
    lib.ddmpc_solve_from_host.restype = C.c_int
    lib.ddmpc_debug_workspace.argtypes = [vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, vp, vp]
    lib.ddmpc_debug_workspace.restype = C.c_int
    for name in ("ddmpc_prepare", "ddmpc_step", "ddmpc_get_gain", "ddmpc_set_option", "ddmpc_pe_guard"):
        getattr(lib, name).restype = C.c_int
    for name in ("ddmpc_create", "ddmpc_destroy", "ddmpc_set_stream", "ddmpc_synchronize", "ddmpc_set_data",
                 "ddmpc_solve", "ddmpc_set_setpoints", "ddmpc_get_solution", "ddmpc_hankel", "ddmpc_cost_model"):
        getattr(lib, name).restype = C.c_int
    if lib.ddmpc_version() != ABI_VERSION:
        raise ImportError("libddmpc.so ABI version %d != expected %d" % (lib.ddmpc_version(), ABI_VERSION))
    _lib = lib
    return lib


def last_error() -> str:
    return load().ddmpc_last_error().decode("utf-8", "replace")


def check(rc: int) -> None:
    if rc != OK:
        raise DDMPCError(rc, last_error())
