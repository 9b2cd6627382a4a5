"""Build + ctypes binding of the compiled CPU restatement (oracle/ddmpc_oracle_c.c).

TEST / MEASUREMENT INFRASTRUCTURE ONLY -- never imported by the product package.

    python -m oracle.oracle_c            # build oracle/_build/libddmpc_oracle.so

The shared object is rebuilt whenever the C source is newer (gcc only; a second or so), so the GPU box
can build it itself if the snapshot's copy is stale.  -march=x86-64-v3 (AVX2 + FMA) keeps one binary valid
on the build container and on the GPU box's EPYC host.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "ddmpc_oracle_c.c")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libddmpc_oracle.so")
CFLAGS = ["-O3", "-march=x86-64-v3", "-fopenmp", "-shared", "-fPIC", "-std=c11"]


def build(force: bool = False) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        tmp = LIB + ".tmp.%d" % os.getpid()
        res = subprocess.run(["gcc"] + CFLAGS + [SRC, "-o", tmp, "-lm"], capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("gcc failed for the C oracle:\n" + res.stderr[-4000:])
        os.replace(tmp, LIB)
    return LIB


class _Spec(C.Structure):
    _fields_ = [("n", C.c_int), ("m", C.c_int), ("p", C.c_int), ("L", C.c_int), ("N", C.c_int),
                ("robust", C.c_int), ("convex", C.c_int), ("tec", C.c_int), ("max_iter", C.c_int),
                ("eps_max", C.c_double), ("lamb_alpha", C.c_double), ("lamb_sigma", C.c_double), ("c", C.c_double),
                ("qdiag", C.POINTER(C.c_double)), ("rdiag", C.POINTER(C.c_double)),
                ("u_s", C.POINTER(C.c_double)), ("y_s", C.POINTER(C.c_double))]


_lib = None


def load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.ddmpc_oracle_c_solve_batch.restype = C.c_int
        _lib.ddmpc_oracle_c_max_threads.restype = C.c_int
    return _lib


def solve_batch(spec, N: int, u_d, y_d, u_past, y_past, threads: int = 1, structured: bool = True, max_iter: int = 50):
    """`spec` is an oracle.ddmpc_oracle.QPSpec with diagonal Q, R.  Arrays: u_d [B,N,m], y_d [B,N,p],
    u_past [B,n*m], y_past [B,n*p].  Returns (u_opt [B,L*m], cost [B], status [B], iters [B])."""
    lib = load()
    qd, rd = np.ascontiguousarray(np.diag(spec.Q), float), np.ascontiguousarray(np.diag(spec.R), float)
    if not (np.array_equal(spec.Q, np.diag(qd)) and np.array_equal(spec.R, np.diag(rd))):
        raise NotImplementedError("the C restatement takes diagonal Q, R")
    us = np.ascontiguousarray(np.asarray(spec.u_s, float).reshape(-1))
    ys = np.ascontiguousarray(np.asarray(spec.y_s, float).reshape(-1))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    s = _Spec(spec.n, spec.m, spec.p, spec.L, int(N), int(bool(spec.robust)),
              int(spec.robust and spec.slack == "convex"), int(bool(spec.tec)), int(max_iter),
              float(spec.eps_max or 0.0), float(spec.lamb_alpha or 0.0), float(spec.lamb_sigma or 0.0),
              float(spec.c or 0.0), dp(qd), dp(rd), dp(us), dp(ys))
    u_d = np.ascontiguousarray(u_d, float); y_d = np.ascontiguousarray(y_d, float)
    u_past = np.ascontiguousarray(u_past, float); y_past = np.ascontiguousarray(y_past, float)
    B = u_d.shape[0]
    assert u_d.shape == (B, N, spec.m) and y_d.shape == (B, N, spec.p)
    assert u_past.shape == (B, spec.n * spec.m) and y_past.shape == (B, spec.n * spec.p)
    u_opt = np.empty((B, spec.L * spec.m)); cost = np.empty(B)
    status = np.empty(B, dtype=np.int32); iters = np.empty(B, dtype=np.int32)
    rc = lib.ddmpc_oracle_c_solve_batch(C.byref(s), B, dp(u_d), dp(y_d), dp(u_past), dp(y_past), dp(u_opt), dp(cost),
                                        status.ctypes.data_as(C.POINTER(C.c_int)),
                                        iters.ctypes.data_as(C.POINTER(C.c_int)), int(threads), int(bool(structured)))
    if rc != 0:
        raise RuntimeError("ddmpc_oracle_c_solve_batch failed (%d)" % rc)
    return u_opt, cost, status, iters


if __name__ == "__main__":
    print(build(force=True))
