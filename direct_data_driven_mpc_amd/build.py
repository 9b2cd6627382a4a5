"""In-tree build of libddmpc.so (hand-written HIP for gfx950, C ABI in include/ddmpc.h).

    python -m direct_data_driven_mpc_amd.build [--force] [--jobs N]

hipcc cross-compiles gfx950 without a GPU.  Each kernel instantiation listed in
csrc/ddmpc_instances.inc is its own translation unit so they build in parallel.
The shared object stays in-tree (git-ignored, but it travels to the GPU box).
"""
from __future__ import annotations

import argparse
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
OBJ_DIR = os.path.join(PKG_DIR, "_build")
LIB_PATH = os.path.join(PKG_DIR, "libddmpc.so")
ARCH = "gfx950"

# -simplifycfg-sink-common=false: the sink-common transform merges per-tile code
# into PHIs of accumulator pointers, which pins the MFMA accumulators in scratch.
COMMON_FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC",
                "-mllvm", "-simplifycfg-sink-common=false", "-I" + CSRC] + os.environ.get("DDMPC_EXTRA_FLAGS", "").split()


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


SKIP_LARGE = os.environ.get("DDMPC_SKIP_LARGE", "") not in ("", "0")   # development only
LARGE_NT = 16


def instances():
    txt = open(os.path.join(CSRC, "ddmpc_instances.inc")).read()
    inst = [(int(a), int(b)) for a, b in re.findall(r"^DDMPC_INSTANCE\((\d+),\s*(\d+)\)", txt, re.M)]
    return [(nt, w) for nt, w in inst if not (SKIP_LARGE and nt > LARGE_NT)]


# what each kind of translation unit is built from (mtime-based rebuild)
INST_DEPS = ["ddmpc_inst.hip", "ddmpc_kernels.hpp"]
API_DEPS = ["ddmpc_api.hip", "ddmpc_aux_kernels.hpp", "ddmpc_kernels.hpp", "ddmpc_instances.inc"]


def _newest(names, extra=()) -> float:
    paths = [os.path.join(CSRC, f) for f in names] + list(extra)
    return max(os.path.getmtime(p) for p in paths)


def _compile(cmd, out):
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (out, " ".join(cmd), res.stderr[-4000:]))
    return out


def build(force: bool = False, jobs: int = 0, verbose: bool = True) -> str:
    """Compile every HIP translation unit for gfx950 and link libddmpc.so."""
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    newest_inst = _newest(INST_DEPS)
    newest_api = _newest(API_DEPS, [os.path.join(os.path.dirname(PKG_DIR), "include", "ddmpc.h")])
    tasks = []
    api_obj = os.path.join(OBJ_DIR, "ddmpc_api_nolarge.o" if SKIP_LARGE else "ddmpc_api.o")
    api_flags = ["-DDDMPC_NO_LARGE"] if SKIP_LARGE else []
    tasks.append(([hipcc] + COMMON_FLAGS + api_flags + ["-c", os.path.join(CSRC, "ddmpc_api.hip"), "-o", api_obj], api_obj))
    for nt, w in instances():
        obj = os.path.join(OBJ_DIR, "ddmpc_inst_%d_%d.o" % (nt, w))
        tasks.append(([hipcc] + COMMON_FLAGS + ["-DDDMPC_INST_NT=%d" % nt, "-DDDMPC_INST_W=%d" % w, "-c",
                                                 os.path.join(CSRC, "ddmpc_inst.hip"), "-o", obj], obj))
    todo = [(c, o) for c, o in tasks if force or not os.path.exists(o) or
            os.path.getmtime(o) < (newest_inst if "ddmpc_inst_" in o else newest_api)]
    todo.sort(key=lambda t: -int(re.search(r"ddmpc_inst_(\d+)_", t[1]).group(1)) if "ddmpc_inst_" in t[1] else 0)  # longest first
    if todo:
        jobs = jobs or min(len(todo), max(1, (os.cpu_count() or 2) - 1), 8)
        if verbose:
            print("[ddmpc build] compiling %d translation unit(s) for %s with %d job(s)" % (len(todo), ARCH, jobs),
                  flush=True)
        with ThreadPoolExecutor(max_workers=jobs) as ex:
            list(ex.map(lambda t: _compile(*t), todo))
    objs = [o for _, o in tasks]
    if todo or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(o) for o in objs):
        _compile([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB_PATH] + objs, LIB_PATH)
        if verbose:
            print("[ddmpc build] linked", LIB_PATH, flush=True)
    return LIB_PATH


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=0)
    a = ap.parse_args(argv)
    print(build(force=a.force, jobs=a.jobs))


if __name__ == "__main__":
    sys.exit(main())
