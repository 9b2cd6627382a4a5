"""In-tree build of libddmpc.so (hand-written HIP for gfx950, C ABI in include/ddmpc.h).

    python -m direct_data_driven_mpc_amd.build [--force] [--jobs N]

hipcc cross-compiles gfx950 without a GPU.  Each kernel instantiation listed in
csrc/ddmpc_instances.inc is its own translation unit so they build in parallel.
The shared object stays in-tree (git-ignored, but it travels to the GPU box).
"""
from __future__ import annotations

import argparse
import hashlib
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
    return [(nt, w) for nt, w in inst if not (SKIP_LARGE and nt > LARGE_NT) and (not ONLY_NT or nt == ONLY_NT)]


# what each kind of translation unit is built from; an object is rebuilt when the hash of these files
# and of its command line differs from the one recorded next to it (content-based: immune to
# checkouts and copies that only change modification times)
INST_DEPS = ["ddmpc_inst.hip", "ddmpc_kernels.hpp", "ddmpc_cold2.hpp"]
API_DEPS = ["ddmpc_api.hip", "ddmpc_rr2.hpp", "ddmpc_rr2_solve.hpp", "ddmpc_rr3.hpp", "ddmpc_aux_kernels.hpp", "ddmpc_workspace_kernels.hpp", "ddmpc_kernels.hpp", "ddmpc_cold2.hpp", "ddmpc_instances.inc"]
ONLY_NT = int(os.environ.get("DDMPC_ONLY_NT", "0") or 0)          # development: build just this instance of the kernels


def _fingerprint(cmd, names, extra=()) -> str:
    h = hashlib.sha256()
    # flags and defines, not the locations (hipcc, -I, source and object paths differ between checkouts)
    h.update("\0".join(a for a in cmd[1:] if not a.startswith(("-I", "/")) and a not in ("-c", "-o")).encode())
    for path in [os.path.join(CSRC, f) for f in names] + list(extra):
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _is_current(obj: str, fp: str) -> bool:
    try:
        with open(obj + ".hash") as fh:
            return os.path.exists(obj) and fh.read().strip() == fp
    except OSError:
        return False


def _compile(cmd, out):
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (out, " ".join(cmd), res.stderr[-4000:]))
    return out


def build(force: bool = False, jobs: int = 0, verbose: bool = True) -> str:
    """Compile every HIP translation unit for gfx950 and link libddmpc.so."""
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    header = os.path.join(os.path.dirname(PKG_DIR), "include", "ddmpc.h")
    tasks = []
    dev = SKIP_LARGE or ONLY_NT
    api_obj = os.path.join(OBJ_DIR, "ddmpc_api_dev.o" if dev else "ddmpc_api.o")
    api_flags = (["-DDDMPC_NO_LARGE"] if SKIP_LARGE else []) + (["-DDDMPC_ONLY_NT=%d" % ONLY_NT] if ONLY_NT else [])
    cmd = [hipcc] + COMMON_FLAGS + api_flags + ["-c", os.path.join(CSRC, "ddmpc_api.hip"), "-o", api_obj]
    tasks.append((cmd, api_obj, _fingerprint(cmd, API_DEPS, [header])))
    for nt, w in instances():
        # 2: the cold-solve kernel, "2r": the same with the iterative-refinement loop compiled in, "2c": the plain kernel with
        # the rank-k treatment of the slack box (controllers with the CONVEX box)
        for gen in ("2", "2r", "2c"):
            obj = os.path.join(OBJ_DIR, "ddmpc_inst%s_%d_%d.o" % (gen, nt, w))
            extra = {"2": [], "2r": ["-DDDMPC_INST_REF=true"], "2c": ["-DDDMPC_INST_CVX=true"]}[gen]
            cmd = [hipcc] + COMMON_FLAGS + ["-DDDMPC_INST_NT=%d" % nt, "-DDDMPC_INST_W=%d" % w] + extra + \
                  ["-c", os.path.join(CSRC, "ddmpc_inst.hip"), "-o", obj]
            tasks.append((cmd, obj, _fingerprint(cmd, INST_DEPS)))
    todo = [t for t in tasks if force or not _is_current(t[1], t[2])]
    todo.sort(key=lambda t: -int(re.search(r"ddmpc_inst2?[rc]?_(\d+)_", t[1]).group(1)) if "ddmpc_inst" in t[1] else 0)  # longest first
    if todo:
        jobs = jobs or min(len(todo), max(1, (os.cpu_count() or 2) - 1), 8)
        if verbose:
            print("[ddmpc build] compiling %d translation unit(s) for %s with %d job(s)" % (len(todo), ARCH, jobs),
                  flush=True)

        def run(t):
            cmd, obj, fp = t
            if os.path.exists(obj + ".hash"):
                os.remove(obj + ".hash")
            _compile(cmd, obj)
            with open(obj + ".hash", "w") as fh:
                fh.write(fp + "\n")
            if verbose:
                print("[ddmpc build]   %s done" % os.path.basename(obj), flush=True)

        with ThreadPoolExecutor(max_workers=jobs) as ex:
            list(ex.map(run, todo))
    objs = [t[1] for t in tasks]
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
