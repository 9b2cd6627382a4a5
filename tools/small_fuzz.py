"""Extended seeded sweep of the register-resident kernels (the 12 cases of tests/test_gpu_parity.py::
test_random_systems_against_oracle continued to `--cases`): random stable plants, m, p in 1..3, every scheme / slack /
terminal-constraint mode, scalar / diagonal / dense weights, against the full-space CPU oracle; cold solve and warm step.

    python tools/small_fuzz.py [--cases 96] [--refine auto|always]
"""
import argparse, sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import test_gpu_parity as T
from direct_data_driven_mpc_amd import _lib as L

ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=96)
ap.add_argument("--refine", default="auto", choices=["auto", "always"]); a = ap.parse_args()
bad = 0
for case in range(12, a.cases):
    try:
        T.test_random_systems_against_oracle(None, case, a.refine)
        print("case %3d ok" % case, flush=True)
    except AssertionError as e:
        bad += 1
        print("case %3d FAILED: %s" % (case, str(e)[:200]), flush=True)
    except L.DDMPCError as e:
        print("case %3d rejected at create: %s" % (case, str(e)[:120]), flush=True)
print("%d of %d additional cases failed" % (bad, a.cases - 12))
