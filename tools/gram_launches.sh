#!/bin/bash
# Dev tool (GPU box, from the repo root): per-launch durations of the two structured-Gram launches ahead of the cold-solve kernel
# (rr2_gram_tiles*_kernel vs ddmpc_gram_tiles_kernel) and of the kernel itself in its three modes, per plant of
# tools/gram_modes_time.py, under rocprofv3.
#   bash tools/gram_launches.sh <outdir>
OUT=${1:-gpurun_out/r5/gram_launches}
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python tools/gram_modes_time.py --steps 3 > "$OUT/run.log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
seq = []
for r in rows:
    n = r["Kernel_Name"]
    if "rocclr" in n or "at::" in n:
        continue
    tag = "stream" if "rr2_gram_tiles" in n else "staged" if "ddmpc_gram_tiles" in n else "refine" if "true, false>" in n else "cold"
    seq.append((tag, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
# the tool runs, per plant and refinement mode: dense, structured (default launch or in-kernel), staged launch (other than 2 / 4 channels)
med = lambda v: sorted(v)[len(v) // 2]
for line in open(sys.argv[1] + "/run.log"):
    if " refinement " in line:
        print(line.rstrip()[:200])
print()
print("per-launch durations in launch order (us, median of each run of equal kernels):")
runs = []
for tag, d in seq:
    if runs and runs[-1][0] == tag and tag != "cold":
        runs[-1][1].append(d)
    else:
        runs.append([tag, [d]])
out, i = [], 0
while i < len(seq):
    # collapse repeating patterns of length 1..3
    for plen in (1, 2, 3):
        pat = [t for t, _ in seq[i:i + plen]]
        j = i
        while [t for t, _ in seq[j:j + plen]] == pat and j + plen <= len(seq):
            j += plen
        reps = (j - i) // plen
        if reps >= 3:
            cols = [[seq[i + k * plen + q][1] for k in range(reps)] for q in range(plen)]
            out.append("  %d x [%s]" % (reps, ", ".join("%s %.0f" % (pat[q], med(cols[q])) for q in range(plen))))
            i = j
            break
    else:
        out.append("  1 x [%s %.0f]" % seq[i])
        i += 1
print("\n".join(out))
PY
