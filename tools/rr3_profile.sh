#!/bin/bash
# Dev tool (GPU box, from the repo root): time + per-kernel trace of the ROBUST scheme at configs[4]'s size (ddmpc_rr3.hpp).
#   bash tools/rr3_profile.sh <outdir>
OUT=${1:-gpurun_out/r5/rr3}
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 200 python tools/cfg5_time.py --robust --warm > "$OUT/time.log" 2>&1
tail -4 "$OUT/time.log"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python tools/cfg5_time.py --robust --steps 3 > "$OUT/stats.log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/stats/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:16]:
    print("%-70s calls %4s total %10.1f us avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3))
PY
