#!/bin/bash
# Dev tool (GPU): kernel traces of the cfg-5 solve for several values of an experiment knob.   bash tools/ab_rr2.sh <tag> <ENVVAR> v1 v2 ...
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; VAR=$2; shift 2
mkdir -p $ROOT/gpurun_out/r4
cd $ROOT
for v in "$@"; do
  export $VAR=$v
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4/ab_${TAG}_$v -- python tools/cfg5_time.py --steps 3 > gpurun_out/r4/ab_${TAG}_$v.log 2>&1 || { echo "variant $v failed"; exit 1; }
  f=$(find gpurun_out/r4/ab_${TAG}_$v -name "*kernel_trace.csv" | head -1)
  echo "== $VAR=$v"; grep "solves/s" gpurun_out/r4/ab_${TAG}_$v.log
  python tools/trace_timeline.py $f rr2_gram 40 | awk '{n[$1]++; t[$1]+=$2} END {for (k in n) printf "%-40s %3d  %8.1f us\n", k, n[k], t[k]}' | sort
done
