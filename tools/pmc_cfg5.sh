#!/bin/bash
# Cache and issue counters of the two launches of a cfg-5 solve (ddmpc_nominal_rr_kernel<1>: factors, <2>: solve on them),
# separate --pmc passes, per-dispatch means printed at the end.  Run through gpurun from the repo root:
#   bash tools/pmc_cfg5.sh
set -e
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_cfg5
mkdir -p $OUT
cd $ROOT
p() { name=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python tools/cfg5_time.py --steps 1 > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
p tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum
p sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
python - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc_cfg5"
for name in ("tcc", "sq"):
    fs = glob.glob(out + "/%s/*/*_counter_collection.csv" % name)
    if not fs: print(name, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "nominal_rr" not in k: continue
        kk = "MODE1" if "ILi1E" in k or "<1>" in k else "MODE2"
        acc[kk][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"], kk) not in seen: seen.add((r["Dispatch_Id"], kk)); cnt[kk] += 1
    for kk in sorted(acc):
        print(name, kk, "dispatches", cnt[kk], {c: "%.4g" % (v / cnt[kk]) for c, v in acc[kk].items()})
PY
