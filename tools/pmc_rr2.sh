#!/bin/bash
# Counters of the phase kernels of a cfg-5 solve (ddmpc_rr2.hpp, ddmpc_rr2_solve.hpp), separate --pmc passes, per-kernel totals
# over the dispatches of three solves at the end.     [RR2_ARGS=--robust] bash tools/pmc_rr2.sh <outdir> [passes...]
# (RR2_ARGS=--robust: the ROBUST scheme with the slack box at that size, ddmpc_rr3.hpp)
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/${1:-gpurun_out/r4/pmc_rr2}
shift
PASSES=${@:-"sq sq2 tcc fetch write"}
mkdir -p $OUT
cd $ROOT
p() { name=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python tools/cfg5_time.py $RR2_ARGS --steps 1 > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
for P in $PASSES; do
  case $P in
    sq) p sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_WAVES ;;
    sq2) p sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU ;;
    tcc) p tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum ;;
    fetch) p fetch FETCH_SIZE ;;
    write) p write WRITE_SIZE ;;
  esac
done
python - "$OUT" $PASSES <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for name in sys.argv[2:]:
    fs = glob.glob(out + "/%s/*/*_counter_collection.csv" % name)
    if not fs: print(name, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ddmpc::", "")
        if "rr" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
        acc[k]["_scratch"] = max(acc[k]["_scratch"], float(r.get("Scratch_Size", 0) or 0))
        acc[k]["_vgpr"] = float(r.get("VGPR_Count", 0) or 0) + float(r.get("Accum_VGPR_Count", 0) or 0)
    for k in sorted(acc):
        n = len(cnt[k])
        print(name, "%-32s dispatches %3d  totals: " % (k, n) + "  ".join("%s=%.4g" % (c, v) for c, v in sorted(acc[k].items())))
    if name in ("fetch", "write"):      # per solve (three solves in the run), in MB; FETCH_SIZE doubled (gfx950: 128-byte requests tallied at 64 B)
        key, mul = ("FETCH_SIZE", 2.0) if name == "fetch" else ("WRITE_SIZE", 1.0)
        tot = 0.0
        for k in sorted(acc):
            mb = mul * acc[k][key] / 3.0 / 1e3
            tot += mb
            print("%s-MB-per-solve %-32s %9.1f" % (name, k, mb))
        print("%s-MB-per-solve %-32s %9.1f" % (name, "TOTAL", tot))
PY
