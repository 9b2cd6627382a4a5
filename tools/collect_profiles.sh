#!/bin/bash
# Collect the per-round evidence on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh <tag> [headline|cfg5|cfg5b|calib|all]        e.g. r05_final headline
# (two parts so that each fits one gpurun call)
# Writes everything under gpurun_out/<tag>/; copy the summaries into profiles/ afterwards.
# Counter passes are separate runs with --pmc only (no trace domains), as the pool requires.
set -e -o pipefail
TAG=${1:-r03_final}
PART=${2:-all}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp

if [ "$PART" = "headline" ] || [ "$PART" = "all" ]; then

timeout -k 10 300 python bench.py --steps 50 --warmup 5 > "$OUT/bench.log" 2>&1
tail -1 "$OUT/bench.log" > "$OUT/bench.json"
echo "[collect] bench done"

timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- \
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-warm > "$OUT/stats.log" 2>&1
echo "[collect] kernel stats done"

pass() {  # name, counters...
    local name=$1; shift
    timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- \
        python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-warm > "$OUT/pmc_$name.log" 2>&1
    echo "[collect] pmc $name done"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES
pass sq2 GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE

# warm step at 32,768 instances (the HBM figure of bench.py's warm_step.roofline): the same bench run WITH its warm section
passw() {  # name, counters...
    local name=$1; shift
    timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_warm_$name" -- \
        python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/pmc_warm_$name.log" 2>&1
    echo "[collect] pmc warm $name done"
}
passw fetch FETCH_SIZE
passw write WRITE_SIZE
timeout -k 10 200 python tools/cold_check.py ref > "$OUT/refine_modes.log" 2>&1

timeout -k 10 120 python tools/phase_stamps.py > "$OUT/phase_stamps.log" 2>&1
timeout -k 10 200 python tools/convex_time.py --stamps > "$OUT/convex_time.log" 2>&1
timeout -k 10 120 python tools/time_host_pipeline.py > "$OUT/host_pipeline.log" 2>&1
timeout -k 10 120 python tools/host_pipeline_order.py >> "$OUT/host_pipeline.log" 2>&1
timeout -k 10 60 tools/substep_probe > "$OUT/substep_probe.log" 2>&1
timeout -k 10 300 python tools/bench_configs.py > "$OUT/other_configs.log" 2>&1
echo "[collect] other configs done"

fi
if [ "$PART" = "cfg5" ] || [ "$PART" = "all" ]; then
# BASELINE configs[4] (nominal, m=p=8, r=608, exact data) and the robust scheme at that size: the global-workspace kernels
timeout -k 10 200 python tools/cfg5_time.py --warm > "$OUT/cfg5_time.log" 2>&1
timeout -k 10 200 python tools/cfg5_time.py --robust --warm >> "$OUT/cfg5_time.log" 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cfg5_stats" -- \
    python tools/cfg5_time.py --steps 3 > "$OUT/cfg5_stats.log" 2>&1
# counters of the phase kernels (ddmpc_rr2.hpp, ddmpc_rr2_solve.hpp): separate --pmc passes, totals per kernel over three solves
bash tools/pmc_rr2.sh "gpurun_out/$TAG/cfg5_pmc" sq sq2 tcc fetch write > "$OUT/cfg5_pmc_totals.txt" 2>&1
# the ROBUST scheme with the slack box at that size (ddmpc_rr3.hpp, round 5): per-kernel durations and counters incl. Scratch_Size
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cfg5r_stats" -- \
    python tools/cfg5_time.py --robust --steps 3 > "$OUT/cfg5r_stats.log" 2>&1
RR2_ARGS=--robust bash tools/pmc_rr2.sh "gpurun_out/$TAG/cfg5r_pmc" sq fetch write > "$OUT/cfg5r_pmc_totals.txt" 2>&1
timeout -k 10 200 python tools/rr3_schedule.py > "$OUT/cfg5r_schedule.log" 2>&1
# the affine-law step (rr2_gain_step_kernel): bytes per launch
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d "$OUT/cfg5_law_$C" -- python tools/cfg5_time.py --warm --steps 2 > "$OUT/cfg5_law_$C.log" 2>&1
done
python - "$OUT" > "$OUT/cfg5_law_pmc.txt" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(out + "/cfg5_law_%s/*/*_counter_collection.csv" % c)
    rows = [r for r in csv.DictReader(open(f[0])) if "rr2_gain_step_kernel" in r["Kernel_Name"]] if f else []
    v = [float(r["Counter_Value"]) for r in rows]
    if v: print("rr2_gain_step_kernel %s: %d launches, mean %.1f KB per launch%s" % (c, len(v), sum(v) / len(v), " (x2 on gfx950 = %.1f MB)" % (2 * sum(v) / len(v) / 1e3) if c == "FETCH_SIZE" else ""))
PY
echo "[collect] cfg5 timing / counters done"
fi
if [ "$PART" = "cfg5b" ] || [ "$PART" = "all" ]; then
timeout -k 10 300 python tools/rr2_check.py --time > "$OUT/rr2_check.log" 2>&1
timeout -k 10 300 python tools/gram_modes_time.py > "$OUT/gram_modes.log" 2>&1
timeout -k 10 200 python tools/cfg5_two_halves.py > "$OUT/cfg5_two_halves.log" 2>&1
timeout -k 10 600 python tools/nominal_fuzz.py --cases 96 --no-svd --large-only > "$OUT/nominal_fuzz.log" 2>&1
timeout -k 10 400 python tools/config5_check.py --check 512 > "$OUT/cfg5_parity.log" 2>&1
timeout -k 10 600 python tools/config5_check.py --robust --check 512 > "$OUT/cfg5size_robust_parity.log" 2>&1
timeout -k 10 400 python tools/large_fuzz.py --cases 12 > "$OUT/large_kernel_fuzz.log" 2>&1
timeout -k 10 120 python tools/pivot_gap_study.py 6 > "$OUT/pivot_gap_study.log" 2>&1
# round 5, second half: trajectories beyond the LDS, the two structured-Gram launches side by side, all of configs[2] with the slack box
timeout -k 10 200 python tools/long_data_time.py > "$OUT/long_data_time.log" 2>&1
bash tools/gram_launches.sh "gpurun_out/$TAG/gram_launches" > "$OUT/gram_launches.log" 2>&1
timeout -k 10 600 python tools/config3_full_parity.py --slack convex > "$OUT/cfg3_convex_full_parity.log" 2>&1
timeout -k 10 900 python tools/small_fuzz.py --cases 96 --refine auto > "$OUT/small_fuzz_auto.log" 2>&1
timeout -k 10 300 python tools/dense_nominal_time.py > "$OUT/dense_nominal_time.log" 2>&1
echo "[collect] cfg5b done"
fi
if [ "$PART" = "calib" ] || [ "$PART" = "all" ]; then
# the AUTO refinement trigger on the benchmark batch and on the 96-case random-plant sweep (bound q, exact residual, off / auto / always)
timeout -k 10 900 python tools/refine_calib.py 96 > "$OUT/refine_calib.log" 2>&1
fi
echo "[collect] $PART done"
