"""Copy the summaries produced by tools/collect_profiles.sh from gpurun_out/<tag>/ into profiles/
(tracked) and print the per-launch numbers that profiles/README.md and bench.py quote.

    python tools/install_profiles.py r01_final
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03_final"
src = os.path.join("gpurun_out", tag)
dst = "profiles"
KERNEL = "ddmpc_cold_solve_kernel2"         # bench.py looks traffic up under this name
MAIN = "ddmpc_cold_solve_kernel2<9, 4, false, false>"   # the plain variant (REF = false, CVX = false): the dominant kernel of the headline run
REFP = "ddmpc_cold_solve_kernel2<9, 4, true, false>"    # the filtered refinement pass of DDMPC_REFINE_AUTO (usually finds nothing to do)


def one(pattern):
    files = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)
    if not files:
        raise SystemExit("missing " + pattern)
    return files[-1]


shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, tag + "_bench.json"))
shutil.copy(one("stats/*/*_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
shutil.copy(os.path.join(src, "phase_stamps.log"), os.path.join(dst, tag + "_phase_stamps.log"))
shutil.copy(os.path.join(src, "other_configs.log"), os.path.join(dst, tag + "_other_configs.log"))

for name in ("cfg5_time.log", "cfg5_phase_stamps.log", "cfg5_parity.log", "cfg5size_robust_parity.log", "large_kernel_fuzz.log",
             "refine_calib.log", "refine_modes.log", "convex_time.log", "host_pipeline.log", "substep_probe.log",
             "cfg5r_pmc_totals.txt", "cfg5r_schedule.log", "pivot_gap_study.log", "long_data_time.log", "gram_launches.log",
             "cfg3_convex_full_parity.log", "small_fuzz_auto.log", "dense_nominal_time.log"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, tag.replace("_final", "") + "_" + name))
if glob.glob(os.path.join(src, "cfg5r_stats/*/*_kernel_stats.csv")):
    shutil.copy(one("cfg5r_stats/*/*_kernel_stats.csv"), os.path.join(dst, tag.replace("_final", "") + "_cfg5size_robust_kernel_stats.csv"))
    for row in csv.DictReader(open(one("cfg5r_stats/*/*_kernel_stats.csv"))):
        if "rr" in row["Name"] or "large_solve" in row["Name"]:
            print("cfg5-size robust rocprof: %-60s %5s calls, avg %9.2f us" % (row["Name"][:60], row["Calls"], float(row["AverageNs"]) / 1e3))
if glob.glob(os.path.join(src, "cfg5_stats/*/*_kernel_stats.csv")):
    shutil.copy(one("cfg5_stats/*/*_kernel_stats.csv"), os.path.join(dst, tag.replace("_final", "") + "_cfg5_kernel_stats.csv"))
    for name in ("cfg5_pmc_totals.txt", "nominal_fuzz.log", "cfg5_law_pmc.txt", "rr2_check.log", "gram_modes.log", "cfg5_two_halves.log"):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(dst, tag.replace("_final", "") + "_" + name))
    for row in csv.DictReader(open(one("cfg5_stats/*/*_kernel_stats.csv"))):
        if "rr" in row["Name"]:
            print("cfg5 rocprof: %-60s %5s calls, avg %9.2f us" % (row["Name"][:60], row["Calls"], float(row["AverageNs"]) / 1e3))

b = json.load(open(os.path.join(src, "bench.json")))
print("bench: %.3e solves/s, %.4f ms/step, kernel %.4f ms, %.2f TFLOP/s = %.1f %% of peak" % (
    b["value"], b["ms_per_step"], b["roofline"]["kernel_ms"], b["roofline"]["achieved"], 100 * b["roofline"]["frac"]))
if "warm_step" in b:
    w = b["warm_step"]
    print("warm : %.3e steps/s, %.1f GB/s = %.1f %% of HBM peak, prepare %.2f ms" % (
        w["value"], w["roofline"]["achieved"], 100 * w["roofline"]["frac"], w["prepare_ms"]))
print("cpu  : %.1f solves/s (%d threads); parity u %.2e cost %.2e" % (
    b["cpu_baseline"]["value"], b["cpu_baseline"]["cores"], b["parity"]["max_rel_err_u"], b["parity"]["max_rel_err_cost"]))
for row in csv.DictReader(open(one("stats/*/*_kernel_stats.csv"))):
    if KERNEL in row["Name"]:
        print("rocprof: %-44s %s calls, avg %.1f us" % (row["Name"][12:56], row["Calls"], float(row["AverageNs"]) / 1e3))

traffic = {}
for name in ("fetch", "write", "sq", "sq2"):
    f = one("pmc_%s/*/*_counter_collection.csv" % name)
    rows = [r for r in csv.DictReader(open(f)) if MAIN in r["Kernel_Name"]]
    with open(os.path.join(dst, "%s_pmc_%s.csv" % (tag, name)), "w", newline="") as out:
        wr = csv.DictWriter(out, fieldnames=list(rows[0].keys()))
        wr.writeheader()
        wr.writerows(rows)
    acc = collections.defaultdict(list)
    for r in rows:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    meta = rows[0]
    print("pmc %-5s VGPR %s scratch %s B/lane LDS %s" % (name, meta["VGPR_Count"], meta["Scratch_Size"], meta["LDS_Block_Size"]))
    for k, v in acc.items():
        print("    %-28s n=%d mean %.6g" % (k, len(v), sum(v) / len(v)))
        if k in ("FETCH_SIZE", "WRITE_SIZE"):
            traffic[k] = sum(v) / len(v)

# HBM bytes per launch for bench.py's roofline.traffic, keyed by the hash of the kernel sources the run was built from
# (FETCH_SIZE x2: gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md "HBM"; both counters are in KB)
if "FETCH_SIZE" in traffic and "WRITE_SIZE" in traffic:
    entry = dict(kernel=KERNEL, code_hash=b["config"]["kernel_source_hash"], batch=b["config"]["global_batch"], slack="none",
                 fetch_kb=traffic["FETCH_SIZE"], write_kb=traffic["WRITE_SIZE"],
                 traffic_bytes=int((2 * traffic["FETCH_SIZE"] + traffic["WRITE_SIZE"]) * 1024), source=tag + "_pmc_fetch.csv / _pmc_write.csv")
    path = os.path.join(dst, "traffic.json")
    tab = json.load(open(path)) if os.path.exists(path) else {"entries": []}
    tab["entries"] = [e for e in tab["entries"] if not (e["kernel"] == KERNEL and e["batch"] == entry["batch"] and e.get("slack") == "none")] + [entry]
    # warm step at 32,768 instances (launches with that many workgroups only: the 4096-instance ones are cache-resident)
    wt = {}
    for name in ("fetch", "write"):
        files = sorted(glob.glob(os.path.join(src, "pmc_warm_%s/*/*_counter_collection.csv" % name)), key=os.path.getmtime)
        if not files:
            continue
        rows = [r for r in csv.DictReader(open(files[-1])) if "ddmpc_warm_step_kernel" in r["Kernel_Name"]
                and int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1) == 32768]
        if rows:
            with open(os.path.join(dst, "%s_warm_step_pmc_%s.csv" % (tag.replace("_final", ""), name)), "w", newline="") as out:
                wr = csv.DictWriter(out, fieldnames=list(rows[0].keys()))
                wr.writeheader()
                wr.writerows(rows)
            vals = [float(r["Counter_Value"]) for r in rows]
            wt[rows[0]["Counter_Name"]] = sum(vals) / len(vals)
    if "FETCH_SIZE" in wt and "WRITE_SIZE" in wt:
        wentry = dict(kernel="ddmpc_warm_step_kernel", code_hash=b["config"]["kernel_source_hash"], batch=32768, slack="none",
                      fetch_kb=wt["FETCH_SIZE"], write_kb=wt["WRITE_SIZE"],
                      traffic_bytes=int((2 * wt["FETCH_SIZE"] + wt["WRITE_SIZE"]) * 1024), source=tag.replace("_final", "") + "_warm_step_pmc_*.csv")
        tab["entries"] = [e for e in tab["entries"] if e["kernel"] != "ddmpc_warm_step_kernel"] + [wentry]
        print("traffic.json: ddmpc_warm_step_kernel (32,768 instances) -> %.1f MB per launch" % (wentry["traffic_bytes"] / 1e6))
    json.dump(tab, open(path, "w"), indent=1)
    print("traffic.json: %s -> %.1f MB per launch (hash %s)" % (KERNEL, entry["traffic_bytes"] / 1e6, entry["code_hash"]))
