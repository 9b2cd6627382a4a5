"""Dev tool (CPU): the launches of the LAST solve in a rocprofv3 --kernel-trace csv, one line per launch (name, duration, grid).

    python tools/trace_timeline.py <..._kernel_trace.csv> [first-kernel-substring] [count]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
key = sys.argv[2] if len(sys.argv) > 2 else "rr2_gram"
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 42
idx = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
s = idx[-1]
tot = 0.0
for r in rows[s:s + cnt]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ddmpc::", "")[:40]
    tot += (en - st) / 1e3
    print("%-42s %8.1f us   grid %sx%s  wg %s  vgpr %s+%s lds %s scratch %s" % (nm, (en - st) / 1e3, r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", ""),
          r.get("Workgroup_Size_X", ""), r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""), r.get("LDS_Block_Size", ""), r.get("Scratch_Size", "")))
print("sum %.1f us" % tot)
