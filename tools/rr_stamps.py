"""Dev tool: per-phase breakdown of ddmpc_nominal_rr_kernel at the cfg-5 size (in-kernel s_memrealtime stamps, median over
the 512 instances of the batch; the phases' proportions are what matters, the clock's unit is nominal).  Reuses the
problem set-up of tools/config5_check.py."""
import sys; sys.path.insert(0,".")
exec(open("tools/config5_check.py").read().split("eng.set_data")[0])
eng.set_data(d["u_d"], d["y_d"])
eng.solve(up, yp)
eng.debug_stamps(True)
eng.solve(up, yp)
st = eng.debug_stamps(False, fetch=True).astype(np.int64)[:, :8]
names = ["gram", "cholG", "fwd+z0", "T form", "cholT", "solve", "out"]
for i, nm in enumerate(names):
    dtk = (st[:, i + 1] - st[:, i]) / 100.0   # 100 MHz -> us
    print("%-8s median %9.1f us" % (nm, np.median(dtk)))
print("total   median %9.1f us" % np.median((st[:, 7] - st[:, 0]) / 100.0))
print(st[:2])
