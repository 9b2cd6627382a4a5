"""Minimal plotting for the batched examples: inputs and outputs of a batch of closed loops as a median line
with a 5-95 % band, instance 0 on top, setpoints dashed.  (The reference draws one controller's trajectories
with setpoint lines, utilities/visualization/data_visualization.py; animation is not reproduced.)"""
import numpy as np


def _limits(series, setpoint):
    """Axis limits from the first (reference) run, widened 3x around the setpoint: a diverging run (the scheme
    without terminal constraints) leaves the frame instead of flattening everything else."""
    ref = series[0][np.isfinite(series[0])]
    lo, hi = np.percentile(ref, [0.5, 99.5])
    lo, hi = min(lo, setpoint), max(hi, setpoint)
    span = max(hi - lo, 1e-6)
    return lo - 1.0 * span, hi + 1.0 * span


def plot_closed_loops(path, runs, u_s, y_s, t0=0, title=None):
    """runs: {label: (u_sys [B,T,m], y_sys [B,T,p])}; writes a PNG to `path`."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    first = next(iter(runs.values()))
    m, p = first[0].shape[2], first[1].shape[2]
    fig, axes = plt.subplots(2, max(m, p), figsize=(5 * max(m, p), 6), squeeze=False, sharex=True)
    colors = plt.rcParams["axes.prop_cycle"].by_key()["color"]
    for ci, (label, (u, y)) in enumerate(runs.items()):
        col = colors[ci % len(colors)]
        t = np.arange(u.shape[1]) + t0
        for row, (x, nchan) in enumerate(((u, m), (y, p))):
            for ch in range(nchan):
                ax = axes[row][ch]
                xs = x[:, :, ch]
                finite = np.where(np.isfinite(xs), xs, np.nan)
                if xs.shape[0] > 1:
                    with np.errstate(all="ignore"):
                        import warnings
                        with warnings.catch_warnings():
                            warnings.simplefilter("ignore", RuntimeWarning)      # columns where every loop has stopped
                            lo, mid, hi = np.nanpercentile(finite, [5, 50, 95], axis=0)
                    ax.fill_between(t, lo, hi, color=col, alpha=0.2, linewidth=0)
                    ax.plot(t, mid, color=col, linewidth=1.0, label=f"{label} (median, 5-95 %)")
                ax.plot(t, finite[0], color=col, linewidth=0.7, linestyle=":" if xs.shape[0] > 1 else "-",
                        label=f"{label} (instance 0)")
    for ch in range(m):
        axes[0][ch].axhline(u_s[ch], color="k", linestyle="--", linewidth=0.8)
        axes[0][ch].set_ylabel(f"u_{ch + 1}")
        axes[0][ch].set_ylim(*_limits([r[0][:, :, ch] for r in runs.values()], u_s[ch]))
    for ch in range(p):
        axes[1][ch].axhline(y_s[ch], color="k", linestyle="--", linewidth=0.8)
        axes[1][ch].set_ylabel(f"y_{ch + 1}")
        axes[1][ch].set_xlabel("time step k")
        axes[1][ch].set_ylim(*_limits([r[1][:, :, ch] for r in runs.values()], y_s[ch]))
    axes[0][0].legend(fontsize=7, loc="best")
    if title:
        fig.suptitle(title, fontsize=10)
    fig.tight_layout()
    fig.savefig(path, dpi=110)
    plt.close(fig)
