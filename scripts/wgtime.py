"""Development: per-workgroup start / end clocks of ONE streaming-kernel launch, and where the launch's time goes when it is held
against the rate the same kernel reaches mid-launch (needs the -DGV_WGTIME build of the library, path in GV_DBG_LIB:
`bash scripts/build_variant.sh wgtime -DGV_WGTIME`).

  python scripts/wgtime.py N M which(atx|ax|atx2|ax2) [layout: 0 = the library's choice (default), 1 two stripe sets, 2 tile]
                           [decomp override: ks=K,taper=T,geo=G,prio=P,occ=2 | cells=C,whole=W,prio=P]

Prints: the decomposition in use, the HIP-event duration of the launch beside the in-kernel span (first start .. last end), the
start / end skew per XCD, the resident-workgroup count over time, and the accounting
    span = bytes / mid_rate + ramp_loss + drain_loss + round_loss
where mid_rate is the aggregate rate of the workgroups that ran wholly inside the full-occupancy window.
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gvamp_amd import capi

capi.LIB_PATH = os.environ["GV_DBG_LIB"]
N, M, which = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
layout = int(sys.argv[4]) if len(sys.argv) > 4 else 0
override = sys.argv[5] if len(sys.argv) > 5 else ""
CELL = 4 * 4096          # bytes of genotypes per cell (4 row groups x one 4 KiB super-block / supertile)

with capi.Shard(N, M) as sh:
    if layout:
        sh.set_layout(False, layout)
        sh.set_kernel_mode(1)
    sh.synth_bed(1234, 5000)
    sh.compute_markers_statistics()
    rng = np.random.default_rng(0)
    x, x2, p, p2, w, w2 = sh.vecM(rng.standard_normal(M)), sh.vecM(rng.standard_normal(M)), sh.vecN(), sh.vecN(), sh.vecM(), sh.vecM()
    sh.ax_dev(x, p); sh.ax_dev(x2, p2)
    f = {"atx": lambda: sh.atx_dev(p, w), "ax": lambda: sh.ax_dev(x, p), "atx2": lambda: sh.atx2_dev(p, p2, w, w2),
         "ax2": lambda: sh.ax2_dev(x, x2, p, p2)}[which]
    for _ in range(3):
        f()                                      # (the first call picks the decomposition: builtin table, cache or autotune)
    if override:
        kw = {k: float(v) for k, v in (t.split("=") for t in override.split(","))}
        if "cells" in kw:
            sh.set_decomp(which, balanced_cells=int(kw["cells"]), whole_quads=int(kw.get("whole", 0)), prio=int(kw.get("prio", 0)),
                          wgs_per_cu=int(kw.get("occ", 0)))
        else:
            sh.set_decomp(which, ks=int(kw.get("ks", 1)), taper=kw.get("taper", 0.0), geo=kw.get("geo", 0.0), prio=int(kw.get("prio", 0)),
                          wgs_per_cu=int(kw.get("occ", 0)), xcd_skew=kw.get("skew", 0.0))
        for _ in range(2):
            f()
    sh.synchronize()
    print("N=%d M=%d %s layout %d  decomposition %s  (%s)" % (N, M, which, sh.get_layout(), sh.decomp()[which], sh.tune_info()[1]))
    L = capi.load()
    L.gv_debug_wgtime.argtypes = [C.c_void_p, C.c_int]
    # HIP events (gv_set_timing(2)) around the streaming kernel of the very launch whose workgroups are clocked
    isax = which.startswith("ax")
    sh.set_timing(2)
    assert L.gv_debug_wgtime(None, 0) == 0          # reset, then ONE launch
    sh.counters(reset=True)
    f()
    sh.synchronize()
    c = sh.counters(reset=True)
    sh.set_timing(0)
    ev_us = 1e3 * (c["ms_ax_kernel"] / max(1, c["n_ax_kernel"]) if isax else c["ms_atx_kernel"] / max(1, c["n_atx_kernel"]))
    n = 16384
    buf = (C.c_ulonglong * (4 * n))()
    assert L.gv_debug_wgtime(buf, n) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).astype(np.int64)
    a = a[a[:, 1] > 0]
    t0 = a[:, 0].min()
    st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0          # wall_clock64: 100 MHz -> us
    dur = en - st
    xcc, cells = a[:, 2] & 0xf, a[:, 2] >> 8
    blk = np.nonzero(np.frombuffer(buf, dtype=np.uint64).reshape(n, 4)[:, 1] > 0)[0]
    tab = {int(r): sorted(set(int(v) for v in xcc[blk % 8 == r])) for r in range(8)}
    print("block index mod 8 -> hardware XCC id (HW_REG_XCC_ID): %s" % ", ".join("%d -> %s" % (r, "/".join(map(str, tab[r]))) for r in range(8)))
    wbytes = cells * CELL
    total = float(wbytes.sum())
    span = float(en.max())
    print("workgroups %d, cells %d = %.3f GB of genotypes; HIP-event duration %.1f us, in-kernel span %.1f us (launch + dispatch + "
          "retire outside the span: %.1f us)" % (len(a), cells.sum(), total / 1e9, ev_us, span, ev_us - span))
    print("whole-launch rate: %.0f GB/s over the events, %.0f GB/s over the span" % (total / ev_us / 1e3, total / span / 1e3))
    print("start  us: p0 %.1f p50 %.1f p90 %.1f p100 %.1f" % tuple(np.percentile(st, [0, 50, 90, 100])))
    print("end    us: p0 %.1f p10 %.1f p50 %.1f p90 %.1f p100 %.1f" % tuple(np.percentile(en, [0, 10, 50, 90, 100])))
    print("dur    us: p0 %.1f p10 %.1f p50 %.1f p90 %.1f p100 %.1f" % tuple(np.percentile(dur, [0, 10, 50, 90, 100])))
    # ---- occupancy: resident workgroups over time; the full-occupancy window [t_full, t_drain]
    ev = np.concatenate([np.stack([st, np.ones_like(st)], 1), np.stack([en, -np.ones_like(en)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    occ = np.cumsum(ev[:, 1])
    peak = int(occ.max())
    t_full = float(ev[np.argmax(occ >= 0.98 * peak), 0])
    # drain: from the last moment the chip held >= 98 % of its peak
    last_full = np.nonzero(occ >= 0.98 * peak)[0][-1]
    t_drain = float(ev[min(last_full + 1, len(ev) - 1), 0])
    print("resident workgroups: peak %d; >= 98 %% of it from %.1f us to %.1f us (ramp %.1f us, drain %.1f us)" %
          (peak, t_full, t_drain, t_full, span - t_drain))
    step = max(5.0, span / 80)
    ts = np.arange(0, span, step)
    print("resident wgs every %.0f us:" % step, " ".join(str(int(((st <= t) & (en > t)).sum())) for t in ts))
    # ---- rate over time under the assumption that a workgroup streams evenly between its stamps
    rate = wbytes / np.maximum(dur, 1e-3)                            # bytes / us
    def streamed(t0_, t1_):
        ov = np.clip(np.minimum(en, t1_) - np.maximum(st, t0_), 0, None)
        return float((rate * ov).sum())
    mid_rate = streamed(t_full, t_drain) / max(t_drain - t_full, 1e-3)     # bytes / us inside the full window
    print("rate inside the full-occupancy window: %.0f GB/s; in its thirds: %s" % (
        mid_rate / 1e3, " ".join("%.0f" % (streamed(t_full + k * (t_drain - t_full) / 3, t_full + (k + 1) * (t_drain - t_full) / 3) /
                                           ((t_drain - t_full) / 3) / 1e3) for k in range(3))))
    ideal = total / mid_rate
    ramp_loss = t_full - streamed(0, t_full) / mid_rate
    drain_loss = (span - t_drain) - streamed(t_drain, span) / mid_rate
    print("accounting (us): bytes / mid-launch rate %.1f + ramp loss %.1f + drain loss %.1f + outside the span %.1f = %.1f (events %.1f)" %
          (ideal, ramp_loss, drain_loss, ev_us - span, ideal + ramp_loss + drain_loss + ev_us - span, ev_us))
    print("as a fraction of the launch: ramp %.1f %%, drain %.1f %%, launch / dispatch / retire %.1f %%" %
          (100 * ramp_loss / ev_us, 100 * drain_loss / ev_us, 100 * (ev_us - span) / ev_us))
    # ---- rounds: workgroups by dispatch order (start time): how the later rounds' starts spread
    order = np.argsort(st)
    r1 = order[:peak]
    print("round 1 (the first %d workgroups): start p50 %.1f p100 %.1f us, dur p10 %.1f p50 %.1f p90 %.1f" % (
        peak, np.median(st[r1]), st[r1].max(), *np.percentile(dur[r1], [10, 50, 90])))
    if len(order) > peak:
        rest = order[peak:]
        print("later workgroups: %d, dur p10 %.1f p50 %.1f p90 %.1f; bytes/us per workgroup p50: round 1 %.1f, later %.1f" % (
            len(rest), *np.percentile(dur[rest], [10, 50, 90]), np.median(rate[r1]), np.median(rate[rest])))
    for k in range(8):
        m = xcc == k
        if m.any():
            print("  xcc %d: %4d wgs %6.3f GB  first start %.1f  last start %.1f  end p50 %.1f  last end %.1f (%.1f before the launch's)  "
                  "dur p50 %.1f" % (k, m.sum(), wbytes[m].sum() / 1e9, st[m].min(), st[m].max(), np.median(en[m]), en[m].max(),
                                    span - en[m].max(), np.median(dur[m])))
