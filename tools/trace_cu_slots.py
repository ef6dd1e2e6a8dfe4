#!/usr/bin/env python3
"""What the warped sweep's workgroup slots do over a launch, from tools/trace_sweep.py's per-wave stamps (gpurun_out/sweep_trace_<cfg>.npz):
per CU, how many waves are resident over the middle of the launch; how long a freed workgroup slot waits for its next workgroup; and whether
the waits come in bursts per XCD (a dispatcher that goes round the XCDs in order stalls every XCD behind a full one)."""
import sys
import numpy as np
t = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/sweep_trace_512.npz")["trace"][0]
start, end, hw = t[..., 0].astype(np.int64), t[..., 1].astype(np.int64), t[..., 2]
run = (end > 0).all(-1)
s, e, h = start[run], end[run], hw[run]
t0 = s.min(); s = (s - t0) * 0.01; e = (e - t0) * 0.01
xcc = ((h >> 32) & 0xf).astype(np.int64); hwid = (h & 0xffffffff).astype(np.int64)
simd = (hwid >> 4) & 3; cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
key = (((xcc * 8 + se) * 2 + sh) * 16 + cu)[:, 0]
mk = e.max()
lo, hi = 0.15 * mk, 0.80 * mk
ws, we = s.min(1), e.max(1)
print("workgroups that ran %d, makespan %.1f us; window %.0f-%.0f us" % (run.sum(), mk, lo, hi))
grid = np.arange(lo, hi, 1.0)
conc = []
for k in np.unique(key):
    m = key == k
    conc.append(((ws[m][None, :] <= grid[:, None]) & (we[m][None, :] > grid[:, None])).sum(1))
conc = np.array(conc)                                     # [CU, time] resident workgroups
print("CUs seen %d; resident workgroups per CU over the window: mean %.2f; share of CU-time at 6 / 5 / 4 / <4 workgroups: %.2f / %.2f / %.2f / %.2f" % (
    conc.shape[0], conc.mean(), (conc == 6).mean(), (conc == 5).mean(), (conc == 4).mean(), (conc < 4).mean()))
# refill gaps: per CU, each workgroup start in the window is matched with the earliest unmatched workgroup end before it
gaps, gap_t, gap_x = [], [], []
for k in np.unique(key):
    m = key == k
    ev = sorted([(x, 1) for x in ws[m]] + [(x, -1) for x in we[m]])
    free = []
    for tm, d in ev:
        if d == -1: free.append(tm)
        elif free and lo < tm < hi:
            g = tm - free.pop(0); gaps.append(g); gap_t.append(tm); gap_x.append(int(xcc[m][0, 0]))
gaps = np.array(gaps); gap_t = np.array(gap_t); gap_x = np.array(gap_x)
print("slot refill wait (workgroup end -> next workgroup's first stamp on that CU): n %d  median %.1f  mean %.1f  p90 %.1f  max %.1f us; share of all slot-time %.3f" % (
    gaps.size, np.median(gaps), gaps.mean(), np.percentile(gaps, 90), gaps.max(), gaps.sum() / (6.0 * conc.shape[0] * (hi - lo))))
dur = (we - ws)
print("workgroup duration: mean %.1f  p10 %.1f  p90 %.1f us" % (dur.mean(), np.percentile(dur, 10), np.percentile(dur, 90)))
# per XCD: are long waits simultaneous?
for x in range(8):
    g = gaps[gap_x == x]
    if g.size: print("  XCD %d: %4d refills, mean wait %.1f us, waits > 20 us: %d" % (x, g.size, g.mean(), (g > 20).sum()))
