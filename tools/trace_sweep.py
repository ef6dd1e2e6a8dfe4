#!/usr/bin/env python3
"""Per-wave timeline of one warped-sweep launch: build/libdfusion_hip_trace.so (tools/build_variant.py trace -DDF_TRACE_WG=1) stamps
every wave's start / end (s_memrealtime, 100 MHz), hardware id and alive-layer count; this writes gpurun_out/sweep_trace_<cfg>.npz and
prints the slot occupancy over time."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from dynamicfusion_amd import capi, build as B
B.LIB_PATH = os.path.join(REPO, "build", "libdfusion_hip_%s.so" % os.environ.get("DF_TRACE_LIB", "trace")); B._stale = lambda: False
from dynamicfusion_amd import Intr, TsdfVolume, WarpField, compute_dists, synth, upload_u16
name = sys.argv[1] if len(sys.argv) > 1 else "512"
rigid = len(sys.argv) > 2 and sys.argv[2] == "rigid"
cfg = synth.CONFIGS[name]; intr = Intr(*cfg.intr)
depth = upload_u16(synth.depth_frame(cfg, 0)); dists = compute_dists(depth, intr); cam = synth.camera_pose(cfg, 1)
vol = TsdfVolume(cfg.dims); vol.setSize([cfg.size] * 3); vol.setTruncDist(cfg.trunc_dist); vol.setMaxWeight(cfg.max_weight); vol.setPose(cfg.volume_pose)
pos, sigma = synth.make_nodes(cfg); dq = synth.node_transforms(cfg, 1)
wf = WarpField(k=cfg.k); wf.init(pos, sigma=sigma, transforms=dq); wf.ensure_index(vol, cfg.k)
run_once = (lambda: vol.integrate(dists, cam, intr)) if rigid else (lambda: vol.integrate_warped(dists, cam, intr, wf))
for _ in range(5): run_once()
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
path = os.path.join(REPO, "gpurun_out", "sweep_trace.bin")
os.environ["DF_TRACE_FILE"] = path
run_once()
del os.environ["DF_TRACE_FILE"]
raw = np.fromfile(path, dtype=np.uint64)
gx, gy, wpw = int(raw[0]), int(raw[1]), int(raw[2])
t = raw[4:].reshape(gy, gx, wpw, 4)
np.savez_compressed(os.path.join(REPO, "gpurun_out", "sweep_trace_%s%s.npz" % (name, "_rigid" if rigid else "")), trace=t)
os.remove(path)
start, end, alive = t[..., 0].astype(np.int64), t[..., 1].astype(np.int64), t[..., 3].astype(np.int64)
run = end > 0                                   # waves of workgroups past the plan's end never stamp
t0 = start[run].min(); start = np.where(run, start - t0, 0); end = np.where(run, end - t0, 0)
tick_us = 0.01
start, end, alive = start[run], end[run], alive[run]
print("waves %d  makespan %.1f us  sum(wave time) %.0f us  mean wave %.1f us  max wave %.1f us" % (
    start.size, end.max() * tick_us, (end - start).sum() * tick_us, (end - start).mean() * tick_us, (end - start).max() * tick_us))
T = int(end.max()) + 1
occ = np.zeros(T + 1, np.int64)
np.add.at(occ, start.ravel(), 1); np.add.at(occ, end.ravel(), -1)
occ = np.cumsum(occ)[:T]
nb = 20
for i in range(nb):
    seg = occ[i * T // nb:(i + 1) * T // nb]
    print("  %5.0f-%5.0f us: %6.0f waves resident" % (i * T // nb * tick_us, (i + 1) * T // nb * tick_us, seg.mean()))
busy = alive > 0
print("waves with work: %d, their time %.0f us; per alive layer %.2f us" % (busy.sum(), (end - start)[busy].sum() * tick_us,
      (end - start)[busy].sum() * tick_us / alive[busy].sum()))
print("workgroups launched %d, that ran %d; last wave start %.1f us" % (gx * gy, int(run.any(-1).sum()), start.max() * tick_us))
