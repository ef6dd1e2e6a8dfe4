#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 --pmc passes: N launches each of the hot kernels at one config."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from dynamicfusion_amd import Intr, TsdfVolume, WarpField, compute_dists, synth, upload_u16
name = sys.argv[1] if len(sys.argv) > 1 else "512"; n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mode = sys.argv[3] if len(sys.argv) > 3 else "tables"
cfg = synth.CONFIGS[name]; intr = Intr(*cfg.intr)
depth = upload_u16(synth.depth_frame(cfg, 0)); dists = compute_dists(depth, intr); cam = synth.camera_pose(cfg, 1)
vol = TsdfVolume(cfg.dims); vol.setSize([cfg.size]*3); vol.setTruncDist(cfg.trunc_dist); vol.setMaxWeight(cfg.max_weight); vol.setPose(cfg.volume_pose)
vol.setRaycastStepFactor(cfg.raycast_step_factor); vol.setGradientDeltaFactor(cfg.gradient_delta_factor)
pos, sigma = synth.make_nodes(cfg); dq = synth.node_transforms(cfg, 1)
wf = WarpField(k=cfg.k, voxel_table=(mode != "lean"), weight_table=(mode in ("tables", "bench"))); wf.init(pos, sigma=sigma, transforms=dq)
pts = torch.empty((cfg.rows, cfg.cols, 4), dtype=torch.float32, device="cuda"); nrm = torch.empty_like(pts)
swept = torch.zeros(1, dtype=torch.int64, device="cuda")
if mode == "bench":
    # round 5: the warped sweep over bench.py's OWN poses -- 3 priming frames, volume cleared, 5 warm-up, then the n timed
    # poses (8 .. 8 + n - 1 with the driver's arguments), every frame with its own depth image and node transforms -- so that
    # roofline.traffic compares like with like.  pmc_summary --last n picks the timed launches.
    wf = WarpField(k=cfg.k); wf.init(pos, sigma=sigma, transforms=synth.node_transforms(cfg, 0))
    seq = list(range(3)) + list(range(3, 8 + n))
    frames = {f: compute_dists(upload_u16(synth.depth_frame(cfg, f)), intr).clone() for f in set(seq)}
    for j, f in enumerate(seq):
        if j == 3: vol.clear()
        if f == 8: wf.debug_counters(swept)
        wf.set_transforms(torch.from_numpy(synth.node_transforms(cfg, f)).cuda())
        vol.integrate_warped(frames[f], synth.camera_pose(cfg, f), intr, wf, sync=False)
    wf.debug_counters(None)
else:
    for i in range(3 + n):        # 3 warm-up launches (on-demand tables, then the blocks' blend models, are made by the first two); pmc_summary --last n
        if i == 3: wf.debug_counters(swept)       # (the PLAN kernel counts the voxels it keeps: the sweep kernel itself is unchanged)
        vol.integrate_warped(dists, cam, intr, wf, sync=False)
    wf.debug_counters(None)
for _ in range(n):
    vol.raycast(cam, intr, pts, nrm)
for _ in range(n):
    vol.integrate(dists, cam, intr, sync=False)
for _ in range(n):
    vol.fetchCloud()
torch.cuda.synchronize()
import json
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
wl = os.path.join(REPO, "gpurun_out", "pmc_workload.json")
doc = json.load(open(wl)) if os.path.exists(wl) else {}
doc[name] = {"n_swept_per_launch": float(swept.item()) / n, "launches": n, "poses": "bench.py's timed poses 8..%d" % (7 + n) if mode == "bench" else "pose 1, static"}
json.dump(doc, open(wl, "w"))
print("done", cfg.name, mode, "swept/launch", float(swept.item()) / n)
