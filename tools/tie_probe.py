#!/usr/bin/env python3
"""Workload for rocprofv3 --kernel-trace: the k-NN paths that can meet exact distance ties -- index build (+ per-voxel tables), the
point queries of a frame (WarpField::warp / KNN on the ray-cast cloud), lean sweep.  Prints wall times; the per-kernel split comes
from the trace."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from dynamicfusion_amd import Intr, TsdfVolume, WarpField, compute_dists, synth, upload_u16
name = sys.argv[1] if len(sys.argv) > 1 else "512"; n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = synth.CONFIGS[name]; intr = Intr(*cfg.intr)
depth = upload_u16(synth.depth_frame(cfg, 0)); dists = compute_dists(depth, intr); cam = synth.camera_pose(cfg, 0)
vol = TsdfVolume(cfg.dims); vol.setSize([cfg.size]*3); vol.setTruncDist(cfg.trunc_dist); vol.setMaxWeight(cfg.max_weight); vol.setPose(cfg.volume_pose)
vol.setRaycastStepFactor(cfg.raycast_step_factor); vol.setGradientDeltaFactor(cfg.gradient_delta_factor)
pos, sigma = synth.make_nodes(cfg); dq = synth.node_transforms(cfg, 1)
def sync(): torch.cuda.synchronize()
wf = WarpField(k=cfg.k); 
t = time.time(); wf.init(pos, sigma=sigma, transforms=dq); sync(); print("set_nodes (incl. tie tree) %.2f ms" % ((time.time() - t) * 1e3))
for i in range(3):
    wf._index_key = None
    t = time.time(); wf.ensure_index(vol, cfg.k); sync(); print("build index + tables %.2f ms" % ((time.time() - t) * 1e3))
vol.integrate_warped(dists, cam, intr, wf)
pts = torch.empty((cfg.rows, cfg.cols, 4), dtype=torch.float32, device="cuda"); nrm = torch.empty_like(pts)
vol.raycast(cam, intr, pts, nrm); sync()
p = pts.reshape(-1, 4)[:, :3].contiguous()
work = p.clone()
for tiling in (0, cfg.cols):
    wf.set_point_tiling(tiling)
    wf.warp(work); sync()
    t = time.time()
    for _ in range(n):
        work.copy_(p); wf.warp(work)
    sync(); print("warp %d points, tiling %d: %.3f ms each (incl. copy)" % (p.shape[0], tiling, (time.time() - t) * 1e3 / n))
    t = time.time()
    for _ in range(n):
        wf.KNN(p, cfg.k)
    sync(); print("knn: %.3f ms each" % ((time.time() - t) * 1e3 / n))
lean = WarpField(k=cfg.k, voxel_table=False); lean.init(pos, sigma=sigma, transforms=dq)
vol.integrate_warped(dists, cam, intr, lean); sync()
t = time.time()
for _ in range(n):
    vol.integrate_warped(dists, cam, intr, lean, sync=False)
sync(); print("lean sweep %.3f ms" % ((time.time() - t) * 1e3 / n))
