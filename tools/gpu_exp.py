import os, sys, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from dynamicfusion_amd import Intr, TsdfVolume, WarpField, compute_dists, synth, upload_u16
from tools.gpu_probe import timeit
cfg = synth.CONFIGS["512"]; intr = Intr(*cfg.intr)
depth = upload_u16(synth.depth_frame(cfg, 0)); dists = compute_dists(depth, intr); cam = synth.camera_pose(cfg, 1)
vol = TsdfVolume(cfg.dims); vol.setSize([cfg.size]*3); vol.setTruncDist(cfg.trunc_dist); vol.setMaxWeight(cfg.max_weight); vol.setPose(cfg.volume_pose)
pos, sigma = synth.make_nodes(cfg); dq = synth.node_transforms(cfg, 1)
wf = WarpField(k=cfg.k); wf.init(pos, sigma=sigma, transforms=dq)
for pipe in ("1", "0", "1"):
    os.environ["DFUSION_ROWS_PIPE"] = pipe
    ms = timeit(lambda: vol.integrate_warped(dists, cam, intr, wf, sync=False), iters=10, warm=2)
    print("rows pipe", pipe, "%.3f ms" % ms)
