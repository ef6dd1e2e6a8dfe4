import os, sys, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from dynamicfusion_amd import Intr, TsdfVolume, WarpField, compute_dists, synth, upload_u16
from tools.gpu_probe import timeit
cfg = synth.CONFIGS["512"]; intr = Intr(*cfg.intr)
depth = upload_u16(synth.depth_frame(cfg, 0)); dists = compute_dists(depth, intr); cam = synth.camera_pose(cfg, 1)
vol = TsdfVolume(cfg.dims); vol.setSize([cfg.size]*3); vol.setTruncDist(cfg.trunc_dist); vol.setMaxWeight(cfg.max_weight); vol.setPose(cfg.volume_pose)
for zc in ("16", "32", "64", "128"):
    os.environ["DFUSION_RIGID_ZCHUNK"] = zc
    for u in ("1", "2", "4"):
        os.environ["DFUSION_RIGID_BATCH"] = u
        ms = timeit(lambda: vol.integrate(dists, cam, intr, sync=False), iters=20, warm=3)
        print("rigid zchunk", zc, "batch", u, ": %.3f ms" % ms)
