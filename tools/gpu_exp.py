import os, sys, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from dynamicfusion_amd import Intr, TsdfVolume, capi, compute_dists, synth, upload_u16
from dynamicfusion_amd.synth import aff12
from tools.gpu_probe import timeit
cfg = synth.CONFIGS["512"]; intr = Intr(*cfg.intr)
depth = upload_u16(synth.depth_frame(cfg, 0)); dists = compute_dists(depth, intr)
vol = TsdfVolume(cfg.dims); vol.setSize([cfg.size]*3); vol.setTruncDist(cfg.trunc_dist); vol.setMaxWeight(cfg.max_weight); vol.setPose(cfg.volume_pose)
for f in range(2): vol.integrate(dists, synth.camera_pose(cfg, f), intr)
buf = torch.empty((1 << 24, 4), dtype=torch.float32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream; nvox = 512**3
for g in ("12", "16", "24", "32"):
    os.environ["DFUSION_EX_BLOCKS_PER_CU"] = g
    ms = timeit(lambda: capi.check(capi.lib().dfusion_extract_cloud(vol.c_volume(), None, capi.floats(aff12(vol.getPose())), buf.data_ptr(), buf.shape[0], cnt.data_ptr(), st)), iters=20)
    print("extract fused blocks/CU", g, ": %.3f ms  %.0f GB/s" % (ms, 4 * nvox / ms / 1e6))
vol.clear()
ms = timeit(lambda: capi.check(capi.lib().dfusion_extract_cloud(vol.c_volume(), None, capi.floats(aff12(vol.getPose())), buf.data_ptr(), buf.shape[0], cnt.data_ptr(), st)), iters=20)
print("extract on EMPTY volume: %.3f ms  %.0f GB/s" % (ms, 4 * nvox / ms / 1e6))
