#!/usr/bin/env python3
"""Same-box A/B of fetchCloud (dfusion_extract_cloud) between the product library and build/libdfusion_hip_TAG.so variants:
tools/ab_extract.py CONFIG TAG [TAG ...]; checks the point SET is the product's."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from dynamicfusion_amd import capi
from dynamicfusion_amd import Intr, TsdfVolume, compute_dists, synth, upload_u16
name, tags = sys.argv[1], sys.argv[2:]
libs = {"product": capi.lib()}
for t in tags: libs[t] = capi.load(os.path.join(REPO, "build", "libdfusion_hip_%s.so" % t), strict=False)
cfg = synth.CONFIGS[name]; intr = Intr(*cfg.intr)
vol = TsdfVolume(cfg.dims); vol.setSize([cfg.size] * 3); vol.setTruncDist(cfg.trunc_dist); vol.setMaxWeight(cfg.max_weight); vol.setPose(cfg.volume_pose)
for f in range(4): vol.integrate(compute_dists(upload_u16(synth.depth_frame(cfg, f)), intr), synth.camera_pose(cfg, f), intr)
buf = torch.empty((1 << 22, 4), dtype=torch.float32, device="cuda")
key = lambda c: np.sort(np.ascontiguousarray(c.cpu().numpy().view(np.uint32)[:, :3]).view([("x", "u4"), ("y", "u4"), ("z", "u4")]).reshape(-1), order=("x", "y", "z"))
ref = None; res = {t: [] for t in libs}
for t in libs:
    capi._lib = libs[t]; c = vol.fetchCloud(buf); k = key(c)
    if ref is None: ref = k
    else: print("%-8s %d points, set %s product" % (t, len(k), "==" if np.array_equal(k, ref) else "!="))
for rnd in range(6):
    for t in libs:
        capi._lib = libs[t]
        cnt = torch.zeros(1, dtype=torch.int64, device="cuda"); st = torch.cuda.current_stream().cuda_stream
        aff = capi.floats(synth.aff12(vol.getPose()))
        run = lambda: capi.check(libs[t].dfusion_extract_cloud(vol.c_volume(), None, aff, buf.data_ptr(), buf.shape[0], cnt.data_ptr(), st))
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize(); res[t].append(e0.elapsed_time(e1) / 20)
X, Y, Z = cfg.dims
for t, v in res.items(): print("%-8s median %.4f ms = %.0f GB/s of volume scan" % (t, float(np.median(v)), 4.0 * X * Y * Z / (np.median(v) * 1e-3) / 1e9))
