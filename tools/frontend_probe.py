#!/usr/bin/env python3
"""HIP-event timings of the depth front-end kernels and the ICP loop (640x480, KinFuParams defaults)."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from dynamicfusion_amd import Intr, frontend, synth, upload_u16

def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

cfg = synth.CONFIGS["512"]; intr = Intr(*cfg.intr)
d0, d1 = upload_u16(synth.depth_frame(cfg, 0)), upload_u16(synth.depth_frame(cfg, 4))
print("bilateral 7x7      : %.4f ms" % timeit(lambda: frontend.depthBilateralFilter(d0, 7, 4.5, 0.04)))
f0 = frontend.depthBilateralFilter(d0, 7, 4.5, 0.04)
print("pyramid 640->320   : %.4f ms" % timeit(lambda: frontend.depthBuildPyramid(f0, 0.04)))
print("point normals      : %.4f ms" % timeit(lambda: frontend.computePointNormals(intr, f0)))
p0, n0 = frontend.computePointNormals(intr, f0)
print("resize pts/normals : %.4f ms" % timeit(lambda: frontend.resizePointsNormals(p0, n0)))
print("truncate           : %.4f ms" % timeit(lambda: frontend.depthTruncation(f0.clone(), 3.0)))
def pyr(d):
    lv = [frontend.depthBilateralFilter(d, 7, 4.5, 0.04)]
    for i in range(1, 3): lv.append(frontend.depthBuildPyramid(lv[-1], 0.04))
    pn = [frontend.computePointNormals(frontend.intr_level(intr, i), lv[i]) for i in range(3)]
    return [a for a, _ in pn], [b for _, b in pn]
v0, nn0 = pyr(d0); v1, nn1 = pyr(d1)
print("front-end (3 lvls) : %.4f ms" % timeit(lambda: pyr(d1), iters=20))
icp = frontend.ProjectiveICP()
est = np.eye(4, dtype=np.float32)
import ctypes as C
from dynamicfusion_amd import capi
from dynamicfusion_amd.synth import aff12
ws = torch.empty(27 * 1200 + 27, dtype=torch.float32, device="cuda"); sums = torch.empty(27, dtype=torch.float32, device="cuda")
d2, mc = icp.thresholds()
for lv in range(3):
    li = frontend.intr_level(intr, lv); rows, cols = nn0[lv].shape[:2]
    fn = lambda: capi.check(capi.lib().dfusion_icp_sums_points(v1[lv].data_ptr(), cols * 16, nn1[lv].data_ptr(), cols * 16, v0[lv].data_ptr(), cols * 16,
            nn0[lv].data_ptr(), cols * 16, cols, rows, capi.floats(aff12(est)), li.as_proj(), d2, mc, ws.data_ptr(), sums.data_ptr(), None,
            C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    print("icp sums level %d   : %.4f ms (2 kernels, no host sync)" % (lv, timeit(fn)))
t0 = time.time(); n = 10
for _ in range(n): ok, aff = icp.estimateTransform(intr, v1, nn1, v0, nn0)
torch.cuda.synchronize(); print("ICP 19 iterations incl. host solves: %.3f ms" % ((time.time() - t0) / n * 1e3))
