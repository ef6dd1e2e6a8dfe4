#!/usr/bin/env python3
"""The reference's FullScan6 extract_kernel + extract_normals_kernel (tsdf_volume.cu:511-795, compiled for the host: a block's threads run as
fibers -- 104 s at this size) against dfusion_extract_cloud / _normals on the WHOLE fused 512^3 volume, with no restatement in between: same
count, same point set, same normals.  Until round 5 this was a test behind DFUSION_SLOW_TESTS=1 (so the driver's run reported one skip);
the suite now runs the same comparison on the 512 x 512 x 64 slab that holds the most surface
(tests/test_gpu_refcu.py::test_fetch_cloud_equals_reference_fullscan6_on_a_512x512x64_slab, 13 s) and the whole-volume run lives here.
    python tools/fullscan6_512.py        (log of round 4's run: profiles/r04_fullscan6_512.txt: 234 074 points, identical)"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import oracle_lib as O
from dynamicfusion_amd import synth
from scene import Scene
from test_gpu_refcu import bits, gpu_frames
F32 = np.float32
cfg = synth.CONFIGS["512"]
sc = Scene(cfg, n_frames=2, with_nodes=False)
vol, intr = gpu_frames(sc, cfg, 2)
ref = vol.download()
cloud = vol.fetchCloud()
normals = vol.fetchNormals(cloud)
torch.cuda.synchronize()
rc, count = O.refcu_extract_cloud(sc.ovol(ref), synth.aff12(sc.pose), 1 << 23)
assert count == cloud.shape[0] and count > 100000
key = lambda a: np.sort(np.ascontiguousarray(bits(a)[:, :3]).view([("x", "u4"), ("y", "u4"), ("z", "u4")]).reshape(-1), order=("x", "y", "z"))
c = cloud.cpu().numpy()
assert np.array_equal(key(c), key(rc[:count]))
rinv = np.linalg.inv(sc.pose[:3, :3].astype(np.float64)).astype(F32)
rn = O.refcu_extract_normals(sc.ovol(ref), synth.aff12(sc.pose), rinv, c[::16], cfg.gradient_delta_factor)
assert np.array_equal(bits(normals.cpu().numpy()[::16])[:, :3], bits(rn)[:, :3])
print("FullScan6 at 512^3: %d points, point set and normals identical with the reference's kernels" % count)
