"""Generates tests/golden/refcu_64.npz from the reference's OWN CUDA kernels compiled for the host
(oracle/_ref/libdfref_cu.so, `make -C oracle ref_cu`; needs /root/reference).  Run from the repo root:
    python tests/golden/make_golden_refcu.py
Scene: 64^3 / 1 m volume with a rotated volume pose, 160x120 depth, 3 frames (tests/test_oracle_refcu.py::make_scene)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import oracle_lib as O                      # noqa: E402
from dynamicfusion_amd import synth          # noqa: E402
from test_oracle_refcu import bits, make_scene   # noqa: E402

assert O.have_refcu(), "build oracle/_ref/libdfref_cu.so first (make -C oracle ref_cu)"
cfg, sc = make_scene(64, rotated=True)
vol = sc.new_volume()
dists = []
for f in range(3):
    dists.append(O.refcu_compute_dists(sc.depths[f], sc.intr))
    O.refcu_integrate(dists[f], sc.ovol(vol), synth.aff12(sc.vol2cam(f)), sc.intr)
tail = (cfg.cols, cfg.rows, cfg.raycast_step_factor, cfg.gradient_delta_factor)
p, n = O.refcu_raycast_points(sc.ovol(vol), synth.aff12(sc.cam2vol(2)), sc.rinv(2), sc.intr, *tail)
d, _ = O.refcu_raycast_depth(sc.ovol(vol), synth.aff12(sc.cam2vol(2)), sc.rinv(2), sc.intr, *tail)
cloud, count = O.refcu_extract_cloud(sc.ovol(vol), synth.aff12(sc.pose), 1 << 20)
out = os.path.join(HERE, "refcu_64.npz")
np.savez_compressed(out, pose=sc.pose, dists=np.stack(dists), volume=vol, points_bits=bits(p), normals_bits=bits(n), depth=d,
                    cloud_count=np.int64(count))
print("wrote", out, os.path.getsize(out), "bytes; hits", int(np.isfinite(p[..., 0]).sum()), "cloud", count)
