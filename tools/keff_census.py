#!/usr/bin/env python3
"""VERDICT r4 #1(b), measured on the CPU before anything is built: how many 8 x 8 x 8 blocks of the headline workload could drop the far
half of their weight record?  A block qualifies for k_eff <= j when, for EVERY voxel of the block, the weights of neighbours j+1..k are
exactly 0.f, or so small that their product with any node component (|rot| <= 1, |node_t| <= t_max) is below half the smallest
subnormal -- only then is leaving them out result-identical whatever the frame's transforms are.
    python tools/keff_census.py [512] [stride]
Weights as the library makes them: (float)exp((double)(-d2 / (2 * sigma * sigma))) with d2 in f32 (warp_field.cpp:238-241).  Blocks are
sampled on a lattice (every `stride`-th block per axis, all 512 voxels of each); 'alive' = within the frame-0 frustum, a proxy for the
verdict pass's alive set (which is a subset of it)."""
import os, sys
import numpy as np
from scipy.spatial import cKDTree
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from dynamicfusion_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else "512"
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg = synth.CONFIGS[name]; k = cfg.k
pos, sigma = synth.make_nodes(cfg)
tree = cKDTree(pos.astype(np.float64))
D = cfg.dims[0]; vs = np.float32(cfg.size / D)
pose = np.asarray(cfg.volume_pose, np.float32); cam = np.asarray(synth.camera_pose(cfg, 0), np.float32)
w2c = np.linalg.inv(cam.astype(np.float64))
fx, fy, cx, cy = cfg.intr
t_max = 0.5                                                    # metres: generous bound on |node_t| (the benchmark's are <= 0.4)
tiny = np.float32(2.0 ** -150) / np.float32(max(1.0, t_max))   # w * c < 2^-150 for every |c| <= max(1, t_max): the product rounds to 0
nb = D // 8
bi = np.arange(0, nb, stride)
stats = {"blocks": 0, "alive": 0}
hist_block = np.zeros(k + 1, np.int64); hist_block_alive = np.zeros(k + 1, np.int64)
vox_hist = np.zeros(k + 1, np.int64); w_last = []
o = np.arange(8, dtype=np.float32)
for bz in bi:
    for by in bi:
        xs = (bi[:, None] * 8 + o[None, :]).reshape(-1)                       # all sampled blocks of the row at once
        X, Y, Z = np.meshgrid(xs, by * 8 + o, bz * 8 + o, indexing="ij")
        p = np.stack([X * vs, Y * vs, Z * vs], -1).reshape(-1, 3).astype(np.float32)
        q = (p.astype(np.float64) @ pose[:3, :3].T.astype(np.float64) + pose[:3, 3]).astype(np.float32)   # canonical position
        d, idx = tree.query(q.astype(np.float64), k=k)
        d2 = (d * d).astype(np.float32)
        sg = sigma[idx].astype(np.float32)
        w = np.exp((-d2 / (np.float32(2) * sg * sg)).astype(np.float64)).astype(np.float32)
        dead = w <= tiny                                                       # neighbour contributes nothing, whatever the transforms
        keff = k - np.cumprod(dead[:, ::-1], axis=1).sum(1)                    # number of leading neighbours that must be kept
        vox_hist += np.bincount(keff, minlength=k + 1)
        keff_b = keff.reshape(len(bi), 8, 8, 8).max(axis=(1, 2, 3))            # per block: the largest k_eff of its voxels
        c = q.reshape(len(bi), 512, 3).mean(1).astype(np.float64) @ w2c[:3, :3].T + w2c[:3, 3]
        u = fx * c[:, 0] / c[:, 2] + cx; v = fy * c[:, 1] / c[:, 2] + cy
        alive = (c[:, 2] > 0.3) & (u > -40) & (u < cfg.cols + 40) & (v > -40) & (v < cfg.rows + 40)
        hist_block += np.bincount(keff_b, minlength=k + 1); hist_block_alive += np.bincount(keff_b[alive], minlength=k + 1)
        stats["blocks"] += len(bi); stats["alive"] += int(alive.sum())
        w_last.append(w[:, -1].reshape(len(bi), 512)[alive].reshape(-1))
w_last = np.concatenate(w_last) if w_last else np.zeros(0, np.float32)
print("config %s: %d nodes, k = %d, sigma %.4f .. %.4f m (2 x mean node spacing), voxel %.4f m" % (name, len(pos), k, sigma.min(), sigma.max(), vs))
print("sampled %d blocks (every %d-th per axis), %d of them inside the frame-0 frustum" % (stats["blocks"], stride, stats["alive"]))
print("blocks by k_eff (largest over the block's 512 voxels), all / in frustum:")
for j in range(k + 1):
    print("  k_eff = %d: %6d (%.2f %%) / %6d (%.2f %%)" % (j, hist_block[j], 100.0 * hist_block[j] / max(1, stats["blocks"]),
                                                        hist_block_alive[j], 100.0 * hist_block_alive[j] / max(1, stats["alive"])))
q4 = hist_block_alive[1:5].sum()
print("in-frustum blocks that could read ONE weight plane (1 <= k_eff <= 4): %d = %.2f %%;  k_eff = 0 (all weights vanish: already skipped by the zero-weight test): %.2f %%"
      % (q4, 100.0 * q4 / max(1, stats["alive"]), 100.0 * hist_block_alive[0] / max(1, stats["alive"])))
if len(w_last):
    pr = np.percentile(w_last.astype(np.float64), [1, 10, 50, 90])
    print("weight of the k-th neighbour over in-frustum voxels: p1 %.3g, p10 %.3g, median %.3g, p90 %.3g" % tuple(pr))
