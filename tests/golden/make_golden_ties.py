#!/usr/bin/env python3
"""Generates tests/golden/knn_ties.npz from the REFERENCE'S OWN nanoflann (oracle/_ref/libdfref.so): node / query sets
in which exact distance ties are the rule, so the answer depends on nanoflann's tree order (which of two equidistant nodes is
found first, nanoflann.hpp:110-131,1200-1254).  Run in the build container (needs /root/reference):
    python tests/golden/make_golden_ties.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as O  # noqa: E402
from dynamicfusion_amd import synth  # noqa: E402

F32 = np.float32


def tie_sets():
    """(name, pos [M,3], queries [N,3]) -- deterministic."""
    rng = np.random.RandomState(424242)
    g = np.stack(np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij"), -1).reshape(-1, 3).astype(F32) * F32(0.25)
    pos = g[rng.permutation(len(g))] + np.array([-0.875, -0.875, 0.125], F32)
    q = np.stack(np.meshgrid(np.arange(-2, 18), np.arange(-2, 18), np.arange(-2, 18), indexing="ij"), -1).reshape(-1, 3).astype(F32) * F32(0.125)
    q = q + np.array([-0.875, -0.875, 0.125], F32)
    yield "grid", pos, q
    # the bench's construction at test size: nodes sampled from pixel-grid hits of the analytic scene, queries = all the hits
    cfg = synth.Config(64, 1.0, cols=160, rows=120, nodes=200, k=8)
    npos, _ = synth.make_nodes(cfg)
    t, pts = synth._hit_points(cfg, synth.camera_pose(cfg, 0))
    qq = pts.reshape(-1, 3).astype(F32)
    yield "surface", npos, qq[np.isfinite(qq).all(1)]
    # duplicated nodes (distance-zero ties between nodes)
    base = rng.uniform(-0.5, 0.5, (60, 3)).astype(F32) + np.array([0, 0, 1.0], F32)
    yield "dups", np.repeat(base, 3, 0)[rng.permutation(180)], (rng.uniform(-0.6, 0.6, (3000, 3)) + np.array([0, 0, 1.0])).astype(F32)


def main():
    assert os.path.isdir("/root/reference/kfusion/src/utils"), "needs the reference checkout"
    out = {}
    for name, pos, q in tie_sets():
        out[name + "_pos"] = pos
        out[name + "_q"] = q
        for k in (4, 8):
            idx, d2 = O.knn(pos, q, k, use_ref=True)
            bi, _ = O.knn(pos, q, k, brute=True)
            out["%s_idx%d" % (name, k)] = idx.astype(np.uint16)
            out["%s_d2_%d" % (name, k)] = d2
            print(name, "k", k, "queries", len(q), "rows where index-order ties differ from nanoflann:", int((bi != idx).any(1).sum()))
    # DQB-warped surface points with transforms (what WarpField::warp returns on a tie-heavy set)
    rng = np.random.RandomState(7)
    pos = out["surface_pos"]
    dq = synth.dq_from_twist(rng.uniform(-0.05, 0.05, (len(pos), 3)).astype(F32), rng.uniform(-0.01, 0.01, (len(pos), 3)).astype(F32))
    sigma = np.full(len(pos), 0.12, F32)
    wp, _ = O.warp_points(pos, dq, sigma, out["surface_q"], None, 8, use_ref=True)
    out.update(surface_dq=dq, surface_sigma=sigma, surface_warp8=wp)
    path = os.path.join(HERE, "knn_ties.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
