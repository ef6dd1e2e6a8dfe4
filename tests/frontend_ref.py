"""Shared helpers for the depth front-end / ICP tests: pyramid construction through any backend and a Gauss-Newton ICP loop
with the solver of dynamicfusion_amd.frontend.ProjectiveICP (so oracle-driven and HIP-driven ICP can be compared)."""
import math

import numpy as np

from dynamicfusion_amd import frontend

F32 = np.float32

# KinFuParams::default_params_dynamicfusion (kinfu.cpp:15-50)
BILATERAL = dict(ksz=7, sigma_spatial=4.5, sigma_depth=0.04)


def level_intr(intr, level):
    div = F32(1 << level)
    return np.array([F32(intr[0]) / div, F32(intr[1]) / div, F32(intr[2]) / div, F32(intr[3]) / div], F32)


def thresholds(dist_thres=0.1, angle_deg=20.0):
    """(dist2_thres, min_cosine) as ComputeIcpHelper computes them (projective_icp.cpp:11-15)."""
    ang = F32(angle_deg) * F32(0.017453293)
    return float(F32(dist_thres) * F32(dist_thres)), float(F32(math.cos(ang)))


def icp_loop(sums_fn, intr, iters=(10, 5, 4)):
    """projective_icp.cpp:129-213 with sums_fn(level, level_intr, affine) -> 27 sums.  Returns (ok, affine, history)."""
    affine = np.eye(4, dtype=F32)
    hist = []
    for level in range(len(iters) - 1, -1, -1):
        li = level_intr(intr, level)
        for _ in range(iters[level]):
            s = sums_fn(level, li, affine)
            A, b = frontend.unpack_icp_sums(s)
            det = float(np.linalg.det(A.astype(np.float64)))
            hist.append((level, float(np.linalg.norm(b)), det))
            if abs(det) < 1e-15 or math.isnan(det):
                return False, affine, hist
            r = np.linalg.solve(A.astype(np.float64), b.astype(np.float64))
            affine = (frontend.rodrigues_affine(r).astype(np.float64) @ affine.astype(np.float64)).astype(F32)
    return True, affine, hist
