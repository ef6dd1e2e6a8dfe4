"""Self-consistency of the depth front-end / ICP oracle (oracle/dfusion_frontend_oracle.c) -- the reference holds no tests or
vectors for these kernels ("parity unpinned"), so the restatement is pinned by independent numpy formulations and by
properties: constant images, planes, known rigid motion."""
import numpy as np

import oracle_lib as O
from dynamicfusion_amd import frontend, synth
from frontend_ref import BILATERAL, icp_loop, level_intr, thresholds

F32 = np.float32
CFG = synth.Config(64, 1.0, cols=160, rows=120, nodes=0, k=4)
INTR = np.array(CFG.intr, F32)


def test_bilateral_constant_and_window_quirk():
    d = np.full((24, 40), 1500, np.uint16)
    out = O.bilateral(d, **BILATERAL)
    assert (out == 1500).all()
    # a lone invalid pixel (0) and its neighbours weigh each other exp(-1500^2 / (2 * 40^2)) ~ 0: both keep their value
    d[10, 10] = 0
    out = O.bilateral(d, **BILATERAL)
    assert out[10, 11] == 1500 and out[10, 10] == 0
    # numpy formulation of imgproc.cu:21-41 at one interior and one border pixel (window is [c-3, min(c+4, n-1)) : the last
    # row / column never enter a window)
    rng = np.random.default_rng(0)
    d = (1000 + rng.integers(0, 60, (24, 40))).astype(np.uint16)
    out = O.bilateral(d, **BILATERAL)
    ss, sd = F32(0.5) / (F32(4.5) * F32(4.5)), F32(0.5) / (F32(40.0) * F32(40.0))
    for (y, x) in ((12, 20), (0, 0), (23, 39), (22, 38)):
        s1 = s2 = 0.0
        for cy in range(max(y - 3, 0), min(y - 3 + 7, 23)):
            for cx in range(max(x - 3, 0), min(x - 3 + 7, 39)):
                w = np.exp(-(float((x - cx) ** 2 + (y - cy) ** 2) * float(ss) + float((int(d[y, x]) - int(d[cy, cx])) ** 2) * float(sd)))
                s1 += float(d[cy, cx]) * w; s2 += w
        assert abs(int(out[y, x]) - s1 / s2) <= 0.5 + 1e-3, (y, x)


def test_truncate_and_pyramid_match_numpy():
    rng = np.random.default_rng(1)
    d = rng.integers(0, 5000, (31, 45)).astype(np.uint16)          # ragged (odd) size
    t = O.truncate_depth(d, 2.5)
    assert np.array_equal(t, np.where(d > 2500, 0, d))
    p = O.depth_pyramid(d, 0.04)
    assert p.shape == (15, 22)
    for (y, x) in ((0, 0), (7, 11), (14, 21)):
        c = int(d[2 * y, 2 * x]); vals = []
        for cy in range(max(0, 2 * y - 2), min(2 * y - 2 + 5, 30)):
            for cx in range(max(0, 2 * x - 2), min(2 * x - 2 + 5, 44)):
                if abs(int(d[cy, cx]) - c) < 120.0:
                    vals.append(int(d[cy, cx]))
        assert p[y, x] == (sum(vals) // len(vals) if vals else 0)
    const = O.depth_pyramid(np.full((16, 16), 777, np.uint16), 0.04)
    assert (const == 777).all()


def test_normals_of_a_fronto_parallel_plane_and_masks():
    d = np.full((20, 30), 2000, np.uint16)
    d[5, 5] = 0
    p, n = O.compute_point_normals(d, INTR)
    good = ~np.isnan(n[..., 0])
    # last row / column and the three pixels that touch the hole have no normal
    assert not good[-1].any() and not good[:, -1].any() and not good[5, 5] and not good[5, 4] and not good[4, 5]
    assert good.sum() == 19 * 29 - 3
    assert np.allclose(n[good][:, :3], [0, 0, -1], atol=1e-6) and (n[good][:, 3] == 0).all()
    assert np.allclose(p[good][:, 2], 2.0) and np.isnan(p[~good]).all()
    x = 7; assert np.isclose(p[3, x, 0], 2.0 * (x - INTR[2]) / INTR[0], rtol=1e-6)
    dm, nm = O.compute_normals_mask_depth(d, INTR)
    assert np.array_equal(np.isnan(nm[..., 0]), ~good) and (nm[..., 3] == 0).all()
    assert np.array_equal(dm == 0, ~good) and (dm[good] == 2000).all()


def test_resize_kernels_match_numpy():
    rng = np.random.default_rng(2)
    d = rng.integers(500, 3000, (12, 18)).astype(np.uint16); d[3, 4] = 0
    p, n = O.compute_point_normals(d, INTR)
    p2, n2 = O.resize_points_normals(p, n)
    assert p2.shape == (6, 9, 4)
    for (y, x) in ((0, 0), (1, 2), (5, 8)):
        blk = p[2 * y:2 * y + 2, 2 * x:2 * x + 2].reshape(4, 4)
        if np.isnan(blk[:, 0]).any():
            assert np.isnan(p2[y, x, :3]).all() and p2[y, x, 3] == 0
        else:
            exp = (((blk[0] + blk[1]) + blk[2]) + blk[3]) * F32(0.25)
            assert np.array_equal(p2[y, x, :3], exp[:3])
    dm, nm = O.compute_normals_mask_depth(d, INTR)
    d2, nn2 = O.resize_depth_normals(dm, nm)
    for (y, x) in ((0, 0), (1, 2), (5, 8)):
        blk = dm[2 * y:2 * y + 2, 2 * x:2 * x + 2].astype(np.int64).ravel()
        if blk[0] * blk[1] != 0 and blk[2] * blk[3] != 0:
            assert d2[y, x] == blk.sum() // 4
        else:
            assert d2[y, x] == 0 and np.isnan(nn2[y, x]).all()


def _pyramids(depth, levels=3):
    d = [O.bilateral(depth, **BILATERAL)]
    for i in range(1, levels):
        d.append(O.depth_pyramid(d[-1], BILATERAL["sigma_depth"]))
    pn = [O.compute_point_normals(d[i], level_intr(INTR, i)) for i in range(levels)]
    return d, [a for a, _ in pn], [b for _, b in pn]


def test_icp_sums_tree_vs_float64_and_gauss_newton_reduces_the_residual():
    """The 27 sums equal a float64 accumulation to fp32 reduction accuracy; Gauss-Newton over them reduces |b| and ends
    at the true relative camera motion in every observable degree of freedom."""
    d0, d1 = synth.depth_frame(CFG, 0), synth.depth_frame(CFG, 6)
    _, v0, n0 = _pyramids(d0)
    _, v1, n1 = _pyramids(d1)
    d2t, mc = thresholds()
    # a generic (non-identity) estimate: with the identity every reprojected pixel centre sits exactly on a pixel boundary and
    # the accept set would hinge on the last bit of u, v
    est = synth.rot_y_about(np.deg2rad(0.4), (0.05, -0.02, 1.0)).astype(F32)
    s, acc = O.icp_sums(v1[0], n1[0], v0[0], n0[0], synth.aff12(est), INTR, d2t, mc)
    assert acc > 5000 and np.isfinite(s).all()
    # float64 re-accumulation of the same rows
    rows = []
    R64, t64 = est[:3, :3].astype(np.float64), est[:3, 3].astype(np.float64)
    for y in range(CFG.rows):
        for x in range(CFG.cols):
            p = v1[0][y, x, :3].astype(np.float64)
            if np.isnan(p[0]):
                continue
            sp = R64 @ p + t64
            if sp[2] <= 0:
                continue
            u = INTR[0] * sp[0] / sp[2] + INTR[2]; v = INTR[1] * sp[1] / sp[2] + INTR[3]
            if not (0 <= u < CFG.cols and 0 <= v < CFG.rows):
                continue
            dp = v0[0][int(v), int(u), :3].astype(np.float64); nd = n0[0][int(v), int(u), :3].astype(np.float64)
            ns = R64 @ n1[0][y, x, :3].astype(np.float64)
            if np.isnan(dp[0]) or ((sp - dp) ** 2).sum() > d2t or abs(float(ns @ nd)) < mc:
                continue
            rows.append(np.concatenate([np.cross(sp, nd), nd, [nd @ (dp - sp)]]))
    R = np.array(rows)
    assert abs(len(rows) - acc) <= 0.003 * acc                     # pixels whose u, v or thresholds tie within rounding
    ref = np.array([(R[:, i] * R[:, j]).sum() for i in range(6) for j in range(i, 7)])
    assert np.allclose(s, ref, rtol=1e-2, atol=1e-2 * np.abs(ref).max())

    def sums_fn(level, li, affine):
        return O.icp_sums(v1[level], n1[level], v0[level], n0[level], synth.aff12(affine), li, d2t, mc)[0]
    ok, aff, hist = icp_loop(sums_fn, INTR)
    assert ok and len(hist) == 19
    assert hist[-1][1] < 0.2 * hist[10][1]                         # |b| at level 0: last iteration << first level-0 iteration
    true = synth.affine_mul(synth.affine_inv(synth.camera_pose(CFG, 0)), synth.camera_pose(CFG, 6))   # curr -> prev
    # sphere over a plane is symmetric about the plane normal through the sphere centre: roll about (nearly) the optical axis is
    # unobservable, the other five degrees of freedom are recovered
    assert np.abs(aff[:3, 3] - true[:3, 3]).max() < 5e-3
    assert np.abs(aff[2, :3] - true[2, :3]).max() < 3e-3 and np.abs(aff[:3, 2] - true[:3, 2]).max() < 3e-3


def test_icp_depth_variant_and_degenerate_input():
    d0, d1 = synth.depth_frame(CFG, 0), synth.depth_frame(CFG, 2)
    f0 = O.bilateral(d0, **BILATERAL); f1 = O.bilateral(d1, **BILATERAL)
    m0, n0 = O.compute_normals_mask_depth(f0, INTR); m1, n1 = O.compute_normals_mask_depth(f1, INTR)
    d2t, mc = thresholds()
    s, acc = O.icp_sums(m1, n1, m0, n0, synth.aff12(np.eye(4, dtype=F32)), INTR, d2t, mc, depth_variant=True)
    assert acc > 5000 and np.isfinite(s).all()
    A, b = frontend.unpack_icp_sums(s)
    assert np.allclose(A, A.T) and np.all(np.linalg.eigvalsh(A.astype(np.float64)) > -1e-3)
    # empty input: every pixel filtered, all sums exactly zero (-> determinant 0 -> estimateTransform returns false)
    z = np.zeros_like(m0)
    s0, acc0 = O.icp_sums(z, n1, m0, n0, synth.aff12(np.eye(4, dtype=F32)), INTR, d2t, mc, depth_variant=True)
    assert acc0 == 0 and not s0.any()
