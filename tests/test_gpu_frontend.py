"""GPU parity of the depth front-end and the projective-ICP reduction: the HIP kernels behind the
C-ABI against oracle/dfusion_frontend_oracle.c on the same inputs -- bit-identical images, bit-identical 27 ICP sums (the
reference's reduction tree is reproduced), hence bit-identical Gauss-Newton trajectories with the same host solver."""
import numpy as np
import pytest
import torch

import oracle_lib as O
from dynamicfusion_amd import Intr, capi, download_u16, frontend, synth, upload_u16
from frontend_ref import BILATERAL, icp_loop, level_intr, thresholds

pytestmark = pytest.mark.gpu
F32 = np.float32
FULL = synth.Config(64, 1.0, cols=640, rows=480, nodes=0, k=4)
RAGGED = synth.Config(64, 1.0, cols=203, rows=117, nodes=0, k=4)        # not a multiple of any block size, odd dims


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def gpu_pyramids(depth_np, intr, levels=3):
    d = [frontend.depthBilateralFilter(upload_u16(depth_np), BILATERAL["ksz"], BILATERAL["sigma_spatial"], BILATERAL["sigma_depth"])]
    for i in range(1, levels):
        d.append(frontend.depthBuildPyramid(d[-1], BILATERAL["sigma_depth"]))
    pn = [frontend.computePointNormals(frontend.intr_level(intr, i), d[i]) for i in range(levels)]
    return d, [a for a, _ in pn], [b for _, b in pn]


def cpu_pyramids(depth_np, intr_np, levels=3):
    d = [O.bilateral(depth_np, **BILATERAL)]
    for i in range(1, levels):
        d.append(O.depth_pyramid(d[-1], BILATERAL["sigma_depth"]))
    pn = [O.compute_point_normals(d[i], level_intr(intr_np, i)) for i in range(levels)]
    return d, [a for a, _ in pn], [b for _, b in pn]


@pytest.mark.parametrize("cfg", [FULL, RAGGED], ids=["640x480", "203x117"])
def test_front_end_images_match_oracle(cfg):
    intr = Intr(*cfg.intr); intr_np = np.array(cfg.intr, F32)
    depth = synth.depth_frame(cfg, 0)
    gd, gv, gn = gpu_pyramids(depth, intr)
    cd, cv, cn = cpu_pyramids(depth, intr_np)
    torch.cuda.synchronize()
    for i in range(3):
        assert np.array_equal(download_u16(gd[i]), cd[i]), "depth level %d" % i
        assert np.array_equal(bits(gv[i].cpu().numpy()), bits(cv[i])) and np.array_equal(bits(gn[i].cpu().numpy()), bits(cn[i]))
    assert (cd[0] != depth).sum() > 0.01 * depth.size                      # the filter did something (curved parts only)
    # truncation (in place)
    t = frontend.depthTruncation(gd[0].clone(), 1.2)
    assert np.array_equal(download_u16(t), O.truncate_depth(cd[0], 1.2)) and (download_u16(t) == 0).sum() > (cd[0] == 0).sum()
    # cloudToDepth (imgproc.cpp:98-103; round 6): the points image back to millimetres -- NaN points (no depth) give 0, values past the
    # ushort range saturate, negative ones give 0 (the reference target's float -> ushort conversion)
    c2d = frontend.cloudToDepth(gv[0])
    assert np.array_equal(download_u16(c2d), O.cloud_to_depth(cv[0])) and (download_u16(c2d) > 0).sum() > 1000
    edge = torch.tensor([[[0, 0, 70.0, 0], [0, 0, -1.0, 0], [0, 0, float("nan"), 0], [0, 0, 65.5349, 0], [0, 0, 0.0009999, 0], [0, 0, float("inf"), 0],
                          [0, 0, 1.2345678, 0], [0, 0, 65.535, 0]]], dtype=torch.float32, device="cuda")
    got_edge = download_u16(frontend.cloudToDepth(edge))
    assert np.array_equal(got_edge, O.cloud_to_depth(edge.cpu().numpy())) and list(got_edge[0][:3]) == [65535, 0, 0] and got_edge[0][5] == 65535
    # USE_DEPTH build: normals + depth mask, resizeDepthNormals
    gm = gd[0].clone()
    gnm = frontend.computeNormalsAndMaskDepth(intr, gm)
    cm, cnm = O.compute_normals_mask_depth(cd[0], intr_np)
    assert np.array_equal(download_u16(gm), cm) and np.array_equal(bits(gnm.cpu().numpy()), bits(cnm))
    gd2, gn2 = frontend.resizeDepthNormals(gm, gnm)
    cd2, cn2 = O.resize_depth_normals(cm, cnm)
    assert np.array_equal(download_u16(gd2), cd2) and np.array_equal(bits(gn2.cpu().numpy()), bits(cn2))
    # resizePointsNormals (the ray-cast pyramid of KinFu::operator(), kinfu.cpp:296-299)
    gp2, gq2 = frontend.resizePointsNormals(gv[0], gn[0])
    cp2, cq2 = O.resize_points_normals(cv[0], cn[0])
    assert np.array_equal(bits(gp2.cpu().numpy()), bits(cp2)) and np.array_equal(bits(gq2.cpu().numpy()), bits(cq2))


@pytest.mark.parametrize("cfg", [FULL, RAGGED], ids=["640x480", "203x117"])
def test_icp_sums_and_trajectory_match_oracle(cfg):
    intr = Intr(*cfg.intr); intr_np = np.array(cfg.intr, F32)
    d0, d1 = synth.depth_frame(cfg, 0), synth.depth_frame(cfg, 6)
    _, gv0, gn0 = gpu_pyramids(d0, intr); _, gv1, gn1 = gpu_pyramids(d1, intr)
    _, cv0, cn0 = cpu_pyramids(d0, intr_np); _, cv1, cn1 = cpu_pyramids(d1, intr_np)
    icp = frontend.ProjectiveICP()
    d2t, mc = icp.thresholds()
    assert (d2t, mc) == thresholds()
    est = synth.rot_y_about(np.deg2rad(0.4), (0.05, -0.02, 1.0)).astype(F32)
    for level in range(3):
        li = frontend.intr_level(intr, level)
        g = icp.sums(li, gv1[level], gn1[level], gv0[level], gn0[level], est)
        c, acc = O.icp_sums(cv1[level], cn1[level], cv0[level], cn0[level], synth.aff12(est), level_intr(intr_np, level), d2t, mc)
        assert icp.last_accepted == acc > 100
        assert np.array_equal(bits(g), bits(c)), "level %d" % level
    # the full Gauss-Newton loop (projective_icp.cpp:129-213): same sums every iteration => same pose, bit for bit
    ok_g, aff_g = icp.estimateTransform(intr, gv1, gn1, gv0, gn0)
    ok_c, aff_c, hist = icp_loop(lambda lv, li, a: O.icp_sums(cv1[lv], cn1[lv], cv0[lv], cn0[lv], synth.aff12(a), li, d2t, mc)[0], intr_np)
    assert ok_g and ok_c and np.array_equal(bits(aff_g), bits(aff_c))
    # the single-enqueue loop (solve + pose update on the device, LU instead of numpy's LAPACK solve): same pose to ~1e-5
    ok_d, aff_d = icp.estimateTransformDevice(intr, gv1, gn1, gv0, gn0)
    assert ok_d and np.abs(aff_d - aff_g).max() < 5e-5      # the roll about the scene's symmetry axis is ill-conditioned and amplifies the solver difference
    true = synth.affine_mul(synth.affine_inv(synth.camera_pose(cfg, 0)), synth.camera_pose(cfg, 6))
    assert np.abs(aff_g[:3, 3] - true[:3, 3]).max() < 1e-2 and np.abs(aff_g[:3, 2] - true[:3, 2]).max() < 1e-2   # sanity vs ground truth


def test_icp_depth_variant_and_degenerate_input():
    cfg = RAGGED
    intr = Intr(*cfg.intr); intr_np = np.array(cfg.intr, F32)
    d0, d1 = synth.depth_frame(cfg, 0), synth.depth_frame(cfg, 3)
    f0, f1 = O.bilateral(d0, **BILATERAL), O.bilateral(d1, **BILATERAL)
    m0, n0 = O.compute_normals_mask_depth(f0, intr_np); m1, n1 = O.compute_normals_mask_depth(f1, intr_np)
    icp = frontend.ProjectiveICP()
    d2t, mc = icp.thresholds()
    est = synth.rot_y_about(np.deg2rad(0.2), (0.0, 0.0, 1.0)).astype(F32)
    up4 = lambda a: torch.from_numpy(a).cuda()
    g = icp.sums(intr, upload_u16(m1), up4(n1), upload_u16(m0), up4(n0), est, depth_variant=True)
    c, acc = O.icp_sums(m1, n1, m0, n0, synth.aff12(est), intr_np, d2t, mc, depth_variant=True)
    assert icp.last_accepted == acc > 1000 and np.array_equal(bits(g), bits(c))
    # nothing to match: all sums exactly zero, determinant 0, estimateTransform reports failure like the reference (:152-156)
    z = torch.zeros_like(upload_u16(m1))
    g0 = icp.sums(intr, z, up4(n1), upload_u16(m0), up4(n0), est, depth_variant=True)
    assert icp.last_accepted == 0 and not g0.any()
    icp1 = frontend.ProjectiveICP(); icp1.setIterationsNum([3])
    ok, _ = icp1.estimateTransform(intr, [z], [up4(n1)], [upload_u16(m0)], [up4(n0)], depth_variant=True)
    assert ok is False
    ok, aff = icp1.estimateTransformDevice(intr, [z], [up4(n1)], [upload_u16(m0)], [up4(n0)], depth_variant=True)
    assert ok is False and np.array_equal(aff, np.eye(4, dtype=F32))
    # argument validation
    assert capi.lib().dfusion_bilateral_filter(z.data_ptr(), 2 * cfg.cols, z.data_ptr(), 2 * cfg.cols, cfg.cols, cfg.rows, 7, 4.5, 0.04, None) == 100001
    assert capi.lib().dfusion_icp_workspace_floats(640, 480) == 27 * 1200


def test_render_kernels_match_oracle_and_reference():
    """dfusion_render_image_points / _depth / _tangent_colors (imgproc.cu:420-583) on a ray-cast of a fused volume: BGRA bytes equal
    the oracle's, and -- where the prebuilt library travelled -- the reference's own kernels run on the host (oracle/cuda_shim)."""
    from test_gpu_parity import make_gpu_volume
    from scene import Scene
    cfg = synth.Config(64, 1.0, cols=320, rows=240, nodes=0, k=8)
    sc = Scene(cfg, n_frames=2, with_nodes=False)
    intr = Intr(*cfg.intr)
    vol = make_gpu_volume(sc)
    for f in range(2):
        vol.integrate(upload_u16(sc.dists[f]), sc.cam_poses[f], intr)
    pts = torch.empty((cfg.rows, cfg.cols, 4), dtype=torch.float32, device="cuda"); nrm = torch.empty_like(pts)
    vol.raycast(sc.cam_poses[1], intr, pts, nrm)
    dep = torch.empty((cfg.rows, cfg.cols), dtype=torch.int16, device="cuda"); dnrm = torch.empty_like(pts)
    vol.raycast(sc.cam_poses[1], intr, dep, dnrm)
    torch.cuda.synchronize()
    hp, hn, hd, hdn = pts.cpu().numpy(), nrm.cpu().numpy(), dep.cpu().numpy().view(np.uint16), dnrm.cpu().numpy()
    for light in ((0.0, 0.0, 0.0), (0.4, -0.3, 0.2)):
        a = frontend.renderImage(pts, nrm, intr, light).cpu().numpy()
        b = frontend.renderImage(dep, dnrm, intr, light).cpu().numpy()
        assert np.array_equal(a, O.render_points(hp, hn, light)) and np.array_equal(b, O.render_depth(hd, hdn, sc.intr, light))
        assert len(np.unique(a[..., 0])) > 30
        if O.have_refcu():
            img = np.zeros((cfg.rows, cfg.cols, 4), np.uint8)
            O.refcu().refcu_render_points(hp.reshape(-1), hn.reshape(-1), cfg.rows, cfg.cols, sc.intr, np.array(light, F32), img.reshape(-1))
            assert np.array_equal(a, img)
    t = frontend.renderTangentColors(nrm).cpu().numpy()
    assert np.array_equal(t, O.render_tangent_colors(hn))
