"""INTEGRATION.md section A, for real: the REFERENCE'S OWN host class drives the HIP kernels.

tests/ref_host_bridge/build_ref_host.py compiles kfusion/src/{tsdf_volume, imgproc, precomp, device_memory}.cpp UNMODIFIED -- where they lie under
/root/reference, against the reference's own headers -- plus ONE added translation unit, tests/ref_host_bridge/hip_bridge.cpp, which
defines the kfusion::device::* functions of the reference's private header (kfusion/src/internal.hpp:104-143) by forwarding them to the
C-ABI of libdfusion_hip.so.  That is the patch a maintainer of the reference would make (delete the .cu files, add the bridge).
  * CPU (where /root/reference exists): the build succeeds, leaves no kfusion::device::* symbol undefined, and the library takes its
    TsdfVolume from the REFERENCE's tsdf_volume.cpp, not from this repository's mirror;
  * GPU: the binary (built in the container, travelled with the snapshot) runs kfusion::cuda::TsdfVolume::{clear, integrate, raycast x2,
    compute_points, compute_normals} and cuda::computeDists of the reference on the MI355X, and every output equals the oracle's bit for
    bit (kfusion/src/tsdf_volume.cpp:89-174,184-220,312-324; kinfu.cpp:226,248,297,398-399);
  * round 6: ALL 20 forwards of the bridge run -- the front-end through the reference's own imgproc.cpp wrappers
    (depthBilateralFilter, depthTruncation, depthBuildPyramid, computeNormalsAndMaskDepth, computePointNormals, resizeDepthNormals,
    resizePointsNormals, renderImage x 2, renderTangentColors, cloudToDepth) with NON-SQUARE intrinsics, TsdfVolume::psdf (tsdf_volume.cpp:266-292) ->
    device::project_and_remove, and the overload nothing in the reference calls, directly -- each output equal to the oracle's bit for
    bit; the binary reports which forwards ran and the test asserts 20 of 20."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O
from dynamicfusion_amd import synth
from scene import Scene, compare_volumes
from test_gpu_cxx_host import cxx_inv, cxx_mul

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_host_bridge"))
import build_ref_host as RB          # noqa: E402  (tests/ref_host_bridge/build_ref_host.py)

F32 = np.float32


@pytest.mark.skipif(not RB.have_reference(), reason="/root/reference not present")
def test_reference_host_layer_builds_unmodified_over_the_bridge():
    app = RB.build()
    assert app and os.path.exists(app) and os.path.exists(RB.LIB)
    assert RB.undefined_device_symbols() == []          # every kfusion::device::* call of the four reference files is bridged
    defined = subprocess.run(["nm", "-C", "--defined-only", "-D", RB.LIB], capture_output=True, text=True).stdout
    for s in ("kfusion::cuda::TsdfVolume::integrate(", "kfusion::cuda::TsdfVolume::raycast(", "kfusion::cuda::TsdfVolume::clear()",
              "kfusion::cuda::TsdfVolume::fetchCloud(", "kfusion::cuda::computeDists(", "kfusion::cuda::DeviceMemory2D::create(",
              "kfusion::device::integrate(", "kfusion::device::raycast(", "kfusion::device::clear_volume(", "kfusion::device::extractCloud("):
        assert s in defined, s
    # ... and the C-ABI is what it links: the hot-path entry points are undefined in it, resolved by libdfusion_hip.so
    undef = subprocess.run(["nm", "-D", "--undefined-only", RB.LIB], capture_output=True, text=True).stdout
    for s in ("dfusion_integrate", "dfusion_raycast_points", "dfusion_raycast_depth", "dfusion_clear", "dfusion_compute_dists", "dfusion_extract_cloud"):
        assert s in undef, s
    ldd = subprocess.run(["ldd", app], capture_output=True, text=True).stdout
    assert "libdfusion_hip.so" in ldd and "libkfusion_refhost.so" in ldd and "libkfusion_hip.so" not in ldd     # not the mirror


@pytest.mark.gpu
def test_reference_tsdf_volume_class_on_the_gpu_matches_oracle(tmp_path):
    app = RB.build()
    if not app:
        pytest.skip("tests/ref_host_bridge/_build/ref_host_frame was not built (needs /root/reference in the build container)")
    cfg = synth.Config(64, 1.0, cols=160, rows=120, nodes=0, k=8)
    frames = 3
    sc = Scene(cfg, n_frames=frames)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(synth.aff12(sc.pose).tobytes())
        f.write(np.asarray(cfg.intr, F32).tobytes())
        for i in range(frames):
            f.write(sc.depths[i].tobytes())
            f.write(synth.aff12(sc.cam_poses[i]).tobytes())
        intr2 = np.array([cfg.intr[0] * 1.0625, cfg.intr[1] * 0.9375, cfg.intr[2] + 1.5, cfg.intr[3] - 2.25], F32)     # fx != fy: both axes of focal_of()
        f.write(intr2.tobytes())
    r = subprocess.run([app, str(cfg.dims[0]), str(cfg.size), str(cfg.cols), str(cfg.rows), str(frames), fin, fout], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ref_host_frame ok" in r.stdout, r.stdout + r.stderr
    rep_line = [ln for ln in r.stdout.splitlines() if ln.startswith("hip_bridge forwards ")]
    assert rep_line, r.stdout
    print(rep_line[-1])
    assert rep_line[-1].split()[2] == "20/20:", rep_line[-1]                 # every kfusion::device::* forward of the bridge ran
    assert "=0" not in rep_line[-1]
    raw = np.fromfile(fout, np.uint8)
    nv, npx = int(np.prod(cfg.dims)), cfg.rows * cfg.cols
    o = 0
    def take(nbytes, dt, shape):
        nonlocal o
        a = raw[o:o + nbytes].view(dt).reshape(shape); o += nbytes
        return a
    vol = take(4 * nv, np.uint32, (cfg.dims[2], cfg.dims[1], cfg.dims[0]))
    pts = take(16 * npx, F32, (cfg.rows, cfg.cols, 4)); nrm = take(16 * npx, F32, (cfg.rows, cfg.cols, 4))
    cdepth = take(2 * npx, np.uint16, (cfg.rows, cfg.cols)); cnrm = take(16 * npx, F32, (cfg.rows, cfg.cols, 4))
    cnt = int(take(8, np.uint64, (1,))[0])
    cloud = take(16 * cnt, F32, (cnt, 4)); cloud_n = take(16 * cnt, F32, (cnt, 4))
    vol_cleared = take(4 * nv, np.uint32, (nv,))
    H, W, h, w = cfg.rows, cfg.cols, cfg.rows // 2, cfg.cols // 2
    bil = take(2 * npx, np.uint16, (H, W)); trunc = take(2 * npx, np.uint16, (H, W)); pyr = take(2 * h * w, np.uint16, (h, w))
    md = take(2 * npx, np.uint16, (H, W)); mn = take(16 * npx, F32, (H, W, 4))
    pc = take(16 * npx, F32, (H, W, 4)); pn = take(16 * npx, F32, (H, W, 4))
    dh = take(2 * h * w, np.uint16, (h, w)); nh = take(16 * h * w, F32, (h, w, 4))
    ph = take(16 * h * w, F32, (h, w, 4)); pnh = take(16 * h * w, F32, (h, w, 4))
    im1 = take(4 * npx, np.uint8, (H, W, 4)); im2 = take(4 * npx, np.uint8, (H, W, 4)); im3 = take(4 * npx, np.uint8, (H, W, 4))
    ro = take(4 * npx, F32, (npx,)); d2_after = take(2 * npx, np.uint16, (H, W))
    p4 = take(16 * npx, F32, (npx, 4)); d3_after = take(2 * npx, np.uint16, (H, W))
    c2d = take(2 * npx, np.uint16, (H, W))
    assert o == len(raw)

    # the oracle on the same inputs, with the host arithmetic of the OpenCV stand-in (= the mirror's: cxx_inv / cxx_mul) for
    # vol2cam = camera_pose.inv() * pose_ (tsdf_volume.cpp:112), cam2vol = pose_.inv() * camera_pose and Rinv (:162-165)
    ref = sc.new_volume()
    for f in range(frames):
        assert np.array_equal(O.compute_dists(sc.depths[f], cfg.intr), sc.dists[f])
        O.integrate(sc.dists[f], ref, sc.ovol(ref), synth.aff12(cxx_mul(cxx_inv(sc.cam_poses[f]), sc.pose)), sc.intr)
    s = compare_volumes(vol, ref)
    assert s["bits_mismatch"] == 0 and int((ref != 0).sum()) > 10000, s          # TsdfVolume::integrate x 3 (through the reference's computeDists)
    cam2vol = cxx_mul(cxx_inv(sc.pose), sc.cam_poses[frames - 1])
    rinv = cxx_inv(cam2vol)[:3, :3]
    rp, rn, _, stats = O.raycast_points(sc.ovol(ref), synth.aff12(cam2vol), rinv, sc.reproj, cfg.cols, cfg.rows, cfg.raycast_step_factor, cfg.gradient_delta_factor)
    assert stats[1] > 1000
    assert np.array_equal(pts.view(np.uint32), rp.view(np.uint32)) and np.array_equal(nrm.view(np.uint32), rn.view(np.uint32))      # raycast(Cloud&, Normals&)
    rd, rdn = O.raycast_depth(sc.ovol(ref), synth.aff12(cam2vol), rinv, sc.reproj, cfg.cols, cfg.rows, cfg.raycast_step_factor, cfg.gradient_delta_factor)[:2]
    assert np.array_equal(cdepth, rd) and np.array_equal(cnrm.view(np.uint32), rdn.view(np.uint32)) and (rd > 0).sum() > 1000      # raycast(Depth&, Normals&)
    rc, n = O.extract_cloud(sc.ovol(ref), synth.aff12(sc.pose), 1 << 22)
    assert n == cnt > 1000                                                     # compute_points: fetchCloud -> device::extractCloud
    order = lambda a: a[np.lexsort(np.ascontiguousarray(a).view(np.uint32).T[::-1])]
    assert np.array_equal(order(cloud).view(np.uint32), order(rc).view(np.uint32))
    rnrm = O.extract_normals(sc.ovol(ref), synth.aff12(sc.pose), cxx_inv(sc.pose)[:3, :3], cloud, cfg.gradient_delta_factor)
    assert np.array_equal(cloud_n.view(np.uint32), rnrm.view(np.uint32))       # compute_normals: fetchNormals -> device::extractNormals
    assert not vol_cleared.any()                                               # TsdfVolume::clear -> device::clear_volume

    # ---- the front-end forwards, through the reference's imgproc.cpp wrappers (non-square intrinsics), vs the oracle, bit for bit
    u = lambda a: np.ascontiguousarray(a).view(np.uint32)
    d_last = sc.depths[frames - 1]
    o_bil = O.bilateral(d_last, 7, 4.5, 0.04)
    assert np.array_equal(bil, o_bil) and (bil != d_last).any()                # cuda::depthBilateralFilter -> device::bilateralFilter
    o_tr = O.truncate_depth(o_bil, 1.2)
    assert np.array_equal(trunc, o_tr) and (o_tr != o_bil).any() and o_tr.any()                 # cuda::depthTruncation -> device::truncateDepth
    assert np.array_equal(pyr, O.depth_pyramid(o_bil, 0.04)) and pyr.any()      # cuda::depthBuildPyramid -> device::depthPyr
    o_md, o_mn = O.compute_normals_mask_depth(o_bil, intr2)
    assert np.array_equal(md, o_md) and np.array_equal(u(mn), u(o_mn)) and np.isfinite(o_mn[..., 0]).sum() > 1000     # computeNormalsAndMaskDepth
    o_pc, o_pn = O.compute_point_normals(o_bil, intr2)
    assert np.array_equal(u(pc), u(o_pc)) and np.array_equal(u(pn), u(o_pn)) and np.isfinite(o_pn[..., 0]).sum() > 1000    # computePointNormals
    # (square intrinsics give different bits: the bridge's focal_of() really reconstructs fx and fy separately)
    assert not np.array_equal(u(O.compute_point_normals(o_bil, np.asarray(cfg.intr, F32))[0]), u(o_pc))
    o_dh, o_nh = O.resize_depth_normals(o_md, o_mn)
    assert np.array_equal(dh, o_dh) and np.array_equal(u(nh), u(o_nh))          # cuda::resizeDepthNormals
    o_ph, o_pnh = O.resize_points_normals(o_pc, o_pn)
    assert np.array_equal(u(ph), u(o_ph)) and np.array_equal(u(pnh), u(o_pnh))  # cuda::resizePointsNormals
    light = np.array([0.3, -0.2, -0.5], F32)
    assert np.array_equal(im1, O.render_depth(o_md, o_mn, intr2, light)) and im1[..., :3].any()     # renderImage(Depth, ...)
    assert np.array_equal(im2, O.render_points(o_pc, o_pn, light)) and im2[..., :3].any()           # renderImage(Cloud, ...)
    assert np.array_equal(im3, O.render_tangent_colors(o_pn)) and im3[..., :3].any()                # renderTangentColors
    # ---- TsdfVolume::psdf -> device::project_and_remove(const PtrStepSz<ushort>&, ...): distances and the dists image after the removal
    dists2 = O.compute_dists(d_last, intr2)
    w4 = np.zeros((npx, 4), F32); w4[:, :3] = pts.reshape(-1, 4)[:, :3]
    o_p4, o_after, o_ro, n_in = O.project_and_remove(dists2, w4, intr2)
    assert n_in > 1000
    assert np.array_equal(u(ro), u(o_ro)) and np.array_equal(d2_after, o_after) and (o_after != dists2).sum() > 1000
    # ---- the non-const overload, called directly on the cast's float4 image as it is (a miss is NaN in all four components and stays
    # untouched; psdf above rebuilt its points with w = 0): the rewritten points and the same removal
    o_p4, o_after3, _, _ = O.project_and_remove(dists2, pts.reshape(-1, 4), intr2)
    assert np.array_equal(u(p4), u(o_p4)) and np.array_equal(d3_after, o_after3) and np.array_equal(o_after3, o_after)
    # ---- cuda::cloudToDepth -> device::cloud_to_depth (round 6: the 20th forward)
    assert np.array_equal(c2d, O.cloud_to_depth(o_pc)) and (c2d > 0).sum() > 1000 and (c2d == 0).any()
