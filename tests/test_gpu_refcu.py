"""GPU parity against the reference's OWN CUDA kernels (not the restatement): the HIP path through the C-ABI vs
  * tests/golden/refcu_64.npz -- vectors produced by kfusion/src/cuda/tsdf_volume.cu + imgproc.cu compiled for the host
    (tests/golden/make_golden_refcu.py), always available;
  * oracle/_ref/libdfref_cu.so run live on the box's CPU (the prebuilt library travels with the snapshot), at 128^3 / 640x480.
Bit-for-bit."""
import os

import numpy as np
import pytest
import torch

import oracle_lib as O
from dynamicfusion_amd import Intr, compute_dists, download_u16, synth, upload_u16
from test_gpu_parity import make_gpu_volume
from test_oracle_refcu import GOLDEN, bits, make_scene

pytestmark = pytest.mark.gpu
F32 = np.float32


def gpu_frames(sc, cfg, frames):
    intr = Intr(*cfg.intr)
    vol = make_gpu_volume(sc)
    for f in range(frames):
        d = compute_dists(upload_u16(sc.depths[f]), intr)
        torch.cuda.synchronize()
        assert np.array_equal(download_u16(d), sc.dists[f])
        vol.integrate(d, sc.cam_poses[f], intr)
    return vol, intr


def gpu_raycast(vol, sc, cfg, intr, f):
    pts = torch.empty((cfg.rows, cfg.cols, 4), dtype=torch.float32, device="cuda")
    nrm = torch.empty_like(pts)
    vol.raycast(sc.cam_poses[f], intr, pts, nrm)
    dep = torch.empty((cfg.rows, cfg.cols), dtype=torch.int16, device="cuda")
    nrm2 = torch.empty_like(pts)
    vol.raycast(sc.cam_poses[f], intr, dep, nrm2)
    torch.cuda.synchronize()
    return pts.cpu().numpy(), nrm.cpu().numpy(), dep.cpu().numpy().view(np.uint16), nrm2.cpu().numpy()


def test_hip_equals_committed_reference_golden():
    g = np.load(GOLDEN)
    cfg, sc = make_scene(64, rotated=True)
    assert np.array_equal(sc.pose, g["pose"]) and np.array_equal(np.stack(sc.dists), g["dists"])
    vol, intr = gpu_frames(sc, cfg, 3)
    assert np.array_equal(vol.download(), g["volume"])                       # tsdf_volume.cu:51-108 on the CPU vs HIP
    p, n, d, _ = gpu_raycast(vol, sc, cfg, intr, 2)
    assert np.array_equal(bits(p), g["points_bits"]) and np.array_equal(bits(n), g["normals_bits"])
    assert np.array_equal(d, g["depth"])
    cloud = vol.fetchCloud()
    assert int(cloud.shape[0]) == int(g["cloud_count"])


@pytest.mark.skipif(not O.have_refcu(), reason="oracle/_ref/libdfref_cu.so did not travel")
def test_hip_equals_reference_kernels_live_128():
    cfg, sc = make_scene(128, rotated=True, cols=640, rows=480)
    vol, intr = gpu_frames(sc, cfg, 3)
    ref = sc.new_volume()
    for f in range(3):
        O.refcu_integrate(sc.dists[f], sc.ovol(ref), synth.aff12(sc.vol2cam(f)), sc.intr)
    assert (ref >> 16).max() == 3
    assert np.array_equal(vol.download(), ref)
    tail = (cfg.cols, cfg.rows, cfg.raycast_step_factor, cfg.gradient_delta_factor)
    for f in (0, 2):
        p, n, d, dn = gpu_raycast(vol, sc, cfg, intr, f)
        rp, rn = O.refcu_raycast_points(sc.ovol(ref), synth.aff12(sc.cam2vol(f)), sc.rinv(f), sc.intr, *tail)
        rd, rdn = O.refcu_raycast_depth(sc.ovol(ref), synth.aff12(sc.cam2vol(f)), sc.rinv(f), sc.intr, *tail)
        assert np.isfinite(rp[..., 0]).sum() > 0.3 * cfg.cols * cfg.rows
        assert np.array_equal(bits(p), bits(rp)) and np.array_equal(bits(n), bits(rn))
        assert np.array_equal(d, rd) and np.array_equal(bits(dn), bits(rdn))
    # FullScan6 + extract_normals_kernel: same point set, same normals
    cloud = vol.fetchCloud()
    normals = vol.fetchNormals(cloud)
    torch.cuda.synchronize()
    rc, count = O.refcu_extract_cloud(sc.ovol(ref), synth.aff12(sc.pose), 1 << 22)
    assert count == cloud.shape[0]
    key = lambda a: np.sort(np.ascontiguousarray(bits(a)[:, :3]).view([("x", "u4"), ("y", "u4"), ("z", "u4")]).reshape(-1), order=("x", "y", "z"))
    c = cloud.cpu().numpy()
    assert np.array_equal(key(c), key(rc))
    rinv = np.linalg.inv(sc.pose[:3, :3].astype(np.float64)).astype(F32)
    rn = O.refcu_extract_normals(sc.ovol(ref), synth.aff12(sc.pose), rinv, c, cfg.gradient_delta_factor)
    assert np.array_equal(bits(normals.cpu().numpy())[:, :3], bits(rn)[:, :3])


@pytest.mark.skipif(not O.have_refcu(), reason="oracle/_ref/libdfref_cu.so did not travel")
def test_hip_equals_reference_kernels_live_512_full_volume():
    """The headline size with nothing in between (VERDICT r2 #1 ii): 640x480 into 512^3 / 3 m, THREE frames of the reference's own
    integrate_kernel (tsdf_volume.cu:51-108, compiled for the host) vs dfusion_integrate -- every one of the 134 M voxels --, then its
    raycast_kernel (Points and Depth variants, :272-405) vs the HIP kernels on that volume."""
    cfg = synth.CONFIGS["512"]
    from scene import Scene
    sc = Scene(cfg, n_frames=3, with_nodes=False)
    vol, intr = gpu_frames(sc, cfg, 3)
    ref = sc.new_volume()
    for f in range(3):
        O.refcu_integrate(sc.dists[f], sc.ovol(ref), synth.aff12(sc.vol2cam(f)), sc.intr)
    assert (ref >> 16).max() == 3 and ((ref >> 16) != 0).sum() > 0.15 * ref.size
    got = vol.download()
    assert np.array_equal(got, ref), "%d of %d voxels differ" % (int((got != ref).sum()), ref.size)
    del got
    tail = (cfg.cols, cfg.rows, cfg.raycast_step_factor, cfg.gradient_delta_factor)
    p, n, d, dn = gpu_raycast(vol, sc, cfg, intr, 2)
    rp, rn = O.refcu_raycast_points(sc.ovol(ref), synth.aff12(sc.cam2vol(2)), sc.rinv(2), sc.intr, *tail)
    rd, rdn = O.refcu_raycast_depth(sc.ovol(ref), synth.aff12(sc.cam2vol(2)), sc.rinv(2), sc.intr, *tail)
    assert np.isfinite(rp[..., 0]).sum() > 0.5 * cfg.cols * cfg.rows
    assert np.array_equal(bits(p), bits(rp)) and np.array_equal(bits(n), bits(rn))
    assert np.array_equal(d, rd) and np.array_equal(bits(dn), bits(rdn))
    # fetchCloud: the reference's FullScan6 runs as fibers on the host (warp-synchronous scan) -- 3 minutes at this size -- so the
    # 512^3 cloud is compared with the restatement, which is pinned to FullScan6 at 64^3 (test_oracle_refcu.py) and live at 128^3 above
    cloud = vol.fetchCloud()
    torch.cuda.synchronize()
    rc, count = O.extract_cloud(sc.ovol(ref), synth.aff12(sc.pose), 1 << 23)
    assert count == cloud.shape[0] and count > 100000
    key = lambda a: np.sort(np.ascontiguousarray(bits(a)[:, :3]).view([("x", "u4"), ("y", "u4"), ("z", "u4")]).reshape(-1), order=("x", "y", "z"))
    assert np.array_equal(key(cloud.cpu().numpy()), key(rc[:count]))


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref/libdfref.so did not travel")
def test_warped_frame_equals_reference_classes_256_full_volume():
    """BASELINE config 1 (256^3 / 1 m, ~500 nodes, k = 4), EVERY voxel: the per-voxel composition driven by the reference's own
    classes -- nanoflann k-NN, WarpField::DQB weights and blend, DualQuaternion::transform, then TsdfIntegrator's arithmetic
    (oracle/ref_glue.cpp ref_integrate_warped: the reference's headers, unmodified) -- vs dfusion_integrate_warped, two frames."""
    from dynamicfusion_amd import WarpField
    from scene import Scene
    cfg = synth.CONFIGS["256"]
    sc = Scene(cfg, n_frames=2)
    intr = Intr(*cfg.intr)
    vol = make_gpu_volume(sc)
    wf = WarpField(k=cfg.k)
    wf.init(sc.pos, sigma=sc.sigma, transforms=sc.dqs[0])
    ref = sc.new_volume()
    Z = cfg.dims[2]
    for f in range(2):
        wf.set_transforms(torch.from_numpy(sc.dqs[f]).cuda())
        vol.integrate_warped(upload_u16(sc.dists[f]), sc.cam_poses[f], intr, wf)
        O.ref_integrate_warped(sc.dists[f], ref, cfg.dims, sc.vs, sc.trunc, cfg.max_weight, synth.aff12(sc.pose), synth.aff12(sc.world2cam(f)),
                               sc.intr, sc.pos, sc.dqs[f], sc.sigma, cfg.k, 0, 0, Z)
    got = vol.download()
    assert (ref >> 16).max() == 2 and ((ref >> 16) != 0).sum() > 0.1 * ref.size
    assert np.array_equal(got, ref), "%d of %d voxels differ" % (int((got != ref).sum()), ref.size)


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref/libdfref.so did not travel")
def test_warped_frame_equals_reference_classes_512_full_volume():
    """The HEADLINE workload (512^3 / 3 m, 2000 nodes, k = 8, 640x480), EVERY voxel (VERDICT r4 #4 i; until round 5 this comparison lived
    only in bench.py's cpu_baseline leg): the per-voxel composition through the reference's own nanoflann, WarpField::DQB and
    DualQuaternion classes (oracle/ref_glue.cpp ref_integrate_warped, OpenMP) vs dfusion_integrate_warped -- the cached, culled,
    pipelined sweep the bench times -- on one frame; ~6 s of the box's host cores."""
    from dynamicfusion_amd import WarpField
    from scene import Scene
    cfg = synth.CONFIGS["512"]
    sc = Scene(cfg, n_frames=1)
    intr = Intr(*cfg.intr)
    vol = make_gpu_volume(sc)
    wf = WarpField(k=cfg.k)
    wf.init(sc.pos, sigma=sc.sigma, transforms=sc.dqs[0])
    n_upd = torch.zeros(1, dtype=torch.int64, device="cuda")
    vol.integrate_warped(upload_u16(sc.dists[0]), sc.cam_poses[0], intr, wf, n_updated=n_upd)
    ref = sc.new_volume()
    O.ref_integrate_warped(sc.dists[0], ref, cfg.dims, sc.vs, sc.trunc, cfg.max_weight, synth.aff12(sc.pose), synth.aff12(sc.world2cam(0)),
                           sc.intr, sc.pos, sc.dqs[0], sc.sigma, cfg.k, 0, 0, cfg.dims[2])
    got = vol.download()
    n_ref = int(((ref >> 16) != 0).sum())
    assert n_ref > 0.15 * ref.size and n_ref == int(n_upd.item())
    assert np.array_equal(got, ref), "%d of %d voxels differ" % (int((got != ref).sum()), ref.size)


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref/libdfref.so did not travel")
def test_warped_sweep_at_the_reference_node_count_equals_reference_classes_512_full_volume():
    """The reference's own node count: WarpField::init keeps every 50th point of the first cloud
    (/root/reference/kfusion/src/warp_field.cpp:41-63) -- ~7 700 nodes for a 640 x 480 frame, four times the headline's density and past
    what an LDS node table holds (5120).  512^3 geometry, M = 8000, k = 8, three frames of a moving camera with changing transforms
    through the product's default path (on-demand tables, look-ahead builds, block models on the side stream): EVERY voxel equals the
    per-voxel composition through the reference's nanoflann / WarpField::DQB / DualQuaternion classes (oracle/ref_glue.cpp,
    arithmetic of warp_field.cpp:203-241), and on the third frame -- the first that can find models -- most kept blocks are swept
    from 4-bit codes (the sub-block unions: a whole block's union fits 16 nodes for one block in four at this density)."""
    from dynamicfusion_amd import WarpField
    from scene import Scene
    base = synth.CONFIGS["512"]
    cfg = synth.Config(base.dims[0], base.size, cols=base.cols, rows=base.rows, nodes=8000, k=8)
    frames = 3
    sc = Scene(cfg, n_frames=frames)
    assert sc.pos.shape[0] == 8000
    intr = Intr(*cfg.intr)
    vol = make_gpu_volume(sc)
    wf = WarpField(k=cfg.k)
    wf.init(sc.pos, sigma=sc.sigma, transforms=sc.dqs[0])
    n_upd = torch.zeros(1, dtype=torch.int64, device="cuda")
    ref = sc.new_volume()
    nl = cfg.dims[2] // 8
    kept = coded = 0
    for f in range(frames):
        wf.set_transforms(torch.from_numpy(sc.dqs[f]).cuda())
        vol.integrate_warped(upload_u16(sc.dists[f]), sc.cam_poses[f], intr, wf, n_updated=n_upd, prefetch="steady")
        a = torch.zeros(nl, dtype=torch.int64, device="cuda"); c = torch.zeros_like(a)
        wf.alive_blocks_per_layer(vol, a); wf.coded_blocks_per_layer(vol, c)
        kept, coded = int(a.sum().item()), int(c.sum().item())
        O.ref_integrate_warped(sc.dists[f], ref, cfg.dims, sc.vs, sc.trunc, cfg.max_weight, synth.aff12(sc.pose), synth.aff12(sc.world2cam(f)),
                               sc.intr, sc.pos, sc.dqs[f], sc.sigma, cfg.k, 0, 0, cfg.dims[2])
    got = vol.download()
    n_ref = int((ref >> 16).sum())
    print("M = 8000: kept blocks %d, coded %d (%.1f %%), updates %d" % (kept, coded, 100.0 * coded / max(kept, 1), n_ref))
    assert n_ref > 0.3 * ref.size and n_ref == int(n_upd.item())
    assert np.array_equal(got, ref), "%d of %d voxels differ" % (int((got != ref).sum()), ref.size)
    assert coded > 0.5 * kept > 0


@pytest.mark.skipif(not O.have_refcu(), reason="oracle/_ref/libdfref_cu.so did not travel")
def test_fetch_cloud_equals_reference_fullscan6_on_a_512x512x64_slab():
    """VERDICT r4 #4 iii: FullScan6 at the headline plane size inside the driver's run.  The reference's extract_kernel +
    extract_normals_kernel (tsdf_volume.cu:511-795, compiled for the host: a block's threads run as fibers -- 104 s for all of 512^3,
    which is why the whole-volume run is tools/fullscan6_512.py, not a test) run on the 64 planes of the fused
    512^3 volume that hold the most surface, taken as a 512 x 512 x 64 volume in its own right (same voxel size, same pose) -- and so does
    dfusion_extract_cloud / _normals: same count, same point set, same normals.  13 s."""
    from dynamicfusion_amd import TsdfVolume
    from scene import Scene
    cfg = synth.CONFIGS["512"]
    sc = Scene(cfg, n_frames=2, with_nodes=False)
    vol, intr = gpu_frames(sc, cfg, 2)
    X, Y, Z = cfg.dims
    ZS = 64
    data = vol.data()                                                   # [Z, Y, X] packed voxels on the device
    near = ((data >> 16) != 0) & ((data & 0xFFFF) != 0x3C00)            # fused and not saturated: where the zero crossings are
    per_slab = near.view(Z // ZS, -1).sum(1)
    z0 = int(per_slab.argmax().item()) * ZS
    sub = TsdfVolume((X, Y, ZS))
    sub.setSize([cfg.size, cfg.size, cfg.size * ZS / Z]); sub.setTruncDist(cfg.trunc_dist); sub.setMaxWeight(cfg.max_weight); sub.setPose(sc.pose)
    sub.setGradientDeltaFactor(cfg.gradient_delta_factor)
    assert np.array_equal(np.asarray(sub.getVoxelSize(), F32), np.asarray(sc.vs, F32)) and sub.getTruncDist() == vol.getTruncDist()
    sub.data().copy_(data[z0:z0 + ZS])
    cloud = sub.fetchCloud()
    normals = sub.fetchNormals(cloud)
    torch.cuda.synchronize()
    host = sub.download()
    ov = O.make_volume(host, (X, Y, ZS), sc.vs, sc.trunc, cfg.max_weight)
    rc, count = O.refcu_extract_cloud(ov, synth.aff12(sc.pose), 1 << 22)
    assert count == cloud.shape[0] and count > 20000, (count, cloud.shape[0], z0)
    key = lambda a: np.sort(np.ascontiguousarray(bits(a)[:, :3]).view([("x", "u4"), ("y", "u4"), ("z", "u4")]).reshape(-1), order=("x", "y", "z"))
    c = cloud.cpu().numpy()
    assert np.array_equal(key(c), key(rc[:count]))
    rinv = np.linalg.inv(sc.pose[:3, :3].astype(np.float64)).astype(F32)
    step = max(1, count // 20000)
    rn = O.refcu_extract_normals(ov, synth.aff12(sc.pose), rinv, c[::step], cfg.gradient_delta_factor)
    assert np.array_equal(bits(normals.cpu().numpy()[::step])[:, :3], bits(rn)[:, :3])
