"""Edge cases of the hot path on the GPU against the oracle: empty inputs, all-invalid / saturated depth, weight saturation,
ragged (odd) image sizes, volumes that are not multiples of the tile sizes, argument validation."""
import numpy as np
import pytest
import torch

import oracle_lib as O
from dynamicfusion_amd import Intr, TsdfVolume, WarpField, capi, compute_dists, download_u16, synth, upload_u16
from scene import Scene, compare_volumes
from test_gpu_parity import make_gpu_volume, make_gpu_warp

pytestmark = pytest.mark.gpu
F32 = np.float32
DF_E_INVALID = 100001


def test_empty_point_sets_are_no_ops():
    sc = Scene(synth.Config(32, 1.0, cols=64, rows=48, nodes=20, k=4), n_frames=1)
    wf = make_gpu_warp(sc)
    z3 = torch.zeros((0, 3), dtype=torch.float32, device="cuda")
    L = capi.lib()
    assert L.dfusion_knn(wf.handle, 4, z3.data_ptr(), 0, None, None, None) in (0, DF_E_INVALID)      # nothing to write
    idx, d2 = wf.KNN(torch.zeros((1, 3), device="cuda"))
    assert idx.shape == (1, 4)
    assert L.dfusion_warp_points(wf.handle, 4, torch.zeros((1, 3), device="cuda").data_ptr(), None, 0, capi.floats(synth.aff12(np.eye(4, dtype=F32))), None) == 0
    d = upload_u16(sc.dists[0])
    assert L.dfusion_project_and_remove(d.data_ptr(), 128, d.clone().data_ptr(), 128, 64, 48, torch.zeros((1, 4), device="cuda").data_ptr(), 0,
                                        capi.floats(sc.intr), None, None, None) == 0
    vol = make_gpu_volume(sc)
    vol.integrate(d, sc.cam_poses[0], Intr(*sc.cfg.intr))
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    buf = torch.zeros((1, 4), dtype=torch.float32, device="cuda")
    capi.check(L.dfusion_extract_cloud(vol.c_volume(), None, capi.floats(synth.aff12(sc.pose)), buf.data_ptr(), 0, cnt.data_ptr(), None))
    _, n_ref = O.extract_cloud(sc.ovol(vol.download()), synth.aff12(sc.pose), 1 << 20)
    assert int(cnt.item()) == n_ref > 0 and not buf.any()                         # capacity 0: counted, nothing written


def test_all_invalid_and_saturated_depth():
    cfg = synth.Config(32, 1.0, cols=64, rows=48, nodes=20, k=4)
    sc = Scene(cfg, n_frames=1)
    intr = Intr(*cfg.intr)
    for depth in (np.zeros((48, 64), np.uint16), np.full((48, 64), 65535, np.uint16)):
        g = download_u16(compute_dists(upload_u16(depth), intr))
        assert np.array_equal(g, O.compute_dists(depth, sc.intr))                 # 65535 mm * lambda overflows half -> inf bits, same on both
        vol = make_gpu_volume(sc)
        n = torch.zeros(1, dtype=torch.int64, device="cuda")
        vol.integrate(upload_u16(g), sc.cam_poses[0], intr, n_updated=n)
        ref = sc.new_volume()
        n_ref = O.integrate(g, ref, sc.ovol(ref), synth.aff12(sc.vol2cam(0)), sc.intr)
        assert int(n.item()) == n_ref and compare_volumes(vol.download(), ref)["bits_mismatch"] == 0
        wf = make_gpu_warp(sc)
        vol.integrate_warped(upload_u16(g), sc.cam_poses[0], intr, wf)
        O.integrate_warped(g, ref, sc.ovol(ref), synth.aff12(sc.pose), synth.aff12(sc.world2cam(0)), sc.intr, sc.pos, sc.dqs[0], sc.sigma, cfg.k)
        assert compare_volumes(vol.download(), ref)["bits_mismatch"] == 0


def test_weight_saturates_at_max_weight():
    cfg = synth.Config(32, 1.0, cols=64, rows=48, nodes=20, k=4)
    sc = Scene(cfg, n_frames=1)
    intr = Intr(*cfg.intr)
    vol = make_gpu_volume(sc); vol.setMaxWeight(3)
    ref = sc.new_volume(); ov = O.make_volume(ref, cfg.dims, sc.vs, sc.trunc, 3)
    wf = make_gpu_warp(sc)
    d = upload_u16(sc.dists[0])
    for _ in range(3):
        vol.integrate(d, sc.cam_poses[0], intr)
        O.integrate(sc.dists[0], ref, ov, synth.aff12(sc.vol2cam(0)), sc.intr)
    for _ in range(3):
        vol.integrate_warped(d, sc.cam_poses[0], intr, wf)
        O.integrate_warped(sc.dists[0], ref, ov, synth.aff12(sc.pose), synth.aff12(sc.world2cam(0)), sc.intr, sc.pos, sc.dqs[0], sc.sigma, cfg.k)
    got = vol.download()
    assert compare_volumes(got, ref)["bits_mismatch"] == 0 and int((got >> 16).max()) == 3


@pytest.mark.parametrize("dims,cols,rows", [((36, 20, 44), 67, 45), ((64, 40, 24), 33, 31)], ids=["36x20x44", "64x40x24"])
def test_ragged_volumes_and_images(dims, cols, rows):
    """Volumes that are not multiples of the sweep tile (32 x 16 x 8) or the 8^3 brick, odd image sizes."""
    cfg = synth.Config(dims, 1.0, cols=cols, rows=rows, nodes=30, k=8)
    sc = Scene(cfg, n_frames=2)
    intr = Intr(*cfg.intr)
    vol = make_gpu_volume(sc)
    wf = make_gpu_warp(sc)
    ref = sc.new_volume()
    for f in range(2):
        wf.set_transforms(torch.from_numpy(sc.dqs[f]).cuda())
        vol.integrate_warped(upload_u16(sc.dists[f]), sc.cam_poses[f], intr, wf)
        O.integrate_warped(sc.dists[f], ref, sc.ovol(ref), synth.aff12(sc.pose), synth.aff12(sc.world2cam(f)), sc.intr, sc.pos, sc.dqs[f], sc.sigma, cfg.k)
    assert compare_volumes(vol.download(), ref)["bits_mismatch"] == 0
    pts = torch.empty((rows, cols, 4), dtype=torch.float32, device="cuda"); nrm = torch.empty_like(pts)
    vol.raycast(sc.cam_poses[1], intr, pts, nrm)
    rp, rn, _, _ = O.raycast_points(sc.ovol(ref), synth.aff12(sc.cam2vol(1)), sc.rinv(1), sc.reproj, cols, rows, cfg.raycast_step_factor,
                                    cfg.gradient_delta_factor)
    assert np.array_equal(pts.cpu().numpy().view(np.uint32), rp.view(np.uint32)) and np.array_equal(nrm.cpu().numpy().view(np.uint32), rn.view(np.uint32))
    cloud = vol.fetchCloud().cpu().numpy()
    rc, n = O.extract_cloud(sc.ovol(ref), synth.aff12(sc.pose), 1 << 20)
    order = lambda a: a[np.lexsort(np.ascontiguousarray(a).view(np.uint32).T[::-1])]
    assert n == len(cloud) and np.array_equal(order(cloud).view(np.uint32), order(rc).view(np.uint32))


def test_argument_validation():
    L = capi.lib()
    with pytest.raises(ValueError):
        TsdfVolume((30, 32, 32))                                                   # dims[0] % 4
    v = TsdfVolume((32, 32, 32))
    bad = capi.DfVolume(None, (capi.C.c_int * 3)(32, 32, 32), (capi.C.c_float * 3)(0.01, 0.01, 0.01), 0.03, 64)
    assert L.dfusion_clear(bad, None, None) == DF_E_INVALID                        # null data
    d = torch.zeros((8, 8), dtype=torch.int16, device="cuda")
    assert L.dfusion_compute_dists(d.data_ptr(), 16, d.data_ptr(), 16, 0, 8, Intr(1, 1, 0, 0).as_proj(), None) == DF_E_INVALID
    slab = capi.DfSlab(8, 8, 4, 8)                                                 # own range outside the stored range
    assert L.dfusion_clear(v.c_volume(), capi.C.byref(slab), None) == DF_E_INVALID
    wf = WarpField(k=4); wf.init(np.zeros((4, 3), F32), sigma=1.0)
    assert L.dfusion_warp_set_nodes(wf.handle, d.data_ptr(), d.data_ptr(), d.data_ptr(), 70000, None) == DF_E_INVALID   # ids are 16-bit
    assert L.dfusion_warp_solve_data_term(wf.handle, 4, d.data_ptr(), d.data_ptr(), 0, 10, 0.0, None, None, None) == DF_E_INVALID
    assert L.dfusion_icp_estimate(None, 1, 0, Intr(1, 1, 0, 0).as_proj(), 0.01, 0.9, d.data_ptr(), d.data_ptr(), None) == DF_E_INVALID
    assert L.dfusion_error_string(DF_E_INVALID) and L.dfusion_error_string(100002) and L.dfusion_abi_version() == 7


@pytest.mark.parametrize("k", [1, 2, 3, 5, 6, 7])
def test_every_neighbour_count_matches_oracle(k):
    """k is a runtime parameter here (KNN_NEIGHBOURS is a compile-time 8 in the reference): the kernels are instantiated for 1..8;
    4 and 8 are covered everywhere else, the rest here -- k-NN, point warp, warped integrate (cached and lean paths)."""
    cfg = synth.Config(48, 1.0, cols=96, rows=72, nodes=40, k=k)
    sc = Scene(cfg, n_frames=1)
    intr = Intr(*cfg.intr)
    wf = make_gpu_warp(sc)
    rng = np.random.default_rng(k)
    q = (sc.pose[:3, 3] + rng.uniform(0, cfg.size, (3000, 3))).astype(F32)
    idx, d2 = wf.KNN(torch.from_numpy(q).cuda())
    ri, rd = O.knn(sc.pos, q, k)
    assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(d2.cpu().numpy().view(np.uint32), rd.view(np.uint32))
    p = torch.from_numpy(q).cuda(); wf.warp(p)
    rp, _ = O.warp_points(sc.pos, sc.dqs[0], sc.sigma, q, None, k)
    assert np.array_equal(p.cpu().numpy().view(np.uint32), rp.view(np.uint32))
    ref = sc.new_volume()
    O.integrate_warped(sc.dists[0], ref, sc.ovol(ref), synth.aff12(sc.pose), synth.aff12(sc.world2cam(0)), sc.intr, sc.pos, sc.dqs[0], sc.sigma, k)
    d = upload_u16(sc.dists[0])
    for kw in ({}, dict(use_table=False), dict(use_weights=False), dict(use_lds=False)):
        vol = make_gpu_volume(sc)
        vol.integrate_warped(d, sc.cam_poses[0], intr, wf, **kw)
        assert compare_volumes(vol.download(), ref)["bits_mismatch"] == 0, kw


@pytest.mark.parametrize("k,sigma", [(8, 0.012), (4, 0.02), (8, 0.03)])
def test_vanishing_blend_weights_match_oracle(k, sigma):
    """Small sigma: most of the volume is many sigma away from every node, so the sweep meets, from the nodes outwards, normal blend
    sums, sums whose squares are denormal (the rotation comes out un-normalised and the near-unit short form must step aside), and
    sums whose squares vanish (norm 0 -> the reference divides by zero -> NaN position -> no update; whole tiles of those are skipped
    from the table build's weight bound).  The whole volume against the oracle, with and without the skip, update counts included."""
    cfg = synth.Config(64, 1.0, cols=96, rows=72, nodes=60, k=k)
    sc = Scene(cfg, n_frames=2)
    sc.sigma = np.full_like(sc.sigma, sigma)
    intr = Intr(*cfg.intr)
    wf = WarpField(k=k)
    wf.init(sc.pos, sigma=sc.sigma, transforms=sc.dqs[0])
    ref = sc.new_volume()
    vols = [make_gpu_volume(sc) for _ in range(3)]
    kws = [dict(), dict(zero_skip=False), dict(cull=False, pipelined=False)]
    n_ref = 0
    n = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in vols]
    for f in range(2):
        wf.set_transforms(torch.from_numpy(sc.dqs[f]).cuda())
        n_ref += O.integrate_warped(sc.dists[f], ref, sc.ovol(ref), synth.aff12(sc.pose), synth.aff12(sc.world2cam(f)), sc.intr, sc.pos, sc.dqs[f],
                                    sc.sigma, k)
        for v, kw, c in zip(vols, kws, n):
            v.integrate_warped(upload_u16(sc.dists[f]), sc.cam_poses[f], intr, wf, n_updated=c, **kw)
    w = ref >> 16
    assert 0 < int((w > 0).sum()) < ref.size // 2            # some voxels near the nodes update, most of the volume cannot
    for v, kw, c in zip(vols, kws, n):
        assert compare_volumes(v.download(), ref)["bits_mismatch"] == 0, kw
        assert int(c.item()) == n_ref, kw


@pytest.mark.parametrize("k", [8, 4])
def test_rotated_volume_pose_matches_oracle(k):
    """The reference's volume pose is a pure translation (kinfu.cpp:27,67) and the sweep has a shortcut for exactly that case
    (R = I: R * p is p itself); a rotated volume goes through the general path -- brick index, per-voxel tables, cull, warped and
    rigid integrate, ray-cast and extraction, all against the oracle."""
    cfg = synth.Config(64, 1.0, cols=96, rows=72, nodes=60, k=k)
    sc = Scene(cfg, n_frames=2)
    centre = (0.0, 0.0, 0.5 + cfg.size / 2)
    sc.pose = synth.affine_mul(synth.rot_y_about(0.21, centre), cfg.volume_pose)
    intr = Intr(*cfg.intr)
    wf = make_gpu_warp(sc)
    ref_w, ref_r = sc.new_volume(), sc.new_volume()
    vw, vw2, vr = make_gpu_volume(sc), make_gpu_volume(sc), make_gpu_volume(sc)
    for f in range(2):
        wf.set_transforms(torch.from_numpy(sc.dqs[f]).cuda())
        d = upload_u16(sc.dists[f])
        O.integrate_warped(sc.dists[f], ref_w, sc.ovol(ref_w), synth.aff12(sc.pose), synth.aff12(sc.world2cam(f)), sc.intr, sc.pos, sc.dqs[f], sc.sigma, k)
        O.integrate(sc.dists[f], ref_r, sc.ovol(ref_r), synth.aff12(sc.vol2cam(f)), sc.intr)
        vw.integrate_warped(d, sc.cam_poses[f], intr, wf)
        vw2.integrate_warped(d, sc.cam_poses[f], intr, wf, cull=False, pipelined=False)
        vr.integrate(d, sc.cam_poses[f], intr)
    assert int((ref_w >> 16).max()) == 2
    assert compare_volumes(vw.download(), ref_w)["bits_mismatch"] == 0
    assert compare_volumes(vw2.download(), ref_w)["bits_mismatch"] == 0
    assert compare_volumes(vr.download(), ref_r)["bits_mismatch"] == 0
    pts = torch.empty((cfg.rows, cfg.cols, 4), dtype=torch.float32, device="cuda"); nrm = torch.empty_like(pts)
    vr.raycast(sc.cam_poses[1], intr, pts, nrm)
    rp, rn, _, stats = O.raycast_points(sc.ovol(ref_r), synth.aff12(sc.cam2vol(1)), sc.rinv(1), sc.reproj, cfg.cols, cfg.rows,
                                        cfg.raycast_step_factor, cfg.gradient_delta_factor)
    gp, gn = pts.cpu().numpy(), nrm.cpu().numpy()
    assert stats[1] > 0 and np.array_equal(np.isnan(gp), np.isnan(rp))
    m = np.isfinite(rp)
    assert np.abs(gp[m] - rp[m]).max() <= 1e-4 and np.abs(gn[m] - rn[m]).max() <= 1e-3


@pytest.mark.parametrize("k,lean", [(8, False), (4, False), (8, True)])
def test_depth_footprint_cull_with_near_occluders_matches_oracle(k, lean):
    """The warped sweep skips a wave's 8x8x8 voxels when nothing in them can update; one of the tests compares the least distance
    from the camera centre any warped voxel of the tile can have with the LARGEST dists value over the pixels the tile can project to
    (max-pyramid of the frame's dists).  Depth images with near occluders, invalid (zero) regions and a far rim make that test fire in
    every combination: the whole volume against the oracle, and against the sweep without the pyramid / without any cull, update
    counts included.  Large node rotations, so that the lateral bound (rotation about the far world origin) is loose while the
    distance bound stays tight."""
    cfg = synth.Config(64, 1.0, cols=96, rows=72, nodes=60, k=k)
    sc = Scene(cfg, n_frames=3)
    sc.dqs = [synth.node_transforms(cfg, f, rot_amp=0.12, trans_amp=0.02) for f in range(3)]
    rng = np.random.default_rng(5)
    for f in range(3):
        d = sc.depths[f].copy()
        d[10:40, 20:60] = 620 + 40 * f                      # a near slab in front of the scene (0.62 m: just inside the volume)
        d[50:72, 0:30] = 0                                  # no measurement
        d[0:8, :] = 1450                                    # far rim
        d[rng.random(d.shape) < 0.01] = 0
        sc.depths[f] = d
        sc.dists[f] = O.compute_dists(d, sc.intr)
    intr = Intr(*cfg.intr)
    wf = WarpField(k=k, voxel_table=not lean)
    wf.init(sc.pos, sigma=sc.sigma, transforms=sc.dqs[0])
    ref = sc.new_volume()
    kws = [dict(), dict(depth_pyramid=False), dict(cull=False), dict(pipelined=False)]
    vols = [make_gpu_volume(sc) for _ in kws]
    n = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in kws]
    n_ref = 0
    for f in range(3):
        wf.set_transforms(torch.from_numpy(sc.dqs[f]).cuda())
        n_ref += O.integrate_warped(sc.dists[f], ref, sc.ovol(ref), synth.aff12(sc.pose), synth.aff12(sc.world2cam(f)), sc.intr, sc.pos, sc.dqs[f],
                                    sc.sigma, k)
        for v, kw, c in zip(vols, kws, n):
            v.integrate_warped(upload_u16(sc.dists[f]), sc.cam_poses[f], intr, wf, n_updated=c, **kw)
    assert int((ref >> 16).max()) == 3 and 0 < n_ref
    for v, kw, c in zip(vols, kws, n):
        assert compare_volumes(v.download(), ref)["bits_mismatch"] == 0, kw
        assert int(c.item()) == n_ref, kw


def test_rigid_scratch_is_kept_per_stream_and_can_be_released():
    """dfusion_integrate keeps its plan / pyramid scratch per (device, stream) (round 3: the stream-ordered allocator gave wrong plans in
    processes that also hipMalloc / hipFree between calls); dfusion_release_scratch frees it, the next call allocates again, on a
    second stream a second buffer is used -- same volume every time."""
    from dynamicfusion_amd import capi
    cfg = synth.Config(64, 1.0, cols=160, rows=120, nodes=0, k=4)
    intr = Intr(*cfg.intr)
    d = compute_dists(upload_u16(synth.depth_frame(cfg, 0)), intr)
    outs = []
    side = torch.cuda.Stream()
    for i in range(4):
        v = TsdfVolume(cfg.dims); v.setSize([cfg.size] * 3); v.setTruncDist(cfg.trunc_dist); v.setMaxWeight(cfg.max_weight); v.setPose(cfg.volume_pose); v.clear()
        torch.cuda.synchronize()
        if i == 1: capi.check(capi.lib().dfusion_release_scratch())
        if i == 3:
            with torch.cuda.stream(side):
                v.integrate(d, synth.camera_pose(cfg, 0), intr)
            side.synchronize()
        else:
            v.integrate(d, synth.camera_pose(cfg, 0), intr)
        outs.append(v.download())
    assert (outs[0] >> 16).max() == 1
    for o in outs[1:]:
        assert np.array_equal(outs[0], o)
    capi.check(capi.lib().dfusion_release_scratch())


def test_rigid_scratch_cache_is_bounded_and_release_keeps_the_current_device():
    """a host that makes a stream per frame must not accumulate one ~84 MB scratch per stream (the cache holds 8 (device,
    stream) pairs, least recently used evicted), cached handles of DESTROYED streams must not be touched by the release, and the release
    must leave the caller's current device as it was.  Twelve short-lived streams, the same volume from each."""
    from dynamicfusion_amd import capi
    cfg = synth.Config(64, 1.0, cols=160, rows=120, nodes=0, k=4)
    intr = Intr(*cfg.intr)
    d = compute_dists(upload_u16(synth.depth_frame(cfg, 0)), intr)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    ref = None
    for i in range(12):
        s = torch.cuda.Stream()
        v = TsdfVolume(cfg.dims); v.setSize([cfg.size] * 3); v.setTruncDist(cfg.trunc_dist); v.setMaxWeight(cfg.max_weight); v.setPose(cfg.volume_pose); v.clear()
        torch.cuda.synchronize()
        with torch.cuda.stream(s):
            v.integrate(d, synth.camera_pose(cfg, 0), intr)
        s.synchronize()
        out = v.download()
        ref = out if ref is None else ref
        assert np.array_equal(out, ref)
        del s, v                                             # the stream handle the cache remembers may now be gone
    dev = torch.cuda.current_device()
    capi.check(capi.lib().dfusion_release_scratch())
    assert torch.cuda.current_device() == dev
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    assert torch.cuda.mem_get_info()[0] >= free0 - (64 << 20)              # nothing of the twelve scratches is left behind
    v = TsdfVolume(cfg.dims); v.setSize([cfg.size] * 3); v.setTruncDist(cfg.trunc_dist); v.setMaxWeight(cfg.max_weight); v.setPose(cfg.volume_pose); v.clear()
    v.integrate(d, synth.camera_pose(cfg, 0), intr)
    assert np.array_equal(v.download(), ref)
    capi.check(capi.lib().dfusion_release_scratch())


def test_march_addressing_generic_form_equals_the_32_bit_one_on_a_tall_volume():
    """The march makes its voxel addresses in 32 bits where the stored planes allow it (z_store_n * Y <= 2^24 and <= 2^30 voxels:
    dfusion_raycast.hip rc_march_addr) and in the generic 64-bit form elsewhere.  A 4 x 4096 x 8192 volume (512 MiB) is past the
    first limit as a whole and inside it as four Z-slabs: the unsharded cast (generic form) and the slab casts merged the way the
    collectives merge them (32-bit form) must agree bit for bit."""
    from dynamicfusion_amd import sharded
    dims = (4, 4096, 8192)
    X, Y, Z = dims
    cols, rows = 160, 120
    intr = Intr(142.6, 142.6, 80.0, 60.0)

    def mkvol(slab=None):
        v = TsdfVolume(dims, slab=slab)
        v.setSize([0.2, 2.0, 2.0]); v.setTruncDist(0.04); v.setMaxWeight(64)
        pose = np.eye(4, dtype=np.float32); pose[:3, 3] = [-0.1, -1.0, 0.5]
        v.setPose(pose)
        return v
    full = mkvol()
    # a surface whose depth varies with y: +1 (weight 1) in front of plane Z/2 + 8 * (y % 64), -1 behind it
    POS, NEG = (1 << 16) | 0x3c00, (1 << 16) | 0xbc00
    zz = torch.arange(Z, device="cuda", dtype=torch.int32)[:, None, None]
    yy = torch.arange(Y, device="cuda", dtype=torch.int32)[None, :, None]
    data = full.data()
    data.copy_(torch.where(zz >= Z // 2 + 8 * (yy % 64), torch.tensor(NEG, dtype=torch.int32, device="cuda"),
                           torch.tensor(POS, dtype=torch.int32, device="cuda")).expand(Z, Y, X))
    cam = np.eye(4, dtype=np.float32)
    p0 = torch.empty((rows, cols, 4), dtype=torch.float32, device="cuda"); n0 = torch.empty_like(p0)
    k0 = torch.empty((rows, cols), dtype=torch.int32, device="cuda")
    full.raycast(cam, intr, p0, n0, keys=k0)
    hits = int(((k0.view(torch.int32) != -1) & ((k0 & 1) == 1)).sum())
    assert hits > 500, hits                                   # (the rays that stay inside the 0.2 m wide volume up to the surface)
    world = 4
    halo = sharded.halo_planes(full.getTruncDist(), full.getRaycastStepFactor(), full.getGradientDeltaFactor(), float(full.getVoxelSize()[2]))
    slabs, k64s = [], []
    for r in range(world):
        z0, zn = sharded.slab_range(Z, r, world)
        v = mkvol(slab=(z0, zn, halo))
        assert (v.z_store_n * Y) <= (1 << 24) < Z * Y        # 32-bit form here, generic form on the whole volume
        v.data().copy_(data[v.z_store0:v.z_store0 + v.z_store_n])
        k64 = torch.empty((rows, cols), dtype=torch.int64, device="cuda")
        v.raycast_march(cam, intr, k64, rank=r)
        slabs.append(v); k64s.append(k64)
    merged = torch.stack(k64s).min(0).values.contiguous()
    ev = torch.where(merged == sharded.KEY_NONE, torch.full_like(merged, 0xFFFFFFFF), (merged >> 39) & 0xFFFFFF).to(torch.int32)
    assert torch.equal(ev, k0), "first events differ"
    acc = torch.zeros((rows, cols, 4), dtype=torch.int32, device="cuda")
    for v in slabs:
        n = torch.empty((rows, cols, 4), dtype=torch.float32, device="cuda")
        v.raycast_shade(cam, intr, merged, None, n)
        acc += n.view(torch.int32)
    p1 = torch.empty_like(p0)
    slabs[0].raycast_points_of_keys(cam, intr, merged, acc.view(torch.float32), p1)
    assert torch.equal(acc, n0.view(torch.int32)), "normals differ"
    assert torch.equal(p1.view(torch.int32), p0.view(torch.int32)), "points differ"
