"""GPU parity tests: the HIP path (through the C-ABI, libdfusion_hip.so) against the CPU oracle on the
same seeded inputs.  Run on the MI355X box with `pytest -m gpu`.

Bar (BASELINE.json north_star / SURVEY.md 8d):
  * integer voxel indexing and weights: bit-exact
  * fused TSDF: |delta| <= 1e-4 after half decode (we additionally require identical half bits on
    >= 99.99 % of voxels and report the rest)
  * k-NN: identical index lists and distances; exact distance ties in nanoflann's tree order (tests/test_gpu_ties.py)
  * ray-cast: identical hit mask, vertex |delta| <= 1e-4 m, normal |delta| <= 1e-3
"""
import os

import numpy as np
import pytest
import torch

import oracle_lib as O
from dynamicfusion_amd import Intr, TsdfVolume, WarpField, capi, compute_dists, download_u16, synth, upload_u16
from scene import Scene, compare_volumes, decode

pytestmark = pytest.mark.gpu
F32 = np.float32

SMALL = synth.Config(64, 1.0, cols=160, rows=120, nodes=100, k=4, name="64^3 small")
MID = synth.Config(128, 1.0, cols=640, rows=480, nodes=500, k=8, name="128^3 mid")


def make_gpu_volume(sc, slab=None):
    cfg = sc.cfg
    vol = TsdfVolume(cfg.dims, slab=slab)
    # NB size before trunc: with KinFu::KinFu's order (kinfu.cpp:102-107: trunc, then size) the clamp of
    # tsdf_volume.cpp:68-73 is applied against the ctor's 3 m default size and never relaxes again; that
    # quirk is kept in the mirror and pinned in tests/test_host_logic.py, but the scenes here want 0.04.
    vol.setSize([cfg.size] * 3)
    vol.setTruncDist(cfg.trunc_dist)
    vol.setMaxWeight(cfg.max_weight)
    vol.setPose(sc.pose)
    vol.setRaycastStepFactor(cfg.raycast_step_factor)
    vol.setGradientDeltaFactor(cfg.gradient_delta_factor)
    assert abs(vol.getTruncDist() - sc.trunc) == 0
    return vol


def make_gpu_warp(sc, f=0, k=None):
    wf = WarpField(k=k or sc.cfg.k)
    wf.init(sc.pos, sigma=sc.sigma, transforms=sc.dqs[f])
    return wf


def assert_volume_parity(gpu_u32, ref_u32, exact=True):
    s = compare_volumes(gpu_u32, ref_u32)
    print("volume parity:", s)
    assert s["weight_mismatch"] == 0 if exact else s["weight_mismatch"] <= 1e-4 * s["n"], s
    if exact:
        assert s["bits_mismatch"] == 0, s
    else:
        assert s["bits_mismatch"] <= 1e-4 * s["n"], s
        assert s["n_dtsdf_gt_1e-4"] <= 1e-4 * s["n"], s
    return s


def test_library_is_the_hip_build():
    assert capi.lib().dfusion_abi_version() == 7
    assert torch.cuda.is_available()


def test_compute_dists_bit_exact():
    sc = Scene(synth.Config(64, 1.0, nodes=0), n_frames=1, with_nodes=False)
    intr = Intr(*sc.cfg.intr)
    d = compute_dists(upload_u16(sc.depths[0]), intr)
    torch.cuda.synchronize()
    assert np.array_equal(download_u16(d), sc.dists[0])


def test_clear_zeroes_every_voxel():
    sc = Scene(SMALL, n_frames=1, with_nodes=False)
    vol = make_gpu_volume(sc)
    vol.data().fill_(0x12345678)
    vol.clear()
    torch.cuda.synchronize()
    assert int(vol.data().abs().max()) == 0


@pytest.mark.parametrize("cfg", [SMALL, synth.CONFIGS["cpu128"]], ids=["64", "128"])
def test_integrate_rigid_bit_exact(cfg):
    sc = Scene(cfg, n_frames=3, with_nodes=False)
    intr = Intr(*cfg.intr)
    vol = make_gpu_volume(sc)
    ref = sc.new_volume()
    n_upd = torch.zeros(1, dtype=torch.int64, device="cuda")
    n_ref = 0
    for f in range(3):
        vol.integrate(upload_u16(sc.dists[f]), sc.cam_poses[f], intr, n_updated=n_upd)
        n_ref += O.integrate(sc.dists[f], ref, sc.ovol(ref), synth.aff12(sc.vol2cam(f)), sc.intr)
    assert_volume_parity(vol.download(), ref)
    assert int(n_upd.item()) == n_ref
    _, w = decode(ref)
    assert w.max() == 3


@pytest.mark.parametrize("cfg", [SMALL, MID], ids=["64", "128"])
def test_integrate_rigid_depth_cull_is_result_identical(cfg):
    """The behind-the-surface skip of the rigid sweep (max-pyramid of dists, conservative per 16-plane sub-chunk) must not change
    a single bit: 3 frames with it, 3 frames without, plus an all-invalid and a one-pixel depth image."""
    sc = Scene(cfg, n_frames=3, with_nodes=False)
    intr = Intr(*cfg.intr)
    L = capi.lib()
    vols = []
    ND, NS, KA = capi.DF_RIGID_NO_DEPTH_CULL, capi.DF_RIGID_NO_SHORT_FORMS, capi.DF_RIGID_KEEP_ALL
    for flags in (0, ND | NS, NS, ND, ND | KA, ND | NS | KA, capi.DF_RIGID_NO_SAT):        # per-call validation switches (dfusion_integrate_ex)
        vol = make_gpu_volume(sc)
        n_upd = torch.zeros(1, dtype=torch.int64, device="cuda")
        for f in range(3):
            vol.integrate(upload_u16(sc.dists[f]), sc.cam_poses[f], intr, n_updated=n_upd, flags=flags)
        empty = np.zeros_like(sc.dists[0])
        vol.integrate(upload_u16(empty), sc.cam_poses[0], intr, n_updated=n_upd, flags=flags)          # Dp == 0 everywhere: nothing updates
        one = empty.copy(); one[cfg.rows // 2, cfg.cols // 2] = sc.dists[0][cfg.rows // 2, cfg.cols // 2]
        vol.integrate(upload_u16(one), sc.cam_poses[0], intr, n_updated=n_upd, flags=flags)
        vols.append((vol.download(), int(n_upd.item())))
    for other in vols[1:]:
        assert np.array_equal(vols[0][0], other[0]) and vols[0][1] == other[1]
    assert (vols[0][0] >> 16).max() == 4


def test_integrate_rigid_slabs_equal_full():
    """Z-slab sharding of the rigid sweep replays vc += zstep and is bit-identical with the full sweep."""
    sc = Scene(SMALL, n_frames=1, with_nodes=False)
    intr = Intr(*SMALL.intr)
    full = make_gpu_volume(sc)
    full.integrate(upload_u16(sc.dists[0]), sc.cam_poses[0], intr)
    parts = []
    Z = SMALL.dims[2]
    for g in range(4):
        v = make_gpu_volume(sc, slab=(g * Z // 4, Z // 4, 0))
        v.integrate(upload_u16(sc.dists[0]), sc.cam_poses[0], intr)
        parts.append(v.download())
    assert np.array_equal(np.concatenate(parts, 0), full.download())


@pytest.mark.parametrize("k", [4, 8])
def test_knn_matches_oracle(k):
    sc = Scene(MID, n_frames=1)
    wf = make_gpu_warp(sc, k=k)
    rng = np.random.RandomState(5)
    q = rng.uniform(-0.6, 0.6, (20000, 3)).astype(F32) + np.array([0, 0, 1.0], F32)
    idx, d2 = wf.KNN(torch.from_numpy(q).cuda(), k)
    torch.cuda.synchronize()
    ridx, rd2 = O.knn(sc.pos, q, k)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    assert np.array_equal(d2.cpu().numpy().view(np.uint32), rd2.view(np.uint32))


def test_knn_reference_fixture_cube_corners():
    """tests/nanoflann_test.cpp:23-49 inputs: 8 cube corners, 5 queries, k = 8: every query is equidistant from groups of corners;
    the order inside a group is nanoflann's tree order on both sides (golden: the reference's own nanoflann)."""
    pos = np.array([[1, 1, 1], [1, 1, -1], [1, -1, 1], [1, -1, -1], [-1, 1, 1], [-1, 1, -1], [-1, -1, 1], [-1, -1, -1]], F32)
    q = np.array([[-1, -1, -1], [0, 0, 0], [1, 1, 1], [2, 2, 2], [3, 3, 3]], F32)
    wf = WarpField(k=8)
    wf.init(pos, sigma=3.0)
    idx, d2 = wf.KNN(torch.from_numpy(q).cuda(), 8)
    ridx, rd2 = O.knn(pos, q, 8)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    assert np.array_equal(d2.cpu().numpy(), rd2)
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "knn.npz"))
    assert np.array_equal(idx.cpu().numpy(), g["corner_idx"])


@pytest.mark.parametrize("sigma_mode", ["spacing", "reference"])
def test_warp_points_matches_oracle(sigma_mode):
    sc = Scene(MID, n_frames=1, sigma_mode=sigma_mode)
    wf = make_gpu_warp(sc)
    rng = np.random.RandomState(6)
    p = (rng.uniform(-0.5, 0.5, (30000, 3)) + np.array([0, 0, 1.0])).astype(F32)
    n = rng.normal(size=(30000, 3)).astype(F32)
    p[17] = np.nan                                       # NaN points are skipped (warp_field.cpp:185)
    live = synth.rot_y_about(0.01, (0, 0, 1))
    wf.setWarpToLive(live)
    pd, nd = torch.from_numpy(p).cuda(), torch.from_numpy(n).cuda()
    wf.warp(pd, nd)
    torch.cuda.synchronize()
    rp, rn = O.warp_points(sc.pos, sc.dqs[0], sc.sigma, p, n, sc.cfg.k, synth.aff12(live))
    gp, gn = pd.cpu().numpy(), nd.cpu().numpy()
    assert np.array_equal(gp.view(np.uint32), rp.view(np.uint32)) and np.array_equal(gn.view(np.uint32), rn.view(np.uint32))


@pytest.mark.parametrize("cfg,sigma_mode", [(SMALL, "spacing"), (SMALL, "reference"), (MID, "spacing")],
                         ids=["64-k4", "64-k4-sigma3", "128-k8"])
def test_integrate_warped_matches_oracle(cfg, sigma_mode):
    sc = Scene(cfg, n_frames=2, sigma_mode=sigma_mode)
    intr = Intr(*cfg.intr)
    vol = make_gpu_volume(sc)
    wf = make_gpu_warp(sc)
    ref = sc.new_volume()
    n_upd = torch.zeros(1, dtype=torch.int64, device="cuda")
    n_ref = 0
    for f in range(2):
        wf.set_transforms(torch.from_numpy(sc.dqs[f]).cuda())
        vol.integrate_warped(upload_u16(sc.dists[f]), sc.cam_poses[f], intr, wf, n_updated=n_upd)
        n_ref += O.integrate_warped(sc.dists[f], ref, sc.ovol(ref), synth.aff12(sc.pose), synth.aff12(sc.world2cam(f)),
                                    sc.intr, sc.pos, sc.dqs[f], sc.sigma, cfg.k)
    assert_volume_parity(vol.download(), ref, exact=True)          # bit for bit (round 1 asserted within 1e-4; every run since has been exact)
    print("n_upd gpu/oracle", int(n_upd.item()), n_ref)
    assert int(n_upd.item()) == n_ref


def test_integrate_warped_cull_is_result_identical():
    sc = Scene(MID, n_frames=1)
    intr = Intr(*MID.intr)
    wf = make_gpu_warp(sc)
    a, b = make_gpu_volume(sc), make_gpu_volume(sc)
    d = upload_u16(sc.dists[0])
    a.integrate_warped(d, sc.cam_poses[0], intr, wf, cull=True)
    b.integrate_warped(d, sc.cam_poses[0], intr, wf, cull=False)
    assert torch.equal(a.data(), b.data())


@pytest.mark.parametrize("cfg", [SMALL, MID], ids=["64-k4", "128-k8"])
def test_voxel_knn_table_equals_on_the_fly_search(cfg):
    """The per-voxel k-NN table cached in HBM (DF_INDEX_VOXEL_TABLE) must give the same volume, bit for bit, as
    re-running the brick-list k-NN every frame."""
    sc = Scene(cfg, n_frames=2)
    intr = Intr(*cfg.intr)
    wf_tab = make_gpu_warp(sc)
    wf_fly = WarpField(k=cfg.k, voxel_table=False)
    wf_fly.init(sc.pos, sigma=sc.sigma, transforms=sc.dqs[0])
    a, b, c, e = make_gpu_volume(sc), make_gpu_volume(sc), make_gpu_volume(sc), make_gpu_volume(sc)
    # every sweep kernel the dispatcher can pick (include/dfusion.h DF_WARP_* validation switches): pipelined (default),
    # batched LDS, global-gather with / without the weight table
    variants = [dict(pipelined=False), dict(use_lds=False), dict(use_lds=False, use_weights=False), dict(zero_skip=False),
                dict(zero_skip=False, cull=False), dict(depth_pyramid=False)]
    others = [make_gpu_volume(sc) for _ in variants]
    for f in range(2):
        d = upload_u16(sc.dists[f])
        for wf in (wf_tab, wf_fly):
            wf.set_transforms(torch.from_numpy(sc.dqs[f]).cuda())
        a.integrate_warped(d, sc.cam_poses[f], intr, wf_tab)
        b.integrate_warped(d, sc.cam_poses[f], intr, wf_fly)
        c.integrate_warped(d, sc.cam_poses[f], intr, wf_tab, use_table=False)
        e.integrate_warped(d, sc.cam_poses[f], intr, wf_tab, use_weights=False)
        for v, kw in zip(others, variants):
            v.integrate_warped(d, sc.cam_poses[f], intr, wf_tab, **kw)
    assert torch.equal(a.data(), b.data()) and torch.equal(a.data(), c.data()) and torch.equal(a.data(), e.data())
    for v in others:
        assert torch.equal(a.data(), v.data())


def test_integrate_warped_identity_nodes_close_to_rigid():
    """Default-constructed node transforms (dual_quaternion.hpp:25-29) give x_w == x_c exactly; the warped
    sweep then differs from the rigid one only by direct-vs-incremental vc rounding."""
    sc = Scene(SMALL, n_frames=1, identity_warp=True)
    intr = Intr(*SMALL.intr)
    wf = make_gpu_warp(sc)
    a, b = make_gpu_volume(sc), make_gpu_volume(sc)
    d = upload_u16(sc.dists[0])
    a.integrate_warped(d, sc.cam_poses[0], intr, wf)
    b.integrate(d, sc.cam_poses[0], intr)
    s = compare_volumes(a.download(), b.download())
    print(s)
    assert s["weight_mismatch"] <= 2e-3 * s["n"]


def _raycast_both(sc, vol, ref, f, want_keys=False, depth_variant=False):
    cfg = sc.cfg
    intr = Intr(*cfg.intr)
    nrm = torch.empty((cfg.rows, cfg.cols, 4), dtype=torch.float32, device="cuda")
    if depth_variant:
        out = torch.empty((cfg.rows, cfg.cols), dtype=torch.int16, device="cuda")
        vol.raycast(sc.cam_poses[f], intr, out, nrm)
        torch.cuda.synchronize()
        rd, rn = O.raycast_depth(sc.ovol(ref), synth.aff12(sc.cam2vol(f)), sc.rinv(f), sc.reproj, cfg.cols, cfg.rows,
                                 cfg.raycast_step_factor, cfg.gradient_delta_factor)
        return download_u16(out), nrm.cpu().numpy(), rd, rn
    pts = torch.empty((cfg.rows, cfg.cols, 4), dtype=torch.float32, device="cuda")
    keys = torch.empty((cfg.rows, cfg.cols), dtype=torch.int32, device="cuda") if want_keys else None
    vol.raycast(sc.cam_poses[f], intr, pts, nrm, keys=keys)
    torch.cuda.synchronize()
    rp, rn, rk, stats = O.raycast_points(sc.ovol(ref), synth.aff12(sc.cam2vol(f)), sc.rinv(f), sc.reproj, cfg.cols, cfg.rows,
                                         cfg.raycast_step_factor, cfg.gradient_delta_factor, want_keys=want_keys)
    return pts.cpu().numpy(), nrm.cpu().numpy(), rp, rn, (keys.cpu().numpy().view(np.uint32) if want_keys else None), rk, stats


def _filled(sc, frames=2):
    """Volume integrated on the CPU (oracle), uploaded to the GPU: isolates ray-cast parity."""
    ref = sc.new_volume()
    for f in range(frames):
        O.integrate(sc.dists[f], ref, sc.ovol(ref), synth.aff12(sc.vol2cam(f)), sc.intr)
    vol = make_gpu_volume(sc)
    vol.upload(ref)
    return vol, ref


@pytest.mark.parametrize("cfg", [SMALL, synth.CONFIGS["cpu128"]], ids=["64", "128"])
def test_raycast_points_matches_oracle(cfg):
    sc = Scene(cfg, n_frames=2, with_nodes=False)
    vol, ref = _filled(sc)
    gp, gn, rp, rn, gk, rk, stats = _raycast_both(sc, vol, ref, 1, want_keys=True)
    assert stats[1] > 0.2 * cfg.cols * cfg.rows          # plenty of hits
    assert np.array_equal(gk, rk)                        # same first event on every ray (integer: exact)
    assert np.array_equal(np.isnan(gp), np.isnan(rp))    # identical hit/miss mask
    m = np.isfinite(rp)
    bit_same = np.array_equal(gp.view(np.uint32), rp.view(np.uint32)) and np.array_equal(gn.view(np.uint32), rn.view(np.uint32))
    print("raycast bit-identical:", bit_same, "max |dv|", np.abs(gp[m] - rp[m]).max(), "max |dn|", np.abs(gn[m] - rn[m]).max())
    assert np.abs(gp[m] - rp[m]).max() <= 1e-4
    assert np.abs(gn[m] - rn[m]).max() <= 1e-3


def test_raycast_from_random_cameras_matches_oracle():
    """Cameras the benchmark never takes: inside the volume, behind it, rolled, looking along an edge or past the volume altogether --
    rays that never enter (tmin >= tmax), enter at t = 0, leave after a step or two, or cross the whole diagonal.  First events,
    points and normals bit for bit (the march's event search only runs for windows with a negative sample, and its voxel addresses
    are clamped: both have to hold for every one of these rays)."""
    cfg = SMALL
    sc = Scene(cfg, n_frames=2, with_nodes=False)
    vol, ref = _filled(sc)
    intr = Intr(*cfg.intr)
    rng = np.random.RandomState(11)
    centre = (sc.pose @ np.array([cfg.size / 2] * 3 + [1.0], np.float32))[:3]
    n_hits = 0
    for i in range(16 + int(os.environ.get("DFUSION_FUZZ_EXTRA", "0"))):          # (profiles/r04_cull_fuzz_extended.txt: one run with 2000 more)
        axis = rng.randn(3); axis /= np.linalg.norm(axis)
        ang = rng.uniform(0.0, [0.3, 1.2, 3.1][i % 3])
        K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
        cam = np.eye(4, dtype=np.float32)
        cam[:3, :3] = R.astype(np.float32)
        # the eye: in front of the volume looking at its centre, or somewhere inside it, or off to a side
        back = R @ np.array([0.0, 0.0, -1.0])
        eye = centre + back * rng.uniform(0.0, 1.6) * cfg.size + (rng.randn(3) * 0.15 * cfg.size if i % 4 == 3 else 0.0)
        cam[:3, 3] = eye.astype(np.float32)
        cam2vol = synth.affine_mul(synth.affine_inv(sc.pose), cam)
        rinv = np.linalg.inv(cam2vol[:3, :3].astype(np.float64)).astype(np.float32)
        pts = torch.empty((cfg.rows, cfg.cols, 4), dtype=torch.float32, device="cuda"); nrm = torch.empty_like(pts)
        keys = torch.empty((cfg.rows, cfg.cols), dtype=torch.int32, device="cuda")
        vol.raycast(cam, intr, pts, nrm, keys=keys)
        rp, rn, rk, stats = O.raycast_points(sc.ovol(ref), synth.aff12(cam2vol), rinv, sc.reproj, cfg.cols, cfg.rows,
                                             cfg.raycast_step_factor, cfg.gradient_delta_factor, want_keys=True)
        assert np.array_equal(keys.cpu().numpy().view(np.uint32), rk), "camera %d: first events differ" % i
        assert np.array_equal(pts.cpu().numpy().view(np.uint32), rp.view(np.uint32)), "camera %d: points differ" % i
        assert np.array_equal(nrm.cpu().numpy().view(np.uint32), rn.view(np.uint32)), "camera %d: normals differ" % i
        n_hits += int(stats[1])
    assert n_hits > 16 * 0.05 * cfg.cols * cfg.rows          # (the cameras do see the surface)
    print("random cameras:", i + 1, "hits", n_hits)


def test_raycast_depth_matches_oracle():
    sc = Scene(SMALL, n_frames=2, with_nodes=False)
    vol, ref = _filled(sc)
    gd, gn, rd, rn = _raycast_both(sc, vol, ref, 1, depth_variant=True)
    assert (rd > 0).sum() > 0.2 * rd.size
    assert np.array_equal(gd, rd)                                             # millimetres, bit for bit
    assert np.array_equal(gn.view(np.uint32), rn.view(np.uint32))


def test_raycast_empty_volume_is_all_nan():
    sc = Scene(SMALL, n_frames=1, with_nodes=False)
    vol = make_gpu_volume(sc)
    pts = torch.zeros((SMALL.rows, SMALL.cols, 4), dtype=torch.float32, device="cuda")
    nrm = torch.zeros_like(pts)
    vol.raycast(sc.cam_poses[0], Intr(*SMALL.intr), pts, nrm)
    assert bool(torch.isnan(pts).all()) and bool(torch.isnan(nrm).all())


def test_errors_are_reported_not_swallowed():
    sc = Scene(SMALL, n_frames=1)
    vol = make_gpu_volume(sc)
    wf = WarpField(k=4)
    wf.init(sc.pos[:2], sigma=1.0)                       # M < k
    with pytest.raises(capi.DfusionError):
        wf.KNN(torch.zeros((4, 3), device="cuda"), 4)
    wf2 = make_gpu_warp(sc)
    rc = capi.lib().dfusion_integrate_warped(
        upload_u16(sc.dists[0]).data_ptr(), SMALL.cols * 2, SMALL.cols, SMALL.rows, vol.c_volume(), None,
        capi.floats(synth.aff12(sc.pose)), capi.floats(synth.aff12(sc.world2cam(0))), Intr(*SMALL.intr).as_proj(),
        wf2.handle, 4, 0, None, None)
    assert rc == 100002                                  # DF_E_NO_INDEX: index not built yet


def test_project_and_remove_and_psdf_match_oracle():
    """device::project_and_remove (tsdf_volume.cu:113-139,163-176) + TsdfVolume::psdf's arithmetic (tsdf_volume.cpp:266-292):
    projected points, removed dists pixels, ro and the inside count are bit-identical with the oracle."""
    cfg = synth.Config(64, 1.0, cols=160, rows=120, nodes=100, k=8)
    sc = Scene(cfg, n_frames=1)
    dists = O.compute_dists(sc.depths[0], sc.intr)
    rng = np.random.default_rng(5)
    n = 40000
    pts = np.zeros((n, 4), F32)
    pts[:, 0] = rng.uniform(-0.9, 0.9, n); pts[:, 1] = rng.uniform(-0.7, 0.7, n); pts[:, 2] = rng.uniform(0.3, 1.6, n)
    pts[::97, 2] = 0.0                       # x/0 -> +-inf / NaN image coordinates => outside
    pts[::101, 1] = np.nan                   # NaN points stay untouched
    pts[::103, 2] = -0.5                     # behind the camera: projects somewhere, reference does not test z
    exp_pts, exp_dists, exp_ro, exp_n = O.project_and_remove(dists, pts, sc.intr)
    d_in = upload_u16(dists); d_out = d_in.clone()
    d_pts = torch.from_numpy(pts).cuda(); d_ro = torch.empty(n, dtype=torch.float32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    capi.check(capi.lib().dfusion_project_and_remove(d_in.data_ptr(), cfg.cols * 2, d_out.data_ptr(), cfg.cols * 2, cfg.cols, cfg.rows,
                                                     d_pts.data_ptr(), n, capi.floats(sc.intr), d_ro.data_ptr(), cnt.data_ptr(), None))
    torch.cuda.synchronize()
    assert int(cnt.item()) == exp_n and 1000 < exp_n < n
    assert np.array_equal(d_pts.cpu().numpy().view(np.uint32), exp_pts.view(np.uint32))
    assert np.array_equal(d_ro.cpu().numpy().view(np.uint32), exp_ro.view(np.uint32))
    got_d = d_out.cpu().numpy().view(np.uint16)
    assert np.array_equal(got_d, exp_dists) and (got_d != dists).sum() > 500
    # the Python mirror of TsdfVolume::psdf (in-place removal from a snapshot)
    from dynamicfusion_amd import Intr, TsdfVolume
    tv = TsdfVolume(cfg.dims)
    d2 = upload_u16(dists)
    ro2 = tv.psdf(torch.from_numpy(pts[:, :3].copy()).cuda(), d2, Intr(*cfg.intr))
    assert np.array_equal(ro2.cpu().numpy().view(np.uint32), exp_ro.view(np.uint32))
    assert np.array_equal(d2.cpu().numpy().view(np.uint16), exp_dists)
    # aliasing the sampled and the zeroed image is refused (racy in the reference)
    assert capi.lib().dfusion_project_and_remove(d_in.data_ptr(), cfg.cols * 2, d_in.data_ptr(), cfg.cols * 2, cfg.cols, cfg.rows,
                                                 d_pts.data_ptr(), n, capi.floats(sc.intr), None, None, None) == 100001


@pytest.mark.parametrize("cfg", [SMALL, MID], ids=["64-k4", "128-k8"])
def test_point_queries_through_the_brick_index_equal_brute_force(cfg):
    """dfusion_knn / dfusion_warp_points use the brick candidate lists once an index exists (a point inside the grid is inside its
    brick's cell, which the lists cover); points outside the grid fall back to the scan in the same launch; NaN points find nothing.
    Same indices, distances and warped coordinates, bit for bit, as before the index was built."""
    sc = Scene(cfg, n_frames=1)
    rng = np.random.default_rng(9)
    n = 50000
    lo = sc.pose[:3, 3].astype(np.float64)
    q = (lo + rng.uniform(-0.2, cfg.size + 0.2, (n, 3))).astype(F32)          # ~30 % outside the volume
    q[::53] = np.nan
    q[7::211, 1] = np.nan
    nrm = rng.normal(size=(n, 3)).astype(F32)
    wf = make_gpu_warp(sc)
    dq = torch.from_numpy(q).cuda()
    i0, d0 = wf.KNN(dq)
    p0, n0 = dq.clone(), torch.from_numpy(nrm).cuda()
    wf.warp(p0, n0)
    vol = make_gpu_volume(sc)
    wf.ensure_index(vol, cfg.k)
    i1, d1 = wf.KNN(dq)
    p1, n1 = dq.clone(), torch.from_numpy(nrm).cuda()
    wf.warp(p1, n1)
    torch.cuda.synchronize()
    assert torch.equal(i0, i1) and torch.equal(d0.view(torch.int32), d1.view(torch.int32))
    assert torch.equal(p0.view(torch.int32), p1.view(torch.int32)) and torch.equal(n0.view(torch.int32), n1.view(torch.int32))
    ri, rd = O.knn(sc.pos, np.nan_to_num(q[:3000], nan=0.3), cfg.k)
    i2, d2 = wf.KNN(torch.from_numpy(np.nan_to_num(q[:3000], nan=0.3)).cuda())
    assert np.array_equal(i2.cpu().numpy(), ri) and np.array_equal(d2.cpu().numpy().view(np.uint32), rd.view(np.uint32))
    # the 8x8-tile processing order (a hint for image-ordered point sets) changes nothing: 49920 = 24 bands of 8 rows x 260 columns...
    for cols, m in ((208, 49920), (640, 46080), (200, 50000)):          # ...whole bands, whole bands, NOT a multiple of 8 rows (hint ignored)
        wf.set_point_tiling(cols)
        i3, d3 = wf.KNN(dq[:m].contiguous())
        p3, n3 = dq[:m].clone(), torch.from_numpy(nrm[:m]).cuda()
        wf.warp(p3, n3)
        assert torch.equal(i3, i0[:m]) and torch.equal(d3.view(torch.int32), d0[:m].view(torch.int32))
        assert torch.equal(p3.view(torch.int32), p0[:m].view(torch.int32)) and torch.equal(n3.view(torch.int32), n0[:m].view(torch.int32))
    wf.set_point_tiling(0)


def test_tables_on_demand_equal_tables_built_at_once():
    """DF_INDEX_TABLES_ON_DEMAND: the per-voxel tables are filled block by block as the launch plans first find blocks alive.  Over a
    camera sweep (new blocks every frame) the volume must equal the one made with tables built at once, and a sweep without verdicts
    (cull off: the missing blocks are built first) and one through the batched kernel must agree too."""
    cfg = synth.CONFIGS["256"]
    frames = [0, 8, 16, 24, 40]                                            # camera_pose(f): 0.25 degrees per frame
    intr = Intr(*cfg.intr)
    pos, sigma = synth.make_nodes(cfg)
    vols = []
    for on_demand in (False, True, True):
        wf = WarpField(k=cfg.k, tables_on_demand=on_demand)
        wf.init(pos, sigma=sigma, transforms=synth.node_transforms(cfg, 0))
        v = TsdfVolume(cfg.dims); v.setSize([cfg.size] * 3); v.setTruncDist(cfg.trunc_dist); v.setMaxWeight(cfg.max_weight); v.setPose(cfg.volume_pose)
        v.clear()
        n = torch.zeros(1, dtype=torch.int64, device="cuda")
        for i, f in enumerate(frames):
            wf.set_transforms(torch.from_numpy(synth.node_transforms(cfg, f)).cuda())
            d = compute_dists(upload_u16(synth.depth_frame(cfg, f)), intr)
            kw = dict()
            if len(vols) == 2 and i == 2: kw = dict(cull=False)             # third variant: completes the tables in mid-sequence,
            if len(vols) == 2 and i == 3: kw = dict(pipelined=False)        # then the batched kernel
            v.integrate_warped(d, synth.camera_pose(cfg, f), intr, wf, n_updated=n, **kw)
        vols.append((v.data().clone(), int(n.item())))
    assert vols[0][1] > 0
    for other in vols[1:]:
        assert torch.equal(vols[0][0], other[0]) and vols[0][1] == other[1]


def test_block_models_shrink_the_swept_set_and_change_nothing():
    """dfusion_warp_blocks.h: the per-block blend models must (i) leave the volume and the update count bit-identical and (ii) actually
    engage -- the launch plan's swept-voxel counter (dfusion_warp_debug_counters) drops.  BASELINE config 1 (256^3, 500 nodes, k = 4);
    tables built at once and on demand; the counter is read on the last of four frames (a block's model serves from the frame after
    the one that made it)."""
    cfg = synth.CONFIGS["256"]
    sc = Scene(cfg, n_frames=4)
    intr = Intr(*cfg.intr)
    L = capi.lib()
    for on_demand in (False, True):
        wf = WarpField(k=cfg.k, tables_on_demand=on_demand)
        wf.init(sc.pos, sigma=sc.sigma, transforms=sc.dqs[0])
        swept, vols, upd = [], [], []
        for kw in (dict(block_model=False), dict(block_model="now")):
            v = make_gpu_volume(sc)
            cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
            for f in range(4):
                wf.set_transforms(torch.from_numpy(sc.dqs[f]).cuda())
                if f == 3: cnt.zero_(); wf.debug_counters(cnt[1:])
                try:
                    # (prefetch="steady": which frame would switch the side stream off depends on host / GPU timing -- the counter below must not)
                    v.integrate_warped(upload_u16(sc.dists[f]), sc.cam_poses[f], intr, wf, n_updated=cnt[:1], prefetch="steady", **kw)
                finally:
                    wf.debug_counters(None)
            upd.append(int(cnt[0].item())); swept.append(int(cnt[1].item())); vols.append(v.data().clone())
        print("on demand %d: swept voxels / updated: ball test %.3f, with block models %.3f" % (on_demand, swept[0] / upd[0], swept[1] / upd[1]))
        assert torch.equal(vols[0], vols[1]) and upd[0] == upd[1] and upd[0] > 0
        assert swept[1] < 0.85 * swept[0] and swept[1] >= upd[1]


@pytest.mark.parametrize("name, prefetch", [("k8", "steady"), ("k8", True), ("k8-256", True), ("256", "steady")])
def test_neighbour_codes_engage_and_change_nothing(name, prefetch):
    """DF_IDX_CODES (round 5): a block with a blend model stores each voxel's neighbours as 4-bit positions in the block's node union
    (2 B a voxel at k = 4, 4 B at k = 8) and the sweep reads those instead of the 16-B record, resolving them through a 16-entry
    per-wave LDS table.  Eight frames of a moving camera with changing transforms, codes on against DF_WARP_NO_CODES: the volume after
    EVERY frame and the update count are identical bit for bit, and from the third frame on kept blocks do carry codes
    (dfusion_warp_coded_blocks).  prefetch=True covers the look-ahead builds racing the plan (the model kernel runs on the side
    stream while the plan kernel decides which blocks are coded: the decision must come from the verdict pass, before the fork).
    k = 4 (BASELINE config 1) has no coded path: no block may claim codes, and the flag changes nothing."""
    cfg = (synth.CONFIGS["256"] if name == "256" else synth.Config(256, 1.5, nodes=700, k=8) if name == "k8-256"
           else synth.Config(128, 1.0, cols=320, rows=240, nodes=300, k=8))
    frames = 8
    sc = Scene(cfg, n_frames=frames)
    intr = Intr(*cfg.intr)
    dists = [upload_u16(d) for d in sc.dists]
    nl = cfg.dims[2] // 8

    def run(codes):
        v = make_gpu_volume(sc)
        wf = WarpField(k=cfg.k)
        wf.init(sc.pos, sigma=sc.sigma, transforms=sc.dqs[0])
        cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
        snaps, coded, kept = [], [], []
        for f in range(frames):
            wf.set_transforms(torch.from_numpy(sc.dqs[f]).cuda())
            v.integrate_warped(dists[f], sc.cam_poses[f], intr, wf, n_updated=cnt, prefetch=prefetch, codes=codes)
            snaps.append(v.data().clone())
            a = torch.zeros(nl, dtype=torch.int64, device="cuda"); c = torch.zeros_like(a)
            wf.alive_blocks_per_layer(v, a); wf.coded_blocks_per_layer(v, c)
            kept.append(int(a.sum().item())); coded.append(int(c.sum().item()))
        return snaps, int(cnt.item()), kept, coded

    a_snaps, a_n, kept, coded = run(True)
    b_snaps, b_n, _, _ = run(False)
    print("%s prefetch=%s: kept blocks %s, of them coded %s" % (name, prefetch, kept, coded))
    assert a_n == b_n > 0
    for f in range(frames):
        assert torch.equal(a_snaps[f], b_snaps[f]), "volume after frame %d differs" % f
    assert coded[0] == 0 and all(c <= k for c, k in zip(coded, kept))
    if cfg.k == 8: assert coded[-1] > 0.5 * kept[-1]
    else: assert not any(coded)
    # third leg (VERDICT r5 #3 i): the coded sweep against the REFERENCE, not only against the uncoded one -- the final volume of the
    # 256^3 case equals the per-voxel composition through the reference's own classes over all 8 frames (~10 s of the box's host cores)
    if name == "k8-256" and O.have_ref():
        ref = sc.new_volume()
        for f in range(frames):
            O.ref_integrate_warped(sc.dists[f], ref, cfg.dims, sc.vs, sc.trunc, cfg.max_weight, synth.aff12(sc.pose), synth.aff12(sc.world2cam(f)),
                                   sc.intr, sc.pos, sc.dqs[f], sc.sigma, cfg.k, 0, 0, cfg.dims[2])
        got = a_snaps[-1].cpu().numpy().view(np.uint32)
        assert int((ref >> 16).sum()) == a_n
        assert np.array_equal(got, ref), "%d of %d voxels differ from the reference's classes" % (int((got != ref).sum()), ref.size)


@pytest.mark.parametrize("k, nodes", [(8, 6000), (4, 6000), (8, 5200)])
def test_node_sets_too_large_for_an_lds_table_keep_the_planned_sweep(k, nodes):
    """Round 6: the pipelined sweep needs no LDS node table, so node sets past 5120 (160 KiB / 32 B) -- where rounds 1-5 fell back to
    the plain gather kernel without verdicts, plan, codes or the prepare / sweep split -- run the same planned sweep: k = 8 from its
    per-wave union copies (and, where a 4 x 4 x 4 sub-block's union overflows 16 nodes, which this density makes common, from the
    16-byte records and the L2), k = 4 gathering from the L2.  Four frames of a moving camera with changing transforms on 128^3:
    every voxel equals the oracle's; the cull changes nothing; and the split API accepts the handle."""
    cfg = synth.Config(128, 1.0, cols=320, rows=240, nodes=nodes, k=k)
    frames = 4
    sc = Scene(cfg, n_frames=frames)
    assert sc.pos.shape[0] == nodes > 5120
    intr = Intr(*cfg.intr)
    dists = [upload_u16(d) for d in sc.dists]
    v, u, w = make_gpu_volume(sc), make_gpu_volume(sc), make_gpu_volume(sc)
    wf = WarpField(k=k)
    wf.init(sc.pos, sigma=sc.sigma, transforms=sc.dqs[0])
    wf2 = WarpField(k=k)
    wf2.init(sc.pos, sigma=sc.sigma, transforms=sc.dqs[0])
    ref = sc.new_volume()
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    nl = cfg.dims[2] // 8
    for f in range(frames):
        wf.set_transforms(torch.from_numpy(sc.dqs[f]).cuda()); wf2.set_transforms(torch.from_numpy(sc.dqs[f]).cuda())
        v.integrate_warped(dists[f], sc.cam_poses[f], intr, wf, n_updated=cnt, prefetch="steady")
        u.integrate_warped(dists[f], sc.cam_poses[f], intr, wf, cull=False)                       # (same handle: tables completed, no verdicts)
        w.integrate_warped_prepare(dists[f], sc.cam_poses[f], intr, wf2, prefetch="steady")      # the split API on a handle of its own
        w.integrate_warped_sweep(wf2, sync=True)
        O.integrate_warped(sc.dists[f], ref, sc.ovol(ref), synth.aff12(sc.pose), synth.aff12(sc.world2cam(f)), sc.intr,
                           sc.pos, sc.dqs[f], sc.sigma, k)
    a = torch.zeros(nl, dtype=torch.int64, device="cuda"); c = torch.zeros_like(a)
    wf.alive_blocks_per_layer(v, a); wf.coded_blocks_per_layer(v, c)
    print("k = %d, M = %d: kept blocks %d, coded %d" % (k, nodes, int(a.sum().item()), int(c.sum().item())))
    s = compare_volumes(v.download(), ref)
    assert s["bits_mismatch"] == 0, s
    assert int((ref >> 16).sum()) == int(cnt.item()) > 0
    assert torch.equal(v.data(), u.data()) and torch.equal(v.data(), w.data())
    if k == 4: assert int(c.sum().item()) == 0


def test_one_handle_switching_k_keeps_its_codes_right():
    """The code tables are allocated the first time a k = 8 sweep wants models -- whatever the handle swept before.  One warp field used
    with k = 4 (tables, models, no codes), then k = 8 on the same geometry (index rebuilt; models AND codes), then k = 4 again and back:
    every k = 8 frame must equal the same frame from a handle that never saw k = 4, and codes must engage in both k = 8 phases."""
    cfg = synth.Config(128, 1.0, cols=320, rows=240, nodes=300, k=8)
    frames = 16
    sc = Scene(cfg, n_frames=frames)
    intr = Intr(*cfg.intr)
    dists = [upload_u16(d) for d in sc.dists]
    ks = [4] * 3 + [8] * 5 + [4] * 3 + [8] * 5
    nl = cfg.dims[2] // 8

    def run(only8):
        v = make_gpu_volume(sc)
        wf = WarpField(k=8)
        wf.init(sc.pos, sigma=sc.sigma, transforms=sc.dqs[0])
        snaps, coded = {}, {}
        for f in range(frames):
            if only8 and ks[f] != 8: continue
            wf.set_transforms(torch.from_numpy(sc.dqs[f]).cuda())
            # (every frame starts from an empty volume: the two runs integrate different frame sets)
            v.clear()
            v.integrate_warped(dists[f], sc.cam_poses[f], intr, wf, k=ks[f], prefetch="steady")
            if ks[f] == 8:
                snaps[f] = v.data().clone()
                c = torch.zeros(nl, dtype=torch.int64, device="cuda"); wf.coded_blocks_per_layer(v, c); coded[f] = int(c.sum().item())
        return snaps, coded

    a, ca = run(False)
    b, cb = run(True)
    print("coded blocks per k = 8 frame, handle that switches k: %s; k = 8 only: %s" % (ca, cb))
    for f in a:
        assert torch.equal(a[f], b[f]), "frame %d differs" % f
    assert ca[7] > 0 and ca[15] > 0 and ca[3] == 0 and ca[11] == 0     # (a rebuilt index starts without models)


def test_prepare_and_sweep_on_two_streams_equal_the_single_call():
    """dfusion_integrate_warped_prepare / _sweep (round 5): the frame's warped integrate split into the part that does not touch the volume
    (pyramid, verdict pass, table / model builds, plan) and the sweep, the former issued on ANOTHER stream beside the previous frame's
    ray-cast -- the pipelining bench.py uses.  Twelve frames of a moving camera with changing transforms: volumes after every frame, update
    counts and every ray-cast image must be the single call's, bit for bit; and the error paths of the split (no plan pending, a plan made
    for another volume, a configuration without a plan) must say so."""
    cfg = synth.Config(128, 1.0, cols=320, rows=240, nodes=300, k=8)
    frames = 12
    sc = Scene(cfg, n_frames=frames)
    intr = Intr(*cfg.intr)
    dists = [upload_u16(d) for d in sc.dists]
    dqs = [torch.from_numpy(q).cuda() for q in sc.dqs]

    def run(split):
        v = make_gpu_volume(sc)
        wf = make_gpu_warp(sc, k=cfg.k)
        cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
        pts = torch.empty((cfg.rows, cfg.cols, 4), dtype=torch.float32, device="cuda"); nrm = torch.empty_like(pts)
        main, prep = torch.cuda.current_stream(), torch.cuda.Stream()
        snaps, casts = [], []
        for f in range(frames):
            if split:
                # nothing orders the prepare half of frame f against the sweep of frame f - 1 here but the library itself: they run side by side
                if f == 0: prep.wait_stream(main)                   # (uploads and the volume's clear were enqueued on the main stream)
                with torch.cuda.stream(prep):
                    wf.set_transforms(dqs[f])
                    v.integrate_warped_prepare(dists[f], sc.cam_poses[f], intr, wf, prefetch="steady")
                v.integrate_warped_sweep(wf, n_updated=cnt)
            else:
                wf.set_transforms(dqs[f])
                v.integrate_warped(dists[f], sc.cam_poses[f], intr, wf, n_updated=cnt, sync=False, prefetch="steady")
            v.raycast(sc.cam_poses[f], intr, pts, nrm)
            snaps.append(v.data().clone()); casts.append(pts.clone())
        torch.cuda.synchronize()
        return snaps, casts, int(cnt.item()), v, wf

    a_snaps, a_casts, a_n, _, _ = run(False)
    b_snaps, b_casts, b_n, v, wf = run(True)
    assert a_n == b_n > 0
    for f in range(frames):
        assert torch.equal(a_snaps[f], b_snaps[f]), "volume after frame %d differs" % f
        assert torch.equal(a_casts[f].view(torch.int32), b_casts[f].view(torch.int32)), "ray-cast of frame %d differs" % f
    # error paths: nothing prepared; a plan for another geometry; the lean configuration has no plan
    with pytest.raises(capi.DfusionError):
        v.integrate_warped_sweep(wf)
    v.integrate_warped_prepare(dists[0], sc.cam_poses[0], intr, wf)
    other = make_gpu_volume(Scene(synth.Config(64, 1.0, cols=320, rows=240, nodes=300, k=8), n_frames=1))
    with pytest.raises(capi.DfusionError):
        other.integrate_warped_sweep(wf)
    v.integrate_warped_sweep(wf)                                  # (the pending plan is still good for ITS volume)
    lean = WarpField(k=cfg.k, voxel_table=False)
    lean.init(sc.pos, sigma=sc.sigma, transforms=sc.dqs[0])
    with pytest.raises(capi.DfusionError):
        v.integrate_warped_prepare(dists[0], sc.cam_poses[0], intr, lean)
    torch.cuda.synchronize()
    # what voids a pending plan: ONE set_transforms after the prepare is the pipelined order and leaves it good -- it writes the
    # alternate node set --; a SECOND would rewrite the set the plan reads, so the sweep must refuse instead of blending with frame t + 2's
    # transforms; set_nodes and a rebuilt index free or re-make what the plan points at
    dq = [torch.from_numpy(sc.dqs[f]).cuda() for f in range(3)]
    wf.set_transforms(dq[0])
    v.integrate_warped_prepare(dists[0], sc.cam_poses[0], intr, wf)
    wf.set_transforms(dq[1])
    v.integrate_warped_sweep(wf, sync=True)                       # still the plan's transforms: accepted
    v.integrate_warped_prepare(dists[1], sc.cam_poses[1], intr, wf)
    wf.set_transforms(dq[2]); wf.set_transforms(dq[0])
    with pytest.raises(capi.DfusionError):
        v.integrate_warped_sweep(wf)
    wf.set_transforms(dq[1])
    v.integrate_warped_prepare(dists[1], sc.cam_poses[1], intr, wf)
    wf.init(sc.pos, sigma=sc.sigma, transforms=sc.dqs[0])       # dfusion_warp_set_nodes
    with pytest.raises(capi.DfusionError):
        v.integrate_warped_sweep(wf)
    torch.cuda.synchronize()
