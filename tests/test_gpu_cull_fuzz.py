"""Randomised stress of the result-identical culls: the sweeps' launch plans (df_sweep_plan_kernel: ball around the tile centre,
frustum side planes, distance bound against the max-pyramid of dists, and the per-block blend models' boxes of dfusion_warp_blocks.h;
df_rigid_plan_kernel: the patch's box against the same) must never drop a voxel that updates.  Random camera poses (inside / outside / beside the volume, tilted), random volume poses, depth
images made of random blocks of near / far / invalid values, strong node motions: the volume with the cull must equal the volume
without it, bit for bit, update counts included.  (The no-cull sweeps are themselves compared with the oracle elsewhere.)"""
import os

import numpy as np
import pytest
import torch

from dynamicfusion_amd import Intr, TsdfVolume, WarpField, capi, compute_dists, synth, upload_u16

# DFUSION_FUZZ_EXTRA=N adds N seeds to each fuzz below (N // 4 to the full-size surface one): the committed long run is
# profiles/r04_cull_fuzz_extended.txt
FUZZ_EXTRA = int(os.environ.get("DFUSION_FUZZ_EXTRA", "0"))
pytestmark = pytest.mark.gpu
F32 = np.float32


def rot(axis, ang):
    a = np.asarray(axis, np.float64); a /= np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)


def random_pose(rng, centre, dist_lo, dist_hi, tilt):
    """camera somewhere around `centre`, looking roughly at it (z forward), rolled and tilted by up to `tilt` rad"""
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    eye = centre + d * rng.uniform(dist_lo, dist_hi)
    z = centre - eye; z /= np.linalg.norm(z)
    up = rng.normal(size=3); x = np.cross(up, z); x /= np.linalg.norm(x); y = np.cross(z, x)
    R = np.stack([x, y, z], 1) @ rot(rng.normal(size=3), rng.uniform(-tilt, tilt))
    m = np.eye(4, dtype=F32); m[:3, :3] = R.astype(F32); m[:3, 3] = eye.astype(F32)
    return m


def random_depth(rng, cols, rows, lo_mm, hi_mm):
    d = np.zeros((rows, cols), np.uint16)
    d[:] = rng.integers(lo_mm, hi_mm)
    for _ in range(rng.integers(3, 12)):
        x0, y0 = rng.integers(0, cols), rng.integers(0, rows)
        w, h = rng.integers(1, cols // 2 + 1), rng.integers(1, rows // 2 + 1)
        d[y0:y0 + h, x0:x0 + w] = 0 if rng.random() < 0.25 else rng.integers(lo_mm, hi_mm)
    d[rng.random(d.shape) < 0.02] = 0
    return d


def make_volume(dims, size, pose, trunc=0.04):
    v = TsdfVolume(dims); v.setSize([size] * 3); v.setTruncDist(trunc); v.setMaxWeight(64); v.setPose(pose)
    return v


@pytest.mark.parametrize("seed", range(12 + FUZZ_EXTRA))
def test_rigid_plan_never_drops_an_update(seed):
    rng = np.random.default_rng(100 + seed)
    dims = [(64, 64, 64), (40, 72, 56), (96, 32, 64)][seed % 3]
    size = float(rng.uniform(0.8, 2.0))
    cols, rows = [(96, 72), (160, 120), (67, 45)][(seed // 3) % 3]
    intr = Intr(*(F32(rng.uniform(0.7, 1.4) * cols), F32(rng.uniform(0.7, 1.4) * cols), F32(cols / 2 + rng.uniform(-8, 8)), F32(rows / 2 + rng.uniform(-8, 8))))
    vol_pose = np.eye(4, dtype=F32); vol_pose[:3, :3] = rot(rng.normal(size=3), rng.uniform(0, 0.6) * (seed % 2)).astype(F32)
    vol_pose[:3, 3] = rng.uniform(-0.5, 0.5, 3).astype(F32)
    centre = (vol_pose[:3, :3].astype(np.float64) @ (np.array([size, size, size]) / 2)) + vol_pose[:3, 3]
    L = capi.lib()
    res = []
    frames = [(random_pose(rng, centre, 0.0 if seed % 4 == 0 else 0.3 * size, 2.2 * size, 0.5), random_depth(rng, cols, rows, 300, int(3500 * size)))
              for _ in range(3)]
    # the plan's tests: both, frustum only, none (every sub-chunk swept); and with the kept scratch poisoned before every call (a read
    # of plan data the call did not write itself would then see 0xFF bytes instead of the previous call's plan)
    for flags in (0, capi.DF_RIGID_NO_DEPTH_CULL, capi.DF_RIGID_NO_DEPTH_CULL | capi.DF_RIGID_KEEP_ALL, capi.DF_RIGID_POISON_SCRATCH):
        v = make_volume(dims, size, vol_pose)
        n = torch.zeros(1, dtype=torch.int64, device="cuda")
        for cam, depth in frames:
            v.integrate(compute_dists(upload_u16(depth), intr), cam, intr, n_updated=n, flags=flags)
        res.append((v.download(), int(n.item())))
    print("rigid fuzz seed", seed, "updates", res[0][1])
    for r in res[1:]:
        assert np.array_equal(res[0][0], r[0]) and res[0][1] == r[1]


@pytest.mark.parametrize("seed", range(24 + FUZZ_EXTRA))
def test_warped_plan_never_drops_an_update(seed):
    rng = np.random.default_rng(500 + seed)
    dims = [(64, 64, 64), (32, 48, 64), (96, 32, 40)][seed % 3]
    size = float(rng.uniform(0.8, 2.0))
    cols, rows = [(96, 72), (160, 120), (67, 45)][(seed // 3) % 3]
    k = [8, 4][seed % 2]
    intr = Intr(*(F32(rng.uniform(0.7, 1.4) * cols), F32(rng.uniform(0.7, 1.4) * cols), F32(cols / 2 + rng.uniform(-8, 8)), F32(rows / 2 + rng.uniform(-8, 8))))
    vol_pose = np.eye(4, dtype=F32)                           # (rotated volume poses take the general kernel: see test_gpu_edge_cases)
    if seed % 4 == 3: vol_pose[:3, :3] = rot(rng.normal(size=3), rng.uniform(0, 0.5)).astype(F32)
    vol_pose[:3, 3] = rng.uniform(-0.5, 0.5, 3).astype(F32)
    ext = np.array(dims, np.float64) / max(dims) * size
    centre = (vol_pose[:3, :3].astype(np.float64) @ (ext / 2)) + vol_pose[:3, 3]
    M = int(rng.integers(30, 120))
    pos = (vol_pose[:3, :3].astype(np.float64) @ (rng.uniform(0.1, 0.9, (M, 3)) * ext).T).T + vol_pose[:3, 3]
    sigma = np.full(M, rng.uniform(0.03, 0.15) * size, F32)
    amp_r, amp_t = [(0.02, 0.005), (0.15, 0.03), (0.5, 0.1)][seed % 3]
    frames = []
    for _ in range(3):
        rv, tv = rng.uniform(-amp_r, amp_r, (M, 3)), rng.uniform(-amp_t, amp_t, (M, 3))
        if seed >= 12:                                        # a coherent motion field (what the block models are tight on) + a little noise
            rv = rng.uniform(-amp_r, amp_r, 3)[None] + 0.1 * rv; tv = rng.uniform(-amp_t, amp_t, 3)[None] + 0.1 * tv
        dq = synth.dq_from_twist(rv.astype(F32), tv.astype(F32))
        frames.append((random_pose(rng, centre, 0.0 if seed % 5 == 0 else 0.3 * size, 2.2 * size, 0.5),
                       random_depth(rng, cols, rows, 300, int(3500 * size)), dq))
    wf = WarpField(k=k, voxel_table=(seed % 6 != 5), tables_on_demand=(seed % 4 < 2))
    wf.init(pos.astype(F32), sigma=sigma, transforms=frames[0][2])
    res = []
    # block models from the first sweep on / by the library's policy (second sweep) / never; image-wide depth bound; no cull at all
    for kw in (dict(block_model="now"), dict(), dict(block_model=False), dict(depth_pyramid=False, block_model="now"), dict(cull=False)):
        v = TsdfVolume(dims); v.setSize(list(ext)); v.setTruncDist(0.04); v.setMaxWeight(64); v.setPose(vol_pose)
        n = torch.zeros(1, dtype=torch.int64, device="cuda")
        for cam, depth, dq in frames:
            wf.set_transforms(torch.from_numpy(dq).cuda())
            v.integrate_warped(compute_dists(upload_u16(depth), intr), cam, intr, wf, n_updated=n, **kw)
        res.append((v.download(), int(n.item())))
    print("warped fuzz seed", seed, "updates", res[0][1])
    for r in res[1:]:
        assert np.array_equal(res[0][0], r[0]) and res[0][1] == r[1]


@pytest.mark.parametrize("seed", range(8 + FUZZ_EXTRA // 4))
def test_block_models_on_surface_node_sets(seed):
    """The block blend models where they are at their tightest and hence their riskiest: a dense node set on the observed surface
    (128^3, 600 nodes, k = 8: unions of 8-16 nodes per block), node motions from nearly rigid to incoherent and up to 0.6 rad,
    cameras that move between frames, tables built at once or on demand.  Volumes and update counts with the models must equal those
    without any cull; and the models must have engaged (fewer swept voxels than the ball test alone)."""
    rng = np.random.default_rng(900 + seed)
    cfg = synth.Config(128, 1.0, cols=320, rows=240, nodes=600, k=8)
    intr = Intr(*cfg.intr)
    pos, sigma = synth.make_nodes(cfg)
    amp_r, amp_t, noise = [(0.05, 0.01, 1.0), (0.3, 0.05, 0.1), (0.6, 0.1, 0.02), (0.1, 0.02, 0.5)][seed % 4]
    frames = []
    for i in range(4):
        rv = rng.uniform(-amp_r, amp_r, 3)[None] + noise * rng.uniform(-amp_r, amp_r, (cfg.nodes, 3))
        tv = rng.uniform(-amp_t, amp_t, 3)[None] + noise * rng.uniform(-amp_t, amp_t, (cfg.nodes, 3))
        f = 6 * i + seed % 48                                 # (the synthetic camera sweep leaves the scene after ~200 frames)
        frames.append((synth.camera_pose(cfg, f), compute_dists(upload_u16(synth.depth_frame(cfg, f)), intr),
                       synth.dq_from_twist(rv.astype(F32), tv.astype(F32))))
    wf = WarpField(k=cfg.k, tables_on_demand=(seed % 2 == 0))
    wf.init(pos, sigma=sigma, transforms=frames[0][2])
    L = capi.lib()
    res = []
    for kw in (dict(block_model="now"), dict(block_model=False), dict(cull=False)):
        v = make_volume(cfg.dims, cfg.size, cfg.volume_pose, cfg.trunc_dist)
        cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
        wf.debug_counters(cnt[1:])
        try:
            for cam, d, dq in frames:
                wf.set_transforms(torch.from_numpy(dq).cuda())
                v.integrate_warped(d, cam, intr, wf, n_updated=cnt[:1], **kw)
        finally:
            wf.debug_counters(None)
        res.append((v.download(), int(cnt[0].item()), int(cnt[1].item())))
    print("surface fuzz seed", seed, "updates", res[0][1], "swept with models / ball / none: %d / %d / %d" % (res[0][2], res[1][2], res[2][2]))
    assert res[0][1] > 0
    for r in res[1:]:
        assert np.array_equal(res[0][0], r[0]) and res[0][1] == r[1]
    assert res[0][2] < res[1][2] <= res[2][2]


@pytest.mark.parametrize("on_demand", [False, True], ids=["eager", "on-demand"])
def test_volumes_that_are_not_whole_blocks(on_demand):
    """Dimensions that are multiples of 4 but not of 8 (the verdicts, models and on-demand builds work on 8 x 8 x 8 blocks, the tables on
    32 x 16 x 8 tiles): partial blocks at three faces.  Every variant of the sweep must agree."""
    rng = np.random.default_rng(77)
    dims, size = (36, 44, 52), 1.0
    cols, rows = 160, 120
    intr = Intr(F32(150.0), F32(150.0), F32(80.0), F32(60.0))
    ext = np.array(dims, np.float64) / max(dims) * size
    pose = np.eye(4, dtype=F32); pose[:3, 3] = (-ext[0] / 2, -ext[1] / 2, 0.4)
    M = 60
    nodes = (rng.uniform(0.1, 0.9, (M, 3)) * ext + pose[:3, 3]).astype(F32)
    wf = WarpField(k=8, tables_on_demand=on_demand)
    frames = []
    for i in range(3):
        dq = synth.dq_from_twist(rng.uniform(-0.1, 0.1, (M, 3)).astype(F32), rng.uniform(-0.02, 0.02, (M, 3)).astype(F32))
        cam = np.eye(4, dtype=F32); cam[:3, :3] = rot((0, 1, 0), 0.05 * i).astype(F32); cam[:3, 3] = (0.02 * i, 0, 0)
        frames.append((cam, random_depth(rng, cols, rows, 500, 1500), dq))
    wf.init(nodes, sigma=np.full(M, 0.08, F32), transforms=frames[0][2])
    res = []
    for kw in (dict(block_model="now"), dict(), dict(block_model=False), dict(cull=False), dict(pipelined=False), dict(use_lds=False)):
        v = TsdfVolume(dims); v.setSize(list(ext)); v.setTruncDist(0.04); v.setMaxWeight(64); v.setPose(pose)
        n = torch.zeros(1, dtype=torch.int64, device="cuda")
        for cam, depth, dq in frames:
            wf.set_transforms(torch.from_numpy(dq).cuda())
            v.integrate_warped(compute_dists(upload_u16(depth), intr), cam, intr, wf, n_updated=n, **kw)
        res.append((v.download(), int(n.item())))
    assert res[0][1] > 0
    for r in res[1:]:
        assert np.array_equal(res[0][0], r[0]) and res[0][1] == r[1]


# ------------------------------------------------------------------------------------------------------------------------------
# where the fuzz above does not reach.  Every case: the sweep with its cull (block models made at once; the library's
# own policy on a FRESH field, which also runs the look-ahead builds on the side stream; the same without the side stream) must leave
# the bits and the update count of the sweep with no cull at all.
def quat_dq(axis, angle, t):
    """[M, 8] dual quaternions from axis / angle / translation, the rotation's scalar part cos(angle / 2) as it comes (<= 0 past pi)"""
    axis = np.asarray(axis, np.float64); axis = axis / np.linalg.norm(axis, axis=-1, keepdims=True)
    rotq = np.concatenate([np.cos(angle / 2)[:, None], axis * np.sin(angle / 2)[:, None]], -1).astype(F32)
    half = np.concatenate([np.zeros((len(rotq), 1), F32), F32(0.5) * np.asarray(t, F32)], -1)
    return np.concatenate([rotq, synth.quat_mul(half, rotq)], -1).astype(F32)      # encodeTranslation, dual_quaternion.hpp:82-85


def depth_from(cfg, pose):
    t, _ = synth._hit_points(cfg, pose)
    valid = np.isfinite(t) & (t >= 0.05) & (t <= 0.5 + cfg.size)
    return np.clip(np.where(valid, np.rint(np.where(valid, t, 0.0) * 1000.0), 0.0), 0, 65535).astype(np.uint16)


def hard_case(name):
    rng = np.random.default_rng({"big_rotations": 1, "negative_scalar": 2, "antipodal": 3, "sigma_spread": 4, "camera_inside_512": 5, "k4_overflow": 6,
                                 "256_2000_nodes": 7}[name])
    k, sig_mul = 8, None
    if name == "camera_inside_512":
        cfg = synth.Config(512, 3.0, nodes=2000, k=8)
    elif name == "256_2000_nodes":
        cfg = synth.Config(256, 1.0, nodes=2000, k=8)
    elif name == "k4_overflow":
        cfg = synth.Config(64, 1.0, cols=160, rows=120, nodes=1500, k=4); k = 4
    else:
        cfg = synth.Config(128, 1.0, cols=320, rows=240, nodes=600, k=8)
    intr = Intr(*cfg.intr)
    if name == "k4_overflow":                                # nodes all through the volume, 8.7 cm apart on average: unions of 20+ per 12.5 cm block
        pos = (rng.uniform(0.05, 0.95, (cfg.nodes, 3)) * cfg.size + cfg.volume_pose[:3, 3]).astype(F32)
        sigma = np.full(cfg.nodes, 0.06, F32)
    else:
        pos, sigma = synth.make_nodes(cfg)
    M = cfg.nodes
    if name == "sigma_spread":                               # dg_w varying 10 x between neighbouring nodes
        sigma = (sigma * rng.choice([0.5, 1.0, 2.5, 5.0], M)).astype(F32)
    frames = []
    for i in range(3):
        ax = rng.normal(size=(M, 3))
        if name in ("big_rotations", "negative_scalar", "antipodal"):
            coherent = rng.normal(size=3)
            ax = coherent[None] + 0.2 * ax
            lo, hi = (np.pi / 2, np.pi) if name == "big_rotations" else (0.0, 0.4)
            ang = rng.uniform(lo, hi) + 0.05 * rng.uniform(-1, 1, M)
            if name == "big_rotations" and i == 2: ang = ang + 0.3                     # a few past pi: scalar parts <= 0
            tv = rng.uniform(-0.03, 0.03, 3)[None] + 0.005 * rng.normal(size=(M, 3))
            dq = quat_dq(ax, ang, tv)
            if name == "negative_scalar": dq[rng.random(M) < (0.5 if i else 1.0)] *= -1     # the same rigid motions, written with w < 0
            if name == "antipodal": dq[::2] *= -1                                            # q and -q on neighbouring nodes: no sign fix in the reference
        else:
            amp = 0.15 if name != "camera_inside_512" else 0.05
            dq = synth.dq_from_twist(rng.uniform(-amp, amp, (M, 3)).astype(F32), rng.uniform(-0.02, 0.02, (M, 3)).astype(F32))
        if name == "camera_inside_512":
            # camera INSIDE the volume, a few centimetres in front of block boundaries: blocks straddle z = 0.05 and the camera plane
            cam = np.eye(4, dtype=F32); cam[:3, :3] = rot((0.2, 1, 0.1), 0.2 * i - 0.2).astype(F32)
            cam[:3, 3] = (0.1 * i - 0.1, 0.05 * i, 0.5 + 0.55 + 0.047 * i)
            depth = depth_from(cfg, cam)
        else:
            cam = synth.camera_pose(cfg, 5 * i)
            if name in ("k4_overflow", "sigma_spread"): depth = random_depth(rng, cfg.cols, cfg.rows, 500, 1500)
            else: depth = synth.depth_frame(cfg, 5 * i)
        frames.append((cam, compute_dists(upload_u16(depth), intr), dq))
    return cfg, intr, pos, sigma, frames, k


@pytest.mark.parametrize("name", ["big_rotations", "negative_scalar", "antipodal", "sigma_spread", "k4_overflow", "256_2000_nodes", "camera_inside_512"])
def test_cull_hard_cases(name):
    cfg, intr, pos, sigma, frames, k = hard_case(name)
    res = []
    for kw in (dict(block_model="now"), dict(), dict(prefetch=False), dict(cull=False)):
        wf = WarpField(k=k)                                   # a fresh field per variant: on-demand tables, the library's model policy
        wf.init(pos, sigma=sigma, transforms=frames[0][2])
        v = make_volume(cfg.dims, cfg.size, cfg.volume_pose, cfg.trunc_dist)
        cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
        wf.debug_counters(cnt[1:])
        for rep in range(2):                                  # twice over the frames: the second pass runs with models / look-ahead tables in place
            for cam, d, dq in frames:
                wf.set_transforms(torch.from_numpy(dq).cuda())
                v.integrate_warped(d, cam, intr, wf, k=k, n_updated=cnt[:1], **kw)
        res.append((v.data().clone(), int(cnt[0].item()), int(cnt[1].item())))
        del wf, v
    print("hard case", name, "updates", res[0][1], "swept now / policy / no side stream / no cull:", [r[2] for r in res])
    assert res[0][1] > 0
    for r in res[1:]:
        assert torch.equal(res[0][0], r[0]) and res[0][1] == r[1]


def test_look_ahead_builds_change_nothing_on_a_moving_camera():
    """12 consecutive frames of a turning camera and a changing warp, tables on demand: with the look-ahead builds (side stream), without
    them, and with no cull at all -- same bits after every frame."""
    cfg = synth.Config(128, 1.0, cols=320, rows=240, nodes=600, k=8)
    intr = Intr(*cfg.intr)
    pos, sigma = synth.make_nodes(cfg)
    frames = [(synth.camera_pose(cfg, 3 * f), compute_dists(upload_u16(synth.depth_frame(cfg, 3 * f)), intr), synth.node_transforms(cfg, 2 * f, rot_amp=0.15, trans_amp=0.03))
              for f in range(12)]
    snaps = []
    for kw in (dict(), dict(prefetch=False), dict(cull=False)):
        wf = WarpField(k=cfg.k); wf.init(pos, sigma=sigma, transforms=frames[0][2])
        v = make_volume(cfg.dims, cfg.size, cfg.volume_pose, cfg.trunc_dist)
        s = []
        for cam, d, dq in frames:
            wf.set_transforms(torch.from_numpy(dq).cuda())
            v.integrate_warped(d, cam, intr, wf, sync=False, **kw)
            s.append(v.data().clone())
        torch.cuda.synchronize()
        snaps.append(s)
    for other in snaps[1:]:
        for a, b in zip(snaps[0], other):
            assert torch.equal(a, b)
