"""BASELINE.json full sizes (256^3 / 500 nodes / k=4, the 512^3 / 2000 nodes / k=8 headline, 1024^3 / 5000 nodes): the oracle
cannot sweep these in seconds, so parity rests on (i) oracle checks on a bounded sample of planes / rays and
(ii) size-independent properties: cull on == cull off, slab-sharded == unsharded, update counts consistent with
weights, reference-header golden vectors reproduced by the HIP k-NN / warp kernels."""
import os

import numpy as np
import pytest
import torch

import oracle_lib as O
from dynamicfusion_amd import Intr, TsdfVolume, WarpField, sharded, synth, upload_u16
from scene import Scene, compare_volumes

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def setup(cfg, slab=None):
    v = TsdfVolume(cfg.dims, slab=slab)
    v.setSize([cfg.size] * 3); v.setTruncDist(cfg.trunc_dist); v.setMaxWeight(cfg.max_weight); v.setPose(cfg.volume_pose)
    v.setRaycastStepFactor(cfg.raycast_step_factor); v.setGradientDeltaFactor(cfg.gradient_delta_factor)
    return v


def test_knn_and_warp_reproduce_reference_header_goldens():
    g = np.load(os.path.join(GOLD, "knn.npz"))
    wf = WarpField(k=8)
    wf.init(g["pos"], sigma=3.0)
    q = torch.from_numpy(g["queries"]).cuda()
    for k, ki, kd in ((4, "idx4", "d2_4"), (8, "idx8", "d2_8")):
        idx, d2 = wf.KNN(q, k)
        assert np.array_equal(idx.cpu().numpy(), g[ki]) and np.array_equal(bits(d2.cpu().numpy()), bits(g[kd]))
    w = np.load(os.path.join(GOLD, "dqb_warp.npz"))
    for tag in ("s3", "s015"):
        for k in (4, 8):
            wf2 = WarpField(k=k)
            wf2.init(w["pos"], sigma=w["sigma_" + tag], transforms=w["dq"])
            p, n = torch.from_numpy(w["points"].copy()).cuda(), torch.from_numpy(w["normals"].copy()).cuda()
            wf2.warp(p, n)
            torch.cuda.synchronize()
            assert np.array_equal(bits(p.cpu().numpy()), bits(w["warp_p_%s_k%d" % (tag, k)]))
            assert np.array_equal(bits(n.cpu().numpy()), bits(w["warp_n_%s_k%d" % (tag, k)]))


@pytest.mark.parametrize("name", ["256", "512", "1024"])
def test_full_size_properties_and_sampled_oracle(name):
    cfg = synth.CONFIGS[name]
    intr = Intr(*cfg.intr)
    sc = Scene(cfg, n_frames=2)
    X, Y, Z = cfg.dims
    wf = WarpField(k=cfg.k)
    wf.init(sc.pos, sigma=sc.sigma, transforms=sc.dqs[0])
    dists = [upload_u16(d) for d in sc.dists]

    # ---- warped integrate: cull on == cull off ; n_upd == sum of weights
    a, b = setup(cfg), setup(cfg)
    n_upd = torch.zeros(1, dtype=torch.int64, device="cuda")
    for f in range(2):
        wf.set_transforms(torch.from_numpy(sc.dqs[f]).cuda())
        a.integrate_warped(dists[f], sc.cam_poses[f], intr, wf, n_updated=n_upd, cull=True, block_model="now")
        b.integrate_warped(dists[f], sc.cam_poses[f], intr, wf, cull=(f == 1), depth_pyramid=False)     # no cull, then the image-wide depth test only
    assert torch.equal(a.data(), b.data())
    weights = (a.data() >> 16) & 0xFFFF
    assert int(weights.sum()) == int(n_upd.item()) and int(weights.max()) == 2
    del b

    # ---- oracle on a bounded sample of planes, both frames: 2 planes in the middle of the volume at 256^3 / 512^3 (40 planes of each
    # are compared by test_warped_and_rigid_planes_bit_exact below), 16 planes -- 8 pairs spread over the depth of the volume, tile-layer
    # boundaries included -- at 1024^3, the size no other test reaches
    pairs = [Z // 2 - 1] if name != "1024" else [Z * (2 * i + 1) // 16 - 1 for i in range(8)]
    n_checked = 0
    for z0 in pairs:
        ref = np.zeros((2, Y, X), np.uint32)
        slab = O.make_slab(z0, 2, z0, 2)
        for f in range(2):
            O.integrate_warped(sc.dists[f], ref, sc.ovol(ref), synth.aff12(sc.pose), synth.aff12(sc.world2cam(f)), sc.intr,
                               sc.pos, sc.dqs[f], sc.sigma, cfg.k, slab=slab)
        s = compare_volumes(a.data()[z0:z0 + 2].cpu().numpy().view(np.uint32), ref)
        print(name, "planes", z0, z0 + 1, "parity:", s)
        assert s["bits_mismatch"] == 0, s
        n_checked += int(((ref >> 16) != 0).sum())
    assert n_checked > 0.05 * 2 * len(pairs) * X * Y                     # the sample lies in the fused part of the volume
    # ---- 1024^3 (BASELINE config 5): ONE CONTIGUOUS band of 64 planes, both frames, through the reference's OWN classes (nanoflann +
    # WarpField::DQB + DualQuaternion, oracle/_ref) -- 8 whole tile layers with everything between their boundaries (VERDICT r4 #4 ii;
    # the full-volume comparisons stop at 512^3: a 1024^3 frame is 43 s of the host's 128 threads)
    if name == "1024" and O.have_ref():
        zb, nb = Z // 2 - 32, 64
        band = np.zeros((nb, Y, X), np.uint32)
        for f in range(2):
            O.ref_integrate_warped(sc.dists[f], band, cfg.dims, sc.vs, sc.trunc, cfg.max_weight, synth.aff12(sc.pose), synth.aff12(sc.world2cam(f)),
                                   sc.intr, sc.pos, sc.dqs[f], sc.sigma, cfg.k, zb, zb, nb)
        got = a.data()[zb:zb + nb].cpu().numpy().view(np.uint32)
        assert ((band >> 16) != 0).sum() > 0.05 * band.size
        assert np.array_equal(got, band), "%d of %d voxels of planes [%d, %d) differ" % (int((got != band).sum()), band.size, zb, zb + nb)

    # ---- slab-sharded (world = 8) integrate + raycast == unsharded
    world = 8
    halo = sharded.halo_planes(sc.trunc, cfg.raycast_step_factor, cfg.gradient_delta_factor, float(sc.vs[2]))
    fp = torch.empty((cfg.rows, cfg.cols, 4), dtype=torch.float32, device="cuda")
    fn = torch.empty_like(fp)
    fk = torch.empty((cfg.rows, cfg.cols), dtype=torch.int32, device="cuda")
    a.raycast(sc.cam_poses[1], intr, fp, fn, keys=fk)
    slabs, k64s = [], []
    for r in range(world):
        zs, zn = sharded.slab_range(Z, r, world)
        v = setup(cfg, slab=(zs, zn, halo))
        for f in range(2):
            wf.set_transforms(torch.from_numpy(sc.dqs[f]).cuda())
            v.integrate_warped(dists[f], sc.cam_poses[f], intr, wf)
        own = slice(v.z_own0 - v.z_store0, v.z_own0 - v.z_store0 + zn)
        assert torch.equal(v.data()[own], a.data()[zs:zs + zn])
        v.data().copy_(a.data()[v.z_store0:v.z_store0 + v.z_store_n])      # halos as the exchange would deliver them
        k64 = torch.empty((cfg.rows, cfg.cols), dtype=torch.int64, device="cuda")
        v.raycast_march(sc.cam_poses[1], intr, k64, rank=r)
        slabs.append(v); k64s.append(k64)
    merged = torch.stack(k64s).min(0).values.contiguous()          # what all_reduce(MIN) computes: the whole merge
    best = torch.where(merged == sharded.KEY_NONE, torch.full_like(merged, 0xFFFFFFFF), (merged >> 39) & 0xFFFFFF)
    assert torch.equal(best, fk.to(torch.int64) & 0xFFFFFFFF)
    acc = torch.zeros((2, cfg.rows, cfg.cols, 4), dtype=torch.int32, device="cuda")
    for v in slabs:
        p, n = torch.empty_like(fp), torch.empty_like(fp)
        v.raycast_shade(sc.cam_poses[1], intr, merged, p, n)
        acc[0] += p.view(torch.int32)
        acc[1] += n.view(torch.int32)
    p3 = torch.empty_like(fp)
    slabs[0].raycast_points_of_keys(sc.cam_poses[1], intr, merged, acc[1].view(torch.float32), p3)     # stage 3: points from keys + summed normals
    del slabs
    assert (~torch.isnan(fp)).float().mean() > 0.3
    assert torch.equal(acc[0], fp.view(torch.int32)) and torch.equal(acc[1], fn.view(torch.int32))
    assert torch.equal(p3.view(torch.int32), fp.view(torch.int32))

    # ---- ray-cast vs the oracle on a sample of rows (oracle casts the full image on the downloaded volume only
    # at 256^3; at 512^3 the download is 512 MiB -- still fine on the GPU box)
    host = a.download()
    rp, rn, rk, _ = O.raycast_points(sc.ovol(host), synth.aff12(sc.cam2vol(1)), sc.rinv(1), sc.reproj, cfg.cols, cfg.rows,
                                     cfg.raycast_step_factor, cfg.gradient_delta_factor, want_keys=True)
    assert np.array_equal(fk.cpu().numpy().view(np.uint32), rk)
    gp, gn = fp.cpu().numpy(), fn.cpu().numpy()
    assert np.array_equal(bits(gp), bits(rp)) and np.array_equal(bits(gn), bits(rn))

    # ---- rigid: slab-sharded == unsharded at full size (replayed vc accumulation); launch plan's tests and short forms on == all off
    r_full = setup(cfg)
    r_full.integrate(dists[0], sc.cam_poses[0], intr)
    from dynamicfusion_amd import capi
    r_nocull = setup(cfg)                                       # every sub-chunk swept, generic arithmetic
    r_nocull.integrate(dists[0], sc.cam_poses[0], intr, flags=capi.DF_RIGID_NO_DEPTH_CULL | capi.DF_RIGID_NO_SHORT_FORMS | capi.DF_RIGID_KEEP_ALL)
    assert torch.equal(r_nocull.data(), r_full.data())
    del r_nocull
    zs, zn = sharded.slab_range(Z, 5, 8)
    r_slab = setup(cfg, slab=(zs, zn, 0))
    r_slab.integrate(dists[0], sc.cam_poses[0], intr)
    assert torch.equal(r_slab.data(), r_full.data()[zs:zs + zn])


def _planes(Z):
    """Sample of Z planes for the oracle: both ends, every 8-plane tile-layer boundary class (last of one / first of the next),
    brick-interior planes, the middle -- 40 planes at 512."""
    base = {0, 1, 2, 7, 8, 9, 15, 16, 31, 32, Z // 8 - 1, Z // 8, Z // 4 - 1, Z // 4, Z // 4 + 3, 3 * Z // 8, Z // 2 - 9, Z // 2 - 8,
            Z // 2 - 1, Z // 2, Z // 2 + 1, Z // 2 + 5, 5 * Z // 8 - 1, 5 * Z // 8, 3 * Z // 4 - 1, 3 * Z // 4, 3 * Z // 4 + 4,
            7 * Z // 8 - 1, 7 * Z // 8, Z - 33, Z - 32, Z - 17, Z - 16, Z - 10, Z - 9, Z - 8, Z - 7, Z - 3, Z - 2, Z - 1}
    return sorted(z for z in base if 0 <= z < Z)


@pytest.mark.parametrize("name", ["256", "512"])
def test_headline_sizes_many_planes_vs_oracle(name):
    """Headline-size parity on a wide plane sample, 3 frames (weights reach 3), BIT-exact: the warped sweep (cached-table path,
    the one the bench times) and the rigid integrate, each against the oracle run on the same planes.  The sample covers the
    first / last planes, planes on both sides of 8-plane tile-layer and brick boundaries, the zero-weight region far from every node
    (low and high z), and -- since every plane cuts the frustum -- its edges."""
    cfg = synth.CONFIGS[name]
    intr = Intr(*cfg.intr)
    frames = 3
    sc = Scene(cfg, n_frames=frames)
    X, Y, Z = cfg.dims
    planes = _planes(Z)
    assert len(planes) >= (32 if Z >= 512 else 30)
    dists = [upload_u16(d) for d in sc.dists]

    # ---- warped sweep
    wf = WarpField(k=cfg.k)
    wf.init(sc.pos, sigma=sc.sigma, transforms=sc.dqs[0])
    vol = setup(cfg)
    n_upd = torch.zeros(1, dtype=torch.int64, device="cuda")
    for f in range(frames):
        wf.set_transforms(torch.from_numpy(sc.dqs[f]).cuda())
        vol.integrate_warped(dists[f], sc.cam_poses[f], intr, wf, n_updated=n_upd)
    torch.cuda.synchronize()
    worst, n_ref, touched, zero_w_planes = 0, 0, 0, 0
    for z in planes:
        ref = np.zeros((1, Y, X), np.uint32)
        slab = O.make_slab(z, 1, z, 1)
        for f in range(frames):
            n_ref += O.integrate_warped(sc.dists[f], ref, sc.ovol(ref), synth.aff12(sc.pose), synth.aff12(sc.world2cam(f)), sc.intr,
                                        sc.pos, sc.dqs[f], sc.sigma, cfg.k, slab=slab)
        got = vol.data()[z:z + 1].cpu().numpy().view(np.uint32)
        bad = int((got != ref).sum())
        worst = max(worst, bad)
        assert bad == 0, (name, "warped plane", z, compare_volumes(got, ref))
        touched += int(((ref >> 16) != 0).sum())
        zero_w_planes += int(not ref.any())
    print(name, "warped: %d planes bit-identical, %d voxels touched on them, %d planes untouched" % (len(planes), touched, zero_w_planes))
    assert touched > 0.05 * len(planes) * X * Y

    # ---- rigid integrate (the reference's actual kernel), same planes: the oracle replays vc += zstep up to the plane
    rig = setup(cfg)
    for f in range(frames):
        rig.integrate(dists[f], sc.cam_poses[f], intr)
    torch.cuda.synchronize()
    wmax = 0
    for z in planes:
        ref = np.zeros((1, Y, X), np.uint32)
        slab = O.make_slab(z, 1, z, 1)
        for f in range(frames):
            O.integrate(sc.dists[f], ref, sc.ovol(ref), synth.aff12(sc.vol2cam(f)), sc.intr, slab=slab)
        got = rig.data()[z:z + 1].cpu().numpy().view(np.uint32)
        assert np.array_equal(got, ref), (name, "rigid plane", z, compare_volumes(got, ref))
        wmax = max(wmax, int((ref >> 16).max()))
    assert wmax == frames
