"""Self-consistency of the oracle on the parts the reference has no vectors for (integrate, raycast,
dists, clear) -- plus the slab (Z-shard) variants the multi-GPU path relies on."""
import numpy as np

import oracle_lib as O
from dynamicfusion_amd import synth
from scene import Scene, decode

F32 = np.float32
CFG = synth.Config(48, 1.0, cols=128, rows=96, nodes=60, k=4)


def _integrated(sc, frames=2):
    vol = sc.new_volume()
    n = 0
    for f in range(frames):
        n += O.integrate(sc.dists[f], vol, sc.ovol(vol), synth.aff12(sc.vol2cam(f)), sc.intr)
    return vol, n


def test_half_roundtrip_and_ties():
    L = O.lib()
    for v in (0.0, 1.0, -1.0, 0.5, 6.1035e-05, 65504.0):
        assert L.orc_half2float(L.orc_float2half(v)) == np.float32(np.float16(v))
    assert L.orc_float2half(1.43505859375) == 0x3dbe           # exact tie -> even (cf. the v_fma_mixlo_f16 trap)


def test_compute_dists_formula():
    sc = Scene(CFG, n_frames=1, with_nodes=False)
    d = sc.depths[0]
    fx, fy, cx, cy = sc.intr
    y, x = 17, 101
    xl = (F32(x) - cx) * (F32(1) / fx)
    yl = (F32(y) - cy) * (F32(1) / fy)
    lam = np.sqrt(F32(xl * xl + yl * yl + F32(1)))
    want = np.float16(F32(F32(d[y, x]) * lam) * F32(0.001)).view(np.uint16)
    assert sc.dists[0][y, x] == want
    assert (sc.dists[0][d == 0] == 0).all()


def test_clear():
    sc = Scene(CFG, n_frames=1, with_nodes=False)
    vol = sc.new_volume()
    vol[:] = 0xdeadbeef
    O.lib().orc_clear(sc.ovol(vol), None)
    assert not vol.any()


def test_integrate_weights_and_truncation():
    sc = Scene(CFG, n_frames=3, with_nodes=False)
    vol, n = _integrated(sc, 3)
    t, w = decode(vol)
    assert n == int(w.astype(np.int64).sum())                   # every update increments exactly one weight
    assert w.max() == 3 and (t <= 1).all() and (t >= -1).all()
    assert (t[w == 0] == 0).all()                               # untouched voxels stay cleared


def test_integrate_max_weight_saturates():
    sc = Scene(CFG, n_frames=1, with_nodes=False)
    vol = sc.new_volume()
    v = O.make_volume(vol, CFG.dims, sc.vs, sc.trunc, 2)
    for _ in range(4):
        O.integrate(sc.dists[0], vol, v, synth.aff12(sc.vol2cam(0)), sc.intr)
    assert decode(vol)[1].max() == 2


def test_integrate_slabs_equal_full():
    sc = Scene(CFG, n_frames=1, with_nodes=False)
    full, n_full = _integrated(sc, 1)
    Z = CFG.dims[2]
    parts, n = [], 0
    for g in range(3):
        z0, zn = g * Z // 3, Z // 3
        part = sc.new_volume(zn)
        n += O.integrate(sc.dists[0], part, sc.ovol(part), synth.aff12(sc.vol2cam(0)), sc.intr, slab=O.make_slab(z0, zn, z0, zn))
        parts.append(part)
    assert n == n_full and np.array_equal(np.concatenate(parts, 0), full)


def test_warped_identity_nodes_equals_direct_rigid():
    """With default-constructed node transforms x_w == x_c exactly, so the warped sweep equals
    a rigid update evaluated with DIRECT (non-incremental) vc = world2cam*(vol2world*v)."""
    sc = Scene(CFG, n_frames=1, identity_warp=True)
    a = sc.new_volume()
    O.integrate_warped(sc.dists[0], a, sc.ovol(a), synth.aff12(sc.pose), synth.aff12(sc.world2cam(0)), sc.intr, sc.pos,
                       sc.dqs[0], sc.sigma, CFG.k)
    b, _ = _integrated(sc, 1)
    ta, wa = decode(a)
    tb, wb = decode(b)
    assert (wa != wb).mean() < 5e-3                             # only projection-boundary voxels may differ
    same = wa == wb
    assert np.abs(ta[same] - tb[same]).max() < 2e-2


def test_warped_slabs_equal_full():
    sc = Scene(CFG, n_frames=1)
    args = (synth.aff12(sc.pose), synth.aff12(sc.world2cam(0)), sc.intr, sc.pos, sc.dqs[0], sc.sigma, CFG.k)
    full = sc.new_volume()
    O.integrate_warped(sc.dists[0], full, sc.ovol(full), *args)
    Z = CFG.dims[2]
    parts = []
    for g in range(2):
        z0, zn = g * Z // 2, Z // 2
        p = sc.new_volume(zn)
        O.integrate_warped(sc.dists[0], p, sc.ovol(p), *args, slab=O.make_slab(z0, zn, z0, zn))
        parts.append(p)
    assert np.array_equal(np.concatenate(parts, 0), full)


def test_raycast_hits_reproduce_depth():
    sc = Scene(CFG, n_frames=2, with_nodes=False)
    vol, _ = _integrated(sc, 2)
    pts, nrm, keys, stats = O.raycast_points(sc.ovol(vol), synth.aff12(sc.cam2vol(0)), sc.rinv(0), sc.reproj, CFG.cols, CFG.rows,
                                             CFG.raycast_step_factor, CFG.gradient_delta_factor, want_keys=True)
    hit = np.isfinite(pts[..., 0])
    assert hit.mean() > 0.3 and int(stats[1]) == int(hit.sum())
    assert (keys[hit] & 1).all() and np.isnan(nrm[~hit]).all()
    m = hit & (sc.depths[0] > 0)
    assert np.median(np.abs(pts[..., 2][m] - sc.depths[0][m] / 1000.0)) < 0.02
    n = nrm[hit][:, :3]
    assert np.abs(np.linalg.norm(n, axis=1) - 1).max() < 1e-3
    dep, _ = O.raycast_depth(sc.ovol(vol), synth.aff12(sc.cam2vol(0)), sc.rinv(0), sc.reproj, CFG.cols, CFG.rows,
                             CFG.raycast_step_factor, CFG.gradient_delta_factor)
    assert np.array_equal(dep > 0, hit & (pts[..., 2] * 1000 >= 1))
    assert np.array_equal(dep[hit], (pts[..., 2][hit] * F32(1000)).astype(np.uint16))


def _two_stage_cast(sc, vol, f, world, halo, cfg):
    """march per slab -> MIN-merge of the 64-bit keys (event | rank | Ts) -> shade per slab -> integer-sum (what sharded.py does)."""
    Z = cfg.dims[2]
    slabs = []
    for r in range(world):
        z0, zn = sharded_mod().slab_range(Z, r, world)
        lo, hi = max(0, z0 - halo), min(Z, z0 + zn + halo)
        slabs.append((np.ascontiguousarray(vol[lo:hi]), O.make_slab(lo, hi - lo, z0, zn)))
    merged = np.full((cfg.rows, cfg.cols), sharded_mod().KEY_NONE, np.int64)
    for r, (part, slab) in enumerate(slabs):
        k, t = O.raycast_march(sc.ovol(part), synth.aff12(sc.cam2vol(f)), sc.reproj, cfg.cols, cfg.rows, cfg.raycast_step_factor, slab=slab)
        merged = np.minimum(merged, sharded_mod().pack_merge_keys(k, t, r))          # all_reduce(MIN)
    best, ts, _ = sharded_mod().unpack_merge_keys(merged)
    acc_p = np.zeros((cfg.rows, cfg.cols, 4), np.uint32)
    acc_n = np.zeros_like(acc_p)
    for part, slab in slabs:
        p, n = O.raycast_shade(sc.ovol(part), synth.aff12(sc.cam2vol(f)), sc.rinv(f), sc.reproj, ts, best, cfg.cols, cfg.rows,
                               cfg.gradient_delta_factor, slab=slab)
        acc_p += p.view(np.uint32)
        acc_n += n.view(np.uint32)
    return best, acc_p, acc_n


def sharded_mod():
    from dynamicfusion_amd import sharded
    return sharded


def test_raycast_two_stage_slab_cast_equals_full():
    """Per-slab march (own planes + halo), MIN-merge of event keys, shade by the slab owning the located vertex
    (which the refinement may extrapolate into another slab, tsdf_volume.cu:389), integer-sum of the outputs:
    reproduces the unsharded cast bit for bit."""
    cfg = synth.Config(96, 1.0, cols=160, rows=120, nodes=0)
    sc = Scene(cfg, n_frames=2, with_nodes=False)
    vol, _ = _integrated(sc, 2)
    fp, fn, fk, _ = O.raycast_points(sc.ovol(vol), synth.aff12(sc.cam2vol(1)), sc.rinv(1), sc.reproj, cfg.cols, cfg.rows,
                                     cfg.raycast_step_factor, cfg.gradient_delta_factor, want_keys=True)
    halo = sharded_mod().halo_planes(sc.trunc, cfg.raycast_step_factor, cfg.gradient_delta_factor, float(sc.vs[2]))
    for world in (2, 6):
        best, acc_p, acc_n = _two_stage_cast(sc, vol, 1, world, halo, cfg)
        assert np.array_equal(best, fk)
        assert np.array_equal(acc_p, fp.view(np.uint32)) and np.array_equal(acc_n, fn.view(np.uint32))


def test_project_and_remove_roundtrip():
    """project_kernel (tsdf_volume.cu:113-139) + psdf (tsdf_volume.cpp:266-292): back-projected pixel centres land on
    their own pixel, read its dists value, remove exactly those pixels; NaN / outside points follow the documented rules."""
    sc = Scene(CFG, n_frames=1, with_nodes=False)
    dists = O.compute_dists(sc.depths[0], sc.intr)
    fx, fy, cx, cy = [float(v) for v in sc.intr]
    ys, xs = np.mgrid[0:CFG.rows:3, 0:CFG.cols:3]
    z = (sc.depths[0][ys, xs].astype(F32) * F32(0.001)).ravel()
    keep = z > 0
    xs, ys, z = xs.ravel()[keep], ys.ravel()[keep], z[keep]
    pts = np.zeros((z.size, 4), F32)
    pts[:, 0] = (xs + 0.5 - cx) / fx * z; pts[:, 1] = (ys + 0.5 - cy) / fy * z; pts[:, 2] = z
    out, after, ro, n_in = O.project_and_remove(dists, pts, sc.intr)
    assert n_in == z.size
    Dp = dists[ys, xs].view(np.float16).astype(F32)
    assert np.array_equal(out[:, 2], Dp)
    assert np.allclose(out[:, 0], (xs + 0.5) * Dp, rtol=1e-5) and np.allclose(out[:, 1], (ys + 0.5) * Dp, rtol=1e-5)
    assert np.allclose(ro, Dp - z, atol=1e-6)                      # b22 == 1 up to one ulp
    removed = np.zeros_like(dists, bool); removed[ys, xs] = True
    assert np.array_equal(after[removed], np.zeros(removed.sum(), np.uint16)) and np.array_equal(after[~removed], dists[~removed])
    # NaN points untouched, outside points -> (qnan, qnan, qnan, 0), ro NaN for both; nothing removed
    odd = np.array([[np.nan, 0, 1, 7], [0, 0, -1e-3, 0], [10, 0, 1, 0], [0, -10, 1, 0], [1, 1, 0, 0]], F32)
    odd[1, :2] = [5, 5]                                           # behind the camera, far off-image
    out2, after2, ro2, n2 = O.project_and_remove(dists, odd, sc.intr)
    assert n2 == 0 and np.array_equal(after2, dists) and np.isnan(ro2).all()
    assert np.array_equal(out2[0].view(np.uint32), odd[0].view(np.uint32))
    assert (out2[1:, :3].view(np.uint32) == 0x7fffffff).all() and (out2[1:, 3] == 0).all()
    # empty input
    _, after3, ro3, n3 = O.project_and_remove(dists, np.zeros((0, 4), F32), sc.intr)
    assert n3 == 0 and ro3.size == 0 and np.array_equal(after3, dists)
