"""world_size-2 (and 3) CPU test of the multi-GPU Z-slab path over the gloo backend.

The collectives layer (dynamicfusion_amd/sharded.py: input broadcast, halo isend/irecv, per-pixel MIN merge of
ray-cast event keys, bit-exact gather of the winner's vertex/normal) is the product code under test; the
per-slab kernels are supplied by the oracle as a stand-in backend (tests may do that; the product path uses the
HIP C-ABI with the same DfSlab semantics, and tests/test_gpu_parity.py checks kernel-vs-oracle slab parity)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from dynamicfusion_amd import sharded, synth
from scene import Scene

CFG = synth.Config(48, 1.0, cols=128, rows=96, nodes=60, k=4)
FRAMES = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _unsharded(sc):
    vol = sc.new_volume()
    for f in range(FRAMES):
        O.integrate_warped(sc.dists[f], vol, sc.ovol(vol), synth.aff12(sc.pose), synth.aff12(sc.world2cam(f)), sc.intr,
                           sc.pos, sc.dqs[f], sc.sigma, CFG.k)
    p, n, _, _ = O.raycast_points(sc.ovol(vol), synth.aff12(sc.cam2vol(FRAMES - 1)), sc.rinv(FRAMES - 1), sc.reproj,
                                  CFG.cols, CFG.rows, CFG.raycast_step_factor, CFG.gradient_delta_factor)
    return vol, p, n


def _worker(rank, world, port, recompute_halo, balanced=False, merge="root", direct=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        O.lib().orc_set_num_threads(2)
        sc = Scene(CFG, n_frames=FRAMES)
        X, Y, Z = CFG.dims
        halo = sharded.halo_planes(sc.trunc, CFG.raycast_step_factor, CFG.gradient_delta_factor, float(sc.vs[2]))
        if balanced:                                   # work-balanced boundaries: decided on rank 0 from the first frame, broadcast
            bt = torch.zeros(world + 1, dtype=torch.int64)
            if rank == 0:
                w = sharded.frustum_plane_weights(CFG.dims, CFG.size, sc.pose, sc.cam_poses[0], CFG.intr, CFG.cols, CFG.rows,
                                                  depth_mm=sc.depths[0], trunc=sc.trunc, margin=0.1, samples=16)
                assert w.shape == (Z,) and (w > 0).all()
                w = w * np.linspace(0.05, 1.0, Z) ** 3     # (this small scene is nearly uniform in depth: make the far planes dear)
                bt.copy_(torch.tensor(sharded.slab_bounds(Z, world, halo, w), dtype=torch.int64))
            dist.broadcast(bt, 0)
            bounds = [int(v) for v in bt]
            sharded.validate_bounds(bounds, Z, halo)
            assert bounds != sharded.slab_bounds(Z, world)          # the scene really asks for unequal slabs
            z0, zn = bounds[rank], bounds[rank + 1] - bounds[rank]
        else:
            z0, zn = sharded.slab_range(Z, rank, world)
        lo, hi = max(0, z0 - halo), min(Z, z0 + zn + halo)
        slab = O.make_slab(lo, hi - lo, z0, zn)
        slab_int = O.make_slab(lo, hi - lo, lo, hi - lo) if recompute_halo else slab     # own := stored planes for the integrate
        vol = np.zeros((hi - lo, Y, X), np.uint32)
        vol_t = torch.from_numpy(vol.view(np.int32))                    # shares memory with `vol`
        pts = nrm = None
        for f in range(FRAMES):
            depth = torch.from_numpy(sc.depths[f].view(np.int16).copy()) if rank == 0 else torch.empty((CFG.rows, CFG.cols), dtype=torch.int16)
            dq = torch.from_numpy(sc.dqs[f].copy()) if rank == 0 else torch.empty((CFG.nodes, 8), dtype=torch.float32)
            if direct:                                                          # (round 6: N - 1 point-to-point sends in one group)
                sharded.coll_broadcast_direct(depth, 0)
                sharded.coll_broadcast_direct(dq, 0)
            else:
                sharded.broadcast_bytes(depth, 0)                               # rank 0 owns the sensor frame ...
                sharded.broadcast_bytes(dq, 0)                                  # ... and the solver's node transforms
            dists = O.compute_dists(depth.numpy().view(np.uint16), sc.intr)
            O.integrate_warped(dists, vol, sc.ovol(vol), synth.aff12(sc.pose), synth.aff12(sc.world2cam(f)), sc.intr,
                               sc.pos, dq.numpy(), sc.sigma, CFG.k, slab=slab_int)
            if not recompute_halo:
                sharded.exchange_halos(vol_t, lo, z0, zn, Z, halo, rank, world)

            def march():
                k, ts = O.raycast_march(sc.ovol(vol), synth.aff12(sc.cam2vol(f)), sc.reproj, CFG.cols, CFG.rows,
                                        CFG.raycast_step_factor, slab=slab)
                return torch.from_numpy(sharded.pack_merge_keys(k, ts, rank))         # the layout dfusion_raycast_march writes

            def shade(k64):
                merged, ts, _ = sharded.unpack_merge_keys(k64.numpy())
                _, n = O.raycast_shade(sc.ovol(vol), synth.aff12(sc.cam2vol(f)), sc.rinv(f), sc.reproj, ts, merged, CFG.cols, CFG.rows,
                                       CFG.gradient_delta_factor, slab=slab)
                return torch.from_numpy(n)

            def points(k64, normals):                                          # rank 0 only: no point crosses a link
                merged, ts, _ = sharded.unpack_merge_keys(k64.numpy())
                return torch.from_numpy(O.raycast_points_of_keys(synth.aff12(sc.cam2vol(f)), sc.rinv(f), sc.reproj, ts, merged, normals.numpy(),
                                                                 CFG.cols, CFG.rows))

            if merge in ("rows", "a2a"):
                per, bands = sharded.row_bands(CFG.rows, world)

                def shade_padded(k64):
                    pad = torch.zeros((world * per, CFG.cols, 4), dtype=torch.float32)
                    pad[:CFG.rows] = shade(k64)
                    return pad

                def points_band(k64, nb, r0, nr):                               # (the oracle takes whole images: embed the band, cut it out again)
                    full = torch.zeros((CFG.rows, CFG.cols, 4), dtype=torch.float32)
                    full[r0:r0 + nr] = nb
                    return points(k64, full)[r0:r0 + nr]

                timer = sharded.StageTimer(cuda=False)
                timer.start()
                pts, nrm, (r0, nr) = sharded.raycast_sharded(march, shade_padded, points_band, rank, world, merge=merge,
                                                             band_out=torch.empty((per, CFG.cols, 4), dtype=torch.float32),
                                                             a2a_recv=torch.empty((world, per, CFG.cols, 4), dtype=torch.float32) if merge == "a2a" else None,
                                                             timer=timer, key_merge="direct" if direct else "ring",
                                                             keys_pad=torch.full((world * per, CFG.cols), sharded.KEY_NONE, dtype=torch.int64) if direct else None,
                                                             keys_recv=torch.empty((world, per, CFG.cols), dtype=torch.int64) if direct else None,
                                                             keys_band=torch.empty((per, CFG.cols), dtype=torch.int64) if direct else None)
                timer.end()
                stages = list(timer.means())
                assert stages == ["march", "all_reduce_min", "shade", "all_to_all" if merge == "a2a" else "reduce_scatter", "points"], stages
            else:
                pts, nrm = sharded.raycast_sharded(march, shade, points, rank, world)
        # every rank checks its slab (own + halo planes) against the unsharded volume; rank 0 checks the merged cast
        full, fp, fn = _unsharded(sc)
        assert np.array_equal(vol, full[lo:hi]), "rank %d: slab (incl. halos) differs from the unsharded volume" % rank
        if merge in ("rows", "a2a"):                                            # every rank holds its band of the merged image
            assert r0 == bands[rank][0] and nr == bands[rank][1] and sum(b[1] for b in bands) == CFG.rows
            assert np.array_equal(pts.numpy().view(np.uint32), fp[r0:r0 + nr].view(np.uint32)), "rank %d: band of the merged vertices differs" % rank
            assert np.array_equal(nrm.numpy().view(np.uint32), fn[r0:r0 + nr].view(np.uint32)), "rank %d: band of the merged normals differs" % rank
        elif rank == 0:
            gp, gn = pts.numpy(), nrm.numpy()
            assert np.array_equal(gp.view(np.uint32), fp.view(np.uint32)), "merged ray-cast vertices differ"
            assert np.array_equal(gn.view(np.uint32), fn.view(np.uint32)), "merged ray-cast normals differ"
            assert np.isfinite(fp[..., 0]).mean() > 0.3
        else:
            assert pts is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("recompute_halo", [False, True], ids=["halo-exchange", "halo-recompute"])
@pytest.mark.parametrize("world", [2, 3])
def test_zslab_pipeline_over_gloo(world, recompute_halo):
    """halo-recompute: every rank integrates its halo planes itself (the integrate is a pure function of the broadcast inputs, so
    the planes come out bit-identical with the neighbour's) -- no halo collective at all; what bench.py does."""
    mp.spawn(_worker, args=(world, _free_port(), recompute_halo), nprocs=world, join=True)


def test_row_bands_cover_the_image_once():
    for rows, world in ((480, 8), (480, 7), (96, 5), (5, 8), (1, 1)):
        per, bands = sharded.row_bands(rows, world)
        assert per * world >= rows and len(bands) == world
        covered = []
        for r0, n in bands:
            assert 0 <= n <= per
            covered += list(range(r0, r0 + n))
        assert covered == list(range(rows))


@pytest.mark.parametrize("world", [2, 3])
def test_zslab_pipeline_row_banded_merge(world):
    """The ray-cast's second collective as a reduce_scatter by pixel rows (bench.py's default, round 4): every rank ends with its band of
    the merged image, bit-identical with the unsharded cast's rows."""
    mp.spawn(_worker, args=(world, _free_port(), True, False, "rows"), nprocs=world, join=True)


@pytest.mark.parametrize("world", [2, 3])
def test_zslab_pipeline_direct_all_to_all_merge(world):
    """Round 5: the second collective without a ring -- every rank sends each other rank ITS band of the normals it made
    (fixed-size pieces: no count exchange), the receiver adds the pieces of its band.  Same bits as the unsharded cast's rows."""
    mp.spawn(_worker, args=(world, _free_port(), True, False, "a2a"), nprocs=world, join=True)


@pytest.mark.parametrize("world", [2, 3])
def test_zslab_pipeline_with_every_collective_direct(world):
    """Round 6: no ring left in the frame -- the inputs go out as N - 1 point-to-point sends, the key merge is an all-to-all of row bands +
    local MIN + all-gather (coll_all_reduce_min_direct), the normals the round-5 all-to-all.  Same bits as the unsharded frame."""
    mp.spawn(_worker, args=(world, _free_port(), True, False, "a2a", True), nprocs=world, join=True)


def test_direct_key_merge_equals_all_reduce_min_on_ragged_bands():
    """One process, three emulated ranks, an image whose rows do not divide by the world: the band-wise minimum of the pieces, gathered, is
    the elementwise minimum of the images; padding rows never reach the image."""
    world, rows, cols = 3, 7, 5
    per, bands = sharded.row_bands(rows, world)
    g = torch.Generator().manual_seed(5)
    imgs = [torch.randint(0, 1 << 40, (rows, cols), generator=g, dtype=torch.int64) for _ in range(world)]
    imgs[1][2, 3] = sharded.KEY_NONE
    pads = [torch.full((world * per, cols), sharded.KEY_NONE, dtype=torch.int64) for _ in range(world)]
    for p, im in zip(pads, imgs):
        p[:rows] = im
    merged = torch.cat([torch.amin(torch.stack([pads[s][r * per:(r + 1) * per] for s in range(world)]), dim=0) for r in range(world)])
    assert torch.equal(merged[:rows], torch.amin(torch.stack(imgs), dim=0))
    assert (merged[rows:] == sharded.KEY_NONE).all()


def test_collective_model_orders_the_merges():
    # the stated model (tools/scale_model.py): the direct all-to-all beats the ring reduce_scatter from 4 ranks on, both beat reduce-to-root
    px16 = 640 * 480 * 16
    for n in (4, 8):
        a, rs, rd = (sharded.collective_model_s(k, px16, n) for k in ("all_to_all", "reduce_scatter", "reduce"))
        assert a < rs < rd
    assert sharded.collective_model_s("all_reduce", 1 << 20, 1) == 0.0
    for n in (2, 4, 8):                                    # the direct forms never lose to the rings they replace
        assert sharded.collective_model_s("all_reduce_direct", 640 * 480 * 8, n) <= sharded.collective_model_s("all_reduce", 640 * 480 * 8, n) + 16e-6
        assert sharded.collective_model_s("broadcast_direct", 680000, n) <= sharded.collective_model_s("broadcast", 680000, n)
    assert sharded.collective_model_s("all_reduce_direct", 640 * 480 * 8, 8) < 0.5 * sharded.collective_model_s("all_reduce", 640 * 480 * 8, 8)


def test_zslab_pipeline_over_gloo_with_work_balanced_slabs():
    """The same pipeline on UNEQUAL slabs (sharded.slab_bounds on frustum_plane_weights: the far ranks get thin slabs), boundaries
    decided on rank 0 and broadcast: still bit-identical with the unsharded frame."""
    mp.spawn(_worker, args=(2, _free_port(), True, True), nprocs=2, join=True)      # (48 planes, halo 11: two ranks leave room to move the boundary)


def test_slab_bounds_balance_and_respect_the_halo():
    w = np.concatenate([np.full(32, 0.02), np.linspace(0.05, 1.0, 96)])
    for world in (2, 3, 4, 8):
        b = sharded.slab_bounds(128, world, 8, w)
        sharded.validate_bounds(b, 128, 8)
        assert all(x % 8 == 0 for x in b) and b == sorted(b)
        share = [w[b[r]:b[r + 1]].sum() / w.sum() for r in range(world)]
        assert max(share) < 2.2 / world                    # no rank far above its fair share (8-plane granularity, 8-plane minimum)
        assert b[1] - b[0] > b[-1] - b[-2]                 # the cheap near planes make a thick slab, the dense far ones a thin one
    assert sharded.slab_bounds(128, 4) == [0, 32, 64, 96, 128]
    with pytest.raises(ValueError):
        sharded.validate_bounds([0, 60, 64, 128], 128, 8)  # a 4-plane slab cannot serve an 8-plane halo


def test_single_rank_is_a_no_op_path():
    p = torch.zeros((4, 4, 4))
    out = sharded.raycast_sharded(lambda: None, lambda k: p, lambda k, n: n, 0, 1)
    assert out[0] is p and out[1] is p
    sharded.exchange_halos(None, 0, 0, 4, 4, 2, 0, 1)


def test_validate_slabs_rejects_unworkable_partitions():
    """A partition that leaves a rank empty or thinner than the halo must raise before any collective."""
    import pytest
    from dynamicfusion_amd import sharded
    sharded.validate_slabs(512, 8, 8)
    sharded.validate_slabs(64, 2, 8)
    with pytest.raises(ValueError):
        sharded.validate_slabs(16, 3, 2)           # slab_range gives rank 2 zero planes
    with pytest.raises(ValueError):
        sharded.validate_slabs(64, 8, 13)          # 8 planes per rank < 13 halo planes
    with pytest.raises(ValueError):
        sharded.validate_slabs(4096, 129, 1)       # rank does not fit the 7-bit tag


def _brute_minmax(Z, world, halo, w, step=8):
    """every partition of Z planes into `world` slabs of multiples of `step` planes, each >= max(halo, step): the smallest largest cost"""
    import itertools
    mn = (max(halo, step) + step - 1) // step * step
    best, arg = None, None
    cuts = range(step, Z, step)
    for c in itertools.combinations(cuts, world - 1):
        b = [0] + list(c) + [Z]
        if any(b[i + 1] - b[i] < mn for i in range(world)):
            continue
        m = max(sharded.slab_costs(b, w, halo))
        if best is None or m < best - 1e-12:
            best, arg = m, b
    return best, arg


def test_minmax_slab_bounds_are_optimal_on_small_volumes_and_valid_on_large_ones():
    """sharded.slab_bounds_minmax (round 6; what bench.py's ranks and tools/scale_model.py cut the volume with): against a brute force over every
    8-aligned partition on small volumes -- the largest rank's cost, halos included, is the minimum --, and on headline-sized ones the
    boundaries pass validate_bounds and never cost more than the equal-shares cut they replace."""
    rng = np.random.default_rng(11)
    for trial in range(40):
        Z = int(rng.choice([64, 96, 128])); world = int(rng.integers(2, 5)); halo = int(rng.choice([0, 3, 8]))
        w = rng.random(Z) ** 3 + 0.02
        if rng.random() < 0.3:
            w[: Z // 3] = 0.02                                                   # (a cheap near third, like the frustum's)
        want, arg = _brute_minmax(Z, world, halo, w)
        if want is None:
            continue
        b = sharded.slab_bounds_minmax(Z, world, halo, w)
        sharded.validate_bounds(b, Z, halo)
        got = max(sharded.slab_costs(b, w, halo))
        assert got <= want * (1 + 1e-9) + 1e-12, (Z, world, halo, b, arg, got, want)
    for world in (2, 4, 8):
        Z, halo = 512, 8
        w = np.concatenate([np.full(160, 0.02), np.linspace(0.05, 1.0, 352) ** 2]) * (1 + 0.1 * rng.random(Z))
        b = sharded.slab_bounds_minmax(Z, world, halo, w)
        sharded.validate_bounds(b, Z, halo)
        assert b[0] == 0 and b[-1] == Z and all(x % 8 == 0 for x in b)
        assert max(sharded.slab_costs(b, w, halo)) <= max(sharded.slab_costs(sharded.slab_bounds(Z, world, halo, w), w, halo)) * (1 + 1e-9)
    # a world that cannot be served (more minimum-thickness slabs than planes) falls back to slab_bounds' answer instead of failing
    assert sharded.slab_bounds_minmax(64, 8, 8, np.ones(64)) == sharded.slab_bounds(64, 8, 8, np.ones(64))


def test_reweight_from_times_is_a_fixed_point_for_consistent_times_and_moves_work_off_the_slow_rank():
    """sharded.reweight_from_times (round 6): times proportional to the swept cost (+ the fixed part) leave the weights' PROFILE alone -- the
    boundaries made from them do not move; a rank that measured twice its share gets its planes made dearer, and the re-cut takes planes
    away from it."""
    rng = np.random.default_rng(3)
    Z, world, halo = 256, 4, 8
    w = np.linspace(0.1, 1.0, Z) * (1 + 0.05 * rng.random(Z))
    b = sharded.slab_bounds_minmax(Z, world, halo, w)
    costs = np.array(sharded.slab_costs(b, w, halo))
    w2 = sharded.reweight_from_times(b, w, halo, 0.045 + 0.01 * costs, 0.045)
    assert np.allclose(w2 / w, (w2 / w)[0]) and np.isfinite(w2).all() and (w2 > 0).all()
    assert sharded.slab_bounds_minmax(Z, world, halo, w2) == b
    t = 0.045 + 0.01 * costs
    t[2] = 0.045 + 0.02 * costs[2]                                               # rank 2 took twice as long as its cost says
    w3 = sharded.reweight_from_times(b, w, halo, t, 0.045)
    assert (w3[b[2]:b[3]] / w[b[2]:b[3]]).min() > (w3[b[0]:b[1]] / w[b[0]:b[1]]).max()
    b3 = sharded.slab_bounds_minmax(Z, world, halo, w3)
    sharded.validate_bounds(b3, Z, halo)
    assert b3[3] - b3[2] < b[3] - b[2]                                           # fewer planes for the slow rank
    # degenerate inputs: a time below the fixed part, a zero-cost slab
    w4 = sharded.reweight_from_times(b, w, halo, [0.01, 0.2, 0.2, 0.2], 0.045)
    assert np.isfinite(w4).all() and (w4 > 0).all()


def test_cxx_slab_bounds_equal_the_python_ones():
    """kfusion::cuda::ZSlabComm::slabBounds / slabBoundsMinMax (what a C++ host cuts the volume with) against sharded.slab_bounds /
    slab_bounds_minmax on random work profiles, worlds 1..8, several halos: the same boundaries, plane for plane.  (`zslab_frame bounds`
    touches no GPU; the binary is one of build()'s host-side products.)"""
    import subprocess
    from dynamicfusion_amd import build
    if not os.path.exists(build.HOST_ZSLAB_APP):
        pytest.skip("host apps not built (python -c 'import __graft_entry__ as g; g.build()')")
    import tempfile
    rng = np.random.default_rng(2)
    with tempfile.TemporaryDirectory() as d:
        for t in range(40):
            Z = int(rng.choice([64, 96, 128, 512, 1000])); world = int(rng.integers(1, 9)); halo = int(rng.choice([0, 3, 8, 11]))
            w = rng.random(Z) ** 3 + 0.02
            path = os.path.join(d, "w.f64")
            w.astype(np.float64).tofile(path)
            for mode, fn in (("bounds", sharded.slab_bounds), ("bounds-minmax", sharded.slab_bounds_minmax)):
                r = subprocess.run([build.HOST_ZSLAB_APP, mode, str(world), str(halo), path], capture_output=True, text=True, timeout=60)
                assert r.returncode == 0, r.stderr
                assert [int(x) for x in r.stdout.strip().split(",")] == fn(Z, world, halo, w), (mode, Z, world, halo)
