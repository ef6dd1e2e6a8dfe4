"""Z-slab sharding of the HIP kernels, emulated on ONE GPU: N slab volumes (own planes + halos), halo copies
in place of ncclSend/Recv, event-key min-merge -- must reproduce the unsharded HIP result bit for bit.
(The collective layer itself is covered over gloo in tests/test_sharded_cpu.py; the driver runs the real
multi-GPU bench.)"""
import numpy as np
import pytest
import torch

from dynamicfusion_amd import Intr, sharded, synth, upload_u16
from scene import Scene
from test_gpu_parity import MID, SMALL, make_gpu_volume, make_gpu_warp

pytestmark = pytest.mark.gpu


def run_sharded(sc, world, frames, k=None, recompute_halo=False):
    cfg = sc.cfg
    intr = Intr(*cfg.intr)
    Z = cfg.dims[2]
    halo = sharded.halo_planes(sc.trunc, cfg.raycast_step_factor, cfg.gradient_delta_factor, float(sc.vs[2]))
    vols = []
    for r in range(world):
        z0, zn = sharded.slab_range(Z, r, world)
        vols.append(make_gpu_volume(sc, slab=(z0, zn, halo)))
    wf = make_gpu_warp(sc, k=k)
    for f in range(frames):
        wf.set_transforms(torch.from_numpy(sc.dqs[f]).cuda())
        d = upload_u16(sc.dists[f])
        for v in vols:
            (v.owning_stored_planes() if recompute_halo else v).integrate_warped(d, sc.cam_poses[f], intr, wf)
        for r in range(0 if recompute_halo else world - 1):   # halo exchange between Z neighbours r <-> r+1 (not needed when recomputed)
            a, b = vols[r], vols[r + 1]
            a_hi = a.z_own0 + a.z_own_n - a.z_store0     # local index one past a's last own plane
            b_lo = b.z_own0 - b.z_store0
            b.data()[0:b_lo].copy_(a.data()[a_hi - b_lo:a_hi])                       # a's top own planes -> b's lower halo
            n_hi = a.data().shape[0] - a_hi
            a.data()[a_hi:a_hi + n_hi].copy_(b.data()[b_lo:b_lo + n_hi])             # b's bottom own planes -> a's upper halo
    f = frames - 1
    # stage 1 per slab, then what all_reduce(MIN) computes: the whole merge (first event, owner, Ts)
    k64s = []
    for r, v in enumerate(vols):
        k64 = torch.empty((cfg.rows, cfg.cols), dtype=torch.int64, device="cuda")
        v.raycast_march(sc.cam_poses[f], intr, k64, rank=r)
        k64s.append(k64)
    merged = torch.stack(k64s).min(0).values.contiguous()
    # the DIRECT key merge (round 6, --key-merge direct): rank r receives every rank's copy of band r of the padded key image and takes the per-key
    # minimum on the device (dfusion_raycast_min_pieces); the bands, gathered, are what all_reduce(MIN) computes
    from dynamicfusion_amd import capi
    per_k, bands_k = sharded.row_bands(cfg.rows, world)
    pads = torch.full((world, world * per_k, cfg.cols), sharded.KEY_NONE, dtype=torch.int64, device="cuda")
    for s_, k64 in enumerate(k64s):
        pads[s_, :cfg.rows] = k64
    gathered = torch.empty((world * per_k, cfg.cols), dtype=torch.int64, device="cuda")
    for r in range(world):
        pieces_k = pads[:, r * per_k:(r + 1) * per_k].contiguous()
        capi.check(capi.lib().dfusion_raycast_min_pieces(pieces_k.data_ptr(), world, pieces_k[0].numel(), gathered[r * per_k:(r + 1) * per_k].data_ptr(), None),
                   "dfusion_raycast_min_pieces")
    assert torch.equal(gathered[:cfg.rows], merged) and bool((gathered[cfg.rows:] == sharded.KEY_NONE).all())
    one = torch.empty((4,), dtype=torch.int64, device="cuda")                      # (an odd count is fine with a single piece; not with several)
    assert capi.lib().dfusion_raycast_min_pieces(merged.data_ptr(), 1, 3, one.data_ptr(), None) == 0 and torch.equal(one[:3], merged.view(-1)[:3])
    assert capi.lib().dfusion_raycast_min_pieces(merged.data_ptr(), 2, 3, one.data_ptr(), None) != 0
    best = torch.where(merged == sharded.KEY_NONE, torch.full_like(merged, 0xFFFFFFFF), (merged >> 39) & 0xFFFFFF)
    acc = torch.zeros((2, cfg.rows, cfg.cols, 4), dtype=torch.int32, device="cuda")
    for v in vols:                                      # stage 2 + what reduce(SUM) of the bit patterns computes
        p = torch.empty((cfg.rows, cfg.cols, 4), dtype=torch.float32, device="cuda")
        n = torch.empty_like(p)
        v.raycast_shade(sc.cam_poses[f], intr, merged, p, n)
        acc[0] += p.view(torch.int32)
        acc[1] += n.view(torch.int32)
    # stage 3: the points need no exchange -- from the merged keys and the SUMMED normals they come out as the sum of the slabs' points
    p3 = torch.empty((cfg.rows, cfg.cols, 4), dtype=torch.float32, device="cuda")
    vols[0].raycast_points_of_keys(sc.cam_poses[f], intr, merged, acc[1].view(torch.float32), p3)
    n_only = torch.empty_like(p3)
    vols[-1].raycast_shade(sc.cam_poses[f], intr, merged, None, n_only)          # (points_dev = NULL is accepted)
    # the row-banded merge (round 4): a rank that holds only ITS band of the summed normals makes exactly its band of the points
    per, bands = sharded.row_bands(cfg.rows, world)
    for r0, nr in bands:
        if nr:
            pb = torch.empty((nr, cfg.cols, 4), dtype=torch.float32, device="cuda")
            vols[0].raycast_points_of_keys(sc.cam_poses[f], intr, merged, acc[1][r0:r0 + nr].view(torch.float32).contiguous(), pb, r0, nr)
            assert torch.equal(pb.view(torch.int32), acc[0][r0:r0 + nr])
    # the DIRECT row-band merge (round 5, --merge a2a): rank r receives every rank's piece of ITS band of the padded normals image and adds
    # them on the device (dfusion_raycast_sum_pieces) -- the band of the summed normals, bit for bit
    shaded = torch.zeros((world, world * per, cfg.cols, 4), dtype=torch.float32, device="cuda")       # [source rank][padded image]
    for s_, v in enumerate(vols):
        v.raycast_shade(sc.cam_poses[f], intr, merged, None, shaded[s_, :cfg.rows])
    for r, (r0, nr) in enumerate(bands):
        pieces = shaded[:, r * per:(r + 1) * per].contiguous()                                        # what the all-to-all delivers to rank r
        band = torch.empty((per, cfg.cols, 4), dtype=torch.float32, device="cuda")
        capi.check(capi.lib().dfusion_raycast_sum_pieces(pieces.data_ptr(), world, pieces[0].numel(), band.data_ptr(), None), "dfusion_raycast_sum_pieces")
        assert torch.equal(band[:nr].view(torch.int32), acc[1][r0:r0 + nr])
        assert not band[nr:].view(torch.int32).any()                                                  # (rows past the image stay zero)
    torch.cuda.synchronize()
    assert torch.equal(p3.view(torch.int32), acc[0])
    return vols, acc[0], acc[1], best


@pytest.mark.parametrize("recompute_halo", [False, True], ids=["halo-exchange", "halo-recompute"])
@pytest.mark.parametrize("cfg,world", [(SMALL, 2), (SMALL, 4), (MID, 4)], ids=["64x2", "64x4", "128x4"])
def test_slab_pipeline_equals_unsharded(cfg, world, recompute_halo):
    frames = 2
    sc = Scene(cfg, n_frames=frames)
    intr = Intr(*cfg.intr)
    full = make_gpu_volume(sc)
    wf = make_gpu_warp(sc)
    for f in range(frames):
        wf.set_transforms(torch.from_numpy(sc.dqs[f]).cuda())
        full.integrate_warped(upload_u16(sc.dists[f]), sc.cam_poses[f], intr, wf)
    fp = torch.empty((cfg.rows, cfg.cols, 4), dtype=torch.float32, device="cuda")
    fn = torch.empty_like(fp)
    fk = torch.empty((cfg.rows, cfg.cols), dtype=torch.int32, device="cuda")
    full.raycast(sc.cam_poses[frames - 1], intr, fp, fn, keys=fk)

    vols, mp, mn, best = run_sharded(sc, world, frames, recompute_halo=recompute_halo)
    ref = full.data()
    for v in vols:                                        # own planes AND exchanged halos equal the unsharded volume
        assert torch.equal(v.data(), ref[v.z_store0:v.z_store0 + v.z_store_n])
    assert torch.equal(best, fk.to(torch.int64) & 0xFFFFFFFF)
    assert (~torch.isnan(fp)).float().mean() > 0.2
    assert torch.equal(mp, fp.view(torch.int32)) and torch.equal(mn, fn.view(torch.int32))     # bit-identical incl. NaN fill


def test_bench_sharded_branch_runs_over_rccl_with_one_rank():
    """bench.py's N > 1 code path -- slab volume, halo-recompute integrate view, the frame-input broadcast (uint8), the merge's
    all_reduce(MIN) on int64 keys, the reduce(SUM) on int32 views, the float64 MAX / SUM of the timing, barrier -- through a REAL
    RCCL process group of one rank (DFUSION_BENCH_FORCE_DIST=1).  RCCL refuses two ranks on one device, so N > 1 itself is only ever
    run by the driver's 8-GPU node; this is the dry run that proves every dtype / op of the frame exists in RCCL before that, and that
    the sharded frame's result is the unsharded one."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = {}
    for mode, env in (("single", {}), ("rccl1", {"DFUSION_BENCH_FORCE_DIST": "1", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})):
        for halo in (("recompute", "exchange") if mode == "rccl1" else ("recompute",)):
            r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--config", "256", "--steps", "4", "--warmup", "1", "--no-extras",
                                "--no-cpu-baseline", "--halo", halo], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
            assert r.stdout.strip().splitlines()[-1].startswith('{"metric"'), r.stdout[-600:]      # the JSON line is the LAST line (driver contract)
            lines[(mode, halo)] = json.loads(r.stdout.strip().splitlines()[-1])
    one = lines[("single", "recompute")]
    for halo in ("recompute", "exchange"):
        d = lines[("rccl1", halo)]
        assert d["config"]["parallelism"] == "zslab1" and d["n_gpus"] == 1
        # same frames, same inputs: the sharded path updates exactly the voxels the unsharded one does
        assert d["roofline"]["n_updated_all_ranks"] == one["roofline"]["n_updated_per_launch"]
        assert d["value"] > 0


def _bench(args, env_extra=None, drop_rank_env=True):
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not (drop_rank_env and k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"))}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py")] + args, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith('{"metric"'), r.stdout[-600:]
    return json.loads(last)


def test_bench_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus N` with NO rank environment (how the driver starts N = 1) must not die on the launch.  On this
    one-GPU box the self-launched ranks share cuda:0 and the collectives run over gloo with host staging (RCCL refuses two ranks on a
    device): the whole sharded frame -- measured re-balance of the slabs, frame-input broadcast, halo recompute, march / merge / shade,
    verify-cull pass -- runs end to end, and updates exactly the voxels the single-GPU run does."""
    common = ["--config", "256", "--steps", "3", "--warmup", "1", "--no-extras", "--no-cpu-baseline"]
    one = _bench(common)
    assert one["n_gpus"] == 1 and one["verify_cull"]["cull_bit_identical"] and one["frame_stats"]["integrate+raycast_ms"]["frames"] >= 40
    for n, extra in ((2, []), (3, ["--halo", "exchange", "--slabs", "balanced"])):
        d = _bench(["--gpus", str(n)] + common + extra)
        assert d["n_gpus"] == n and d["config"]["parallelism"] == "zslab%d" % n and "oversubscribed" in d
        assert d["roofline"]["n_updated_all_ranks"] >= one["roofline"]["n_updated_per_launch"]      # (+ the halo planes every rank also integrates)
        if not extra:
            assert d["roofline"]["n_updated_all_ranks"] > one["roofline"]["n_updated_per_launch"]
            rb = d["config"]["rebalance"]
            assert rb["to"][0] == 0 and rb["to"][-1] == 256 and sum(rb["alive_blocks_per_8_planes"]) > 0
        assert d["verify_cull"]["cull_bit_identical"] and d["verify_cull"]["updates_with_cull"] == d["verify_cull"]["updates_without_cull"] > 0
        assert d["value"] > 0
        # round 5: what the first multi-GPU hardware run will be read from -- per rank, per stage; the ranks that took part;
        # the model's predicted collective times; the other merges and the halo exchange timed in the same launch
        sd = d["scaling_detail"]
        assert sd["rccl_ranks_seen"] == n and len(sd["devices"]) == n
        for st in ("broadcast", "integrate_warped", "march", "all_reduce_min", "shade", "points"):
            assert st in sd["stages"] and len(sd["per_rank_ms"][st]) == n and all(v > 0 for v in sd["per_rank_ms"][st]), st
        assert set(sd["variants"]) == {"merge=rows", "merge=a2a", "merge=root", "halo=exchange", "key_merge=ring", "bcast=ring"}
        assert d["config"]["key_merge"].startswith("direct") and "point-to-point" in d["config"]["inputs"]
        assert sd["predicted_collective_ms"]["all_reduce_min"] < sd["predicted_collective_ms"]["all_reduce_min(ring)"] + 0.02
        assert "all_to_all" in sd["variants"]["merge=a2a"]["per_rank_ms"] and "halo_exchange" in sd["variants"]["halo=exchange"]["per_rank_ms"]
        assert all(v["ms_per_frame"] > 0 for v in sd["variants"].values())
        assert sd["predicted_collective_ms"]["all_to_all"] > 0 and sd["predicted_collective_ms"]["all_reduce_min"] > sd["predicted_collective_ms"]["broadcast"]


def test_bench_force_dist_self_launch_is_one_real_rccl_rank():
    """the same launcher with one rank: torch.distributed.run -> init_process_group("nccl") -> every collective of the frame over RCCL"""
    d = _bench(["--gpus", "1", "--config", "256", "--steps", "3", "--warmup", "1", "--no-extras", "--no-cpu-baseline"], {"DFUSION_BENCH_FORCE_DIST": "1"})
    assert d["n_gpus"] == 1 and d["config"]["parallelism"] == "zslab1" and "oversubscribed" not in d
    assert d["verify_cull"]["cull_bit_identical"] and d["value"] > 0


def test_alive_block_profile_of_slabs_adds_up_to_the_unsharded_one():
    """dfusion_warp_alive_blocks (the measured re-balance's input): per 8-plane layer, the blocks a sweep's verdict pass kept.  With the
    ball test alone (no block models: a verdict then depends on the block and the frame only) the profiles of two slabs, each counting
    its OWN layers, add up to the unsharded volume's; and re-cutting the slabs by it moves the boundary towards the thin side."""
    sc = Scene(MID, n_frames=1)
    intr = Intr(*MID.intr)
    Z = MID.dims[2]
    d = upload_u16(sc.dists[0])

    def profile(slab):
        v = make_gpu_volume(sc, slab=slab) if slab else make_gpu_volume(sc)
        wf = make_gpu_warp(sc)
        v.integrate_warped(d, sc.cam_poses[0], intr, wf, block_model=False)
        out = torch.zeros(Z // 8, dtype=torch.int64, device="cuda")
        wf.alive_blocks_per_layer(v, out)
        return out.cpu().numpy()
    full = profile(None)
    halves = profile((0, Z // 2, 0)) + profile((Z // 2, Z // 2, 0))
    assert full.sum() > 0 and np.array_equal(full, halves)
    b = sharded.slab_bounds(Z, 2, 8, sharded.layer_weights_to_planes(full, Z))
    assert b[0] == 0 and b[2] == Z and b[1] % 8 == 0
    w = sharded.layer_weights_to_planes(full, Z)
    assert abs(w[:b[1]].sum() - w[b[1]:].sum()) <= abs(w[:Z // 2].sum() - w[Z // 2:].sum()) + 1e-9


@pytest.mark.parametrize("cuts", [(0, 12, 37, 64), (0, 4, 60, 64), (0, 29, 35, 64)], ids=["12-37", "4-60", "29-35"])
def test_slabs_that_cut_through_layers_equal_the_unsharded_sweep(cuts):
    """Slab boundaries that are NOT multiples of 8 planes (the C-ABI takes any DfSlab): the pipelined sweep deals a workgroup's work out
    in half-layer cells (round 4), so an own range that ends inside a layer, or inside its first half, clips the first / last cell of some
    wave's segment -- every plane must still be swept exactly once, with and without the cull."""
    sc = Scene(SMALL, n_frames=2)
    intr = Intr(*SMALL.intr)
    full = make_gpu_volume(sc)
    wf = make_gpu_warp(sc)
    for f in range(2):
        wf.set_transforms(torch.from_numpy(sc.dqs[f]).cuda())
        full.integrate_warped(upload_u16(sc.dists[f]), sc.cam_poses[f], intr, wf)
    for kw in (dict(), dict(cull=False)):
        for a, b in zip(cuts[:-1], cuts[1:]):
            v = make_gpu_volume(sc, slab=(a, b - a, 0))
            w2 = make_gpu_warp(sc)
            n = torch.zeros(1, dtype=torch.int64, device="cuda")
            for f in range(2):
                w2.set_transforms(torch.from_numpy(sc.dqs[f]).cuda())
                v.integrate_warped(upload_u16(sc.dists[f]), sc.cam_poses[f], intr, w2, n_updated=n, **kw)
            assert torch.equal(v.data(), full.data()[a:b]), (a, b, kw)
            assert int(n.item()) == int(((v.data() >> 16) & 0xffff).sum().item())       # every update counted once (weights start at 0, 2 frames)


def test_slab_march_from_random_cameras_equals_the_unsharded_cast():
    """Z-slab march + key merge + owner's shade from cameras that look across the slabs instead of along them (rolled, from the side,
    from inside the volume): the first events, normals and points of the unsharded cast, bit for bit."""
    from test_gpu_parity import _filled
    cfg = SMALL
    sc = Scene(cfg, n_frames=2, with_nodes=False)
    full, ref = _filled(sc)
    intr = Intr(*cfg.intr)
    Z = cfg.dims[2]
    world = 3
    halo = sharded.halo_planes(sc.trunc, cfg.raycast_step_factor, cfg.gradient_delta_factor, float(sc.vs[2]))
    slabs = []
    for r in range(world):
        z0, zn = sharded.slab_range(Z, r, world)
        v = make_gpu_volume(sc, slab=(z0, zn, halo))
        v.data().copy_(full.data()[v.z_store0:v.z_store0 + v.z_store_n])
        slabs.append(v)
    rng = np.random.RandomState(5)
    centre = (sc.pose @ np.array([cfg.size / 2] * 3 + [1.0], np.float32))[:3]
    for i in range(8):
        axis = rng.randn(3); axis /= np.linalg.norm(axis)
        ang = rng.uniform(0.2, 3.0)
        K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
        cam = np.eye(4, dtype=np.float32)
        cam[:3, :3] = R.astype(np.float32)
        cam[:3, 3] = (centre + (R @ np.array([0.0, 0.0, -1.0])) * rng.uniform(0.0, 1.5) * cfg.size).astype(np.float32)
        p0 = torch.empty((cfg.rows, cfg.cols, 4), dtype=torch.float32, device="cuda"); n0 = torch.empty_like(p0)
        k0 = torch.empty((cfg.rows, cfg.cols), dtype=torch.int32, device="cuda")
        full.raycast(cam, intr, p0, n0, keys=k0)
        k64s = []
        for r, v in enumerate(slabs):
            k64 = torch.empty((cfg.rows, cfg.cols), dtype=torch.int64, device="cuda")
            v.raycast_march(cam, intr, k64, rank=r)
            k64s.append(k64)
        merged = torch.stack(k64s).min(0).values.contiguous()
        ev = torch.where(merged == sharded.KEY_NONE, torch.full_like(merged, 0xFFFFFFFF), (merged >> 39) & 0xFFFFFF).to(torch.int32)
        assert torch.equal(ev, k0), "camera %d: first events differ" % i
        acc = torch.zeros((cfg.rows, cfg.cols, 4), dtype=torch.int32, device="cuda")
        for v in slabs:
            n = torch.empty((cfg.rows, cfg.cols, 4), dtype=torch.float32, device="cuda")
            v.raycast_shade(cam, intr, merged, None, n)
            acc += n.view(torch.int32)
        p1 = torch.empty_like(p0)
        slabs[0].raycast_points_of_keys(cam, intr, merged, acc.view(torch.float32), p1)
        assert torch.equal(acc, n0.view(torch.int32)), "camera %d: normals differ" % i
        assert torch.equal(p1.view(torch.int32), p0.view(torch.int32)), "camera %d: points differ" % i
