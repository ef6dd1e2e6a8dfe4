#!/usr/bin/env python3
"""bench.py -- frames/sec of the DynamicFusion hot path on MI355X.

A "step" = one frame of the hot path on device-resident synthetic inputs:
    set node transforms -> compute_dists -> integrate_warped (per-voxel k-NN + DQB + TSDF update)
    -> raycast (Points), [N>1: + broadcast of the frame inputs, redundant halo integrate, ray-cast merge].
Workload = BASELINE.json configs[2] (headline): 640x480 depth -> 512^3 TSDF (3 m), ~2000 warp nodes,
k = 8.  N>1 shards the SAME volume by Z-slab (strong scaling).

Prints ONE JSON line (rank 0) with `roofline` (integrate kernel, HIP-event timed inside the timed
region) and `cpu_baseline` (the oracle timed on a bounded sample of the same workload, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from dynamicfusion_amd import Intr, TsdfVolume, WarpField, capi, compute_dists, sharded, synth, upload_u16  # noqa: E402

HBM_PEAK_GBPS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="512", choices=sorted(synth.CONFIGS))
    ap.add_argument("--frames", type=int, default=4, help="distinct synthetic frames cycled through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rigid", action="store_true", help="(kept for old command lines; the extra kernels are timed by default)")
    ap.add_argument("--halo", choices=["recompute", "exchange"], default="recompute",
                    help="N>1: how a rank gets its neighbours' boundary planes for the ray-cast -- recompute them (default: every rank "
                         "integrates its halo planes too, no collective) or exchange them after the integrate (paired isend/irecv over RCCL, "
                         "the north star's wording; one more collective per frame, 2*H fewer planes to sweep)")
    ap.add_argument("--slabs", choices=["balanced", "uniform"], default="balanced",
                    help="N>1: Z-slab boundaries -- equal shares of the integrate's WORK (planes weighted by how much of them lies inside the frustum "
                         "and in front of the first frame's surface; the far slabs are thin) or equal plane counts")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra objects (rigid_integrate, extract_cloud, kinfu_frame)")
    ap.add_argument("--no-kinfu", action="store_true", help="skip the kinfu_frame extra (it runs a child process; use under profilers)")
    return ap.parse_args()


def measured_copy_gbps(nbytes=1 << 30, iters=10):
    src = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(src)
    src.zero_()
    L = capi.lib()
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        capi.check(L.dfusion_copy_bandwidth_probe(dst.data_ptr(), src.data_ptr(), nbytes, st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        capi.check(L.dfusion_copy_bandwidth_probe(dst.data_ptr(), src.data_ptr(), nbytes, st))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return 2.0 * nbytes / (ms * 1e-3) / 1e9        # read + write bytes


def cpu_baseline(cfg, sc_inputs, vol_u32, target_s=10.0, gpu_after=None):
    """The reference-style CPU path timed on the host cores on a BOUNDED sample of the same frame: per-voxel warped
    integrate through the reference's OWN nanoflann k-NN + DQB + transform classes (oracle/_ref, one tree / result set per OpenMP
    thread -- the reference's WarpField is single-threaded and not re-entrant) on a band of Z planes in the middle of the volume,
    scaled to all planes, + the full ray-cast (oracle restatement, pinned to the reference's kernel).  kind = "reference".
    Also reported: the same band on ONE thread (how the reference itself would run it) and the oracle port (restated nanoflann).
    gpu_after: the GPU's volume after integrating THE SAME frame from the same start volume -- the band the reference classes just
    computed is compared with it voxel for voxel (`integrate_bit_identical`: reference vs HIP at the headline size, no oracle between)."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_lib as O
    depth, _, pose, cam_pose, pos, sigma, dq = sc_inputs
    intr = np.array(cfg.intr, np.float32)
    dists = O.compute_dists(depth, intr)
    vs = np.array([np.float32(cfg.size) / np.float32(d) for d in cfg.dims], np.float32)
    trunc = float(max(np.float32(cfg.trunc_dist), np.float32(2.1) * vs.max()))
    X, Y, Z = cfg.dims
    world2cam = synth.affine_inv(cam_pose)
    have_ref = O.have_ref()

    last_ref = {}

    def timed_band(n_planes, mode, threads=0):
        z0 = (Z - n_planes) // 2
        sample = np.ascontiguousarray(vol_u32[z0:z0 + n_planes]).copy()
        t0 = time.time()
        if mode == "reference":
            _, used = O.ref_integrate_warped(dists, sample, cfg.dims, vs, trunc, cfg.max_weight, synth.aff12(pose), synth.aff12(world2cam), intr,
                                             pos, dq, sigma, cfg.k, z0, z0, n_planes, threads=threads)
            if threads == 0:
                last_ref["z0"], last_ref["band"] = z0, sample
        else:
            ov = O.make_volume(sample, cfg.dims, vs, trunc, cfg.max_weight)
            O.integrate_warped(dists, sample, ov, synth.aff12(pose), synth.aff12(world2cam), intr, pos, dq, sigma, cfg.k,
                               slab=O.make_slab(z0, n_planes, z0, n_planes))
            used = int(O.lib().orc_num_threads())
        return time.time() - t0, used

    def sized_band(mode, budget_s):
        planes, (t, used) = 4, timed_band(4, mode)
        for _ in range(2):                                  # grow the band until it is about budget_s of work (or the whole volume)
            if t >= 0.5 * budget_s or planes >= Z:
                break
            planes = int(min(Z, max(planes + 4, 4 * round(planes * budget_s / max(t, 1e-3) / 4))))
            t, used = timed_band(planes, mode)
        return planes, t, used

    kind = "reference" if have_ref else "port"
    planes, t_band, cores = sized_band(kind, target_s)
    t_int = t_band * (Z / planes)
    full = O.make_volume(vol_u32, cfg.dims, vs, trunc, cfg.max_weight)
    cam2vol = synth.affine_mul(synth.affine_inv(pose), cam_pose)
    rinv = np.linalg.inv(cam2vol[:3, :3].astype(np.float64)).astype(np.float32)
    reproj = np.array([np.float32(1) / np.float32(cfg.intr[0]), np.float32(1) / np.float32(cfg.intr[1]), cfg.intr[2], cfg.intr[3]], np.float32)
    t0 = time.time()
    _, _, _, rc_stats = O.raycast_points(full, synth.aff12(cam2vol), rinv, reproj, cfg.cols, cfg.rows, cfg.raycast_step_factor,
                                         cfg.gradient_delta_factor)
    t_ray = time.time() - t0
    # SURVEY.md 8(d) algorithmic bytes of the ray-cast: 4 B per nearest-voxel fetch (steps + 1 per ray), 256 B per hit (2 + 6 trilinear
    # evaluations x 8 taps x 4 B), 32 B per pixel of output -- steps and hits counted by the oracle on the same volume and pose
    rc_bytes = 4.0 * (float(rc_stats[0]) + cfg.cols * cfg.rows) + 256.0 * float(rc_stats[1]) + 32.0 * cfg.cols * cfg.rows
    out = {"value": 1.0 / (t_int + t_ray), "unit": "frames/s", "cores": cores, "kind": kind,
           "raycast_algorithmic_bytes": rc_bytes, "raycast_steps": int(rc_stats[0]), "raycast_hits": int(rc_stats[1]),
           "sample": "per-voxel warped integrate through the reference's own nanoflann + DQB classes (oracle/_ref, OpenMP, one tree per thread) on "
                     "%d of %d Z planes in %.1f s on %d threads, scaled x%.1f (est %.2f s/frame) + full %dx%d ray-cast (%.2f s)"
                     % (planes, Z, t_band, cores, Z / planes, t_int, cfg.cols, cfg.rows, t_ray)}
    if gpu_after is not None and "band" in last_ref:
        z0, band = last_ref["z0"], last_ref["band"]
        diff = band != gpu_after[z0:z0 + band.shape[0]]
        out["integrate_bit_identical"] = bool(not diff.any())
        out["integrate_compare"] = {"what": "volume planes [%d, %d) after this frame: the reference's WarpField::DQB + DualQuaternion::transform classes "
                                            "composed with TsdfIntegrator (oracle/_ref, unmodified headers) vs dfusion_integrate_warped, same start volume"
                                            % (z0, z0 + band.shape[0]),
                                    "voxels_compared": int(band.size), "voxels_differing": int(diff.sum()),
                                    "voxels_updated_in_band": int((band != vol_u32[z0:z0 + band.shape[0]]).sum())}
    if have_ref:
        # one thread, as the reference's own (non-re-entrant) WarpField would run it: one plane is ~0.2-0.5 s
        t1, _ = timed_band(2, "reference", threads=1)
        out["single_thread"] = {"value": 1.0 / (t1 * Z / 2 + t_ray * cores), "unit": "frames/s", "cores": 1,
                                "sample": "same path on 1 thread, 2 of %d planes in %.2f s (est %.0f s/frame integrate; ray-cast scaled by the thread count)" % (Z, t1, t1 * Z / 2)}
        # the oracle port (restated nanoflann, what the parity tests run), a short band
        p_planes, p_t, p_cores = sized_band("port", 3.0)
        out["port"] = {"value": 1.0 / (p_t * Z / p_planes + t_ray), "unit": "frames/s", "cores": p_cores,
                       "sample": "oracle restatement on %d planes in %.1f s" % (p_planes, p_t)}
    return out


def reference_warp_baseline(cfg, pts_dev, pos, sigma, dq, wf):
    """The reference's own CPU path for the per-frame warp (kinfu.cpp:356-383 -> WarpField::warp, warp_field.cpp:180-195: nanoflann
    k-NN + DQB + transform per point, single-threaded and not re-entrant) -- the reference's headers compiled unmodified into
    oracle/_ref/libdfref.so -- timed on the valid ray-cast points of the last frame, beside dfusion_warp_points on the same points.
    Returns None when the reference build is not present (it is built where /root/reference exists and travels with the snapshot)."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_lib as O
    if not O.have_ref():
        return None
    p = pts_dev.reshape(-1, 4)[:, :3].contiguous()
    p = p[~torch.isnan(p).any(dim=1)].contiguous()
    host = p.cpu().numpy().astype(np.float32)
    t0 = time.time()
    ref_out, _ = O.warp_points(pos, dq, sigma, host, None, cfg.k, use_ref=True)
    t_ref = time.time() - t0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    work = p.clone(); wf.warp(work)                         # warm-up (and the result to compare)
    same = bool(np.array_equal(work.cpu().numpy().view(np.uint32), ref_out.view(np.uint32)))
    e0.record()
    for _ in range(10):
        work.copy_(p); wf.warp(work)
    e1.record(); torch.cuda.synchronize()
    e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e2.record()
    for _ in range(10):
        work.copy_(p)
    e3.record(); torch.cuda.synchronize()
    ms_gpu = (e0.elapsed_time(e1) - e2.elapsed_time(e3)) / 10
    return {"kind": "reference", "what": "WarpField::warp of the frame's valid ray-cast points: reference nanoflann + DQB code (oracle/_ref) on one host "
                                         "thread vs dfusion_warp_points; called twice per frame by KinFu::dynamicfusion",
            "points": int(host.shape[0]), "cpu_seconds": t_ref, "cpu_threads": 1, "gpu_ms": ms_gpu, "bit_identical": same}


def kinfu_frame_ms(cfg, frames=12):
    """Wall clock of kfusion::KinFu::operator() in dynamicfusion_amd/host/kinfu_headless on the synthetic sequence."""
    import re
    import subprocess
    import tempfile
    from dynamicfusion_amd import build
    build.build_host()
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            f.write(np.asarray(cfg.intr, np.float32).tobytes())
            for i in range(frames):
                f.write(synth.depth_frame(cfg, i).tobytes())
        r = subprocess.run([build.HOST_KINFU_APP, str(cfg.cols), str(cfg.rows), str(frames), str(cfg.dims[0]), str(cfg.size), fin, fout],
                           capture_output=True, text=True, timeout=300)
    m = re.search(r"(\d+) warp nodes, (\d+) surface points, ([0-9.]+) ms/frame", r.stdout)
    if r.returncode != 0 or not m:
        raise RuntimeError((r.stdout + r.stderr)[-200:])
    return {"ms": float(m.group(3)), "warp_nodes": int(m.group(1)), "surface_points": int(m.group(2)), "frames": frames,
            "what": "C++ kfusion::KinFu::operator() per frame, wall clock: bilateral, pyramid, normals, 19 ICP iterations, ray-cast, "
                    "2x WarpField::warp, psdf, fusion, extract, ray-cast, resize (device-resident data flow)"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # DFUSION_BENCH_FORCE_DIST=1: take the N > 1 code path (RCCL process group, slab volume, broadcast, the merge collectives) with
    # whatever WORLD_SIZE is -- with one rank it is an RCCL dry run of every collective, dtype and op of the sharded frame
    # (tests/test_gpu_sharded.py runs it), since an 8-GPU node is only ever seen by the driver
    dist_on = world > 1 or os.environ.get("DFUSION_BENCH_FORCE_DIST") == "1"
    if dist_on:
        torch.cuda.set_device(local_rank)
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    assert args.gpus == world, "--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world)
    dev = torch.device("cuda", torch.cuda.current_device())

    cfg = synth.CONFIGS[args.config]
    intr = Intr(*cfg.intr)
    X, Y, Z = cfg.dims
    F = args.frames

    # ---- synthetic inputs, resident in HBM before the timed region
    depths_np = [synth.depth_frame(cfg, f) for f in range(F)]
    depths = [upload_u16(d, dev) for d in depths_np]
    cam_poses = [synth.camera_pose(cfg, f) for f in range(F)]
    pos, sigma = synth.make_nodes(cfg)
    dqs_np = [synth.node_transforms(cfg, f) for f in range(F)]
    dqs = [torch.from_numpy(d).to(dev) for d in dqs_np]

    vs_z = cfg.size / Z
    trunc_eff = max(cfg.trunc_dist, 2.1 * vs_z)
    halo = sharded.halo_planes(trunc_eff, cfg.raycast_step_factor, cfg.gradient_delta_factor, vs_z)
    slab_bounds = None
    if dist_on:
        # the partition is decided once, from the first sensor frame, on rank 0 (which owns the sensor) and broadcast: every rank must
        # hold the SAME boundaries before the first slab is allocated
        bt = torch.zeros(world + 1, dtype=torch.int64, device=dev)
        if rank == 0:
            wts = None
            if args.slabs == "balanced" and world > 1:
                wts = sharded.frustum_plane_weights(cfg.dims, cfg.size, cfg.volume_pose, cam_poses[0], cfg.intr, cfg.cols, cfg.rows,
                                                    depth_mm=depths_np[0], trunc=trunc_eff, margin=0.3)
            bt.copy_(torch.tensor(sharded.slab_bounds(Z, world, halo, wts), dtype=torch.int64))
        dist.broadcast(bt, 0)
        slab_bounds = [int(v) for v in bt.cpu()]
        sharded.validate_bounds(slab_bounds, Z, halo)      # same verdict on every rank, before the first data collective
        z_own0, z_own_n = slab_bounds[rank], slab_bounds[rank + 1] - slab_bounds[rank]
        vol = TsdfVolume(cfg.dims, device=dev, slab=(z_own0, z_own_n, halo))
    else:
        vol = TsdfVolume(cfg.dims, device=dev)
    vol.setSize([cfg.size] * 3); vol.setTruncDist(cfg.trunc_dist); vol.setMaxWeight(cfg.max_weight)
    vol.setPose(cfg.volume_pose)
    vol.setRaycastStepFactor(cfg.raycast_step_factor); vol.setGradientDeltaFactor(cfg.gradient_delta_factor)
    vol.clear()

    # N > 1: every rank also integrates its halo planes (a pure function of the broadcast inputs: bit-identical with the neighbour's
    # planes), so the frame has NO halo collective; `vol_int` is the same blob seen as owner of all its stored planes.
    vol_int = vol.owning_stored_planes() if (dist_on and args.halo == "recompute") else vol
    wf = WarpField(k=cfg.k, device=dev)
    wf.init(pos, sigma=sigma, transforms=dqs_np[0])
    t0_index = time.time()
    wf.ensure_index(vol_int, cfg.k)
    torch.cuda.synchronize()

    dists = torch.empty_like(depths[0])
    keys = torch.empty((cfg.rows, cfg.cols), dtype=torch.int64, device=dev) if dist_on else None
    out2 = torch.empty((2, cfg.rows, cfg.cols, 4), dtype=torch.float32, device=dev)
    pts, nrm = out2[0], out2[1]
    # frame inputs travel as ONE byte bundle (depth image + node transforms): one ncclBroadcast per frame
    n_depth = cfg.rows * cfg.cols * 2
    bundle = torch.empty(n_depth + cfg.nodes * 32, dtype=torch.uint8, device=dev)
    depth_in = bundle[:n_depth].view(torch.int16).view(cfg.rows, cfg.cols)
    dq_in = bundle[n_depth:].view(torch.float32).view(cfg.nodes, 8)

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]

    def step(i, timed_idx=None):
        f = i % F
        if dist_on:                                    # rank 0 owns the sensor frame and the solver output
            if rank == 0:
                depth_in.copy_(depths[f]); dq_in.copy_(dqs[f])
            dist.broadcast(bundle, 0)
            d_in, q_in = depth_in, dq_in
        else:
            d_in, q_in = depths[f], dqs[f]
        wf.set_transforms(q_in)
        compute_dists(d_in, intr, dists)
        if timed_idx is not None: ev[timed_idx][0].record()
        vol_int.integrate_warped(dists, cam_poses[f], intr, wf, sync=False)
        if timed_idx is not None: ev[timed_idx][1].record()
        if dist_on and args.halo == "exchange":
            sharded.exchange_halos(vol.data(), vol.z_store0, vol.z_own0, vol.z_own_n, Z, halo, rank, world)
        if dist_on:
            out = sharded.raycast_sharded(lambda: vol.raycast_march(cam_poses[f], intr, keys, rank),
                                          lambda mk: vol.raycast_shade(cam_poses[f], intr, mk, None, nrm)[1],
                                          lambda mk, n: vol.raycast_points_of_keys(cam_poses[f], intr, mk, n, pts),
                                          rank, world, collectives=True)
        else:
            vol.raycast(cam_poses[f], intr, pts, nrm)
            out = (pts, nrm)
        if timed_idx is not None: ev[timed_idx][2].record()
        return out

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    # The per-node-set work: with tables on demand (the mirrors' default) `ensure_index` above only builds the brick index; the tables,
    # then the blend models, of the blocks the launch plans find alive are made by the first sweeps.  Three untimed frames pay that
    # here, whatever --warmup is (what newly alive blocks cost as the camera moves stays inside the timed frames); the volume is
    # cleared again.  `index_build_once_s` covers index + these frames.
    for i in range(3):
        step(i)
    vol.clear()
    barrier()
    t_index = time.time() - t0_index
    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, timed_idx=i)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_int = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    ms_ray = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
    # per-frame spread (integrate + ray-cast between HIP events on the launch stream) and the reference-shaped frame
    # (KinFu::dynamicfusion integrates once and ray-casts twice)
    per_frame = np.array([e[0].elapsed_time(e[2]) for e in ev], np.float64)
    frame_stats = {"integrate+raycast_ms": {"p10": float(np.percentile(per_frame, 10)), "median": float(np.median(per_frame)),
                                            "p90": float(np.percentile(per_frame, 90))},
                   "reference_shaped_ms": ms_int + 2.0 * ms_ray}

    # ---- algorithmic bytes of one integrate launch: 8*N_upd + 2*W*H + 48*M.
    # N_upd counted by the kernel itself (parity-checked against the oracle's count in tests/), untimed pass.
    n_upd = torch.zeros(2, dtype=torch.int64, device=dev)          # [0] updated, [1] swept (the plan's alive cells, dfusion_debug_warp_counters)
    capi.check(capi.lib().dfusion_debug_warp_counters(n_upd[1:].data_ptr()))
    for f in range(F):
        vol.integrate_warped(compute_dists(depths[f], intr, dists), cam_poses[f], intr, wf, n_updated=n_upd[:1], sync=False)
    torch.cuda.synchronize()
    capi.check(capi.lib().dfusion_debug_warp_counters(None))
    n_upd_launch = float(n_upd[0].item()) / F
    n_swept_launch = float(n_upd[1].item()) / F
    if dist_on:
        t = torch.tensor([n_upd_launch], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        n_upd_total = float(t.item())
    else:
        n_upd_total = n_upd_launch
    alg_bytes = 8.0 * n_upd_launch + 2.0 * cfg.cols * cfg.rows + 48.0 * cfg.nodes
    achieved = alg_bytes / (ms_int * 1e-3) / 1e9

    extra = {}
    if not args.no_extras and not dist_on:
        vol2 = TsdfVolume(cfg.dims, device=dev)
        vol2.setSize([cfg.size] * 3); vol2.setTruncDist(cfg.trunc_dist); vol2.setMaxWeight(cfg.max_weight); vol2.setPose(cfg.volume_pose)
        nr = torch.zeros(2, dtype=torch.int64, device=dev)       # [0] updated, [1] swept (dfusion_debug_rigid_counters)
        for f in range(2):
            vol2.integrate(dists, cam_poses[f], intr, sync=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20):
            vol2.integrate(dists, cam_poses[i % F], intr, sync=False)
        e1.record()
        torch.cuda.synchronize()
        capi.check(capi.lib().dfusion_debug_rigid_counters(nr[1:].data_ptr()))
        for f in range(F):
            vol2.integrate(dists, cam_poses[f], intr, n_updated=nr[:1], sync=False)
        torch.cuda.synchronize()
        capi.check(capi.lib().dfusion_debug_rigid_counters(None))
        ms_r = e0.elapsed_time(e1) / 20
        b_r = 8.0 * float(nr[0].item()) / F + 2.0 * cfg.cols * cfg.rows
        extra["rigid_integrate"] = {"kernel": "df_integrate_rigid_kernel<2, true, false> (+ df_rigid_plan_kernel, df_pyramid_tiles_kernel)",
                                    "ms": ms_r, "achieved_GBps": b_r / (ms_r * 1e-3) / 1e9, "frac_of_peak": b_r / (ms_r * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                    "n_updated": float(nr[0].item()) / F, "n_swept": float(nr[1].item()) / F,
                                    "swept_over_updated": float(nr[1].item()) / max(float(nr[0].item()), 1.0),
                                    "algorithmic_bytes_per_launch": b_r}
        del vol2
        # surface extraction (kinfu.cpp:398-399) on the fused volume: a pure HBM scan, 4 B/voxel
        st = torch.cuda.current_stream().cuda_stream
        buf = torch.empty((1 << 22, 4), dtype=torch.float32, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        aff = capi.floats(synth.aff12(vol.getPose()))
        ex = lambda: capi.check(capi.lib().dfusion_extract_cloud(vol.c_volume(), None, aff, buf.data_ptr(), buf.shape[0], cnt.data_ptr(), st))
        for _ in range(3):
            ex()
        cnt.zero_()
        e0.record()
        for _ in range(20):
            ex()
        e1.record()
        torch.cuda.synchronize()
        ms_e = e0.elapsed_time(e1) / 20
        n_pts = float(cnt.item()) / 20
        b_e = 4.0 * X * Y * Z + 16.0 * n_pts
        sink = torch.zeros(4, dtype=torch.int32, device=dev)
        src = torch.zeros(1 << 30, dtype=torch.uint8, device=dev)
        rd = lambda: capi.check(capi.lib().dfusion_read_bandwidth_probe(src.data_ptr(), 1 << 30, sink.data_ptr(), st))
        rd(); e0.record()
        for _ in range(10):
            rd()
        e1.record(); torch.cuda.synchronize()
        read_gbps = (1 << 30) / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e9
        extra["extract_cloud"] = {"kernel": "df_extract_kernel<4>", "ms": ms_e, "points": n_pts, "bound": "hbm",
                                  "achieved_GBps": b_e / (ms_e * 1e-3) / 1e9, "frac_of_peak": b_e / (ms_e * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                  "measured_read_GBps": read_gbps}
        del buf, src
        # ---- what the per-voxel caches buy and cost: the timed frame leaves the k-NN search and the weights in the
        # 48 B/voxel cache, which is valid while node POSITIONS stand (the reference's node set is static after init; the solver
        # only writes transforms).  A frame after the node set changed pays the rebuild first; the lean path never caches.
        try:
            reb = []
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                wf.set_nodes(*wf._keep)                   # positions "changed": tree replica + invalidated index
                wf.ensure_index(vol_int, cfg.k)           # brick lists + per-voxel k-NN / weight tables
                vol.integrate_warped(dists, cam_poses[0], intr, wf, sync=False)
                vol.raycast(cam_poses[0], intr, pts, nrm)
                torch.cuda.synchronize(); reb.append((time.perf_counter() - t0) * 1e3)
            extra["frame_nodes_changed_ms"] = {"ms": float(np.median(reb)), "what": "set_nodes (incl. the nanoflann tree replica, built on the host) + brick index "
                                               "+ per-voxel k-NN / weight tables rebuilt + the frame itself; wall clock, median of 3"}
            lean = WarpField(k=cfg.k, device=dev, voxel_table=False)
            lean.init(pos, sigma=sigma, transforms=dqs_np[0])
            lean.ensure_index(vol, cfg.k)
            vol.integrate_warped(dists, cam_poses[0], intr, lean, sync=False)
            e0.record()
            for i in range(5):
                vol.integrate_warped(dists, cam_poses[i % F], intr, lean, sync=False)
            e1.record(); torch.cuda.synchronize()
            extra["lean_path_ms"] = {"integrate_warped_ms": e0.elapsed_time(e1) / 5, "what": "no per-voxel cache (brick candidate lists only, ~80 MB): exact top-k "
                                     "over ~150 candidates per voxel and the weights' exp() every frame (df_warp_brick_kernel)"}
            del lean
        except Exception as e:
            extra["frame_nodes_changed_ms"] = {"error": repr(e)[:200]}
        # one whole KinFu::operator() frame (front-end, ICP, dynamicfusion, ray-cast) through the C++ mirror, for context
        if not args.no_kinfu:
            try:
                extra["kinfu_frame"] = kinfu_frame_ms(cfg)
            except Exception as e:      # the extras never fail the bench line
                extra["kinfu_frame"] = {"error": str(e)[:200]}

    # kernel that ran + its per-voxel cache footprint; PMC traffic comes from the committed rocprofv3 --pmc passes
    lds_ok = cfg.nodes * 32 <= 160 * 1024
    if lds_ok and cfg.k in (4, 8):
        axis_aligned = bool(np.array_equal(np.asarray(cfg.volume_pose, np.float32).reshape(4, 4)[:3, :3], np.eye(3, dtype=np.float32)))
        kernel_name = "df_warp_rows_pipe_kernel<%d, 2, %d, %s>" % (cfg.k, 1024 if cfg.nodes * 32 > 80 * 1024 else 512, "true" if axis_aligned else "false")
    elif lds_ok:
        kernel_name = "df_warp_rows_lds_kernel<%d, true, 2>" % cfg.k
    else:
        kernel_name = "df_warp_rows_kernel<%d, true, 4>" % cfg.k
    table_bytes = int(X) * Y * vol.z_own_n * cfg.k * 6
    traffic, traffic_src = None, None
    pmc_file = os.path.join(REPO, "profiles", "pmc_latest.json")
    if not dist_on and os.path.exists(pmc_file):
        try:
            pm = json.load(open(pmc_file))
            ent = pm.get(args.config, {}).get(kernel_name.split("<")[0])
            # the counters belong to ONE build of the kernel: the file carries the sha256 of the source it was taken from, and a
            # figure from another build is not reported
            from dynamicfusion_amd import build as _build
            src_sha = _build.kernel_source_sha("df_warp_rows_pipe_kernel")
            if ent and ent.get("source_sha256") == src_sha:
                traffic, traffic_src = ent["hbm_bytes_per_launch"], "profiles/pmc_latest.json (%s)" % ent.get("how", "rocprofv3 --pmc")
            elif ent:
                traffic_src = "profiles/pmc_latest.json is from another build of the sweep's sources (sha256 %s...): traffic dropped" % str(ent.get("source_sha256"))[:12]
        except Exception:
            pass
    if rank == 0:
        copy_gbps = measured_copy_gbps() if not dist_on else None
        out = {
            "metric": "frames/sec integrate+raycast, %dx%d->%d^3 TSDF" % (cfg.cols, cfg.rows, cfg.dims[0]),
            "value": args.steps / elapsed,
            "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32 (fp16 TSDF / u16 weight storage, f64 exp + normalise as in the reference)",
            "data": "synthetic",
            "config": {"workload": cfg.name, "volume_dims": list(cfg.dims), "volume_size_m": cfg.size,
                       "depth": [cfg.cols, cfg.rows], "warp_nodes": cfg.nodes, "k": cfg.k,
                       "parallelism": "zslab%d" % world if dist_on else "single", "halo_planes": halo if dist_on else 0,
                       "slab_bounds": slab_bounds, "slabs": args.slabs if dist_on else None,
                       "halo": (("integrated redundantly by every rank, no halo collective" if args.halo == "recompute" else
                                 "exchanged after the integrate (paired isend/irecv of %d planes per side)" % halo) if dist_on else None),
                       "frame": "set_transforms + compute_dists + integrate_warped + raycast_points"},
            "kernel_ms": {"integrate_warped": ms_int, "raycast(+merge)": ms_ray, "index_build_once_s": t_index},
            "frame_stats": frame_stats,
            "roofline": {"kernel": kernel_name, "bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes, "n_updated_per_launch": n_upd_launch,
                         "n_swept_per_launch": n_swept_launch, "swept_over_updated": (n_swept_launch / n_upd_launch) if n_upd_launch else None,
                         "n_updated_all_ranks": n_upd_total, "measured_copy_GBps": copy_gbps,
                         "knn_cache_bytes": table_bytes,
                         # the same achieved rate against the copy rate measured on this box, what the kernel really moves per
                         # launch (PMC) against the algorithmic bytes, and the rate of that real traffic -- the sweep's actual HBM load
                         "frac_of_measured_copy": (achieved / copy_gbps) if copy_gbps else None,
                         "traffic_over_algorithmic": (traffic / alg_bytes) if traffic else None,
                         "traffic_GBps": (traffic / (ms_int * 1e-3) / 1e9) if traffic else None,
                         "traffic_frac_of_measured_copy": (traffic / (ms_int * 1e-3) / 1e9 / copy_gbps) if (traffic and copy_gbps) else None,
                         "note": "achieved = SURVEY 8(d) algorithmic bytes (8*N_upd + 2*W*H + 48*M) / HIP-event time; the sweep "
                                 "also streams its per-voxel k-NN + weight cache (48 B/voxel at k=8), see DESIGN.md"},
        }
        out.update(extra)
        if not dist_on and not args.no_cpu_baseline:
            vol_host = vol.download()
            # the same frame on the GPU from the same start volume (for cpu_baseline.integrate_bit_identical)
            wf.set_transforms(dqs[0])
            vol.integrate_warped(compute_dists(depths[0], intr, dists), cam_poses[0], intr, wf)
            vol_after = vol.download()
            out["cpu_baseline"] = cpu_baseline(cfg, (depths_np[0], None, cfg.volume_pose, cam_poses[0], pos, sigma, dqs_np[0]), vol_host,
                                               gpu_after=vol_after)
            del vol_after
            rcb = out["cpu_baseline"].pop("raycast_algorithmic_bytes")
            out["raycast"] = {"kernel": "df_raycast_kernel<0>", "ms": ms_ray, "algorithmic_bytes": rcb,
                              "steps": out["cpu_baseline"].pop("raycast_steps"), "hits": out["cpu_baseline"].pop("raycast_hits"),
                              "achieved_GBps": rcb / (ms_ray * 1e-3) / 1e9,
                              "note": "gather-latency bound (one dependent 4-byte fetch per march step); reported, no roofline target (SURVEY 8d)"}
            try:
                wf.set_transforms(dqs[0])
                rw = reference_warp_baseline(cfg, pts, pos, sigma, dqs_np[0], wf)
                if rw:
                    out["cpu_baseline"]["reference_warp"] = rw
            except Exception as e:                      # the reference build is optional; never lose the bench line over it
                out["cpu_baseline"]["reference_warp"] = {"error": repr(e)[:200]}
        line = json.dumps(out)
    if dist_on:
        dist.destroy_process_group()
    if rank == 0:
        # the ONE JSON line is the last thing on stdout: RCCL writes its version banner through C stdio, which a pipe buffers until
        # exit -- flush that first
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(line, flush=True)


if __name__ == "__main__":
    main()
