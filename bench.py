#!/usr/bin/env python3
"""bench.py -- frames/sec of the DynamicFusion hot path on MI355X.

A "step" = one frame of the hot path on device-resident synthetic inputs:
    set node transforms -> compute_dists -> integrate_warped (per-voxel k-NN + DQB + TSDF update)
    -> raycast (Points), [N>1: + broadcast of the frame inputs, redundant halo integrate, ray-cast merge].
Workload = BASELINE.json configs[2] (headline): 640x480 depth -> 512^3 TSDF (3 m), ~2000 warp nodes,
k = 8.  N>1 shards the SAME volume by Z-slab (strong scaling).

The camera goes somewhere: frame f looks from 0.25 deg * f (KinFu::operator() never revisits a pose,
/root/reference/kfusion/src/kinfu.cpp:274-297), priming, warm-up and timed frames are consecutive frames of ONE monotone sweep, so
the blocks the moving camera brings into view pay for their on-demand k-NN / weight tables and blend models INSIDE the timed frames.
(`--frames F` cycles F poses instead -- round 3's steady-state loop; the default run reports it as frame_stats.steady_state_loop_ms.)

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches its own ranks (torch.distributed.run, one per
GPU, rendezvous on 127.0.0.1), prints rank 0's ONE JSON line last and exits with the ranks' status; under torchrun it is a rank.
A box with fewer GPUs than ranks runs the same code over gloo with host-staged collectives (a smoke of the code path, flagged
`oversubscribed`); a box with no GPU runs the launcher and the frame's collectives only (`dry_run`, value null).

Prints ONE JSON line (rank 0) with `roofline` (integrate kernel, HIP-event timed inside the timed
region) and `cpu_baseline` (the oracle timed on a bounded sample of the same workload, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
N_PRIME = 3                 # untimed frames after the index build (first tables, first blend models), volume cleared after them
MIN_STAT_FRAMES = 40        # frame_stats percentiles are taken over at least this many consecutive frames of the sweep


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="512", choices=["cpu128", "256", "512", "1024"])
    ap.add_argument("--nodes", type=int, default=0,
                    help="warp nodes instead of the config's (0 = the config's): e.g. --nodes 8000, the density the reference's own WarpField::init "
                         "(every 50th point of the first cloud, warp_field.cpp:41-63) gives a 640 x 480 frame")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="the default N = 1 headline run also measures BASELINE configs 2 (256^3 / 500 nodes / k = 4) and 5 (1024^3 / 1280 x 960 / 5000 nodes, "
                         "if >= 70 GB of HBM are free) and the headline geometry at 8000 nodes, each in a child process after the headline's timed region "
                         "(`other_configs` in the line); this switch skips them")
    ap.add_argument("--frames", type=int, default=0,
                    help="0 (default): a monotone camera sweep, one new pose per frame (0.25 deg / frame); F > 0: F synthetic frames cycled "
                         "(a steady-state loop: every table and block model exists before the timed region)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rigid", action="store_true", help="(kept for old command lines; the extra kernels are timed by default)")
    ap.add_argument("--halo", choices=["recompute", "exchange", "both"], default="recompute",
                    help="N>1: how a rank gets its neighbours' boundary planes for the ray-cast -- recompute them (default: every rank "
                         "integrates its halo planes too, no collective) or exchange them after the integrate (paired isend/irecv over RCCL, "
                         "the north star's wording; one more collective per frame, 2*H fewer planes to sweep); `both` = `recompute` for the "
                         "headline, and the exchange form timed as a variant in the same launch (scaling_detail.variants -- what N>1 runs do by default)")
    ap.add_argument("--slabs", choices=["balanced", "uniform", "measured"], default="measured",
                    help="N>1: Z-slab boundaries -- `balanced`: equal shares of an a-priori work model (planes weighted by how much of them lies "
                         "inside the frustum and in front of the first frame's surface); `measured` (default): start from `balanced`, then "
                         "re-cut once after the priming frames from the verdict pass's alive-block counts per plane (what the sweep really "
                         "visits), slabs re-allocated; `uniform`: equal plane counts")
    ap.add_argument("--merge", choices=["rows", "a2a", "root"], default="a2a",
                    help="N>1: the ray-cast's second collective -- `a2a` (default since round 5): every rank finishes its band of pixel rows, the "
                         "normals reach it by ONE direct all-to-all of fixed-size pieces + a local sum (no ring, no count exchange; the stated "
                         "collective model puts it at 0.026 ms against 0.093 for the ring at N = 8); `rows` (round 4's default): the same bands by "
                         "reduce_scatter (1/N of the bytes lands on a rank); `root`: reduce(SUM) to rank 0, which makes "
                         "the whole image.  N>1 runs time the other forms as variants after the timed region (scaling_detail.variants)")
    ap.add_argument("--key-merge", choices=["direct", "ring"], default="direct",
                    help="N>1: the ray-cast's FIRST collective, the per-pixel MIN of the 64-bit keys -- `direct` (default since round 6): all-to-all of "
                         "the keys' row bands + local minimum (dfusion_raycast_min_pieces) + all-gather of the merged bands: two exchanges of one step "
                         "each over the pairwise xGMI links (stated model at N = 8: 0.046 ms); `ring`: one ncclAllReduce(MIN) (2 (N - 1) ring steps: "
                         "0.128 ms in the same model).  Same bits; the other form is timed as a variant")
    ap.add_argument("--bcast", choices=["direct", "ring"], default="direct",
                    help="N>1: how the frame's inputs (depth + node transforms, one 0.68 MB bundle) leave rank 0 -- `direct` (default since round 6): "
                         "N - 1 point-to-point sends in one group, each on its own link; `ring`: ncclBroadcast")
    ap.add_argument("--no-variants", action="store_true", help="N>1: skip the untimed variant passes (the other merges, the halo exchange)")
    ap.add_argument("--long-frames", type=int, default=200,
                    help="frame_stats.long_sweep: the sweep goes on (untimed by the wall clock, HIP events per frame) until this many consecutive frames "
                         "have been measured -- the headline's 20 poses move +-6 %% with which poses they are (VERDICT r4 #7); 0 = off")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra objects (rigid_integrate, extract_cloud, kinfu_frame)")
    ap.add_argument("--no-kinfu", action="store_true", help="skip the kinfu_frame extra (it runs a child process; use under profilers)")
    ap.add_argument("--no-verify-cull", action="store_true",
                    help="skip the untimed pass that re-integrates every timed frame with the launch plan's cull switched off, from the same "
                         "start volume, and compares the volumes bit for bit (cull_bit_identical)")
    ap.add_argument("--pipeline", action="store_true",
                    help="N = 1: pipeline the frames over TWO streams (round 5): the part of the warped integrate that does not touch the volume "
                         "-- set_transforms, compute_dists, dists pyramid, verdict pass, plan: dfusion_integrate_warped_prepare -- goes on a second "
                         "stream, where it runs beside the PREVIOUS frame's sweep and ray-cast; the sweep (dfusion_integrate_warped_sweep) and the ray-cast stay "
                         "on the first.  Same results, bit for bit (tests/test_gpu_parity.py).  Measured, same box, interleaved: +1 to +2.4 %% frames/s -- the "
                         "prepare half's small kernels cost the sweep they run beside 30-80 us of workgroup slots, nearly what hiding them gains -- so it is "
                         "NOT the default (profiles/r05_ab_pipeline.txt)")
    ap.add_argument("--no-pipeline", action="store_true", help="(the default; kept for A/B command lines)")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="A/B switch: DF_WARP_NO_PREFETCH on every warped integrate (no look-ahead table / model builds on the handle's side stream)")
    return ap.parse_args(argv)


def config_of(args):
    from dynamicfusion_amd import synth
    base = synth.CONFIGS[args.config]
    if not args.nodes or args.nodes == base.nodes:
        return base
    return synth.Config(base.dims[0], base.size, cols=base.cols, rows=base.rows, nodes=args.nodes, k=base.k,
                        name="%s, but %d warp nodes" % (base.name, args.nodes))


def other_configs(args):
    """BASELINE configs 2 and 5 and the headline geometry at the reference's node density, measured by this same file in child processes
    (own HIP context, own tables) AFTER the headline's timed region: driver-run evidence for the configs the headline is not (VERDICT r5 #3 ii).
    Each entry: frames/s, the integrate's HIP-event time, its roofline fraction, swept / updated voxels, coded blocks, and whether the
    cull-off replay of its timed frames was bit-identical."""
    import subprocess
    free_gb = torch.cuda.mem_get_info()[0] / 1e9
    runs = [("256", ["--config", "256", "--steps", "20", "--warmup", "5"]),
            ("512_nodes8000", ["--config", "512", "--nodes", "8000", "--steps", "20", "--warmup", "5"]),
            ("1024", ["--config", "1024", "--steps", "10", "--warmup", "2"])]
    res = {}
    for tag, a in runs:
        if tag == "1024" and free_gb < 70.0:
            res[tag] = {"skipped": "%.0f GB of HBM free, 70 needed (4 GiB volume + 51 GiB of per-voxel tables)" % free_gb}
            continue
        t0 = time.time()
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__)] + a + ["--no-extras", "--no-cpu-baseline", "--long-frames", "0", "--no-other-configs"],
                               capture_output=True, text=True, timeout=600)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
            if p.returncode != 0 or not line:
                res[tag] = {"error": "rc %d: %s" % (p.returncode, p.stderr.strip()[-300:])}
                continue
            d = json.loads(line[-1])
            fs, rf = d.get("frame_stats", {}), d.get("roofline", {})
            res[tag] = {"workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"], "steps": d["steps"], "ms_per_step": d["ms_per_step"],
                        "integrate_warped_ms": d["kernel_ms"]["integrate_warped"], "raycast_ms": d["kernel_ms"]["raycast(+merge)"],
                        "kernel": rf.get("kernel"), "frac": rf.get("frac"), "achieved_GBps": rf.get("achieved"),
                        "swept_over_updated": rf.get("swept_over_updated"), "kept_blocks": fs.get("last_pose_kept_blocks"),
                        "coded_blocks": fs.get("last_pose_coded_blocks"), "verify_cull": d.get("verify_cull"), "wall_s": time.time() - t0}
        except Exception as e:              # (an extra never loses the headline's line)
            res[tag] = {"error": repr(e)[:300]}
    return res


def self_launch(args):
    """`python bench.py --gpus N` without a rank environment: become the launcher.  Re-executes this file under torch.distributed.run
    (one process per GPU, rendezvous on 127.0.0.1 at a free port), relays everything the ranks print except rank 0's JSON line to
    stderr, prints that line LAST on stdout and returns the ranks' exit status (non-zero if any rank failed or no line came)."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, DFUSION_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(args.gpus, 1))))
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    out, _ = p.communicate()
    line = None
    for ln in out.splitlines():
        if ln.startswith('{"metric"'):
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)
    sys.stderr.flush()
    if line is not None:
        print(line, flush=True)
    if p.returncode != 0:
        return p.returncode
    return 0 if line is not None else 1


def dry_run(args, rank, world):
    """No GPU on this box: run what can be run without one -- the launcher, the rendezvous, the slab partition and EVERY collective of
    the sharded frame with the frame's shapes, dtypes and ops (over gloo, host tensors) -- and say so.  No kernel runs and nothing is
    measured: value is null.  (The HIP path has no CPU fallback; this exists so that `bench.py --gpus N` is exercised end to end by
    the CPU test suite.)"""
    import torch
    import torch.distributed as dist
    from dynamicfusion_amd import sharded, synth
    cfg = synth.CONFIGS[args.config]
    if world > 1 or os.environ.get("DFUSION_BENCH_FORCE_DIST") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    X, Y, Z = cfg.dims
    vs_z = cfg.size / Z
    halo = sharded.halo_planes(max(cfg.trunc_dist, 2.1 * vs_z), cfg.raycast_step_factor, cfg.gradient_delta_factor, vs_z)
    bt = torch.zeros(world + 1, dtype=torch.int64)
    if rank == 0:
        wts = None
        if args.slabs != "uniform" and world > 1:
            wts = sharded.frustum_plane_weights(cfg.dims, cfg.size, cfg.volume_pose, synth.camera_pose(cfg, 0), cfg.intr, cfg.cols, cfg.rows,
                                                depth_mm=synth.depth_frame(cfg, 0), trunc=max(cfg.trunc_dist, 2.1 * vs_z), margin=0.3)
        bt.copy_(torch.tensor(sharded.slab_bounds(Z, world, halo, wts), dtype=torch.int64))
    if dist.is_initialized():
        dist.broadcast(bt, 0)
    bounds = [int(v) for v in bt]
    sharded.validate_bounds(bounds, Z, halo)
    bundle = torch.zeros(cfg.rows * cfg.cols * 2 + cfg.nodes * 32, dtype=torch.uint8)
    keys = torch.full((cfg.rows, cfg.cols), sharded.KEY_NONE, dtype=torch.int64)
    nrm = torch.zeros((cfg.rows, cfg.cols, 4), dtype=torch.float32)
    alive = torch.zeros(Z // 8 if Z % 8 == 0 else Z, dtype=torch.int64)
    t0 = time.perf_counter()
    for _ in range(args.warmup + args.steps):
        if dist.is_initialized():
            dist.broadcast(bundle, 0)
            dist.all_reduce(keys, op=dist.ReduceOp.MIN)
            dist.reduce(nrm.view(torch.int32), dst=0, op=dist.ReduceOp.SUM)
            per, _ = sharded.row_bands(cfg.rows, world)           # the row-banded forms of the second collective (--merge rows / a2a)
            pad = torch.zeros((world * per, cfg.cols, 4), dtype=torch.float32)
            sharded.coll_reduce_scatter_rows(pad.view(torch.int32), torch.empty((per, cfg.cols, 4), dtype=torch.int32))
            sharded.coll_all_to_all_rows(pad.view(torch.int32), torch.empty((world, per, cfg.cols, 4), dtype=torch.int32))
            sharded.coll_broadcast_direct(bundle, 0)              # round 6: the direct forms of the first two collectives
            sharded.coll_all_reduce_min_direct(torch.full((world * per, cfg.cols), sharded.KEY_NONE, dtype=torch.int64),
                                               torch.empty((world, per, cfg.cols), dtype=torch.int64), torch.empty((per, cfg.cols), dtype=torch.int64))
    if dist.is_initialized():
        dist.all_reduce(alive, op=dist.ReduceOp.SUM)       # the re-balance's per-plane alive counts
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "frames/sec integrate+raycast, %dx%d->%d^3 TSDF" % (cfg.cols, cfg.rows, cfg.dims[0]), "value": None,
                          "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
                          "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "data": "synthetic",
                          "dry_run": "no GPU visible: launcher, rendezvous, slab partition and the frame's collectives (gloo, host tensors) only; "
                                     "no kernel ran, nothing was measured",
                          "config": {"workload": cfg.name, "parallelism": "zslab%d" % world, "halo_planes": halo, "slab_bounds": bounds}}), flush=True)
    return 0


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

    def sized_band(mode, budget_s, min_planes=4):
        planes, (t, used) = 4, timed_band(4, mode)
        for _ in range(2):                                  # grow the band until it is about budget_s of work (or the whole volume)
            if (t >= 0.5 * budget_s and planes >= min_planes) or planes >= Z:
                break
            planes = int(min(Z, max(planes + 4, min_planes, 4 * round(planes * budget_s / max(t, 1e-3) / 4))))
            t, used = timed_band(planes, mode)
        return planes, t, used

    kind = "reference" if have_ref else "port"
    planes, t_band, cores = sized_band(kind, target_s, min_planes=64 if Z >= 1024 else 4)      # (1024^3: at least 64 planes through the reference classes)
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


def cxx_headline_ms(cfg, prime, warmup, steps, depths_np, cam_poses, pos, sigma):
    """The headline frame through the C++ mirror (the north star's host is C++): dynamicfusion_amd/host/headless_frame `bench`
    runs the same poses, depth images and node transforms -- prime + warmup untimed, `steps` timed, everything resident on the device before
    its clock starts -- through kfusion::WarpField::setTransformsDevice + cuda::computeDists + cuda::TsdfVolume::integrateAsync(..., warp) +
    raycast(Points), wall clock between two device synchronises."""
    import re
    import subprocess
    import tempfile
    from dynamicfusion_amd import build
    build.build_host()
    n = prime + warmup + steps
    with tempfile.TemporaryDirectory() as td:
        fin = os.path.join(td, "in.bin")
        with open(fin, "wb") as f:
            f.write(synth.aff12(cfg.volume_pose).tobytes())
            f.write(np.asarray(cfg.intr, np.float32).tobytes())
            for i in range(n):
                f.write(np.ascontiguousarray(depths_np[i], np.uint16).tobytes())
                f.write(synth.aff12(cam_poses[i]).tobytes())
            f.write(np.ascontiguousarray(pos, np.float32).tobytes())
            for i in range(n):
                f.write(np.ascontiguousarray(synth.node_transforms(cfg, i), np.float32).tobytes())
            f.write(np.ascontiguousarray(sigma, np.float32).tobytes())
        r = subprocess.run([build.HOST_APP, "bench", str(cfg.dims[0]), str(cfg.size), str(cfg.cols), str(cfg.rows), str(n), str(cfg.nodes), str(cfg.k),
                            str(prime), str(warmup), fin], capture_output=True, text=True, timeout=600)
    m = re.search(r"cxx_host_ms_per_frame ([0-9.]+) over (\d+) frames", r.stdout)
    if r.returncode != 0 or not m:
        raise RuntimeError((r.stdout + r.stderr)[-300:])
    return {"cxx_host_ms_per_frame": float(m.group(1)), "frames": int(m.group(2)),
            "what": "the same %d timed poses through the C++ mirror (libkfusion_hip.so over the C-ABI): WarpField::setTransformsDevice + computeDists + "
                    "TsdfVolume::integrateAsync(dists, pose, intr, warp) + raycast(Points), wall clock; compare with ms_per_step (Python mirror + ctypes)" % int(m.group(2))}


def _imports():
    global torch, dist, Intr, TsdfVolume, WarpField, capi, compute_dists, sharded, synth, upload_u16
    import torch
    import torch.distributed as dist
    from dynamicfusion_amd import Intr, TsdfVolume, WarpField, capi, compute_dists, sharded, synth, upload_u16


def main():
    args = parse()
    force_dist = os.environ.get("DFUSION_BENCH_FORCE_DIST") == "1"
    if (args.gpus > 1 or force_dist) and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    _imports()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        # never die on this: the rank environment decides (the driver's 8-GPU run must produce a line whichever way it launches)
        print("bench.py: --gpus %d but WORLD_SIZE=%d: running %d rank(s)" % (args.gpus, world, world), file=sys.stderr)
    if os.environ.get("DFUSION_BENCH_TEST_FAIL_RANK") == str(rank):          # test hook: the launcher must report a failing rank
        raise RuntimeError("DFUSION_BENCH_TEST_FAIL_RANK: rank %d fails on purpose" % rank)
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n_dev == 0:
        return dry_run(args, rank, world)
    # DFUSION_BENCH_FORCE_DIST=1: take the N > 1 code path (RCCL process group, slab volume, broadcast, the merge collectives) with
    # whatever WORLD_SIZE is -- with one rank it is an RCCL dry run of every collective, dtype and op of the sharded frame
    # (tests/test_gpu_sharded.py runs it), since an 8-GPU node is only ever seen by the driver
    dist_on = world > 1 or force_dist
    # fewer GPUs than ranks (a one-GPU box asked for N ranks): RCCL refuses two ranks on one device, so the ranks share the devices
    # round-robin and the collectives run over gloo with host staging -- the same sharded code, as a smoke of the path, not a timing
    oversub = dist_on and world > n_dev
    if dist_on:
        torch.cuda.set_device(local_rank % n_dev)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
        if oversub:
            sharded.set_host_staging(True)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    cfg = config_of(args)
    intr = Intr(*cfg.intr)
    X, Y, Z = cfg.dims
    # ---- the frame sequence: N_PRIME priming + warm-up + timed + (so that the percentiles have >= MIN_STAT_FRAMES samples) extra frames,
    # all consecutive poses of one sweep unless --frames F asks for F cycled poses
    n_extra = max(0, MIN_STAT_FRAMES - args.steps)
    n_seq = N_PRIME + args.warmup + args.steps + n_extra
    monotone = args.frames <= 0
    n_long = max(0, args.long_frames - args.steps - n_extra) if (monotone and args.config != "1024") else 0     # frames of the sweep past the percentile window
    F = args.frames if args.frames > 0 else n_seq + n_long

    # ---- synthetic inputs, resident in HBM before the timed region (rank 0 owns the sensor and the solver output)
    cam_poses = [synth.camera_pose(cfg, f) for f in range(F)]
    pos, sigma = synth.make_nodes(cfg)
    dqs_np0 = synth.node_transforms(cfg, 0)
    depth0_np = synth.depth_frame(cfg, 0)
    if rank == 0 or not dist_on:
        # (the synthetic sensor is numpy on one core, 0.4 s a frame: the sweep's frames are made on a thread pool)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=max(1, min(32, (os.cpu_count() or 8) // max(world, 1)))) as pool:
            depths_np = [depth0_np] + list(pool.map(lambda f: synth.depth_frame(cfg, f), range(1, F)))
        depths = [upload_u16(d, dev) for d in depths_np]
        dqs = [torch.from_numpy(synth.node_transforms(cfg, f)).to(dev) for f in range(F)]
    else:
        depths, dqs = None, None

    vs_z = cfg.size / Z
    trunc_eff = max(cfg.trunc_dist, 2.1 * vs_z)
    halo = sharded.halo_planes(trunc_eff, cfg.raycast_step_factor, cfg.gradient_delta_factor, vs_z)
    slab_bounds = None

    def make_volume(bounds):
        if dist_on:
            v = TsdfVolume(cfg.dims, device=dev, slab=(bounds[rank], bounds[rank + 1] - bounds[rank], halo))
        else:
            v = TsdfVolume(cfg.dims, device=dev)
        v.setSize([cfg.size] * 3); v.setTruncDist(cfg.trunc_dist); v.setMaxWeight(cfg.max_weight)
        v.setPose(cfg.volume_pose)
        v.setRaycastStepFactor(cfg.raycast_step_factor); v.setGradientDeltaFactor(cfg.gradient_delta_factor)
        v.clear()
        # N > 1: every rank also integrates its halo planes (a pure function of the broadcast inputs: bit-identical with the
        # neighbour's planes), so the frame has NO halo collective; `vi` is the same blob seen as owner of all its stored planes.
        vi = v.owning_stored_planes() if (dist_on and args.halo in ("recompute", "both")) else v
        return v, vi

    def share_bounds(b):
        bt = torch.zeros(world + 1, dtype=torch.int64, device=dev)
        if rank == 0:
            bt.copy_(torch.tensor(b, dtype=torch.int64))
        sharded.coll_broadcast(bt, 0)
        b = [int(v) for v in bt.cpu()]
        sharded.validate_bounds(b, Z, halo)                 # same verdict on every rank, before the first data collective
        return b

    if dist_on:
        # the partition is decided on rank 0 (which owns the sensor) and broadcast: every rank must hold the SAME boundaries before
        # the first slab is allocated
        b0 = None
        if rank == 0:
            wts = None
            if args.slabs != "uniform" and world > 1:
                wts = sharded.frustum_plane_weights(cfg.dims, cfg.size, cfg.volume_pose, cam_poses[0], cfg.intr, cfg.cols, cfg.rows,
                                                    depth_mm=depth0_np, trunc=trunc_eff, margin=0.3)
            b0 = sharded.slab_bounds(Z, world, halo, wts)
        slab_bounds = share_bounds(b0)
    vol, vol_int = make_volume(slab_bounds)

    wf = WarpField(k=cfg.k, device=dev)
    wf.init(pos, sigma=sigma, transforms=dqs_np0)
    t0_index = time.time()
    wf.ensure_index(vol_int, cfg.k)
    torch.cuda.synchronize()

    dists = torch.empty((cfg.rows, cfg.cols), dtype=torch.int16, device=dev)
    dists2 = [dists, torch.empty_like(dists)]
    # the key image lives in the first rows of a buffer of world * per rows (whole row bands for the direct key merge; the rows past the image
    # stay KEY_NONE)
    keys_pad = torch.full((world * sharded.row_bands(cfg.rows, world)[0], cfg.cols), sharded.KEY_NONE, dtype=torch.int64, device=dev) if dist_on else None
    keys = keys_pad[:cfg.rows] if dist_on else None
    out2 = torch.empty((2, cfg.rows, cfg.cols, 4), dtype=torch.float32, device=dev)
    pts, nrm = out2[0], out2[1]
    rows_merge = dist_on          # (the band buffers exist whenever the frame is sharded: the variant passes use them too)
    if rows_merge:       # the normals buffer the reduce_scatter splits: world * per rows (rows past the image stay zero), + this rank's band
        per_rows, _ = sharded.row_bands(cfg.rows, world)
        nrm_pad = torch.zeros((world * per_rows, cfg.cols, 4), dtype=torch.float32, device=dev)
        nrm_band = torch.empty((per_rows, cfg.cols, 4), dtype=torch.float32, device=dev)
        pts_band = torch.empty((per_rows, cfg.cols, 4), dtype=torch.float32, device=dev)
        a2a_recv = torch.empty((world, per_rows, cfg.cols, 4), dtype=torch.float32, device=dev)     # the direct merge's N pieces of this rank's band
        keys_recv = torch.empty((world, per_rows, cfg.cols), dtype=torch.int64, device=dev)         # the direct key merge's N pieces of this rank's band
        keys_band = torch.empty((per_rows, cfg.cols), dtype=torch.int64, device=dev)
    # frame inputs travel as ONE byte bundle (depth image + node transforms): one ncclBroadcast per frame
    n_depth = cfg.rows * cfg.cols * 2
    bundle = torch.empty(n_depth + cfg.nodes * 32, dtype=torch.uint8, device=dev)
    bcast_scratch = torch.empty(world * bundle.numel(), dtype=torch.uint8, device=dev) if (dist_on and rank == 0) else None     # the N copies rank 0's direct send reads
    depth_in = bundle[:n_depth].view(torch.int16).view(cfg.rows, cfg.cols)
    dq_in = bundle[n_depth:].view(torch.float32).view(cfg.nodes, 8)

    halo_main = "recompute" if args.halo == "both" else args.halo
    # the direct forms of the first two collectives, tried ONCE before anything is timed: a torch / RCCL build that refuses one of the calls
    # (the same exception on every rank) sends the run back to the ring forms instead of ending it; config says which ran and why
    direct_fallback = None
    if dist_on and (args.key_merge == "direct" or args.bcast == "direct"):
        try:
            if args.bcast == "direct": sharded.coll_broadcast_direct(bundle, 0, scratch=bcast_scratch)
            if args.key_merge == "direct": sharded.coll_all_reduce_min_direct(keys_pad, keys_recv, keys_band)
            torch.cuda.synchronize()
        except (RuntimeError, NotImplementedError, ValueError) as e:
            direct_fallback = str(e)[:300]
            args.key_merge = "ring"; args.bcast = "ring"
        keys_pad.fill_(sharded.KEY_NONE)
    # frames pipelined across two streams (N = 1): see --no-pipeline
    pipeline = (not dist_on) and args.pipeline and (not args.no_pipeline) and cfg.k in (4, 8)
    s_main = torch.cuda.current_stream()
    s_prep = torch.cuda.Stream(device=dev) if pipeline else None
    # the prepare half of frame n may run BESIDE the sweep of frame n - 1 (the library double-buffers what that sweep reads of the handle);
    # what the caller owns of the sweep's inputs -- the dists image -- is double-buffered here: frame n uses dists2[n & 1], written once the
    # sweep of frame n - 2 is done (ev_sweep[n & 1])
    ev_sweep = [torch.cuda.Event(), torch.cuda.Event()] if pipeline else None
    ev_prep = torch.cuda.Event() if pipeline else None       # this frame's prepare half is done
    pipe_n = [0]
    if pipeline:
        s_prep.wait_stream(s_main)
        ev_sweep[0].record(s_main); ev_sweep[1].record(s_main)

    def step(i, ev=None, timer=None, merge=None, halo_mode=None, key_merge=None, bcast=None):
        """one frame.  timer: a sharded.StageTimer (the stages of this frame are marked); merge / halo_mode: a variant of the frame's
        second collective / of how the halo planes are made (default: the command line's)."""
        f = i % F
        merge = args.merge if merge is None else merge
        key_merge = args.key_merge if key_merge is None else key_merge
        bcast = args.bcast if bcast is None else bcast
        km = dict(key_merge=key_merge, keys_pad=keys_pad, keys_recv=keys_recv, keys_band=keys_band) if dist_on else {}
        halo_mode = halo_main if halo_mode is None else halo_mode
        mark = timer.mark if timer is not None else (lambda name: None)
        if timer is not None: timer.start()
        if dist_on:                                    # rank 0 owns the sensor frame and the solver output
            if rank == 0:
                depth_in.copy_(depths[f]); dq_in.copy_(dqs[f])
            if bcast == "direct": sharded.coll_broadcast_direct(bundle, 0, scratch=bcast_scratch)
            else: sharded.coll_broadcast(bundle, 0)
            mark("broadcast")
            d_in, q_in = depth_in, dq_in
        else:
            d_in, q_in = depths[f], dqs[f]
        if pipeline:
            # the half of the frame that does not touch the volume, on the second stream: it runs beside the PREVIOUS frame's sweep and
            # ray-cast (what that sweep still reads of the handle -- node arrays, launch plan -- is double-buffered inside the library)
            # (the waits are made HERE, before the timing events, so that [3] -> [4] is the prepare half's own run time and [0] -> [1] the sweep
            # kernel's; the library orders the same things itself -- dfusion.h -- and finds nothing left to wait for)
            n = pipe_n[0]; pipe_n[0] += 1
            dn = dists2[n & 1]
            with torch.cuda.stream(s_prep):
                s_prep.wait_event(ev_sweep[n & 1])
                if ev is not None: ev[3].record()
                wf.set_transforms(q_in)
                compute_dists(d_in, intr, dn)
                vol.integrate_warped_prepare(dn, cam_poses[f], intr, wf, prefetch=not args.no_prefetch)
                if ev is not None: ev[4].record()
                ev_prep.record()
            s_main.wait_event(ev_prep)
            if ev is not None: ev[0].record()
            vol.integrate_warped_sweep(wf)
            mark("integrate_sweep")
            if ev is not None: ev[1].record()
            ev_sweep[n & 1].record(s_main)
        else:
            wf.set_transforms(q_in)
            compute_dists(d_in, intr, dists)
            mark("set_transforms+compute_dists")
            if ev is not None: ev[0].record()
            # (halo recompute: the integrate owns every stored plane; halo exchange: its own planes, the halos come from the neighbours)
            (vol_int if halo_mode == "recompute" else vol).integrate_warped(dists, cam_poses[f], intr, wf, sync=False, prefetch=not args.no_prefetch)
            mark("integrate_warped")
            if ev is not None: ev[1].record()
        if dist_on and halo_mode == "exchange":
            sharded.exchange_halos(vol.data(), vol.z_store0, vol.z_own0, vol.z_own_n, Z, halo, rank, world)
            mark("halo_exchange")
        if dist_on and merge in ("rows", "a2a"):
            def shade_padded(mk):
                vol.raycast_shade(cam_poses[f], intr, mk, None, nrm_pad[:cfg.rows])
                return nrm_pad
            out = sharded.raycast_sharded(lambda: vol.raycast_march(cam_poses[f], intr, keys, rank), shade_padded,
                                          lambda mk, nb, r0, nr: vol.raycast_points_of_keys(cam_poses[f], intr, mk, nb, pts_band[:nr], r0, nr),
                                          rank, world, collectives=True, merge=merge, band_out=nrm_band, a2a_recv=a2a_recv, timer=timer, **km)
        elif dist_on:
            out = sharded.raycast_sharded(lambda: vol.raycast_march(cam_poses[f], intr, keys, rank),
                                          lambda mk: vol.raycast_shade(cam_poses[f], intr, mk, None, nrm)[1],
                                          lambda mk, n: vol.raycast_points_of_keys(cam_poses[f], intr, mk, n, pts),
                                          rank, world, collectives=True, timer=timer, **km)
        else:
            vol.raycast(cam_poses[f], intr, pts, nrm)
            mark("raycast")
            out = (pts, nrm)
        if ev is not None: ev[2].record()
        if timer is not None: timer.end()
        return out

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def events(n):          # per frame: [0] before the integrate (sweep), [1] after it, [2] after the ray-cast; pipelined: [3], [4] around the prepare half on its stream
        return [tuple(torch.cuda.Event(enable_timing=True) for _ in range(5 if pipeline else 3)) for _ in range(n)]

    def int_ms(e):          # the integrate of one frame: prepare (set_transforms + compute_dists + pyramid + verdict + plan, on its stream) + sweep
        return e[0].elapsed_time(e[1]) + (e[3].elapsed_time(e[4]) if pipeline else 0.0)

    # The per-node-set work: with tables on demand (the mirrors' default) `ensure_index` above only builds the brick index; the tables,
    # then the blend models, of the blocks the launch plans find alive are made by the first sweeps.  N_PRIME untimed frames pay for
    # the FIRST alive set here, whatever --warmup is; what the moving camera brings in afterwards is paid inside the warm-up and
    # timed frames.  The volume is cleared again.  `index_build_once_s` covers index + these frames.
    for i in range(N_PRIME):
        step(i)
    rebalance = None
    if dist_on and world > 1 and args.slabs == "measured":
        # ---- re-cut the slabs ONCE from what the verdict pass found alive: the a-priori frustum weights do not
        # describe a sweep the block verdicts have thinned.  Every rank counts the alive 8x8x8 blocks of its OWN layers, one
        # all_reduce(SUM) makes the global per-layer profile, every rank computes the same new boundaries, slabs and index are re-made.
        layers = torch.zeros(Z // 8, dtype=torch.int64, device=dev)
        wf.alive_blocks_per_layer(vol, layers)
        sharded.coll_all_reduce(layers, dist.ReduceOp.SUM)
        w_layer = layers.cpu().numpy().astype(np.float64)
        # (round 6: the boundaries minimise the LARGEST rank's cost, halo planes it integrates itself included -- sharded.slab_bounds_minmax)
        w_planes = sharded.layer_weights_to_planes(w_layer, Z)
        count_halo = halo_main == "recompute"
        new_bounds = sharded.slab_bounds_minmax(Z, world, halo, w_planes, count_halo=count_halo)
        sharded.validate_bounds(new_bounds, Z, halo)
        rebalance = {"from": slab_bounds, "to": new_bounds, "alive_blocks_per_8_planes": [int(v) for v in w_layer]}

        def recut(b):
            nonlocal slab_bounds, vol, vol_int
            slab_bounds = b
            del vol, vol_int
            vol, vol_int = make_volume(slab_bounds)
            wf.ensure_index(vol_int, cfg.k)
            for i in range(N_PRIME):
                step(i)

        if new_bounds != slab_bounds:
            recut(new_bounds)
        # ---- ... and ONCE MORE from what the ranks measure (round 6, VERDICT r5 #7 i): four frames, HIP events around every rank's
        # integrate, one all_reduce(SUM) of a vector in which each rank fills its own slot; the planes a rank owns are re-weighted by
        # (its time - the launch-sized part) / (the weight it swept) and the boundaries made again (sharded.reweight_from_times)
        ev_b = events(4)
        for i in range(4):
            step(N_PRIME + i, ev_b[i])
        torch.cuda.synchronize()
        tvec = torch.zeros(world, dtype=torch.float64, device=dev)
        tvec[rank] = float(np.mean([int_ms(e) for e in ev_b]))
        sharded.coll_all_reduce(tvec, dist.ReduceOp.SUM)
        times_ms = [float(v) for v in tvec.cpu()]
        FIXED_MS = 0.045                                      # verdict pass + plan + pyramid + launches of a slab's integrate (profiles/r06_frame_trace.txt)
        w2 = sharded.reweight_from_times(slab_bounds, w_planes, halo if count_halo else 0, times_ms, FIXED_MS)
        bounds2 = sharded.slab_bounds_minmax(Z, world, halo, w2, count_halo=count_halo)
        sharded.validate_bounds(bounds2, Z, halo)
        rebalance["integrate_ms_per_rank_after_first_recut"] = times_ms
        rebalance["second_recut_to"] = bounds2
        if bounds2 != slab_bounds and max(times_ms) > 1.08 * float(np.mean(times_ms)):
            recut(bounds2)
            rebalance["to"] = bounds2
    vol.clear()
    barrier()
    t_index = time.time() - t0_index
    i0 = N_PRIME
    for i in range(args.warmup):
        step(i0 + i)
    verify = not args.no_verify_cull
    vol_start = vol.data().clone() if verify else None
    i0 += args.warmup
    ev = events(args.steps)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i0 + i, ev[i])
    barrier()
    elapsed = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        sharded.coll_all_reduce(t, dist.ReduceOp.MAX)
        elapsed = float(t.item())
    vol_end = vol.data().clone() if verify else None
    timed_frames = [(i0 + i) % F for i in range(args.steps)]

    # ---- the sweep goes on (untimed by the wall clock, HIP events per frame) until the percentiles have MIN_STAT_FRAMES samples; these
    # frames also carry the per-stage events (sharded.StageTimer: every collective and kernel stage of the frame, per rank) -- not the
    # timed ones, whose wall clock they would stretch by a dozen event records per frame
    ev_more = events(n_extra)
    stage_timer = sharded.StageTimer()
    for i in range(n_extra):
        step(i0 + args.steps + i, ev_more[i], timer=stage_timer)
    # ---- ... and on, to args.long_frames consecutive frames of the sweep (frame_stats.long_sweep)
    ev_long = events(n_long)
    for i in range(n_long):
        step(i0 + args.steps + n_extra + i, ev_long[i])
    torch.cuda.synchronize()
    ms_int = float(np.mean([int_ms(e) for e in ev]))
    ms_ray = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
    ms_prep = float(np.mean([e[3].elapsed_time(e[4]) for e in ev])) if pipeline else None
    # per-frame spread (integrate + ray-cast between HIP events on the launch stream) and the reference-shaped frame
    # (KinFu::dynamicfusion integrates once and ray-casts twice)
    per_frame = np.array([int_ms(e) + e[1].elapsed_time(e[2]) for e in ev + ev_more], np.float64)
    per_int = np.array([int_ms(e) for e in ev + ev_more], np.float64)
    frame_stats = {"integrate+raycast_ms": {"p10": float(np.percentile(per_frame, 10)), "median": float(np.median(per_frame)),
                                            "p90": float(np.percentile(per_frame, 90)), "max": float(per_frame.max()), "frames": int(per_frame.size)},
                   "integrate_warped_ms": {"p10": float(np.percentile(per_int, 10)), "median": float(np.median(per_int)),
                                           "p90": float(np.percentile(per_int, 90)), "max": float(per_int.max())},
                   "trajectory": ("monotone sweep, 0.25 deg per frame: %d priming + %d warm-up + %d timed + %d more frames, poses %d..%d timed"
                                  % (N_PRIME, args.warmup, args.steps, n_extra, i0, i0 + args.steps - 1)) if monotone else "%d poses cycled" % F,
                   "reference_shaped_ms": ms_int + 2.0 * ms_ray}
    if n_long:
        lf = np.array([int_ms(e) + e[1].elapsed_time(e[2]) for e in ev + ev_more + ev_long], np.float64)
        frame_stats["long_sweep"] = {"frames": int(lf.size), "integrate+raycast_ms_mean": float(lf.mean()), "p10": float(np.percentile(lf, 10)),
                                     "median": float(np.median(lf)), "p90": float(np.percentile(lf, 90)),
                                     "what": "HIP-event integrate + ray-cast time of %d consecutive frames of the same sweep starting at the first timed pose "
                                             "(the headline times the first %d of them; later poses look past the scene and sweep less)" % (lf.size, args.steps)}
    if monotone:
        # round 3's headline for comparison: the last four poses cycled -- every table and block model they need exists
        loop = [n_seq - 4 + j for j in range(4)]
        for j in range(4):
            step(loop[j])
        barrier()
        t1 = time.perf_counter()
        for j in range(20):
            step(loop[j % 4])
        barrier()
        frame_stats["steady_state_loop_ms"] = 1e3 * (time.perf_counter() - t1) / 20

    # ---- algorithmic bytes of one integrate launch: 8*N_upd + 2*W*H + 48*M, over the TIMED frames' poses.
    # N_upd counted by the kernel itself (parity-checked against the oracle's count in tests/), untimed pass; whether a voxel updates
    # depends on the frame's geometry and depth only, not on what the volume holds.
    n_upd = torch.zeros(2, dtype=torch.int64, device=dev)          # [0] updated, [1] swept (the plan's alive cells, dfusion_warp_debug_counters)

    def replay(cull, counters):
        """the timed frames' integrates again (inputs as rank 0 broadcast them: every rank keeps what it needs to replay locally)"""
        for f in timed_frames:
            if dist_on:
                if rank == 0:
                    depth_in.copy_(depths[f]); dq_in.copy_(dqs[f])
                sharded.coll_broadcast(bundle, 0)
                d_in, q_in = depth_in, dq_in
            else:
                d_in, q_in = depths[f], dqs[f]
            wf.set_transforms(q_in)
            vol_int.integrate_warped(compute_dists(d_in, intr, dists), cam_poses[f], intr, wf, n_updated=counters, cull=cull, sync=False,
                                     prefetch=not args.no_prefetch)
        torch.cuda.synchronize()

    wf.debug_counters(n_upd[1:])
    replay(True, n_upd[:1])
    wf.debug_counters(None)
    n_upd_launch = float(n_upd[0].item()) / len(timed_frames)
    n_swept_launch = float(n_upd[1].item()) / len(timed_frames)
    if dist_on:
        t = torch.tensor([n_upd_launch], dtype=torch.float64, device=dev)
        sharded.coll_all_reduce(t, dist.ReduceOp.SUM)
        n_upd_total = float(t.item())
    else:
        n_upd_total = n_upd_launch
    alg_bytes = 8.0 * n_upd_launch + 2.0 * cfg.cols * cfg.rows + 48.0 * cfg.nodes
    achieved = alg_bytes / (ms_int * 1e-3) / 1e9
    frame_stats["timed_poses_mean_swept_voxels"] = n_swept_launch          # (this rank's planes; what roofline.traffic's profile is compared with)
    frame_stats["timed_poses_mean_updated_voxels"] = n_upd_launch
    try:        # how many of the blocks the last replayed pose kept were read as 4-bit neighbour codes (dfusion_warp_coded_blocks; this rank's planes)
        if Z % 8 == 0:
            kept_l = torch.zeros(Z // 8, dtype=torch.int64, device=dev); coded_l = torch.zeros_like(kept_l)
            wf.alive_blocks_per_layer(vol, kept_l); wf.coded_blocks_per_layer(vol, coded_l)
            kept_n, coded_n = int(kept_l.sum().item()), int(coded_l.sum().item())
            frame_stats["last_pose_kept_blocks"] = kept_n
            frame_stats["last_pose_coded_blocks"] = coded_n
    except Exception as e:      # (a measurement extra never fails the bench line)
        frame_stats["last_pose_coded_blocks_error"] = str(e)[:120]

    # ---- per rank, per stage: HIP-event means over the n_extra frames after the timed region, gathered on rank 0; and,
    # N > 1, the OTHER forms of the frame's collectives timed back to back in the same launch (wall clock between barriers + stages)
    def gather(obj):
        if not dist_on:
            return [obj]
        lst = [None] * world
        dist.all_gather_object(lst, obj)
        return lst

    scaling_detail = None
    if n_extra > 0:
        per_rank = gather(stage_timer.means())
        stages = list(per_rank[0].keys())
        scaling_detail = {"stages": stages, "per_rank_ms": {k: [float(pr.get(k, float("nan"))) for pr in per_rank] for k in stages},
                          "frames": n_extra, "how": "HIP events on each rank's launch stream around every stage, mean over the %d frames that follow the "
                                                     "timed region (the same sweep; not the timed frames, which carry 3 events each)" % n_extra}
    if pipeline and scaling_detail is not None and n_extra > 0:
        scaling_detail["per_rank_ms"]["prepare (second stream: set_transforms + compute_dists + pyramid + verdict pass + plan; beside the previous frame's ray-cast)"] = \
            [float(np.mean([e[3].elapsed_time(e[4]) for e in ev_more]))]
        scaling_detail["pipelined"] = True
    if dist_on and scaling_detail is not None:
        ones = torch.ones(1, dtype=torch.int64, device=dev)
        sharded.coll_all_reduce(ones, dist.ReduceOp.SUM)
        scaling_detail["rccl_ranks_seen"] = int(ones.item())
        scaling_detail["backend"] = dist.get_backend()
        scaling_detail["devices"] = gather("%s:%d" % (torch.cuda.get_device_name(dev), dev.index))
        px = cfg.rows * cfg.cols
        sizes = {"broadcast": px * 2 + cfg.nodes * 32, "all_reduce_min": px * 8, "reduce_scatter": px * 16, "all_to_all": px * 16, "reduce": px * 16,
                 "halo_exchange": halo * X * Y * 4, "broadcast(ring)": px * 2 + cfg.nodes * 32, "all_reduce_min(ring)": px * 8}
        kinds = {"broadcast": "broadcast_direct" if args.bcast == "direct" else "broadcast",
                 "all_reduce_min": "all_reduce_direct" if args.key_merge == "direct" else "all_reduce", "reduce_scatter": "reduce_scatter",
                 "all_to_all": "all_to_all", "reduce": "reduce", "halo_exchange": "halo", "broadcast(ring)": "broadcast", "all_reduce_min(ring)": "all_reduce"}
        scaling_detail["predicted_collective_ms"] = {k: 1e3 * sharded.collective_model_s(kinds[k], sizes[k], world) for k in sizes}
        scaling_detail["predicted_how"] = ("tools/scale_model.py's stated model: %.0f us per call + %.0f us per ring step + bytes on the busiest link / %.0f GB/s "
                                           "(rings: N - 1 steps, all_reduce 2 (N - 1); all_to_all, the halo exchange and the direct forms of broadcast / key merge: one step per exchange)"
                                           % (sharded.T_LAUNCH * 1e6, sharded.T_HOP * 1e6, sharded.LINK_GBPS))
        if world > 1 and not args.no_variants:
            variants = {}
            n_var = max(4, min(10, args.steps))
            base = i0 + args.steps                       # (the poses right after the timed ones, every variant the same ones)
            todo = [("merge=%s" % m, dict(merge=m)) for m in ("rows", "a2a", "root")] + [("halo=exchange", dict(halo_mode="exchange"))]
            todo += [("key_merge=%s" % ("ring" if args.key_merge == "direct" else "direct"), dict(key_merge="ring" if args.key_merge == "direct" else "direct")),
                     ("bcast=%s" % ("ring" if args.bcast == "direct" else "direct"), dict(bcast="ring" if args.bcast == "direct" else "direct"))]
            for name, kw in todo:
                # (a variant that this RCCL / torch build refuses -- the same exception on every rank -- is recorded, not fatal: the
                # headline has been measured by now and its line must still be printed)
                try:
                    tm = sharded.StageTimer()
                    for i in range(2):
                        step(base + i, **kw)
                    barrier()
                    t1 = time.perf_counter()
                    for i in range(n_var):
                        step(base + i, timer=tm, **kw)
                    barrier()
                    dt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
                    sharded.coll_all_reduce(dt, dist.ReduceOp.MAX)
                    pr = gather(tm.means())
                    variants[name] = {"ms_per_frame": 1e3 * float(dt.item()) / n_var, "frames": n_var,
                                      "per_rank_ms": {k: [float(r.get(k, float("nan"))) for r in pr] for k in pr[0].keys()}}
                except (RuntimeError, NotImplementedError, ValueError) as e:
                    variants[name] = {"error": str(e)[:300]}
            scaling_detail["variants"] = variants
            scaling_detail["variants_how"] = ("after the timed region, %d frames each (poses %d..%d, every variant the same; wall clock between barriers, "
                                              "max over ranks, incl. the stage events): the headline's form is merge=%s, halo=%s, key_merge=%s, bcast=%s"
                                              % (n_var, base, base + n_var - 1, args.merge, halo_main, args.key_merge, args.bcast))

    # ---- the launch plan's cull, verified on the timed frames themselves: the same frames from the same start volume with the cull
    # switched off (every voxel goes through the reference's own tests, tsdf_volume.cu:77-93) must leave the same bits
    verify_cull = None
    if verify:
        vol.data().copy_(vol_start)
        n_nc = torch.zeros(1, dtype=torch.int64, device=dev)
        replay(False, n_nc)
        # (the planes this rank's integrate owns: with --halo exchange the halo planes come from the neighbours, not from the replay)
        o0 = vol_int.z_own0 - vol_int.z_store0
        got, want = vol.data()[o0:o0 + vol_int.z_own_n], vol_end[o0:o0 + vol_int.z_own_n]
        same = bool(torch.equal(got, want))
        n_diff = 0 if same else int((got != want).sum().item())
        flag = torch.tensor([1 if same else 0, int(n_nc.item()), int(n_upd[0].item()), n_diff], dtype=torch.int64, device=dev)
        if dist_on:
            mn = flag[:1].clone(); sharded.coll_all_reduce(mn, dist.ReduceOp.MIN)
            sharded.coll_all_reduce(flag, dist.ReduceOp.SUM)
            flag[0] = mn[0]
        flag = [int(v) for v in flag.cpu()]
        verify_cull = {"cull_bit_identical": bool(flag[0]), "frames": len(timed_frames), "voxels_differing": flag[3],
                       "updates_with_cull": flag[2], "updates_without_cull": flag[1],
                       "what": "the timed frames re-integrated from the volume they started on with DF_WARP_NO_CULL (no launch plan: every voxel of the "
                               "slab is warped and tested); volumes compared bit for bit, update counts side by side"}
        vol.data().copy_(vol_end)
        del vol_start, vol_end

    extra = {}
    if not args.no_extras and not dist_on:
        vol2 = TsdfVolume(cfg.dims, device=dev)
        vol2.setSize([cfg.size] * 3); vol2.setTruncDist(cfg.trunc_dist); vol2.setMaxWeight(cfg.max_weight); vol2.setPose(cfg.volume_pose)
        nr = torch.zeros(2, dtype=torch.int64, device=dev)       # [0] updated, [1] swept (dfusion_integrate_ex)
        tf = timed_frames
        for f in tf[:2]:
            vol2.integrate(dists, cam_poses[f], intr, sync=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20):
            vol2.integrate(dists, cam_poses[tf[i % len(tf)]], intr, sync=False)
        e1.record()
        torch.cuda.synchronize()
        for i in range(20):
            vol2.integrate(dists, cam_poses[tf[i % len(tf)]], intr, n_updated=nr[:1], n_swept=nr[1:], sync=False)
        torch.cuda.synchronize()
        ms_r = e0.elapsed_time(e1) / 20
        b_r = 8.0 * float(nr[0].item()) / 20 + 2.0 * cfg.cols * cfg.rows
        extra["rigid_integrate"] = {"kernel": "df_integrate_rigid_kernel<2, true, false> (+ df_rigid_plan_kernel, df_pyramid_tiles_kernel)",
                                    "ms": ms_r, "achieved_GBps": b_r / (ms_r * 1e-3) / 1e9, "frac_of_peak": b_r / (ms_r * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                    "n_updated": float(nr[0].item()) / 20, "n_swept": float(nr[1].item()) / 20,
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
            lean.init(pos, sigma=sigma, transforms=dqs_np0)
            lean.ensure_index(vol, cfg.k)
            vol.integrate_warped(dists, cam_poses[0], intr, lean, sync=False)
            e0.record()
            for i in range(5):
                vol.integrate_warped(dists, cam_poses[tf[i % len(tf)]], intr, lean, sync=False)
            e1.record(); torch.cuda.synchronize()
            extra["lean_path_ms"] = {"integrate_warped_ms": e0.elapsed_time(e1) / 5, "what": "no per-voxel cache (brick candidate lists only, ~80 MB): exact top-k "
                                     "over ~150 candidates per voxel and the weights' exp() every frame (df_warp_brick_kernel)"}
            del lean
        except Exception as e:
            extra["frame_nodes_changed_ms"] = {"error": repr(e)[:200]}
        # the headline frame itself through the C++ mirror
        if not args.no_kinfu and cfg.dims[0] == cfg.dims[1] == cfg.dims[2]:
            try:
                extra["cxx_host"] = cxx_headline_ms(cfg, N_PRIME, args.warmup, args.steps, depths_np, cam_poses, pos, sigma)
                extra["cxx_host"]["python_over_cxx"] = (1e3 * elapsed / args.steps) / extra["cxx_host"]["cxx_host_ms_per_frame"]
            except Exception as e:      # the extras never fail the bench line
                extra["cxx_host"] = {"error": str(e)[:200]}
        # one whole KinFu::operator() frame (front-end, ICP, dynamicfusion, ray-cast) through the C++ mirror, for context
        if not args.no_kinfu:
            try:
                extra["kinfu_frame"] = kinfu_frame_ms(cfg)
            except Exception as e:      # the extras never fail the bench line
                extra["kinfu_frame"] = {"error": str(e)[:200]}

    # kernel that ran + its per-voxel cache footprint; PMC traffic comes from the committed rocprofv3 --pmc passes
    # (dfusion_warp.hip, df_integrate_warped_impl: k = 8 runs the union-copy sweep at every node count, 256 threads, one plane per batch;
    # k = 4 keeps the LDS node table where it fits -- 1024 threads past 80 KiB --, and gathers from the L2 otherwise)
    lds_ok = cfg.nodes * 32 <= 160 * 1024
    axis_aligned = bool(np.array_equal(np.asarray(cfg.volume_pose, np.float32).reshape(4, 4)[:3, :3], np.eye(3, dtype=np.float32)))
    if cfg.k == 8:
        kernel_name = "df_warp_rows_pipe_kernel<8, 1, 256, %s, false>" % ("true" if axis_aligned else "false")
    elif cfg.k == 4:
        wide = lds_ok and cfg.nodes * 32 > 80 * 1024
        kernel_name = "df_warp_rows_pipe_kernel<4, 2, %d, %s, %s>" % (1024 if wide else 512, "true" if axis_aligned else "false", "true" if lds_ok else "false")
    elif lds_ok:
        kernel_name = "df_warp_rows_lds_kernel<%d, true, 2>" % cfg.k
    else:
        kernel_name = "df_warp_rows_kernel<%d, true, 4>" % cfg.k
    table_bytes = int(X) * Y * vol.z_own_n * cfg.k * 6
    traffic, traffic_src, compute = None, None, None
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
                c = ent.get("counters", {})
                if all(k in c for k in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES")) and ent.get("n_swept_per_launch"):
                    # The compute ceiling beside the HBM one (same file, same sha stamp).  One VALU wave64 instruction occupies a
                    # SIMD's 16 lanes for 4 cycles, so a SIMD issues at most clock/4 of them per second; 1024 SIMDs.  The counters
                    # are per LAUNCH of tools/pmc_run.py's frame (its swept-voxel count rides along); SQ_ACTIVE_INST_VALU is
                    # summed over the 4 SIMDs of a CU in units of 4 cycles and SQ_BUSY_CYCLES over 32 shader-engine halves
                    # ("in counters"), hence the two scalings.
                    clock_hz, simds = 2.4e9, 1024
                    peak_issue = simds * clock_hz / 4.0
                    launch_cycles = c["SQ_BUSY_CYCLES"] / 32.0
                    insts = c["SQ_INSTS_VALU"]
                    compute = {"valu_insts_per_swept_voxel": insts * 64.0 / ent["n_swept_per_launch"],
                               "valu_wave_insts_per_launch": insts, "salu_wave_insts_per_launch": c.get("SQ_INSTS_SALU"),
                               "valu_busy_frac": 4.0 * c["SQ_ACTIVE_INST_VALU"] / simds / launch_cycles,
                               "peak_issue_rate": peak_issue, "unit": "VALU wave64 instructions/s (1024 SIMDs x 2.4 GHz / 4 cycles)",
                               # this run's frames: the profiled instructions per swept voxel x the voxels THIS run's launches sweep, over
                               # this run's HIP-event integrate time (which includes the verdict pass and the plan, ~5 %)
                               "achieved_issue_rate": insts / ent["n_swept_per_launch"] * n_swept_launch / (ms_int * 1e-3),
                               "waves_per_simd": (c["SQ_WAVE_CYCLES"] * 4.0 / simds / launch_cycles) if "SQ_WAVE_CYCLES" in c else None,
                               "n_swept_per_launch_profiled": ent["n_swept_per_launch"],
                               "source": "profiles/pmc_latest.json (rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES, separate pass; "
                                         "same source sha256 as traffic)"}
                    compute["achieved_frac"] = compute["achieved_issue_rate"] / peak_issue
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
                       "halo": (("recompute: integrated redundantly by every rank, no halo collective (the north star words it as an RCCL halo exchange: that form is --halo exchange, "
                                 "timed in the same launch as scaling_detail.variants; 2 x %d planes of sweep cost less than 2 x %.1f MiB over xGMI at this size)" % (halo, halo * X * Y * 4 / 2.0 ** 20)
                                 if halo_main == "recompute" else
                                 "exchanged after the integrate (paired isend/irecv of %d planes per side)" % halo) if dist_on else None),
                       "raycast_merge": (("all_reduce(MIN) of the keys + reduce_scatter(SUM) of the normals by pixel rows: every rank finishes its band "
                                          "of %d rows, the image stays row-sharded" % sharded.row_bands(cfg.rows, world)[0]) if args.merge == "rows" else
                                         ("all_reduce(MIN) of the keys + one direct all-to-all of the normals' row bands (fixed-size pieces, no counts) + a local "
                                          "sum: every rank finishes its band of %d rows" % sharded.row_bands(cfg.rows, world)[0]) if args.merge == "a2a" else
                                         "all_reduce(MIN) of the keys + reduce(SUM) of the normals to rank 0") if dist_on else None,
                       "direct_collectives_refused": direct_fallback,
                       "key_merge": (("direct: all-to-all of the keys' row bands + local minimum + all-gather (two one-step exchanges instead of a ring "
                                      "all_reduce; the `all_reduce(MIN)` named in raycast_merge is made this way)") if args.key_merge == "direct" else
                                     "ring: ncclAllReduce(MIN)") if dist_on else None,
                       "inputs": (("rank 0 sends depth + node transforms to every rank point-to-point, one group (each copy on its own xGMI link)"
                                   if args.bcast == "direct" else "ncclBroadcast of depth + node transforms from rank 0") if dist_on else None),
                       "frame": "set_transforms + compute_dists + integrate_warped + raycast_points" +
                                (" (frames pipelined over two streams: the volume-free half of frame f + 1 runs beside the sweep and ray-cast of frame f)" if pipeline else "")},
            "kernel_ms": {"integrate_warped": ms_int, "raycast(+merge)": ms_ray, "index_build_once_s": t_index,
                          "integrate_prepare_on_side_stream": ms_prep,
                          "pipelined": ("set_transforms + compute_dists + dfusion_integrate_warped_prepare (pyramid, verdict pass, plan) on a second stream, beside the "
                                        "previous frame's sweep and ray-cast; integrate_warped = that + the sweep, each between its own HIP events, so ms_per_step < "
                                        "integrate_warped + raycast") if pipeline else None},
            "frame_stats": frame_stats,
            "scaling_detail": scaling_detail,
            "roofline": {"kernel": kernel_name, "bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_source": traffic_src, "compute": compute,
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
        if verify_cull is not None:
            out["verify_cull"] = verify_cull
        if rebalance is not None:
            out["config"]["rebalance"] = rebalance
        if oversub:
            out["oversubscribed"] = ("%d ranks on %d GPU(s): gloo with host-staged collectives, ranks share devices -- a smoke of the "
                                     "sharded code path, NOT a measurement" % (world, n_dev))
        if not dist_on and not args.no_cpu_baseline:
            vol_host = vol.download()
            # the same frame on the GPU from the same start volume (for cpu_baseline.integrate_bit_identical)
            fr = timed_frames[0]                        # the first timed pose of the sweep
            dq_fr = synth.node_transforms(cfg, fr)
            wf.set_transforms(dqs[fr])
            vol.integrate_warped(compute_dists(depths[fr], intr, dists), cam_poses[fr], intr, wf)
            vol_after = vol.download()
            out["cpu_baseline"] = cpu_baseline(cfg, (depths_np[fr], None, cfg.volume_pose, cam_poses[fr], pos, sigma, dq_fr), vol_host,
                                               gpu_after=vol_after)
            out["cpu_baseline"]["frame"] = fr
            del vol_after
            rcb = out["cpu_baseline"].pop("raycast_algorithmic_bytes")
            out["raycast"] = {"kernel": "df_raycast_kernel<0>", "ms": ms_ray, "algorithmic_bytes": rcb,
                              "steps": out["cpu_baseline"].pop("raycast_steps"), "hits": out["cpu_baseline"].pop("raycast_hits"),
                              "achieved_GBps": rcb / (ms_ray * 1e-3) / 1e9,
                              "note": "issue-bound while every wave marches (19 VALU instructions + one 4-byte gather per step, 4 steps in flight), gather latency in its tail; reported, no roofline target (SURVEY 8d)"}
            try:
                rw = reference_warp_baseline(cfg, pts, pos, sigma, dq_fr, wf)
                if rw:
                    out["cpu_baseline"]["reference_warp"] = rw
            except Exception as e:                      # the reference build is optional; never lose the bench line over it
                out["cpu_baseline"]["reference_warp"] = {"error": repr(e)[:200]}
        if not dist_on and args.config == "512" and not args.nodes and not args.no_other_configs and not args.no_extras:
            out["other_configs"] = other_configs(args)
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


    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
