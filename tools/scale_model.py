#!/usr/bin/env python3
"""Predicted frame time of the Z-slab sharded frame at N = 1 / 2 / 4 / 8 GPUs -- the table the driver's SCALE run is to be checked
against (VERDICT r2 #5 ii; no multi-GPU node is available to the build).

Per-rank KERNEL times are measured, on the one GPU there is: for every rank r of N the slab it would own (own planes + the halo planes
it integrates itself, bench.py's default --halo recompute) is built and timed alone -- warped integrate, march, shade -- and the frame
takes the SLOWEST rank (the far slabs hold most of the frustum).  COLLECTIVE times are a model, stated here so that it can be wrong in a
checkable way:
    t(collective) = launches * T_LAUNCH + steps * T_HOP + bytes_on_the_busiest_link / LINK_GBPS
with xGMI ~153 GB/s per link and direction pair (guide), of which a ring step sustains LINK_GBPS = 100; T_HOP = 5 us per ring step
(RCCL's LL/LL128 protocols for messages of a few MB), T_LAUNCH = 15 us per collective call (host enqueue + kernel start):
    broadcast  (depth + transforms, 0.68 MB):  N - 1 ring steps, bytes = size          [direct, round 6: one step, bytes = size]
    all_reduce MIN of the int64 keys (2.46 MB): 2 (N - 1) steps, bytes = 2 (N - 1) / N * size   [direct: 2 launches x (1 step, size / N)]
    reduce SUM of the normals to rank 0 (4.92 MB; the points follow from the merged keys): N - 1 steps, bytes = size
Usage:  python tools/scale_model.py [CONFIG] [balanced|uniform]     (prints a markdown table, writes gpurun_out/scale_model_<cfg>_<kind>.json)
`measured` (bench.py's default, round 4): slab boundaries re-cut from the verdict pass's alive-block counts per 8-plane layer after three
priming frames on the unsharded volume (what bench.py's ranks all-reduce); `balanced`: from the a-priori frustum_plane_weights;
`uniform`: equal plane counts.
"""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from dynamicfusion_amd import Intr, TsdfVolume, WarpField, compute_dists, sharded, synth, upload_u16  # noqa: E402

T_LAUNCH, T_HOP, LINK_GBPS = sharded.T_LAUNCH, sharded.T_HOP, sharded.LINK_GBPS      # (the model lives in dynamicfusion_amd/sharded.py: bench.py prints it next to what it measures)
collective = sharded.collective_model_s


def timeit(fn, n=30):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "512"
    cfg = synth.CONFIGS[name]; intr = Intr(*cfg.intr)
    X, Y, Z = cfg.dims
    vs_z = cfg.size / Z
    halo = sharded.halo_planes(max(cfg.trunc_dist, 2.1 * vs_z), cfg.raycast_step_factor, cfg.gradient_delta_factor, vs_z)
    dists = compute_dists(upload_u16(synth.depth_frame(cfg, 0)), intr)
    cam = synth.camera_pose(cfg, 1)
    pos, sigma = synth.make_nodes(cfg)
    dq = torch.from_numpy(synth.node_transforms(cfg, 1)).cuda()
    px = cfg.rows * cfg.cols
    sizes = {"broadcast": px * 2 + cfg.nodes * 32, "all_reduce": px * 8, "reduce": px * 16}        # (the reduce sums the normals only)
    wts = sharded.frustum_plane_weights(cfg.dims, cfg.size, cfg.volume_pose, synth.camera_pose(cfg, 0), cfg.intr, cfg.cols, cfg.rows,
                                        depth_mm=synth.depth_frame(cfg, 0), trunc=max(cfg.trunc_dist, 2.1 * vs_z), margin=0.3)
    kind = sys.argv[2] if len(sys.argv) > 2 else "measured"
    if kind == "measured":                                  # the profile bench.py's ranks measure and sum: alive blocks per 8-plane layer
        v0 = TsdfVolume(cfg.dims); v0.setSize([cfg.size] * 3); v0.setTruncDist(cfg.trunc_dist); v0.setMaxWeight(cfg.max_weight); v0.setPose(cfg.volume_pose)
        w0 = WarpField(k=cfg.k); w0.init(pos, sigma=sigma, transforms=synth.node_transforms(cfg, 0))
        for f in range(3):
            w0.set_transforms(torch.from_numpy(synth.node_transforms(cfg, f)).cuda())
            v0.integrate_warped(compute_dists(upload_u16(synth.depth_frame(cfg, f)), intr), synth.camera_pose(cfg, f), intr, w0)
        layers = torch.zeros(Z // 8, dtype=torch.int64, device="cuda")
        w0.alive_blocks_per_layer(v0, layers)
        wts = sharded.layer_weights_to_planes(layers.cpu().numpy(), Z)
        del v0, w0
        torch.cuda.empty_cache()
    rows = []
    for n in (1, 2, 4, 8):
      # `measured` (round 6): boundaries that minimise the largest rank's cost, halos included (slab_bounds_minmax), then ONE more
      # re-cut from the per-rank integrate times this loop has just measured (sharded.reweight_from_times) -- what bench.py's ranks do
      bounds = (sharded.slab_bounds_minmax(Z, n, halo, wts) if kind == "measured" else sharded.slab_bounds(Z, n, halo, wts if kind == "balanced" else None)) if n > 1 else [0, Z]
      for attempt in range(2 if (kind == "measured" and n > 1) else 1):
        worst = {"integrate": 0.0, "march": 0.0, "shade": 0.0, "sum": 0.0}
        per_rank = []
        for r in range(n):
            z0, zn = bounds[r], bounds[r + 1] - bounds[r]
            vol = TsdfVolume(cfg.dims, slab=(z0, zn, halo if n > 1 else 0))
            vol.setSize([cfg.size] * 3); vol.setTruncDist(cfg.trunc_dist); vol.setMaxWeight(cfg.max_weight); vol.setPose(cfg.volume_pose)
            vol.setRaycastStepFactor(cfg.raycast_step_factor); vol.setGradientDeltaFactor(cfg.gradient_delta_factor)
            vol.clear()
            vint = vol.owning_stored_planes() if n > 1 else vol
            wf = WarpField(k=cfg.k); wf.init(pos, sigma=sigma, transforms=synth.node_transforms(cfg, 0)); wf.ensure_index(vint, cfg.k)
            wf.set_transforms(dq)
            for _ in range(3):
                vint.integrate_warped(dists, cam, intr, wf, sync=False)
            keys = torch.empty((cfg.rows, cfg.cols), dtype=torch.int64, device="cuda")
            out = torch.empty((2, cfg.rows, cfg.cols, 4), dtype=torch.float32, device="cuda")
            t_i = timeit(lambda: (wf.set_transforms(dq), vint.integrate_warped(dists, cam, intr, wf, sync=False)))
            t_m = timeit(lambda: vol.raycast_march(cam, intr, keys, r))
            t_s = timeit(lambda: (vol.raycast_shade(cam, intr, keys, None, out[1]), vol.raycast_points_of_keys(cam, intr, keys, out[1], out[0])))   # (rank 0's share)
            per_rank.append((t_i, t_m, t_s))
            if t_i + t_m + t_s > worst["sum"]:
                worst = {"integrate": t_i, "march": t_m, "shade": t_s, "sum": t_i + t_m + t_s}
            del wf, vol, vint
            torch.cuda.empty_cache()
        if attempt == 0 and kind == "measured" and n > 1:
            t_ms = [1e3 * p[0] for p in per_rank]
            first = {"bounds": bounds, "per_rank_integrate_ms": t_ms}
            b2 = sharded.slab_bounds_minmax(Z, n, halo, sharded.reweight_from_times(bounds, wts, halo, t_ms, 0.045))
            if b2 == bounds or max(t_ms) <= 1.08 * float(np.mean(t_ms)):
                break
            bounds = b2
      if True:
        comm = {k: collective(k, sizes[k], n) for k in sizes}
        t_frame = worst["sum"] + sum(comm.values())
        comm["all_to_all"] = collective("all_to_all", sizes["reduce"], n)      # round 5: the direct form of the second collective (bench.py --merge a2a)
        t_frame_a2a = t_frame - comm["reduce"] + comm["all_to_all"]
        # round 6: no ring left -- the inputs as N - 1 point-to-point sends, the key merge as all-to-all of row bands + local MIN + all-gather
        # (bench.py's defaults: --bcast direct --key-merge direct --merge a2a)
        comm["broadcast_direct"] = collective("broadcast_direct", sizes["broadcast"], n)
        comm["all_reduce_direct"] = collective("all_reduce_direct", sizes["all_reduce"], n)
        t_frame_direct = worst["sum"] + comm["broadcast_direct"] + comm["all_reduce_direct"] + comm["all_to_all"]
        rows.append({"n": n, "halo": halo if n > 1 else 0, "slabs": kind, "bounds": bounds, "kernels_ms": {k: 1e3 * v for k, v in worst.items()},
                     "collectives_ms": {k: 1e3 * v for k, v in comm.items()}, "frame_ms": 1e3 * t_frame, "frames_per_s": 1.0 / t_frame, "frame_ms_a2a": 1e3 * t_frame_a2a, "frame_ms_direct": 1e3 * t_frame_direct,
                     "per_rank_integrate_ms": [1e3 * p[0] for p in per_rank],
                     "first_cut": first if (kind == "measured" and n > 1) else None})
    base = rows[0]["frames_per_s"]
    print("| N | planes swept by the slowest rank's kernels: integrate / march / shade (ms, measured on one GPU) | broadcast / all_reduce(MIN) / reduce(SUM) (ms, model) | frame (ms) | frames/s | speed-up |")
    print("|---|---|---|---|---|---|")
    for r in rows:
        k, c = r["kernels_ms"], r["collectives_ms"]
        print("| %d | %.3f / %.3f / %.3f | %.3f / %.3f / %.3f | %.3f | %.0f | %.2fx |" % (
            r["n"], k["integrate"], k["march"], k["shade"], c["broadcast"], c["all_reduce"], c["reduce"], r["frame_ms"], r["frames_per_s"],
            r["frames_per_s"] / base))
    print("second collective as ONE direct all-to-all of the row bands (--merge a2a): " +
          ", ".join("N = %d: %.3f ms -> frame %.3f ms (%.2fx)" % (r["n"], r["collectives_ms"]["all_to_all"], r["frame_ms_a2a"], rows[0]["frame_ms"] / r["frame_ms_a2a"]) for r in rows))
    print("every collective direct (bench.py's defaults since round 6: --bcast direct --key-merge direct --merge a2a): " +
          ", ".join("N = %d: %.3f + %.3f + %.3f ms -> frame %.3f ms (%.2fx)" % (r["n"], r["collectives_ms"]["broadcast_direct"], r["collectives_ms"]["all_reduce_direct"],
                                                                              r["collectives_ms"]["all_to_all"], r["frame_ms_direct"], rows[0]["frame_ms"] / r["frame_ms_direct"]) for r in rows))
    for r in rows:
        print("N = %d (%s slabs %s) integrate per rank (ms):" % (r["n"], kind, r["bounds"]), " ".join("%.3f" % v for v in r["per_rank_integrate_ms"]))
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump({"config": name, "model": {"T_LAUNCH_us": T_LAUNCH * 1e6, "T_HOP_us": T_HOP * 1e6, "LINK_GBPS": LINK_GBPS, "bytes": sizes},
               "rows": rows}, open(os.path.join(REPO, "gpurun_out", "scale_model_%s_%s.json" % (name, kind)), "w"), indent=1)


if __name__ == "__main__":
    main()
