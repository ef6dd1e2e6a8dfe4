"""Z-slab sharding of the TSDF volume over the GPUs of one node (one process per GPU, RCCL via
torch.distributed).  The reference is single-GPU; this is the only parallelism of the path
(SURVEY.md 8e):

  integrate   embarrassingly parallel per voxel: each rank updates the planes it owns; inputs (depth,
              node transforms) are broadcast from rank 0; no exchange.
  halo        after integrate every rank sends its H boundary planes to each Z neighbour
              (paired isend/irecv = ncclSend/ncclRecv over one xGMI link per neighbour).
  raycast     every rank marches ALL rays on the same global step lattice but only evaluates steps
              whose `curr` sample lies in a plane it owns, emitting per pixel ONE 64-bit key: first
              event (step, hit / back-face break), rank, and -- for a hit -- the refined ray parameter
              Ts as f32 bits (include/dfusion.h).  Merge = ONE per-pixel MIN over ranks (all_reduce MIN
              on int64): it names the first event along every ray and hands every rank the winner's
              Ts, from which the vertex origin + direction * Ts is recomputed locally, bit for bit as
              the unsharded cast computes it.  The refinement can extrapolate the vertex into ANOTHER
              rank's slab (tsdf_volume.cu:389), so the normal is computed by the rank that owns the
              vertex and the final point/normal bits are summed to rank 0 (every other rank contributes
              integer zero: bit-identical with the unsharded cast).

The collectives go through `torch.distributed`, so the same code runs over RCCL on GPUs and over gloo
in the world_size-2 CPU tests (tests/test_sharded_cpu.py), where a stand-in backend supplies the
per-slab kernels.
"""
import math

import numpy as np
import torch
import torch.distributed as dist

# Host staging of the collectives: a process group whose backend cannot take device tensors (gloo, used when a box has fewer
# GPUs than ranks -- bench.py's oversubscribed smoke mode and the CPU tests) gets them as host copies, copied back afterwards.
# Off (the default) the device tensors go to the backend as they are (RCCL).
HOST_STAGED = False


def set_host_staging(on):
    global HOST_STAGED
    HOST_STAGED = bool(on)


def _staged(t):
    return HOST_STAGED and t.is_cuda


def coll_broadcast(t, src=0, group=None):
    if _staged(t):
        c = t.cpu(); dist.broadcast(c, src, group=group); t.copy_(c)
    else:
        dist.broadcast(t, src, group=group)


def coll_all_reduce(t, op, group=None):
    if _staged(t):
        c = t.cpu(); dist.all_reduce(c, op=op, group=group); t.copy_(c)
    else:
        dist.all_reduce(t, op=op, group=group)


def coll_reduce(t, dst, op, group=None):
    if _staged(t):
        c = t.cpu(); dist.reduce(c, dst=dst, op=op, group=group); t.copy_(c)
    else:
        dist.reduce(t, dst=dst, op=op, group=group)


def row_bands(rows, world):
    """Equal bands of pixel rows for the row-banded merge: rank r finishes rows [r * per, min((r + 1) * per, rows)); the normals buffer
    that is reduce-scattered has world * per rows (the rows past the image stay zero)."""
    per = (rows + world - 1) // world
    return per, [(min(rows, r * per), max(0, min(rows, (r + 1) * per) - min(rows, r * per))) for r in range(world)]


def coll_reduce_scatter_rows(full, out, group=None):
    """out[per, ...] <- rank's band of the elementwise SUM over the ranks of full[world * per, ...] (int32 views).  RCCL: one
    ncclReduceScatter.  gloo has no reduce-scatter (and host-staged runs go through gloo): all_reduce + slice there -- the same result."""
    rank = dist.get_rank(group)
    per = out.shape[0]
    if dist.get_backend(group) == "nccl" and not _staged(full):
        dist.reduce_scatter_tensor(out, full, op=dist.ReduceOp.SUM, group=group)
    else:
        c = full.cpu() if full.is_cuda else full.clone()
        dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
        out.copy_(c[rank * per:(rank + 1) * per])


def coll_all_to_all_rows(full, recv, group=None):
    """recv[s] <- rank s's copy of THIS rank's band: full is [world * per, ...] (band d = what goes to rank d), recv [world, per, ...].
    RCCL: one ncclSend / ncclRecv group (all_to_all_single with equal splits) -- every byte crosses ONE xGMI link once, the seven links
    of a rank in parallel, no ring.  gloo / host-staged runs: all_to_all_single where the backend has it, else paired isend / irecv."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = recv.shape[1]
    assert full.shape[0] == world * per and recv.shape[0] == world
    if dist.get_backend(group) == "nccl" and not _staged(full):
        dist.all_to_all_single(recv.view(world * per, *recv.shape[2:]), full, group=group)
        return
    src = full.cpu() if full.is_cuda else full
    dst = torch.empty((world * per,) + tuple(full.shape[1:]), dtype=full.dtype)
    try:
        dist.all_to_all_single(dst, src.contiguous(), group=group)
    except (RuntimeError, NotImplementedError):                    # (a gloo build without alltoall: the same exchange as point-to-point pairs)
        ops = []
        for r in range(world):
            if r == rank:
                dst[r * per:(r + 1) * per].copy_(src[r * per:(r + 1) * per])
            else:
                ops.append(dist.P2POp(dist.isend, src[r * per:(r + 1) * per].contiguous(), r, group))
                ops.append(dist.P2POp(dist.irecv, dst[r * per:(r + 1) * per], r, group))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    recv.copy_(dst.view(recv.shape))


def coll_all_reduce_min_direct(keys_pad, recv, band, group=None):
    """keys_pad[world * per, cols] int64 <- the per-key MIN over the ranks, WITHOUT a ring (round 6): ncclAllReduce on a ring is 2 (N - 1)
    sequential steps; the GPUs of an xGMI node are linked pairwise, so the same result is two exchanges of ONE step each --
      (1) all-to-all of the row bands: rank r receives every rank's copy of band r (recv [world, per, cols]; fixed-size pieces),
      (2) r takes the per-key minimum of its N pieces (dfusion_raycast_min_pieces on the GPU) into band [per, cols],
      (3) all-gather of the merged bands back into keys_pad on every rank.
    Rows past the image must hold KEY_NONE (or any value: they are never read back as pixels)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = band.shape[0]
    assert keys_pad.dtype == torch.int64 and keys_pad.shape[0] == world * per and recv.shape[0] == world and recv.shape[1] == per
    coll_all_to_all_rows(keys_pad, recv, group=group)
    if recv.is_cuda:
        from . import capi
        capi.check(capi.lib().dfusion_raycast_min_pieces(recv.data_ptr(), world, recv[0].numel(), band.data_ptr(), torch.cuda.current_stream().cuda_stream),
                   "dfusion_raycast_min_pieces")
    else:
        torch.amin(recv, dim=0, out=band)
    if dist.get_backend(group) == "nccl" and not _staged(keys_pad):
        dist.all_gather_into_tensor(keys_pad, band, group=group)
    else:
        pieces = [torch.empty(band.shape, dtype=band.dtype) for _ in range(world)]
        dist.all_gather(pieces, band.cpu() if band.is_cuda else band.contiguous(), group=group)
        keys_pad.copy_(torch.cat(pieces, dim=0))


def coll_broadcast_direct(t, src=0, group=None, scratch=None):
    """rank `src`'s tensor to every rank as N - 1 point-to-point copies in ONE group, each on its own xGMI link, all links of `src` at once --
    one step instead of the N - 1 of a ring broadcast; what the frame inputs (0.68 MB) want.  RCCL: one all_to_all_single with uneven
    splits (the grouped ncclSend / ncclRecv the other direct exchanges use, on the group's own communicator): `src` sends the whole tensor
    to every rank, everybody else sends nothing; `scratch` (optional, N * t.numel() elements on `src`) holds the N copies the call reads.
    gloo / host-staged: the same as isend / irecv pairs on CPU copies."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if dist.get_backend(group) == "nccl" and not _staged(t):
        flat = t.view(-1)
        n = flat.numel()
        if rank == src:
            rep = scratch.view(-1)[:world * n] if scratch is not None else torch.empty(world * n, dtype=t.dtype, device=t.device)
            rep.view(world, n).copy_(flat.unsqueeze(0).expand(world, n))
            out = torch.empty(n, dtype=t.dtype, device=t.device)               # (its own copy comes back: the input is not also an output)
            dist.all_to_all_single(out, rep, output_split_sizes=[n if r == src else 0 for r in range(world)], input_split_sizes=[n] * world, group=group)
        else:
            dist.all_to_all_single(flat, flat[:0], output_split_sizes=[n if r == src else 0 for r in range(world)], input_split_sizes=[0] * world, group=group)
        return
    if world == 1:
        return
    buf = t.cpu() if t.is_cuda else t
    if rank == src:
        ops = [dist.P2POp(dist.isend, buf, r, group) for r in range(world) if r != src]
    else:
        ops = [dist.P2POp(dist.irecv, buf, src, group)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    if rank != src and buf is not t:
        t.copy_(buf)


# ---- the collective model of tools/scale_model.py, here so that bench.py can print the PREDICTED time of every
# collective next to the one it measures: t = launches * T_LAUNCH + steps * T_HOP + bytes on the busiest link / LINK_GBPS
T_LAUNCH, T_HOP, LINK_GBPS = 15e-6, 5e-6, 100.0


def collective_model_s(kind, size, n):
    """kind: broadcast | all_reduce | reduce | reduce_scatter (rings: N - 1 or 2 (N - 1) steps) | all_to_all (direct: one step, size / N
    per link, all links of a rank in parallel) | all_reduce_direct | broadcast_direct (round 6: the same results as direct exchanges) |
    halo (one paired send / receive of `size` bytes per side)."""
    if n == 1:
        return 0.0
    if kind == "broadcast":
        steps, b = n - 1, size
    elif kind == "all_reduce":
        steps, b = 2 * (n - 1), 2.0 * (n - 1) / n * size
    elif kind == "reduce_scatter":
        steps, b = n - 1, (n - 1.0) / n * size
    elif kind == "all_to_all":
        steps, b = 1, size / n
    elif kind == "all_reduce_direct":          # all-to-all of the bands + all-gather of the merged bands: two launches of one step, size / N per link each
        return 2 * (T_LAUNCH + T_HOP + (size / n) / (LINK_GBPS * 1e9))
    elif kind == "broadcast_direct":           # N - 1 sends from one rank, each on its own link
        steps, b = 1, size
    elif kind == "halo":
        steps, b = 1, size
    else:
        steps, b = n - 1, size
    return T_LAUNCH + steps * T_HOP + b / (LINK_GBPS * 1e9)


class StageTimer:
    """Per-stage times of a frame: mark(name) closes the stage `name` (everything enqueued on the current stream since the previous
    mark).  HIP events on the launch stream -- torch.distributed's collectives make that stream wait for them before returning, so a
    mark after a collective is after its completion -- or the host clock without a GPU.  means() -> {stage: mean ms}, in first-seen order."""

    def __init__(self, cuda=None):
        self.cuda = torch.cuda.is_available() if cuda is None else cuda
        self.frames, self.cur = [], None

    def _now(self):
        if self.cuda:
            e = torch.cuda.Event(enable_timing=True); e.record(); return e
        import time
        return time.perf_counter()

    def start(self):
        self.cur = [("", self._now())]

    def mark(self, name):
        if self.cur is not None:
            self.cur.append((name, self._now()))

    def end(self):
        if self.cur is not None:
            self.frames.append(self.cur); self.cur = None

    def means(self):
        if self.cuda:
            torch.cuda.synchronize()
        acc, order = {}, []
        for fr in self.frames:
            for (_, a), (name, b) in zip(fr[:-1], fr[1:]):
                ms = a.elapsed_time(b) if self.cuda else 1e3 * (b - a)
                if name not in acc:
                    acc[name] = []; order.append(name)
                acc[name].append(ms)
        return {k: float(np.mean(acc[k])) for k in order}


NO_EVENT = 0xFFFFFFFF                 # per-slab event key (step << 1 | hit): none
KEY_NONE = 0x7FFFFFFFFFFFFFFF         # merge key: no event (include/dfusion.h DF_RC_KEY_NONE)
MAX_RANKS = 128                       # the merge key carries the rank in 7 bits


def pack_merge_keys(event_keys, ts, rank):
    """numpy: per-slab event keys (uint32, NO_EVENT = none) + Ts (float32) -> int64 merge keys, the layout dfusion_raycast_march
    writes: [0 | step k : 23 | hit : 1 | rank : 7 | Ts bits : 32]."""
    k = np.asarray(event_keys, np.uint32).astype(np.int64)
    t = np.ascontiguousarray(ts, np.float32).view(np.uint32).astype(np.int64)
    out = (((k << 7) | int(rank)) << 32) | t
    out[np.asarray(event_keys, np.uint32) == NO_EVENT] = KEY_NONE
    return out


def unpack_merge_keys(keys64):
    """-> (event keys uint32 with NO_EVENT for none, Ts float32, rank)"""
    k64 = np.asarray(keys64, np.int64)
    none = k64 == KEY_NONE
    ev = ((k64 >> 39) & 0xFFFFFF).astype(np.uint32)
    ev[none] = NO_EVENT
    ts = (k64 & 0xFFFFFFFF).astype(np.uint32).view(np.float32)
    return ev, np.where(none, np.float32(0), ts), ((k64 >> 32) & 0x7F).astype(np.int32)


def slab_range(Z, rank, world):
    """Planes owned by `rank`: contiguous, multiples of 8 where possible (brick-aligned)."""
    per = (Z // 8 + world - 1) // world * 8 if Z % 8 == 0 else (Z + world - 1) // world
    lo = min(Z, rank * per)
    hi = min(Z, lo + per)
    return lo, hi - lo


def frustum_plane_weights(dims, size, volume_pose, camera_pose, intr, cols, rows, samples=48, depth_mm=None, trunc=0.04, margin=0.0):
    """Relative cost of sweeping each Z plane of the volume for a camera at `camera_pose`: the share of the plane that lies inside
    the image frustum (a `samples` x `samples` lattice of its voxel centres projected with the frame's intrinsics), plus a floor
    for the planes the launch plans skip.  The integrate's work is proportional to the voxels in front of the camera and inside
    the frustum, and a frustum's cross-section grows with the square of the depth: equal-plane slabs leave the far rank with
    several times the near rank's work (measured: 0.73 vs 0.32 ms at two ranks, 512^3 -- DESIGN.md section 5).
    depth_mm (optional, uint16 [rows, cols], the first sensor frame): samples that the integrate's own test would skip -- no depth
    at their pixel, or more than `trunc` behind the observed surface (tsdf_volume.cu:86-91) -- do not count either: the planes
    behind the scene cost a launch plan, not a sweep.  margin (metres): how far the warp field can move a voxel -- the warped
    sweep must keep everything within that distance of the frustum and of the surface, so its work reaches that much further."""
    X, Y, Z = [int(d) for d in dims]
    vs = np.asarray(size, np.float64).reshape(-1)[:3] / np.array([X, Y, Z], np.float64) if np.ndim(size) else np.full(3, float(size)) / np.array([X, Y, Z], np.float64)
    v2c = np.linalg.inv(np.asarray(camera_pose, np.float64)) @ np.asarray(volume_pose, np.float64)
    fx, fy, cx, cy = [float(v) for v in intr]
    gx = (np.arange(samples) + 0.5) * (X / samples) * vs[0]
    gy = (np.arange(samples) + 0.5) * (Y / samples) * vs[1]
    px, py = np.meshgrid(gx, gy)
    w = np.empty(Z, np.float64)
    for z in range(Z):
        p = np.stack([px.ravel(), py.ravel(), np.full(px.size, z * vs[2]), np.ones(px.size)])
        c = v2c @ p
        zc = c[2]
        with np.errstate(divide="ignore", invalid="ignore"):
            u = fx * c[0] / zc + cx
            v = fy * c[1] / zc + cy
        with np.errstate(divide="ignore", invalid="ignore"):
            mu, mv = margin * fx / zc, margin * fy / zc                    # the margin in pixels at this depth
        inside = (zc > -margin) & (u >= -mu) & (u < cols + mu) & (v >= -mv) & (v < rows + mv)
        if margin > 0:
            inside &= zc > 0.05
        if depth_mm is not None:
            ui = np.clip(np.nan_to_num(u, nan=0.0, posinf=0.0, neginf=0.0), 0, cols - 1).astype(np.int64)
            vi = np.clip(np.nan_to_num(v, nan=0.0, posinf=0.0, neginf=0.0), 0, rows - 1).astype(np.int64)
            d = np.asarray(depth_mm)[vi, ui].astype(np.float64) * 1e-3
            lam = np.sqrt(((ui - cx) / fx) ** 2 + ((vi - cy) / fy) ** 2 + 1.0)            # compute_dists, imgproc.cu:259-272
            rng = np.sqrt(c[0] ** 2 + c[1] ** 2 + zc ** 2)
            inside &= (d > 0) & (d * lam - rng >= -(trunc + margin))
        w[z] = inside.mean()
    return w + 0.02                                       # (plan kernels and culled planes are not free)


def slab_bounds(Z, world, halo=0, weights=None):
    """Slab boundaries b[0] = 0 < b[1] < ... < b[world] = Z (multiples of 8 where Z allows): rank r owns planes [b[r], b[r + 1]).
    weights = None: equal plane counts (slab_range).  Otherwise planes are dealt out so that every rank gets about the same share
    of `weights` (one non-negative number per plane, e.g. frustum_plane_weights), every slab keeping at least max(halo, 8) planes."""
    if weights is None:
        return [slab_range(Z, r, world)[0] for r in range(world)] + [Z]
    w = np.asarray(weights, np.float64).reshape(-1)
    assert w.size == Z and (w >= 0).all()
    step = 8 if Z % 8 == 0 else 1
    min_planes = max(int(halo), step)
    min_planes = (min_planes + step - 1) // step * step
    cum = np.concatenate([[0.0], np.cumsum(w)])
    total = cum[-1] if cum[-1] > 0 else 1.0
    b = [0]
    for r in range(1, world):
        target = total * r / world
        z = int(np.searchsorted(cum, target))
        z = int(round(z / step)) * step
        z = max(z, b[-1] + min_planes)                     # this slab is thick enough ...
        z = min(z, Z - (world - r) * min_planes)           # ... and so can every later one be
        b.append(z)
    b.append(Z)
    return b


def slab_costs(bounds, weights, halo=0):
    """What each rank sweeps under halo RECOMPUTE (bench.py's default): the weights of its own planes AND of the `halo` planes it
    integrates either side of them (clipped to the volume)."""
    w = np.asarray(weights, np.float64).reshape(-1)
    cum = np.concatenate([[0.0], np.cumsum(w)])
    Z = w.size
    return [float(cum[min(Z, bounds[r + 1] + halo)] - cum[max(0, bounds[r] - halo)]) for r in range(len(bounds) - 1)]


def slab_bounds_minmax(Z, world, halo, weights, count_halo=True):
    """Boundaries that minimise the LARGEST per-rank cost (slab_costs: own planes + the halo planes the rank integrates itself) instead of
    equalising the own planes' shares (slab_bounds): interior ranks carry two halos, the outer ones one, and at 8 ranks the 16 halo planes are
    a quarter of a 64-plane slab -- equal own shares left max / mean = 1.21 on the headline scene (profiles/r05_scale_model_512_measured.txt).
    Exact for the given granularity (multiples of 8 planes): a binary search on the bound T with a greedy left-to-right packing -- rank r
    takes planes until one more step would put its cost above T."""
    w = np.asarray(weights, np.float64).reshape(-1)
    assert w.size == Z and (w >= 0).all()
    step = 8 if Z % 8 == 0 else 1
    min_planes = (max(int(halo), step) + step - 1) // step * step
    H = int(halo) if count_halo else 0
    cum = np.concatenate([[0.0], np.cumsum(w)])

    def cost(z0, z1):
        return cum[min(Z, z1 + H)] - cum[max(0, z0 - H)]

    def pack(T):
        b = [0]
        for r in range(world - 1):
            z0 = b[-1]
            hi = Z - (world - 1 - r) * min_planes               # every later rank keeps its minimum
            z1 = z0 + min_planes
            if z1 > hi:
                return None
            while z1 + step <= hi and cost(z0, z1 + step) <= T:
                z1 += step
            if cost(z0, z1) > T and z1 > z0 + min_planes:       # (cannot happen: the loop only grows while under T)
                return None
            b.append(z1)
        b.append(Z)
        return b if cost(b[-2], Z) <= T and all(cost(b[r], b[r + 1]) <= T for r in range(world)) else None

    lo, hi = 0.0, float(cum[-1]) + 1.0
    best = pack(hi)
    if best is None:                                        # (the minimum slab thickness alone does not fit: equal shares)
        return slab_bounds(Z, world, halo, weights)
    for _ in range(48):
        mid = 0.5 * (lo + hi)
        b = pack(mid)
        if b is None:
            lo = mid
        else:
            hi, best = mid, b
    return best


def reweight_from_times(bounds, weights, halo, times_ms, fixed_ms=0.0):
    """Second re-cut (round 6): per-plane weights corrected by what the ranks MEASURED.  Rank r swept cost_r = slab_costs(...)[r] of the
    current weights in times_ms[r] (HIP events around its integrate, a few frames' mean); of that, fixed_ms is launch-sized work that
    does not move with the slab.  The planes a rank owns are rescaled by (t_r - fixed) / cost_r -- what a unit of the a-priori weight
    really costs there (far slabs hold more coded, denser blocks; a slab cut through a layer pays the whole layer's verdicts) -- and
    the boundaries are then made again from the corrected weights.  One step of a fixed-point iteration; bench.py takes one."""
    w = np.asarray(weights, np.float64).reshape(-1).copy()
    costs = slab_costs(bounds, w, halo)
    t = np.maximum(np.asarray(times_ms, np.float64) - float(fixed_ms), 1e-6)
    rate = np.array([t[r] / max(costs[r], 1e-12) for r in range(len(costs))])
    rate /= rate.mean()
    for r in range(len(costs)):
        w[bounds[r]:bounds[r + 1]] *= rate[r]
    return w


def layer_weights_to_planes(w_layer, Z):
    """per-8-plane-layer work (e.g. alive block counts, WarpField.alive_blocks_per_layer summed over the ranks) -> one weight per plane
    for slab_bounds, with the same small floor frustum_plane_weights gives culled planes (plan kernels are not free)."""
    w = np.repeat(np.asarray(w_layer, np.float64).reshape(-1), 8)[:Z]
    if w.size < Z:
        w = np.concatenate([w, np.zeros(Z - w.size)])
    peak = w.max() if w.size and w.max() > 0 else 1.0
    return w / peak + 0.02


def validate_bounds(bounds, Z, halo):
    """The checks of validate_slabs for an explicit boundary list (slab_bounds)."""
    world = len(bounds) - 1
    if world < 1 or world > MAX_RANKS:
        raise ValueError("world size %d: the merge key carries the rank in 7 bits (1..%d ranks)" % (world, MAX_RANKS))
    if bounds[0] != 0 or bounds[-1] != Z:
        raise ValueError("slab boundaries must run from 0 to %d" % Z)
    for r in range(world):
        n = bounds[r + 1] - bounds[r]
        if n <= 0:
            raise ValueError("slab boundaries leave rank %d without planes" % r)
        if world > 1 and n < halo:
            raise ValueError("rank %d owns %d planes, fewer than the %d halo planes its neighbours need" % (r, n, halo))


def validate_slabs(Z, world, halo):
    """Raise -- identically on every rank, BEFORE any collective is entered -- if the partition cannot work: a rank without planes,
    or a slab thinner than the halo its neighbour needs from it (exchange_halos would otherwise die inside a collective while the
    other ranks block in it)."""
    if world < 1 or world > MAX_RANKS:
        raise ValueError("world size %d: the merge key carries the rank in 7 bits (1..%d ranks)" % (world, MAX_RANKS))
    for r in range(world):
        lo, n = slab_range(Z, r, world)
        if n <= 0:
            raise ValueError("Z-slab partition of %d planes over %d ranks leaves rank %d without planes" % (Z, world, r))
        if world > 1 and n < halo:
            raise ValueError("rank %d owns %d planes, fewer than the %d halo planes its neighbours need" % (r, n, halo))


def halo_planes(trunc_dist, step_factor, delta_factor, voxel_z):
    """Planes of the neighbour a slab must hold: the march's `next` sample can be one time_step beyond
    `curr` (tsdf_volume.cu:378-380), trilinear taps read g+1 (:236-243), gradient probes reach
    +-gradient_delta (:413-423)."""
    return int(math.ceil(trunc_dist * step_factor / voxel_z + delta_factor)) + 2


def broadcast_bytes(t, src=0, group=None):
    """Broadcast any contiguous tensor as raw bytes (RCCL/gloo have no 16-bit integer type; the depth image
    is uint16)."""
    coll_broadcast(t.view(torch.uint8), src, group=group)


def exchange_halos(vol_tensor, z_store0, z_own0, z_own_n, Z, halo, rank, world, group=None):
    """vol_tensor: [z_store_n, Y, X] int32 (own planes + halos).  Sends own boundary planes to both Z
    neighbours and receives theirs into the halo planes."""
    if world == 1:
        return
    ops = []
    lo_local = z_own0 - z_store0
    if rank > 0:                       # lower neighbour: send my first `halo` own planes, receive its last ones
        n_lo = z_own0 - z_store0       # halo planes I hold below
        assert n_lo == halo and z_own_n >= halo, "slab thinner than the halo"
        ops.append(dist.P2POp(dist.isend, vol_tensor[lo_local:lo_local + halo], rank - 1, group))
        ops.append(dist.P2POp(dist.irecv, vol_tensor[0:n_lo], rank - 1, group))
    if rank < world - 1:
        hi_local = lo_local + z_own_n
        n_hi = vol_tensor.shape[0] - hi_local
        assert n_hi == halo and z_own_n >= halo, "slab thinner than the halo"
        ops.append(dist.P2POp(dist.isend, vol_tensor[hi_local - halo:hi_local], rank + 1, group))
        ops.append(dist.P2POp(dist.irecv, vol_tensor[hi_local:hi_local + n_hi], rank + 1, group))
    if HOST_STAGED and vol_tensor.is_cuda:
        host_ops, backs = [], []
        for op in ops:
            c = op.tensor.cpu()
            host_ops.append(dist.P2POp(op.op, c, op.peer, group))
            if op.op is dist.irecv:
                backs.append((op.tensor, c))
        for r in dist.batch_isend_irecv(host_ops):
            r.wait()
        for t, c in backs:
            t.copy_(c)
        return
    for r in dist.batch_isend_irecv(ops):
        r.wait()


def raycast_sharded(march_fn, shade_fn, points_fn, rank, world, dst=0, group=None, collectives=None, merge="root", band_out=None,
                    a2a_recv=None, timer=None, key_merge="ring", keys_pad=None, keys_recv=None, keys_band=None):
    """Sharded ray-cast (include/dfusion.h: dfusion_raycast_march / _shade / _points_of_keys).

    march_fn()                  -> keys64 int64 [rows, cols]: the merge keys of this slab (first event | rank | Ts bits)
    shade_fn(keys64)            -> normals float32 [rows (+ padding), cols, 4]; all-zero bits for pixels this slab does not resolve
    points_fn(keys64, normals)  -> points float32 [rows, cols, 4], called on rank `dst` only, with the summed normals
    Returns (points, normals) of the merged cast on rank `dst`, (None, None) elsewhere.

    merge = "root": TWO collectives per frame: all_reduce(MIN) of the int64 keys (2.4 MB at 640x480) -- which also delivers the
    winners' Ts -- and reduce(SUM) of the NORMAL bits to rank `dst` (4.9 MB; every summand but one is integer zero, so the sum is the
    owner's value, NaN fill included).  The points cross no link: vertex = origin + direction * Ts from the pixel, and whether a
    hit stands is in the normal's 4th component (round 2 exchanged vertices, 4.9 MB, and points, another 4.9 MB).
    merge = "rows" (round 4): the second collective is a reduce_scatter by PIXEL ROWS instead: rank r receives the summed normals of
    its band of rows only (4.9 MB / world lands on a rank, none of them a hot spot) and finishes the band itself --
    points_fn(keys64, normals_band, row0, nrows) -> points of the band; shade_fn must return a buffer of world * per rows (row_bands),
    band_out an int32-viewable [per, cols, 4] float tensor.  Returns (points_band, normals_band, (row0, nrows)) on EVERY rank: the image
    stays row-sharded for a row-sharded consumer (DESIGN.md section 5 says what that consumer is).
    merge = "a2a" (round 5): the same row bands, but every pixel's normal travels DIRECTLY from the rank that made it to the rank that
    finishes its row: one all-to-all of the band-sized pieces of the (padded) normals image -- each piece crosses one xGMI link once,
    the seven links of a rank in parallel: N - 1 times fewer sequential steps than the ring behind reduce_scatter, the same bytes per
    link -- and the receiver adds the N pieces of its band (integer adds; every summand but one is zero, so the sum is the owner's bits).
    No counts are exchanged: the pieces have a fixed size.  a2a_recv: [world, per, cols, 4] float buffer for the pieces.
    key_merge = "direct" (round 6): the FIRST collective without a ring as well (coll_all_reduce_min_direct): march_fn must return the
    first `rows` rows of keys_pad [world * per, cols] int64 (or anything, which is then copied there); keys_recv [world, per, cols] and
    keys_band [per, cols] are its buffers.  The stage is marked "all_reduce_min" either way.
    collectives: None = only when world > 1; True = also with one rank (an RCCL dry run of the dtypes and ops).
    timer: a StageTimer; the stages march / all_reduce_min / shade / reduce_scatter | all_to_all | reduce / points are marked."""
    on = world > 1 if collectives is None else collectives
    mark = timer.mark if timer is not None else (lambda name: None)
    keys64 = march_fn()
    mark("march")
    if on and key_merge == "direct":
        rows_img = keys64.shape[0]
        assert keys_pad is not None and keys_recv is not None and keys_band is not None and keys_pad.shape[0] >= rows_img
        if keys64.data_ptr() != keys_pad.data_ptr():
            keys_pad[:rows_img].copy_(keys64)
        coll_all_reduce_min_direct(keys_pad, keys_recv, keys_band, group=group)
        keys64 = keys_pad[:rows_img]
        mark("all_reduce_min")
    elif on:
        coll_all_reduce(keys64, dist.ReduceOp.MIN, group=group)
        mark("all_reduce_min")
    normals = shade_fn(keys64)
    mark("shade")
    if merge in ("rows", "a2a"):
        rows = keys64.shape[0]
        per, bands = row_bands(rows, world)
        row0, nrows = bands[rank]
        if on:
            assert normals.shape[0] == world * per and band_out is not None and band_out.shape[0] == per
            if merge == "a2a":
                assert a2a_recv is not None and a2a_recv.shape[0] == world and a2a_recv.shape[1] == per
                coll_all_to_all_rows(normals.view(torch.int32), a2a_recv.view(torch.int32), group=group)
                if a2a_recv.is_cuda:           # dfusion_raycast_sum_pieces: the N pieces of this rank's band added (integer adds: the owner's bits)
                    from . import capi
                    capi.check(capi.lib().dfusion_raycast_sum_pieces(a2a_recv.data_ptr(), world, a2a_recv[0].numel(), band_out.data_ptr(),
                                                                     torch.cuda.current_stream().cuda_stream), "dfusion_raycast_sum_pieces")
                else:
                    torch.sum(a2a_recv.view(torch.int32), dim=0, out=band_out.view(torch.int32))
                mark("all_to_all")
            else:
                coll_reduce_scatter_rows(normals.view(torch.int32), band_out.view(torch.int32), group=group)
                mark("reduce_scatter")
            nb = band_out[:nrows]
        else:
            nb = normals[row0:row0 + nrows]
        out = points_fn(keys64, nb, row0, nrows)
        mark("points")
        return out, nb, (row0, nrows)
    if on:
        coll_reduce(normals.view(torch.int32), dst, dist.ReduceOp.SUM, group=group)
        mark("reduce")
        if rank != dst:
            return None, None
    out = points_fn(keys64, normals)
    mark("points")
    return out, normals
