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


def _worker(rank, world, port, recompute_halo):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        O.lib().orc_set_num_threads(2)
        sc = Scene(CFG, n_frames=FRAMES)
        X, Y, Z = CFG.dims
        halo = sharded.halo_planes(sc.trunc, CFG.raycast_step_factor, CFG.gradient_delta_factor, float(sc.vs[2]))
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
            sharded.broadcast_bytes(depth, 0)                                   # rank 0 owns the sensor frame ...
            sharded.broadcast_bytes(dq, 0)                                      # ... and the solver's node transforms
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
                p, n = O.raycast_shade(sc.ovol(vol), synth.aff12(sc.cam2vol(f)), sc.rinv(f), sc.reproj, ts, merged, CFG.cols, CFG.rows,
                                       CFG.gradient_delta_factor, slab=slab)
                return torch.from_numpy(np.stack([p, n]))

            pts, nrm = sharded.raycast_sharded(march, shade, rank, world)
        # every rank checks its slab (own + halo planes) against the unsharded volume; rank 0 checks the merged cast
        full, fp, fn = _unsharded(sc)
        assert np.array_equal(vol, full[lo:hi]), "rank %d: slab (incl. halos) differs from the unsharded volume" % rank
        if rank == 0:
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


def test_single_rank_is_a_no_op_path():
    p = torch.zeros((4, 4, 4))
    out = sharded.raycast_sharded(lambda: None, lambda k: (p, p), 0, 1)
    assert out[0] is p
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
