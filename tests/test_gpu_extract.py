"""surface extraction (fetchCloud / fetchNormals) -- HIP vs oracle.  The reference's output order is
atomics-dependent, so clouds are compared as SETS (lexicographically sorted), bit for bit."""
import numpy as np
import pytest
import torch

import oracle_lib as O
from dynamicfusion_amd import Intr, sharded, synth, upload_u16
from scene import Scene
from test_gpu_parity import MID, SMALL, make_gpu_volume

pytestmark = pytest.mark.gpu


def sort_rows(a):
    a = np.ascontiguousarray(a)
    return a[np.lexsort(a.view(np.uint32).T[::-1])]


def filled(sc, frames=2):
    ref = sc.new_volume()
    for f in range(frames):
        O.integrate(sc.dists[f], ref, sc.ovol(ref), synth.aff12(sc.vol2cam(f)), sc.intr)
    vol = make_gpu_volume(sc)
    vol.upload(ref)
    return vol, ref


@pytest.mark.parametrize("cfg", [SMALL, MID], ids=["64", "128"])
def test_fetch_cloud_and_normals_match_oracle(cfg):
    sc = Scene(cfg, n_frames=2, with_nodes=False)
    vol, ref = filled(sc)
    cloud = vol.fetchCloud()
    normals = vol.fetchNormals(cloud)
    torch.cuda.synchronize()
    rp, n = O.extract_cloud(sc.ovol(ref), synth.aff12(sc.pose), 1 << 22)
    assert n == vol.last_cloud_count_ == cloud.shape[0] and n > 1000
    gp = cloud.cpu().numpy()
    assert np.array_equal(sort_rows(gp).view(np.uint32), sort_rows(rp).view(np.uint32))
    rinv = np.linalg.inv(sc.pose[:3, :3].astype(np.float64)).astype(np.float32)
    rn = O.extract_normals(sc.ovol(ref), synth.aff12(sc.pose), rinv, gp, cfg.gradient_delta_factor)   # same point order as the GPU
    gn = normals.cpu().numpy()
    assert np.array_equal(np.isnan(gn), np.isnan(rn)) and np.isfinite(rn[:, 0]).mean() > 0.5
    assert np.array_equal(gn.view(np.uint32), rn.view(np.uint32))


def test_fetch_cloud_capacity_and_count():
    sc = Scene(SMALL, n_frames=2, with_nodes=False)
    vol, ref = filled(sc)
    small = torch.empty((100, 4), dtype=torch.float32, device="cuda")
    out = vol.fetchCloud(small)
    _, n = O.extract_cloud(sc.ovol(ref), synth.aff12(sc.pose), 1)
    assert out.shape[0] == 100 and vol.last_cloud_count_ == n            # count reports everything found, writes are capped


def test_fetch_cloud_empty_volume():
    sc = Scene(SMALL, n_frames=1, with_nodes=False)
    vol = make_gpu_volume(sc)
    assert vol.fetchCloud().shape[0] == 0


def test_fetch_cloud_slabs_equal_full():
    sc = Scene(MID, n_frames=2, with_nodes=False)
    vol, ref = filled(sc)
    full = sort_rows(vol.fetchCloud().cpu().numpy())
    Z = MID.dims[2]
    parts = []
    for r in range(4):
        z0, zn = sharded.slab_range(Z, r, 4)
        v = make_gpu_volume(sc, slab=(z0, zn, 1))                          # 1 halo plane for the +z neighbour
        v.upload(ref[v.z_store0:v.z_store0 + v.z_store_n])
        parts.append(v.fetchCloud().cpu().numpy())
    assert np.array_equal(sort_rows(np.concatenate(parts)).view(np.uint32), full.view(np.uint32))
