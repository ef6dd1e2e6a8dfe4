"""Host-side mirror of kfusion::WarpField over the C-ABI (node store + k-NN + DQB on the GPU).

Mirrors /root/reference/kfusion/include/kfusion/warp_field.hpp:41-88 for the hot-path methods:
init / getNodes / KNN / warp / setWarpToLive / buildKDTree (here: the GPU brick index).
energy_data (the Opt / Ceres data term of the reference) is a conjugate-gradient solve on the GPU (dfusion_warp_solve_data_term).
"""
import ctypes as C

import numpy as np
import torch

from . import capi
from .synth import aff12, identity_dq
from .tsdf_volume import _ptr, _stream

F32 = np.float32
KNN_NEIGHBOURS = 8           # warp_field.hpp:10 (compile-time there, runtime here)


class WarpField:
    def __init__(self, k=KNN_NEIGHBOURS, device="cuda", voxel_table=True, weight_table=True, tables_on_demand=True):
        self.k = int(k)
        # fill the per-voxel tables block by block as the sweeps' launch plans first need them (DF_INDEX_TABLES_ON_DEMAND) instead of
        # all at once when the index is built; same results
        self.tables_on_demand = bool(tables_on_demand)
        self.voxel_table = bool(voxel_table)     # cache the per-voxel k-NN in HBM (k*2 B/voxel); False = re-rank per frame
        self.weight_table = bool(weight_table) and self.voxel_table   # also cache the k blend weights (k*4 B/voxel)
        self.device = torch.device(device)
        h = C.c_void_p()
        capi.check(capi.lib().dfusion_warp_create(C.byref(h)), "dfusion_warp_create")
        self.handle = h
        self.warp_to_live_ = np.eye(4, dtype=F32)      # warp_field.cpp:26
        self.M = 0
        self._index_key = None
        self._keep = []

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                capi.lib().dfusion_warp_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ---- WarpField::init(std::vector<Vec3f>) (warp_field.cpp:68-88): identity transforms, dg_w = sigma
    def init(self, vertices, sigma=3.0, transforms=None):
        pos = np.ascontiguousarray(vertices, F32).reshape(-1, 3)
        M = pos.shape[0]
        dq = identity_dq(M) if transforms is None else np.ascontiguousarray(transforms, F32).reshape(M, 8)
        sig = np.full(M, sigma, F32) if np.isscalar(sigma) else np.ascontiguousarray(sigma, F32).reshape(M)
        self.set_nodes(torch.from_numpy(pos).to(self.device), torch.from_numpy(dq).to(self.device),
                       torch.from_numpy(sig).to(self.device))

    def set_nodes(self, pos_dev, dq_dev, sigma_dev):
        self.M = int(pos_dev.shape[0])
        self._keep = [pos_dev, dq_dev, sigma_dev]
        capi.check(capi.lib().dfusion_warp_set_nodes(self.handle, _ptr(pos_dev), _ptr(dq_dev), _ptr(sigma_dev), self.M,
                                                     _stream()), "dfusion_warp_set_nodes")
        self._index_key = None

    def set_transforms(self, dq_dev):
        """What the warp optimiser writes back every frame (CombinedSolver.h:189-197)."""
        capi.check(capi.lib().dfusion_warp_set_transforms(self.handle, _ptr(dq_dev), _stream()),
                   "dfusion_warp_set_transforms")

    def setWarpToLive(self, pose):     # warp_field.cpp:302-305
        self.warp_to_live_ = np.asarray(pose, F32).reshape(4, 4).copy()

    # ---- buildKDTree (warp_field.cpp:275-282) -> exact k-NN brick index for one volume geometry
    def ensure_index(self, volume, k):
        key = (volume.getDims(), tuple(volume.getVoxelSize().tolist()), volume.getPose().tobytes(),
               (volume.z_own0, volume.z_own_n), int(k))
        if self._index_key is not None and self._index_key[:4] == key[:4] and (
                self._index_key[4] == k or (not self.voxel_table and self._index_key[4] >= k)):
            return
        capi.check(capi.lib().dfusion_warp_build_index(self.handle, volume.c_volume(), volume.c_slab(),
                                                       capi.floats(aff12(volume.getPose())), int(k),
                                                       (capi.DF_INDEX_VOXEL_TABLE if self.voxel_table else 0) |
                                                       (capi.DF_INDEX_WEIGHT_TABLE if self.weight_table else 0) |
                                                       (capi.DF_INDEX_TABLES_ON_DEMAND if self.tables_on_demand else 0), _stream()),
                   "dfusion_warp_build_index")
        self._index_key = key

    # ---- WarpField::KNN (warp_field.cpp:247-251), batched
    def KNN(self, queries_dev, k=None):
        k = self.k if k is None else k
        n = int(queries_dev.shape[0])
        idx = torch.empty((n, k), dtype=torch.int32, device=self.device)
        d2 = torch.empty((n, k), dtype=torch.float32, device=self.device)
        capi.check(capi.lib().dfusion_knn(self.handle, k, _ptr(queries_dev), n, _ptr(idx), _ptr(d2), _stream()),
                   "dfusion_knn")
        return idx, d2

    # ---- WarpField::warp (warp_field.cpp:180-195), in place on device [N,3] tensors
    def warp(self, points_dev, normals_dev=None, k=None):
        k = self.k if k is None else k
        n = int(points_dev.shape[0])
        capi.check(capi.lib().dfusion_warp_points(self.handle, k, _ptr(points_dev),
                                                  _ptr(normals_dev) if normals_dev is not None else None, n,
                                                  capi.floats(aff12(self.warp_to_live_)), _stream()),
                   "dfusion_warp_points")

    # ---- WarpField::energy_data (warp_field.cpp:117-163) / WarpFieldOptimiser::optimiseWarpData: the data term, on the GPU

    def debug_counters(self, swept_dev=None):
        """Measurement hook (include/dfusion.h dfusion_warp_debug_counters): while set (device int64[1]), every warped integrate through
        THIS field adds the voxels its launch plan keeps; None switches it off."""
        capi.check(capi.lib().dfusion_warp_debug_counters(self.handle, _ptr(swept_dev) if swept_dev is not None else None),
                   "dfusion_warp_debug_counters")

    def alive_blocks_per_layer(self, volume, layers_dev):
        """layers_dev (device int64 [Z / 8]) += the 8x8x8 blocks the last sweep's verdict pass kept, per 8-plane layer, for the layers
        inside `volume`'s OWN planes (Z-slab re-balancing; include/dfusion.h dfusion_warp_alive_blocks)."""
        capi.check(capi.lib().dfusion_warp_alive_blocks(self.handle, int(volume.z_own0), int(volume.z_own_n), _ptr(layers_dev),
                                                        int(layers_dev.numel()), _stream()), "dfusion_warp_alive_blocks")
        return layers_dev

    def coded_blocks_per_layer(self, volume, layers_dev):
        """The kept blocks among those that had 4-bit neighbour codes (include/dfusion.h dfusion_warp_coded_blocks)."""
        capi.check(capi.lib().dfusion_warp_coded_blocks(self.handle, int(volume.z_own0), int(volume.z_own_n), _ptr(layers_dev),
                                                        int(layers_dev.numel()), _stream()), "dfusion_warp_coded_blocks")

    def set_point_tiling(self, image_cols):
        """Locality hint (include/dfusion.h dfusion_warp_set_point_tiling): point queries are row-major images `image_cols` wide and are
        processed in 8x8 pixel tiles per wave; 0 = off.  Results do not change."""
        capi.check(capi.lib().dfusion_warp_set_point_tiling(self.handle, int(image_cols)), "dfusion_warp_set_point_tiling")

    def energy_data(self, canonical_dev, live_dev, iters=100, lam=0.0, k=None):
        """Least-squares update of the node translations so that canonical + sum_i w_i T_i meets live (device [N,3] tensors).
        Returns (dq [M,8] device tensor of the updated transforms, energy [before, after] device tensor)."""
        k = self.k if k is None else k
        n = int(canonical_dev.shape[0])
        dq = torch.empty((self.M, 8), dtype=torch.float32, device=self.device)
        en = torch.zeros(2, dtype=torch.float32, device=self.device)
        capi.check(capi.lib().dfusion_warp_solve_data_term(self.handle, k, _ptr(canonical_dev), _ptr(live_dev), n, int(iters), float(lam),
                                                           _ptr(dq), _ptr(en), _stream()), "dfusion_warp_solve_data_term")
        return dq, en
