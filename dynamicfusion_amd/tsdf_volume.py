"""Host-side mirror of kfusion::cuda::TsdfVolume over the C-ABI (include/dfusion.h).

Same method names, argument meaning and defaults as
/root/reference/kfusion/include/kfusion/cuda/tsdf_volume.hpp:11-100 and
/root/reference/kfusion/src/tsdf_volume.cpp, so tests read like calls on the reference class.
Device memory, streams and (in sharded.py) collectives come from torch; all compute is in
libdfusion_hip.so.  Poses are 4x4 float32 numpy matrices (cv::Affine3f); images are torch CUDA
tensors: Depth/Dists int16-viewed uint16 [rows, cols], Cloud/Normals float32 [rows, cols, 4].
"""
import ctypes as C

import numpy as np
import torch

from . import capi
from .synth import aff12, affine_inv, affine_mul

F32 = np.float32


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr())


class Intr:
    """kfusion::Intr (kfusion/include/kfusion/types.hpp:20-27)."""

    def __init__(self, fx, fy, cx, cy):
        self.fx, self.fy, self.cx, self.cy = (float(F32(v)) for v in (fx, fy, cx, cy))

    def as_proj(self):      # device::Projector(fx, fy, cx, cy), precomp.cpp:42
        return capi.floats([self.fx, self.fy, self.cx, self.cy])

    def as_reproj(self):    # device::Reprojector: finv = 1.f/f, precomp.cpp:55
        return capi.floats([F32(1) / F32(self.fx), F32(1) / F32(self.fy), self.cx, self.cy])


def upload_u16(arr, device="cuda"):
    """numpy uint16 [rows, cols] -> device tensor (stored as int16; the kernels read the bits)."""
    return torch.from_numpy(np.ascontiguousarray(arr).view(np.int16)).to(device)


def download_u16(t):
    return t.detach().cpu().numpy().view(np.uint16)


def compute_dists(depth, intr, dists=None):
    """kfusion::cuda::computeDists (kfusion/src/imgproc.cpp:87-91)."""
    rows, cols = depth.shape
    if dists is None:
        dists = torch.empty_like(depth)
    capi.check(capi.lib().dfusion_compute_dists(_ptr(depth), cols * 2, _ptr(dists), cols * 2, cols, rows,
                                                intr.as_proj(), _stream()), "dfusion_compute_dists")
    return dists


class TsdfVolume:
    """kfusion::cuda::TsdfVolume.  `slab=(z_own0, z_own_n, halo)` makes this object one Z-slab shard
    (own planes + `halo` planes each side, clipped to the volume) of a larger volume."""

    def __init__(self, dims, device="cuda", slab=None, allocate=True):
        # ctor defaults: tsdf_volume.cpp:7-14
        self.trunc_dist_ = float(F32(0.03))
        self.max_weight_ = 128
        self.size_ = np.full(3, 3.0, F32)
        self.pose_ = np.eye(4, dtype=F32)
        self.gradient_delta_factor_ = 0.75
        self.raycast_step_factor_ = 0.75
        self.device = torch.device(device)
        self.slab_ = slab
        self.data_ = None
        self.allocate_ = allocate      # False: host parameter object only (no device blob; compute calls raise)
        self.create(dims)

    # ---- tsdf_volume.cpp:32-39
    def create(self, dims):
        self.dims_ = tuple(int(d) for d in dims)
        X, Y, Z = self.dims_
        if X % 4:
            raise ValueError("dims[0] must be a multiple of 4 (reference asserts % 32, kinfu.cpp:97)")
        if self.slab_ is None:
            self.z_store0, self.z_store_n, self.z_own0, self.z_own_n = 0, Z, 0, Z
        else:
            z_own0, z_own_n, halo = self.slab_
            lo, hi = max(0, z_own0 - halo), min(Z, z_own0 + z_own_n + halo)
            self.z_store0, self.z_store_n, self.z_own0, self.z_own_n = lo, hi - lo, z_own0, z_own_n
        self.setTruncDist(self.trunc_dist_)
        if self.allocate_:
            self.data_ = torch.empty((self.z_store_n, Y, X), dtype=torch.int32, device=self.device)
            self.clear()

    def owning_stored_planes(self):
        """A second handle on the SAME device blob whose `own` range is the whole stored range (own planes + halos): integrating
        through it makes a Z-slab shard compute its halo planes itself instead of receiving them from its neighbours."""
        import copy
        v = copy.copy(self)
        v.slab_ = (self.z_store0, self.z_store_n, 0)
        v.z_own0, v.z_own_n = self.z_store0, self.z_store_n
        return v

    def getDims(self):
        return self.dims_

    def getVoxelSize(self):       # tsdf_volume.cpp:54-57 (float division)
        return np.array([F32(self.size_[i]) / F32(self.dims_[i]) for i in range(3)], F32)

    def data(self):
        return self.data_

    def getSize(self):
        return self.size_.copy()

    def setSize(self, size):      # tsdf_volume.cpp:63-64
        self.size_ = np.asarray(size, F32).reshape(3).copy()
        self.setTruncDist(self.trunc_dist_)

    def getTruncDist(self):
        return self.trunc_dist_

    def setTruncDist(self, distance):   # tsdf_volume.cpp:68-73 (clamp to >= 2.1 * max voxel edge)
        vsz = self.getVoxelSize()
        max_coeff = max(vsz[0], vsz[1], vsz[2])
        self.trunc_dist_ = float(max(F32(distance), F32(2.1) * F32(max_coeff)))

    def getMaxWeight(self):
        return int(self.max_weight_)

    def setMaxWeight(self, weight):
        self.max_weight_ = weight

    def getPose(self):
        return self.pose_.copy()

    def setPose(self, pose):
        self.pose_ = np.asarray(pose, F32).reshape(4, 4).copy()

    def getRaycastStepFactor(self):
        return self.raycast_step_factor_

    def setRaycastStepFactor(self, factor):
        self.raycast_step_factor_ = float(factor)

    def getGradientDeltaFactor(self):
        return self.gradient_delta_factor_

    def setGradientDeltaFactor(self, factor):
        self.gradient_delta_factor_ = float(factor)

    def swap(self, data):         # tsdf_volume.cpp:87
        self.data_, other = data, self.data_
        return other

    def applyAffine(self, affine):   # tsdf_volume.cpp:88
        self.pose_ = affine_mul(np.asarray(affine, F32), self.pose_)

    # ---- C-ABI views
    def c_volume(self):
        if self.data_ is None:
            raise capi.DfusionError("TsdfVolume has no device blob (allocate=False)")
        v = capi.DfVolume()
        v.data = self.data_.data_ptr()
        v.dims[:] = self.dims_
        v.voxel_size[:] = [float(x) for x in self.getVoxelSize()]
        v.trunc_dist = self.trunc_dist_
        v.max_weight = int(self.max_weight_)
        return v

    def c_slab(self):
        if self.slab_ is None:
            return None
        return C.byref(capi.DfSlab(self.z_store0, self.z_store_n, self.z_own0, self.z_own_n))

    # ---- tsdf_volume.cpp:89-102
    def clear(self):
        capi.check(capi.lib().dfusion_clear(self.c_volume(), self.c_slab(), _stream()), "dfusion_clear")

    # ---- tsdf_volume.cpp:110-122
    def integrate(self, dists, camera_pose, intr, n_updated=None, sync=True, flags=0, n_swept=None):
        """flags: capi.DF_RIGID_* validation switches of THIS call (0 = the product path); n_swept: device int64[1], += the voxels
        the sweep put through the projective sample."""
        vol2cam = affine_mul(affine_inv(np.asarray(camera_pose, F32)), self.pose_)
        rows, cols = dists.shape
        if flags == 0 and n_swept is None:
            capi.check(capi.lib().dfusion_integrate(_ptr(dists), cols * 2, cols, rows, self.c_volume(), self.c_slab(),
                                                    capi.floats(aff12(vol2cam)), intr.as_proj(),
                                                    _ptr(n_updated) if n_updated is not None else None, _stream()),
                       "dfusion_integrate")
        else:
            capi.check(capi.lib().dfusion_integrate_ex(_ptr(dists), cols * 2, cols, rows, self.c_volume(), self.c_slab(),
                                                       capi.floats(aff12(vol2cam)), intr.as_proj(), int(flags),
                                                       _ptr(n_updated) if n_updated is not None else None,
                                                       _ptr(n_swept) if n_swept is not None else None, _stream()),
                       "dfusion_integrate_ex")
        if sync:                   # device::integrate ends with cudaDeviceSynchronize (tsdf_volume.cu:160)
            torch.cuda.current_stream().synchronize()

    # ---- the north-star composition; no reference method of this name exists
    def integrate_warped(self, dists, camera_pose, intr, warp_field, k=None, n_updated=None, cull=True, sync=True,
                         use_table=True, use_weights=True, use_lds=True, pipelined=True, zero_skip=True, depth_pyramid=True, block_model=True,
                         prefetch=True, codes=True):
        """block_model: True = the library's policy (models built the second time a weight table is swept), "now" = at the first
        sweep, False = never (DF_WARP_NO_BLOCK_MODEL).  prefetch: True = the library's policy (look-ahead builds on the handle's side
        stream, switched off on a scene at rest from an unsynchronised report), False = DF_WARP_NO_PREFETCH, "steady" = on in every
        frame (DF_WARP_STEADY_PREFETCH: reproducible swept-voxel counters).  codes: False = DF_WARP_NO_CODES (modelled blocks read the full
        16-B neighbour record instead of their 4-bit codes)."""
        k = warp_field.k if k is None else k
        world2cam = affine_mul(affine_inv(np.asarray(camera_pose, F32)), warp_field.warp_to_live_)
        warp_field.ensure_index(self, k)
        rows, cols = dists.shape
        capi.check(capi.lib().dfusion_integrate_warped(
            _ptr(dists), cols * 2, cols, rows, self.c_volume(), self.c_slab(), capi.floats(aff12(self.pose_)),
            capi.floats(aff12(world2cam)), intr.as_proj(), warp_field.handle, k,
            (0 if cull else capi.DF_WARP_NO_CULL) | (0 if use_table else capi.DF_WARP_NO_TABLE) |
            (0 if use_weights else capi.DF_WARP_NO_WEIGHT_TABLE) | (0 if use_lds else capi.DF_WARP_NO_LDS) |
            (0 if pipelined else capi.DF_WARP_NO_PIPELINE) | (0 if zero_skip else capi.DF_WARP_NO_ZERO_SKIP) |
            (0 if depth_pyramid else capi.DF_WARP_NO_DEPTH_PYRAMID) | (capi.DF_WARP_STEADY_PREFETCH if prefetch == "steady" else 0 if prefetch else capi.DF_WARP_NO_PREFETCH) |
            (0 if codes else capi.DF_WARP_NO_CODES) |
            (capi.DF_WARP_BLOCK_MODEL_NOW if block_model == "now" else 0 if block_model else capi.DF_WARP_NO_BLOCK_MODEL),
            _ptr(n_updated) if n_updated is not None else None, _stream()), "dfusion_integrate_warped")
        if sync:
            torch.cuda.current_stream().synchronize()

    # ---- the same frame in two calls (include/dfusion.h dfusion_integrate_warped_prepare / _sweep): `prepare` does everything that does
    # not touch the volume and may run on another stream (beside the previous frame's ray-cast); `sweep` waits for it on the device
    def integrate_warped_prepare(self, dists, camera_pose, intr, warp_field, k=None, prefetch=True, block_model=True, codes=True):
        k = warp_field.k if k is None else k
        world2cam = affine_mul(affine_inv(np.asarray(camera_pose, F32)), warp_field.warp_to_live_)
        warp_field.ensure_index(self, k)
        rows, cols = dists.shape
        capi.check(capi.lib().dfusion_integrate_warped_prepare(
            _ptr(dists), cols * 2, cols, rows, self.c_volume(), self.c_slab(), capi.floats(aff12(self.pose_)),
            capi.floats(aff12(world2cam)), intr.as_proj(), warp_field.handle, k,
            (capi.DF_WARP_STEADY_PREFETCH if prefetch == "steady" else 0 if prefetch else capi.DF_WARP_NO_PREFETCH) |
            (0 if codes else capi.DF_WARP_NO_CODES) |
            (capi.DF_WARP_BLOCK_MODEL_NOW if block_model == "now" else 0 if block_model else capi.DF_WARP_NO_BLOCK_MODEL),
            _stream()), "dfusion_integrate_warped_prepare")

    def integrate_warped_sweep(self, warp_field, n_updated=None, sync=False):
        capi.check(capi.lib().dfusion_integrate_warped_sweep(self.c_volume(), self.c_slab(), warp_field.handle,
                                                             _ptr(n_updated) if n_updated is not None else None, _stream()),
                   "dfusion_integrate_warped_sweep")
        if sync:
            torch.cuda.current_stream().synchronize()

    def _raycast_args(self, camera_pose):
        cam2vol = affine_mul(affine_inv(self.pose_), np.asarray(camera_pose, F32))      # tsdf_volume.cpp:162
        Rinv = np.linalg.inv(cam2vol[:3, :3].astype(np.float64)).astype(F32)            # :165 inv(DECOMP_SVD)
        return capi.floats(aff12(cam2vol)), capi.floats(Rinv.reshape(-1))

    # ---- tsdf_volume.cpp:131-174 : Cloud variant if `points` is float [rows, cols, 4], Depth variant if
    # it is a 16-bit [rows, cols] image
    def raycast(self, camera_pose, intr, points, normals, keys=None):
        aff, Rinv = self._raycast_args(camera_pose)
        rows, cols = normals.shape[0], normals.shape[1]
        L = capi.lib()
        if points.dtype == torch.float32:
            capi.check(L.dfusion_raycast_points(self.c_volume(), self.c_slab(), aff, Rinv, intr.as_reproj(), _ptr(points),
                                                cols * 16, _ptr(normals), cols * 16, cols, rows,
                                                self.raycast_step_factor_, self.gradient_delta_factor_,
                                                _ptr(keys) if keys is not None else None, _stream()),
                       "dfusion_raycast_points")
        else:
            capi.check(L.dfusion_raycast_depth(self.c_volume(), self.c_slab(), aff, Rinv, intr.as_reproj(), _ptr(points),
                                               cols * 2, _ptr(normals), cols * 16, cols, rows,
                                               self.raycast_step_factor_, self.gradient_delta_factor_, _stream()),
                       "dfusion_raycast_depth")

    # ---- Z-slab two-stage cast (dfusion_raycast_march / _shade; no reference counterpart)
    def raycast_march(self, camera_pose, intr, keys64, rank=0):
        """keys64: int64 [rows, cols] <- merge key [step k : 23 | hit : 1 | rank : 7 | Ts bits : 32] (include/dfusion.h), DF_RC_KEY_NONE
        where this slab saw no event: a MIN over ranks is the whole merge."""
        aff, _ = self._raycast_args(camera_pose)
        rows, cols = keys64.shape
        capi.check(capi.lib().dfusion_raycast_march(self.c_volume(), self.c_slab(), aff, intr.as_reproj(), cols, rows,
                                                    self.raycast_step_factor_, int(rank), _ptr(keys64), _stream()),
                   "dfusion_raycast_march")
        return keys64

    def raycast_shade(self, camera_pose, intr, merged_keys64, points, normals):
        """points may be None: only the normals cross GPUs, the points follow from the merged keys (raycast_points_of_keys)."""
        aff, Rinv = self._raycast_args(camera_pose)
        rows, cols = merged_keys64.shape
        capi.check(capi.lib().dfusion_raycast_shade(self.c_volume(), self.c_slab(), aff, Rinv, intr.as_reproj(),
                                                    _ptr(merged_keys64), _ptr(points) if points is not None else None, cols * 16,
                                                    _ptr(normals), cols * 16, cols, rows, self.gradient_delta_factor_,
                                                    _stream()), "dfusion_raycast_shade")
        return points, normals

    def raycast_points_of_keys(self, camera_pose, intr, merged_keys64, normals, points, row0=0, nrows=None):
        """Stage 3 of the sharded cast, on the rank that wants the image: points from the merged keys (Ts) and the summed normals.
        row0 / nrows: only the band of pixel rows [row0, row0 + nrows) -- `normals` / `points` are then the BAND's tensors
        ([nrows, cols, 4]), merged_keys64 the whole image's."""
        aff, Rinv = self._raycast_args(camera_pose)
        rows, cols = merged_keys64.shape
        nrows = rows - row0 if nrows is None else nrows
        capi.check(capi.lib().dfusion_raycast_points_of_keys_rows(aff, Rinv, intr.as_reproj(), _ptr(merged_keys64), _ptr(normals), cols * 16,
                                                                  _ptr(points), cols * 16, cols, rows, int(row0), int(nrows), _stream()),
                   "dfusion_raycast_points_of_keys_rows")
        return points

    # ---- tsdf_volume.cpp:181-218 fetchCloud / fetchNormals (device tensors; count read back like the reference does)
    def fetchCloud(self, cloud_buffer=None):
        """cloud_buffer: float32 [capacity, 4] device tensor (default capacity 256^3, tsdf_volume.cpp:184, capped by the
        voxel count).  Returns a view of the first min(found, capacity) points."""
        if cloud_buffer is None:
            cap = min(256 * 256 * 256, int(np.prod(self.dims_)))
            cloud_buffer = torch.empty((cap, 4), dtype=torch.float32, device=self.device)
        count = torch.zeros(1, dtype=torch.int64, device=self.device)
        capi.check(capi.lib().dfusion_extract_cloud(self.c_volume(), self.c_slab(), capi.floats(aff12(self.pose_)),
                                                    _ptr(cloud_buffer), cloud_buffer.shape[0], _ptr(count), _stream()),
                   "dfusion_extract_cloud")
        n = int(count.item())                               # cudaMemcpyFromSymbol(output_count), tsdf_volume.cu:815
        self.last_cloud_count_ = n
        return cloud_buffer[:min(n, cloud_buffer.shape[0])]

    def fetchNormals(self, cloud, normals=None):
        n = int(cloud.shape[0])
        if normals is None:
            normals = torch.empty((n, 4), dtype=torch.float32, device=self.device)
        Rinv = np.linalg.inv(self.pose_[:3, :3].astype(np.float64)).astype(F32)        # tsdf_volume.cpp:214
        capi.check(capi.lib().dfusion_extract_normals(self.c_volume(), self.c_slab(), capi.floats(aff12(self.pose_)),
                                                      capi.floats(Rinv.reshape(-1)), _ptr(cloud), n,
                                                      self.gradient_delta_factor_, _ptr(normals), _stream()),
                   "dfusion_extract_normals")
        return normals

    # ---- tsdf_volume.cpp:266-292 psdf (device::project_and_remove + the per-point K^-1 arithmetic, fused on the GPU)
    def psdf(self, warped, dists, intr, return_points=False):
        """warped: float32 [n, 3] or [n, 4] device tensor of camera-frame points; dists: u16 [rows, cols] device tensor,
        pixels hit by a warped point are zeroed IN PLACE (sampled from a snapshot taken first, see include/dfusion.h).
        Returns ro [n] (device), optionally also the projected points [n, 4]."""
        n = int(warped.shape[0])
        pts = torch.zeros((n, 4), dtype=torch.float32, device=self.device)
        pts[:, :3] = warped[:, :3]
        ro = torch.empty(n, dtype=torch.float32, device=self.device)
        snapshot = dists.clone()
        rows, cols = dists.shape
        capi.check(capi.lib().dfusion_project_and_remove(_ptr(snapshot), cols * 2, _ptr(dists), cols * 2, cols, rows,
                                                         _ptr(pts), n, intr.as_proj(), _ptr(ro), None, _stream()),
                   "dfusion_project_and_remove")
        return (ro, pts) if return_points else ro

    # ---- convenience for tests
    def download(self):
        """uint32 numpy [z_store_n, Y, X] : lo16 = half tsdf bits, hi16 = weight."""
        return self.data_.detach().cpu().numpy().view(np.uint32)

    def upload(self, arr_u32):
        self.data_.copy_(torch.from_numpy(np.ascontiguousarray(arr_u32).view(np.int32)).reshape(self.data_.shape))
