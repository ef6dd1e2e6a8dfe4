"""ctypes bindings for the parity checkers in oracle/ (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product package (dynamicfusion_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
u16p = np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")


class Volume(C.Structure):
    """device::TsdfVolume, kfusion/src/internal.hpp:29-49 (same layout as DfVolume)."""
    _fields_ = [("data", C.c_void_p), ("dims", C.c_int * 3), ("voxel_size", C.c_float * 3),
                ("trunc_dist", C.c_float), ("max_weight", C.c_int)]


class Slab(C.Structure):
    _fields_ = [("z_store0", C.c_int), ("z_store_n", C.c_int), ("z_own0", C.c_int), ("z_own_n", C.c_int)]


def build(force=False):
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    src = os.path.join(ORACLE_DIR, "dfusion_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    ref = os.path.join(ORACLE_DIR, "_ref", "libdfref.so")
    glue = os.path.join(ORACLE_DIR, "ref_glue.cpp")
    if os.path.isdir("/root/reference/kfusion/src/utils") and (
            force or not os.path.exists(ref) or os.path.getmtime(ref) < os.path.getmtime(glue)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)


    refcu = os.path.join(ORACLE_DIR, "_ref", "libdfref_cu.so")
    deps = [os.path.join(ORACLE_DIR, f) for f in ("ref_cu_glue.cpp", "gen_ref_cu.sh", "cuda_shim/cuda_runtime_api.h",
                                                   "cuda_shim/cuda_shim_runtime.cpp")]
    if os.path.isdir("/root/reference/kfusion/src/cuda") and (
            force or not os.path.exists(refcu) or os.path.getmtime(refcu) < max(os.path.getmtime(d) for d in deps)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "ref_cu"], stdout=subprocess.DEVNULL)


_lib = None
_ref = None
_refcu = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(os.path.join(ORACLE_DIR, "liboracle.so"))
        L.orc_float2half.restype = C.c_uint16
        L.orc_float2half.argtypes = [C.c_float]
        L.orc_half2float.restype = C.c_float
        L.orc_half2float.argtypes = [C.c_uint16]
        L.orc_compute_dists.restype = None
        L.orc_compute_dists.argtypes = [u16p, C.c_size_t, u16p, C.c_size_t, C.c_int, C.c_int, f32p]
        L.orc_clear.restype = None
        L.orc_clear.argtypes = [Volume, C.POINTER(Slab)]
        L.orc_integrate.restype = C.c_uint64
        L.orc_integrate.argtypes = [u16p, C.c_size_t, C.c_int, C.c_int, Volume, C.POINTER(Slab), f32p, f32p]
        L.orc_integrate_warped.restype = C.c_uint64
        L.orc_integrate_warped.argtypes = [u16p, C.c_size_t, C.c_int, C.c_int, Volume, C.POINTER(Slab), f32p, f32p,
                                           f32p, f32p, f32p, f32p, C.c_int, C.c_int]
        L.orc_knn.restype = None
        L.orc_knn.argtypes = [f32p, C.c_int, f32p, C.c_int, C.c_int, i32p, f32p]
        L.orc_knn_brute.restype = None
        L.orc_knn_brute.argtypes = [f32p, C.c_int, f32p, C.c_int, C.c_int, i32p, f32p]
        L.orc_nanoflann_tree_info.restype = None
        L.orc_nanoflann_tree_info.argtypes = [f32p, C.c_int, i32p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_dqb.restype = None
        L.orc_dqb.argtypes = [f32p, f32p, f32p, C.c_int, C.c_int, f32p, C.c_int, f32p]
        L.orc_warp_points.restype = None
        L.orc_warp_points.argtypes = [f32p, f32p, f32p, C.c_int, C.c_int, f32p, C.c_void_p, C.c_int, f32p]
        L.orc_raycast_points.restype = None
        L.orc_raycast_points.argtypes = [Volume, C.POINTER(Slab), f32p, f32p, f32p, f32p, C.c_size_t, f32p,
                                         C.c_size_t, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.orc_raycast_depth.restype = None
        L.orc_raycast_depth.argtypes = [Volume, C.POINTER(Slab), f32p, f32p, f32p, u16p, C.c_size_t, f32p,
                                        C.c_size_t, C.c_int, C.c_int, C.c_float, C.c_float]
        L.orc_raycast_march.restype = None
        L.orc_raycast_march.argtypes = [Volume, C.POINTER(Slab), f32p, f32p, C.c_int, C.c_int, C.c_float, u32p, f32p]
        L.orc_raycast_points_of_keys.restype = None
        L.orc_raycast_points_of_keys.argtypes = [f32p, f32p, f32p, f32p, u32p, f32p, C.c_size_t, f32p, C.c_size_t, C.c_int, C.c_int]
        L.orc_raycast_shade.restype = None
        L.orc_raycast_shade.argtypes = [Volume, C.POINTER(Slab), f32p, f32p, f32p, f32p, u32p, f32p, C.c_size_t, f32p, C.c_size_t,
                                        C.c_int, C.c_int, C.c_float]
        L.orc_extract_cloud.restype = C.c_uint64
        L.orc_extract_cloud.argtypes = [Volume, C.POINTER(Slab), f32p, f32p, C.c_uint64]
        L.orc_project_and_remove.restype = C.c_uint64
        L.orc_project_and_remove.argtypes = [u16p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, f32p, C.c_uint64, f32p, C.c_void_p]
        sz = C.c_size_t
        L.orc_bilateral.restype = None
        L.orc_bilateral.argtypes = [u16p, sz, u16p, sz, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]
        L.orc_truncate_depth.restype = None
        L.orc_truncate_depth.argtypes = [u16p, sz, C.c_int, C.c_int, C.c_float]
        L.orc_cloud_to_depth.restype = None
        L.orc_cloud_to_depth.argtypes = [f32p, sz, u16p, sz, C.c_int, C.c_int]
        L.orc_depth_pyramid.restype = None
        L.orc_depth_pyramid.argtypes = [u16p, sz, C.c_int, C.c_int, u16p, sz, C.c_float]
        L.orc_compute_normals_mask_depth.restype = None
        L.orc_compute_normals_mask_depth.argtypes = [u16p, sz, f32p, sz, C.c_int, C.c_int, f32p]
        L.orc_compute_point_normals.restype = None
        L.orc_compute_point_normals.argtypes = [u16p, sz, f32p, sz, f32p, sz, C.c_int, C.c_int, f32p]
        L.orc_resize_depth_normals.restype = None
        L.orc_resize_depth_normals.argtypes = [u16p, sz, f32p, sz, C.c_int, C.c_int, u16p, sz, f32p, sz]
        L.orc_resize_points_normals.restype = None
        L.orc_resize_points_normals.argtypes = [f32p, sz, f32p, sz, C.c_int, C.c_int, f32p, sz, f32p, sz]
        u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
        L.orc_render_points.restype = None
        L.orc_render_points.argtypes = [f32p, sz, f32p, sz, C.c_int, C.c_int, f32p, u8p, sz]
        L.orc_render_depth.restype = None
        L.orc_render_depth.argtypes = [u16p, sz, f32p, sz, C.c_int, C.c_int, f32p, f32p, u8p, sz]
        L.orc_render_tangent_colors.restype = None
        L.orc_render_tangent_colors.argtypes = [f32p, sz, C.c_int, C.c_int, u8p, sz]
        L.orc_icp_sums_points.restype = None
        L.orc_icp_sums_points.argtypes = [f32p, sz, f32p, sz, f32p, sz, f32p, sz, C.c_int, C.c_int, f32p, f32p, C.c_float, C.c_float,
                                          f32p, C.POINTER(C.c_int)]
        L.orc_icp_sums_depth.restype = None
        L.orc_icp_sums_depth.argtypes = [u16p, sz, f32p, sz, u16p, sz, f32p, sz, C.c_int, C.c_int, f32p, f32p, C.c_float, C.c_float,
                                         f32p, C.POINTER(C.c_int)]
        L.orc_solve_data_term.restype = None
        L.orc_solve_data_term.argtypes = [f32p, f32p, f32p, C.c_int, C.c_int, f32p, f32p, C.c_int, C.c_int, C.c_float, f32p, f32p]
        L.orc_extract_normals.restype = None
        L.orc_extract_normals.argtypes = [Volume, C.POINTER(Slab), f32p, f32p, f32p, C.c_uint64, C.c_float, f32p]
        for name in ("orc_quat_mul",):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [f32p, f32p, f32p]
        L.orc_quat_normalize.restype = None
        L.orc_quat_normalize.argtypes = [f32p, f32p]
        L.orc_quat_encode_rotation.restype = None
        L.orc_quat_encode_rotation.argtypes = [C.c_float] * 4 + [f32p]
        L.orc_quat_rotate_xyz.restype = None
        L.orc_quat_rotate_xyz.argtypes = [f32p, f32p]
        L.orc_node_translation.restype = None
        L.orc_node_translation.argtypes = [f32p, f32p]
        L.orc_dq_from_twist.restype = None
        L.orc_dq_from_twist.argtypes = [f32p, f32p, f32p]
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def have_ref():
    build()
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libdfref.so"))


def ref():
    """The reference's own headers compiled here (oracle/_ref/libdfref.so)."""
    global _ref
    if _ref is None:
        build()
        R = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libdfref.so"))
        R.ref_knn.restype = None
        R.ref_knn.argtypes = [f32p, C.c_int, f32p, C.c_int, C.c_int, i32p, f32p]
        R.ref_dqb.restype = None
        R.ref_dqb.argtypes = [f32p, f32p, f32p, C.c_int, C.c_int, f32p, C.c_int, f32p]
        R.ref_warp_points.restype = None
        R.ref_warp_points.argtypes = [f32p, f32p, f32p, C.c_int, C.c_int, f32p, C.c_void_p, C.c_int]
        R.ref_quat_encode_rotation.restype = None
        R.ref_quat_encode_rotation.argtypes = [C.c_float] * 4 + [f32p]
        R.ref_quat_rotate_xyz.restype = None
        R.ref_quat_rotate_xyz.argtypes = [f32p, f32p]
        R.ref_quat_mul.restype = None
        R.ref_quat_mul.argtypes = [f32p, f32p, f32p]
        R.ref_quat_dot.restype = C.c_float
        R.ref_quat_dot.argtypes = [f32p, f32p]
        R.ref_quat_normalize.restype = None
        R.ref_quat_normalize.argtypes = [f32p, f32p]
        R.ref_dq_euler.restype = None
        R.ref_dq_euler.argtypes = [C.c_float] * 6 + [f32p, f32p]
        R.ref_dq_from_twist.restype = None
        R.ref_dq_from_twist.argtypes = [f32p, f32p, f32p]
        R.ref_dq_get_translation.restype = None
        R.ref_dq_get_translation.argtypes = [f32p, f32p]
        R.ref_dq_transform.restype = None
        R.ref_dq_transform.argtypes = [f32p, f32p]
        R.ref_nanoflann_version.restype = C.c_int
        R.ref_integrate_warped.restype = C.c_uint64
        R.ref_integrate_warped.argtypes = [u16p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p,
                                           C.c_float, C.c_int, f32p, f32p, f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int,
                                           C.POINTER(C.c_int)]
        _ref = R
    return _ref


def have_refcu():
    build()
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libdfref_cu.so"))


def refcu():
    """The reference's own CUDA kernels (tsdf_volume.cu, imgproc.cu, proj_icp.cu) compiled for the host through
    oracle/cuda_shim (oracle/_ref/libdfref_cu.so, `make -C oracle ref_cu`)."""
    global _refcu
    if _refcu is None:
        build()
        R = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libdfref_cu.so"))
        u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
        i, f = C.c_int, C.c_float
        R.refcu_clear.argtypes = [Volume]
        R.refcu_integrate.argtypes = [u16p, i, i, Volume, f32p, f32p]
        R.refcu_raycast_points.argtypes = [Volume, f32p, f32p, f32p, i, i, f, f, f32p, f32p]
        R.refcu_raycast_depth.argtypes = [Volume, f32p, f32p, f32p, i, i, f, f, u16p, f32p]
        R.refcu_project_and_remove.argtypes = [u16p, i, i, f32p, i, i, f32p]
        R.refcu_extract_normals.argtypes = [Volume, f32p, f32p, f32p, C.c_uint64, f, f32p]
        R.refcu_extract_cloud.argtypes = [Volume, f32p, f32p, C.c_uint64]
        R.refcu_extract_cloud.restype = C.c_uint64
        R.refcu_compute_dists.argtypes = [u16p, i, i, f32p, u16p]
        R.refcu_bilateral.argtypes = [u16p, i, i, i, f, f, u16p]
        R.refcu_truncate_depth.argtypes = [u16p, i, i, f]
        R.refcu_depth_pyramid.argtypes = [u16p, i, i, f, u16p]
        R.refcu_cloud_to_depth.argtypes = [f32p, i, i, u16p]
        R.refcu_compute_normals_mask_depth.argtypes = [u16p, i, i, f32p, f32p]
        R.refcu_compute_point_normals.argtypes = [u16p, i, i, f32p, f32p, f32p]
        R.refcu_resize_depth_normals.argtypes = [u16p, f32p, i, i, u16p, f32p]
        R.refcu_resize_points_normals.argtypes = [f32p, f32p, i, i, f32p, f32p]
        R.refcu_render_points.argtypes = [f32p, f32p, i, i, f32p, f32p, u8p]
        R.refcu_render_depth.argtypes = [u16p, f32p, i, i, f32p, f32p, u8p]
        R.refcu_render_tangent_colors.argtypes = [f32p, i, i, u8p]
        R.refcu_icp_sums_points.argtypes = [f32p, f32p, f32p, f32p, i, i, f32p, f32p, f, f, f32p]
        for n in dir(R):
            pass
        for name in ("refcu_clear", "refcu_integrate", "refcu_raycast_points", "refcu_raycast_depth", "refcu_project_and_remove",
                     "refcu_extract_normals", "refcu_compute_dists", "refcu_bilateral", "refcu_truncate_depth", "refcu_depth_pyramid", "refcu_cloud_to_depth",
                     "refcu_compute_normals_mask_depth", "refcu_compute_point_normals", "refcu_resize_depth_normals",
                     "refcu_resize_points_normals", "refcu_render_points", "refcu_render_depth", "refcu_render_tangent_colors",
                     "refcu_icp_sums_points"):
            getattr(R, name).restype = None
        _refcu = R
    return _refcu


# ---- the reference's own kernels, numpy level (same argument conventions as the oracle helpers below) ---------------
def refcu_compute_dists(depth_u16, intr):
    d = _u16(depth_u16); rows, cols = d.shape
    out = np.zeros_like(d)
    refcu().refcu_compute_dists(d, rows, cols, f32(intr), out)
    return out


def refcu_integrate(dists, volume, vol2cam, intr):
    rows, cols = dists.shape
    refcu().refcu_integrate(_u16(dists), rows, cols, volume, f32(vol2cam).reshape(-1), f32(intr))


def refcu_raycast_points(volume, cam2vol, Rinv, intr, cols, rows, step_factor, delta_factor):
    pts = np.empty((rows, cols, 4), np.float32); nrm = np.empty((rows, cols, 4), np.float32)
    refcu().refcu_raycast_points(volume, f32(cam2vol).reshape(-1), f32(Rinv).reshape(-1), f32(intr), rows, cols, step_factor, delta_factor,
                                 pts.reshape(-1), nrm.reshape(-1))
    return pts, nrm


def refcu_raycast_depth(volume, cam2vol, Rinv, intr, cols, rows, step_factor, delta_factor):
    dep = np.empty((rows, cols), np.uint16); nrm = np.empty((rows, cols, 4), np.float32)
    refcu().refcu_raycast_depth(volume, f32(cam2vol).reshape(-1), f32(Rinv).reshape(-1), f32(intr), rows, cols, step_factor, delta_factor,
                                dep, nrm.reshape(-1))
    return dep, nrm


def refcu_extract_cloud(volume, aff, capacity):
    pts = np.zeros((capacity, 4), np.float32)
    n = int(refcu().refcu_extract_cloud(volume, f32(aff).reshape(-1), pts.reshape(-1), capacity))
    return pts[:min(n, capacity)], n


def refcu_extract_normals(volume, aff, Rinv, points, gradient_delta_factor):
    points = np.ascontiguousarray(points, np.float32)
    out = np.empty_like(points)
    refcu().refcu_extract_normals(volume, f32(aff).reshape(-1), f32(Rinv).reshape(-1), points.reshape(-1), points.shape[0],
                                  gradient_delta_factor, out.reshape(-1))
    return out


# ------------------------------------------------------------------ numpy-level helpers
def make_volume(vol_u32, dims, voxel_size, trunc_dist, max_weight):
    """vol_u32: C-contiguous uint32 array [nz_store, Y, X] (lo16 = half tsdf, hi16 = weight)."""
    v = Volume()
    v.data = vol_u32.ctypes.data
    v.dims[:] = [int(d) for d in dims]
    v.voxel_size[:] = [float(np.float32(s)) for s in voxel_size]
    v.trunc_dist = float(np.float32(trunc_dist))
    v.max_weight = int(max_weight)
    return v


def make_slab(z_store0, z_store_n, z_own0, z_own_n):
    return Slab(int(z_store0), int(z_store_n), int(z_own0), int(z_own_n))


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def compute_dists(depth_u16, intr):
    rows, cols = depth_u16.shape
    out = np.zeros_like(depth_u16)
    lib().orc_compute_dists(np.ascontiguousarray(depth_u16), cols * 2, out, cols * 2, cols, rows, f32(intr))
    return out


def integrate(dists, vol_u32, volume, vol2cam, proj, slab=None):
    rows, cols = dists.shape
    return int(lib().orc_integrate(dists, cols * 2, cols, rows, volume, C.byref(slab) if slab else None,
                                   f32(vol2cam).reshape(-1), f32(proj)))


def integrate_warped(dists, vol_u32, volume, vol2world, world2cam, proj, pos, dq, sigma, k, slab=None):
    rows, cols = dists.shape
    pos, dq, sigma = f32(pos), f32(dq), f32(sigma)
    return int(lib().orc_integrate_warped(dists, cols * 2, cols, rows, volume, C.byref(slab) if slab else None,
                                          f32(vol2world).reshape(-1), f32(world2cam).reshape(-1), f32(proj),
                                          pos.reshape(-1), dq.reshape(-1), sigma, pos.shape[0], k))


def ref_integrate_warped(dists, vol_u32, dims, vs, trunc, max_weight, vol2world, world2cam, proj, pos, dq, sigma, k, z_store0, z0, zn,
                         threads=0):
    """Per-voxel warped integrate through the reference's own nanoflann / DQB classes (oracle/_ref), OpenMP with one tree per
    thread.  vol_u32: [z_store_n, Y, X] blob starting at plane z_store0.  Returns (n_updated, threads_used)."""
    rows, cols = dists.shape
    used = C.c_int(0)
    n = ref().ref_integrate_warped(np.ascontiguousarray(dists), cols * 2, cols, rows, vol_u32.ctypes.data, int(dims[0]), int(dims[1]),
                                   int(z_store0), int(z0), int(zn), f32(vs), float(np.float32(trunc)), int(max_weight),
                                   f32(vol2world).reshape(-1), f32(world2cam).reshape(-1), f32(proj), f32(pos).reshape(-1),
                                   f32(dq).reshape(-1), f32(sigma), int(len(pos)), int(k), int(threads), C.byref(used))
    return int(n), used.value


def raycast_points(volume, cam2vol, Rinv, reproj, cols, rows, step_factor, delta_factor, slab=None, want_keys=False):
    pts = np.empty((rows, cols, 4), np.float32)
    nrm = np.empty((rows, cols, 4), np.float32)
    keys = np.empty((rows, cols), np.uint32) if want_keys else None
    stats = np.zeros(2, np.uint64)
    lib().orc_raycast_points(volume, C.byref(slab) if slab else None, f32(cam2vol).reshape(-1), f32(Rinv).reshape(-1),
                             f32(reproj), pts.reshape(-1), cols * 16, nrm.reshape(-1), cols * 16, cols, rows,
                             step_factor, delta_factor, keys.ctypes.data if want_keys else None, stats.ctypes.data)
    return pts, nrm, keys, stats


def raycast_march(volume, cam2vol, reproj, cols, rows, step_factor, slab=None):
    """-> (event keys uint32 [rows, cols], Ts float32 [rows, cols]: the refined ray parameter of hits, 0 elsewhere)"""
    keys = np.empty((rows, cols), np.uint32)
    ts = np.empty((rows, cols), np.float32)
    lib().orc_raycast_march(volume, C.byref(slab) if slab else None, f32(cam2vol).reshape(-1), f32(reproj), cols, rows,
                            step_factor, keys.reshape(-1), ts.reshape(-1))
    return keys, ts


def raycast_shade(volume, cam2vol, Rinv, reproj, ts, merged_keys, cols, rows, delta_factor, slab=None):
    pts = np.empty((rows, cols, 4), np.float32)
    nrm = np.empty((rows, cols, 4), np.float32)
    lib().orc_raycast_shade(volume, C.byref(slab) if slab else None, f32(cam2vol).reshape(-1), f32(Rinv).reshape(-1), f32(reproj),
                            np.ascontiguousarray(ts, np.float32).reshape(-1), np.ascontiguousarray(merged_keys, np.uint32).reshape(-1),
                            pts.reshape(-1), cols * 16, nrm.reshape(-1), cols * 16, cols, rows, delta_factor)
    return pts, nrm


def raycast_points_of_keys(cam2vol, Rinv, reproj, ts, merged_keys, normals, cols, rows):
    pts = np.empty((rows, cols, 4), np.float32)
    lib().orc_raycast_points_of_keys(f32(cam2vol).reshape(-1), f32(Rinv).reshape(-1), f32(reproj), np.ascontiguousarray(ts, np.float32).reshape(-1),
                                     np.ascontiguousarray(merged_keys, np.uint32).reshape(-1), np.ascontiguousarray(normals, np.float32).reshape(-1),
                                     cols * 16, pts.reshape(-1), cols * 16, cols, rows)
    return pts


def raycast_depth(volume, cam2vol, Rinv, reproj, cols, rows, step_factor, delta_factor, slab=None):
    dep = np.empty((rows, cols), np.uint16)
    nrm = np.empty((rows, cols, 4), np.float32)
    lib().orc_raycast_depth(volume, C.byref(slab) if slab else None, f32(cam2vol).reshape(-1), f32(Rinv).reshape(-1),
                            f32(reproj), dep, cols * 2, nrm.reshape(-1), cols * 16, cols, rows, step_factor,
                            delta_factor)
    return dep, nrm


def extract_cloud(volume, aff, capacity, slab=None):
    pts = np.zeros((capacity, 4), np.float32)
    n = int(lib().orc_extract_cloud(volume, C.byref(slab) if slab else None, f32(aff).reshape(-1), pts.reshape(-1), capacity))
    return pts[:min(n, capacity)], n


def extract_normals(volume, aff, Rinv, points, gradient_delta_factor, slab=None):
    points = np.ascontiguousarray(points, np.float32)
    out = np.empty_like(points)
    lib().orc_extract_normals(volume, C.byref(slab) if slab else None, f32(aff).reshape(-1), f32(Rinv).reshape(-1),
                              points.reshape(-1), points.shape[0], gradient_delta_factor, out.reshape(-1))
    return out


def project_and_remove(dists, points, proj, remove=True, want_ro=True):
    """Returns (new_points [n,4], dists_after (copy with removed pixels zeroed, or None), ro [n] or None, n_inside)."""
    rows, cols = dists.shape
    dists = np.ascontiguousarray(dists)
    pts = np.array(points, np.float32, copy=True).reshape(-1, 4)
    out = dists.copy() if remove else None
    ro = np.empty(pts.shape[0], np.float32) if want_ro else None
    n = lib().orc_project_and_remove(dists, cols * 2, out.ctypes.data if remove else None, cols * 2, cols, rows,
                                     pts.reshape(-1), pts.shape[0], f32(proj), ro.ctypes.data if want_ro else None)
    return pts, out, ro, int(n)


def knn(pos, queries, k, use_ref=False, brute=False):
    """use_ref: the reference's own nanoflann (oracle/_ref); brute: exhaustive index-order scan (ties -> lower index);
    default: the oracle's restatement of nanoflann (tie order of the reference)."""
    pos, queries = f32(pos), f32(queries)
    n = queries.shape[0]
    idx = np.empty((n, k), np.int32)
    d2 = np.empty((n, k), np.float32)
    fn = ref().ref_knn if use_ref else (lib().orc_knn_brute if brute else lib().orc_knn)
    fn(pos.reshape(-1), pos.shape[0], queries.reshape(-1), n, k, idx.reshape(-1), d2.reshape(-1))
    return idx, d2


def dqb(pos, dq, sigma, points, k, use_ref=False):
    pos, dq, sigma, points = f32(pos), f32(dq), f32(sigma), f32(points)
    out = np.empty((points.shape[0], 8), np.float32)
    fn = ref().ref_dqb if use_ref else lib().orc_dqb
    fn(pos.reshape(-1), dq.reshape(-1), sigma, pos.shape[0], k, points.reshape(-1), points.shape[0], out.reshape(-1))
    return out


def warp_points(pos, dq, sigma, points, normals, k, warp_to_live=None, use_ref=False):
    pos, dq, sigma = f32(pos), f32(dq), f32(sigma)
    pts = f32(points).copy()
    nrm = f32(normals).copy() if normals is not None else None
    nptr = nrm.ctypes.data if nrm is not None else None
    if use_ref:
        ref().ref_warp_points(pos.reshape(-1), dq.reshape(-1), sigma, pos.shape[0], k, pts.reshape(-1), nptr, pts.shape[0])
    else:
        if warp_to_live is None:
            warp_to_live = np.concatenate([np.eye(3, dtype=np.float32).reshape(-1), np.zeros(3, np.float32)])
        lib().orc_warp_points(pos.reshape(-1), dq.reshape(-1), sigma, pos.shape[0], k, pts.reshape(-1), nptr,
                              pts.shape[0], f32(warp_to_live).reshape(-1))
    return pts, nrm


# ---- depth front-end + ICP (oracle/dfusion_frontend_oracle.c) -------------------------------------------------------
def _u16(a):
    return np.ascontiguousarray(a, np.uint16)


def bilateral(depth, ksz, sigma_spatial, sigma_depth):
    depth = _u16(depth); rows, cols = depth.shape
    out = np.zeros_like(depth)
    lib().orc_bilateral(depth, cols * 2, out, cols * 2, cols, rows, ksz, sigma_spatial, sigma_depth)
    return out


def truncate_depth(depth, max_dist):
    out = _u16(depth).copy(); rows, cols = out.shape
    lib().orc_truncate_depth(out, cols * 2, cols, rows, max_dist)
    return out


def cloud_to_depth(cloud):
    c = f32(cloud); rows, cols = c.shape[:2]
    out = np.zeros((rows, cols), np.uint16)
    lib().orc_cloud_to_depth(c.reshape(-1), cols * 16, out, cols * 2, cols, rows)
    return out


def depth_pyramid(depth, sigma_depth):
    depth = _u16(depth); rows, cols = depth.shape
    out = np.zeros((rows // 2, cols // 2), np.uint16)
    lib().orc_depth_pyramid(depth, cols * 2, cols, rows, out, (cols // 2) * 2, sigma_depth)
    return out


def compute_normals_mask_depth(depth, intr):
    d = _u16(depth).copy(); rows, cols = d.shape
    n = np.zeros((rows, cols, 4), np.float32)
    lib().orc_compute_normals_mask_depth(d, cols * 2, n.reshape(-1), cols * 16, cols, rows, f32(intr))
    return d, n


def compute_point_normals(depth, intr):
    d = _u16(depth); rows, cols = d.shape
    p = np.zeros((rows, cols, 4), np.float32); n = np.zeros_like(p)
    lib().orc_compute_point_normals(d, cols * 2, p.reshape(-1), cols * 16, n.reshape(-1), cols * 16, cols, rows, f32(intr))
    return p, n


def resize_depth_normals(depth, normals):
    d = _u16(depth); rows, cols = d.shape
    nrm = f32(normals)
    do = np.zeros((rows // 2, cols // 2), np.uint16); no = np.zeros((rows // 2, cols // 2, 4), np.float32)
    lib().orc_resize_depth_normals(d, cols * 2, nrm.reshape(-1), cols * 16, cols, rows, do, (cols // 2) * 2, no.reshape(-1), (cols // 2) * 16)
    return do, no


def resize_points_normals(points, normals):
    p, nrm = f32(points), f32(normals); rows, cols = p.shape[:2]
    po = np.zeros((rows // 2, cols // 2, 4), np.float32); no = np.zeros_like(po)
    lib().orc_resize_points_normals(p.reshape(-1), cols * 16, nrm.reshape(-1), cols * 16, cols, rows, po.reshape(-1), (cols // 2) * 16,
                                    no.reshape(-1), (cols // 2) * 16)
    return po, no


def render_points(points, normals, light):
    p, n = f32(points), f32(normals); rows, cols = p.shape[:2]
    img = np.zeros((rows, cols, 4), np.uint8)
    lib().orc_render_points(p.reshape(-1), cols * 16, n.reshape(-1), cols * 16, cols, rows, f32(light), img.reshape(-1), cols * 4)
    return img


def render_depth(depth, normals, intr, light):
    d, n = _u16(depth), f32(normals); rows, cols = d.shape
    img = np.zeros((rows, cols, 4), np.uint8)
    lib().orc_render_depth(d, cols * 2, n.reshape(-1), cols * 16, cols, rows, f32(intr), f32(light), img.reshape(-1), cols * 4)
    return img


def render_tangent_colors(normals):
    n = f32(normals); rows, cols = n.shape[:2]
    img = np.zeros((rows, cols, 4), np.uint8)
    lib().orc_render_tangent_colors(n.reshape(-1), cols * 16, cols, rows, img.reshape(-1), cols * 4)
    return img


def icp_sums(curr, ncurr, prev, nprev, aff, intr, dist2_thres, min_cosine, depth_variant=False):
    """Returns (sums[27] f32, accepted)."""
    ncurr, nprev = f32(ncurr), f32(nprev)
    rows, cols = nprev.shape[:2]
    out = np.zeros(27, np.float32); acc = C.c_int(0)
    if depth_variant:
        lib().orc_icp_sums_depth(_u16(curr), cols * 2, ncurr.reshape(-1), cols * 16, _u16(prev), cols * 2, nprev.reshape(-1), cols * 16,
                                 cols, rows, f32(aff).reshape(-1), f32(intr), dist2_thres, min_cosine, out, C.byref(acc))
    else:
        lib().orc_icp_sums_points(f32(curr).reshape(-1), cols * 16, ncurr.reshape(-1), cols * 16, f32(prev).reshape(-1), cols * 16,
                                  nprev.reshape(-1), cols * 16, cols, rows, f32(aff).reshape(-1), f32(intr), dist2_thres, min_cosine, out,
                                  C.byref(acc))
    return out, acc.value


def solve_data_term(pos, dq, sigma, canonical, live, k, iters, lam=0.0):
    """Returns (dq_out [M, 8], energy [before, after])."""
    pos, dq, sigma = f32(pos), f32(dq), f32(sigma)
    canonical, live = f32(canonical), f32(live)
    out = np.zeros_like(dq); en = np.zeros(2, np.float32)
    lib().orc_solve_data_term(pos.reshape(-1), dq.reshape(-1), sigma, pos.shape[0], k, canonical.reshape(-1), live.reshape(-1),
                              canonical.shape[0], iters, lam, out.reshape(-1), en)
    return out, en
