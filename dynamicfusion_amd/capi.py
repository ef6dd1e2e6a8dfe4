"""ctypes binding of the C-ABI in include/dfusion.h (libdfusion_hip.so).

There is NO fallback: if the HIP library is missing or an entry point fails, this raises.  The
product path never touches oracle/.
"""
import ctypes as C
import os

from . import build as _build

_lib = None


class DfVolume(C.Structure):
    """include/dfusion.h DfVolume == device::TsdfVolume (kfusion/src/internal.hpp:29-49)."""
    _fields_ = [("data", C.c_void_p), ("dims", C.c_int * 3), ("voxel_size", C.c_float * 3),
                ("trunc_dist", C.c_float), ("max_weight", C.c_int)]


class DfIcpLevel(C.Structure):
    _fields_ = [("curr", C.c_void_p), ("curr_pitch", C.c_size_t), ("ncurr", C.c_void_p), ("ncurr_pitch", C.c_size_t),
                ("prev", C.c_void_p), ("prev_pitch", C.c_size_t), ("nprev", C.c_void_p), ("nprev_pitch", C.c_size_t),
                ("cols", C.c_int), ("rows", C.c_int), ("iters", C.c_int)]


class DfSlab(C.Structure):
    _fields_ = [("z_store0", C.c_int), ("z_store_n", C.c_int), ("z_own0", C.c_int), ("z_own_n", C.c_int)]


DF_WARP_NO_CULL = 1
DF_WARP_NO_TABLE = 2
DF_WARP_NO_WEIGHT_TABLE = 4
DF_WARP_NO_LDS = 8
DF_WARP_NO_PIPELINE = 16
DF_WARP_NO_ZERO_SKIP = 32
DF_WARP_NO_DEPTH_PYRAMID = 64
DF_WARP_NO_BLOCK_MODEL = 128
DF_WARP_BLOCK_MODEL_NOW = 256
DF_WARP_NO_PREFETCH = 512
DF_WARP_STEADY_PREFETCH = 1024
DF_WARP_NO_CODES = 2048
DF_RIGID_NO_DEPTH_CULL = 1
DF_RIGID_NO_SHORT_FORMS = 2
DF_RIGID_KEEP_ALL = 4
DF_RIGID_NO_SAT = 8
DF_RIGID_POISON_SCRATCH = 16
DF_INDEX_VOXEL_TABLE = 1
DF_INDEX_WEIGHT_TABLE = 2
DF_INDEX_TABLES_ON_DEMAND = 4

# every symbol include/dfusion.h declares (tests check the library exports all of them)
SYMBOLS = [
    "dfusion_abi_version", "dfusion_error_string", "dfusion_clear", "dfusion_compute_dists", "dfusion_project_and_remove", "dfusion_integrate",
    "dfusion_raycast_points", "dfusion_raycast_depth", "dfusion_raycast_march", "dfusion_raycast_shade", "dfusion_extract_cloud",
    "dfusion_extract_normals", "dfusion_warp_create", "dfusion_warp_destroy",
    "dfusion_warp_set_nodes", "dfusion_warp_set_transforms", "dfusion_warp_build_index", "dfusion_knn",
    "dfusion_warp_points", "dfusion_integrate_warped", "dfusion_copy_bandwidth_probe", "dfusion_read_bandwidth_probe",
    "dfusion_bilateral_filter", "dfusion_truncate_depth", "dfusion_depth_pyramid", "dfusion_compute_normals_mask_depth",
    "dfusion_compute_point_normals", "dfusion_resize_depth_normals", "dfusion_resize_points_normals",
    "dfusion_icp_workspace_floats", "dfusion_icp_sums_points", "dfusion_icp_sums_depth", "dfusion_transform_points", "dfusion_warp_solve_data_term", "dfusion_warp_index_info", "dfusion_icp_estimate", "dfusion_release_scratch", "dfusion_raycast_points_of_keys",
    "dfusion_selftest_exact_forms", "dfusion_warp_set_point_tiling", "dfusion_integrate_ex", "dfusion_warp_debug_counters", "dfusion_warp_alive_blocks", "dfusion_warp_coded_blocks", "dfusion_raycast_points_of_keys_rows", "dfusion_raycast_sum_pieces", "dfusion_raycast_min_pieces", "dfusion_integrate_warped_prepare", "dfusion_integrate_warped_sweep",
    "dfusion_render_image_points", "dfusion_render_image_depth", "dfusion_render_tangent_colors", "dfusion_cloud_to_depth",
]


class DfusionError(RuntimeError):
    pass


def library_path():
    return _build.LIB_PATH


def lib():
    """Load (building if stale and hipcc is present) libdfusion_hip.so.  Raises if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        import torch  # noqa: F401  -- load torch's libamdhip64 first so one HIP runtime serves both
    except Exception:  # pragma: no cover
        pass
    path = _build.LIB_PATH
    try:
        path = _build.build_library()
    except Exception as e:
        if not os.path.exists(path):
            raise DfusionError("libdfusion_hip.so is missing and could not be built: %s" % e)
    _lib = load(path)
    return _lib


def load(path, strict=True):
    """dlopen a build of the library and declare its entry points.  strict=False (tools comparing against OLDER builds only)
    tolerates entry points the build does not have yet."""
    L = C.CDLL(path)
    if not strict:
        class _Missing:
            missing = True
        for s in SYMBOLS:
            if not hasattr(L, s):
                setattr(L, s, _Missing())
    fp = C.POINTER(C.c_float)
    vp = C.c_void_p
    L.dfusion_abi_version.restype = C.c_int
    L.dfusion_error_string.restype = C.c_char_p
    L.dfusion_error_string.argtypes = [C.c_int]
    L.dfusion_clear.argtypes = [DfVolume, C.POINTER(DfSlab), vp]
    L.dfusion_compute_dists.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.c_int, C.c_int, fp, vp]
    L.dfusion_project_and_remove.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.c_int, C.c_int, vp, C.c_ulonglong, fp, vp, vp, vp]
    L.dfusion_integrate.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, DfVolume, C.POINTER(DfSlab), fp, fp, vp, vp]
    L.dfusion_raycast_points.argtypes = [DfVolume, C.POINTER(DfSlab), fp, fp, fp, vp, C.c_size_t, vp, C.c_size_t,
                                         C.c_int, C.c_int, C.c_float, C.c_float, vp, vp]
    L.dfusion_raycast_depth.argtypes = [DfVolume, C.POINTER(DfSlab), fp, fp, fp, vp, C.c_size_t, vp, C.c_size_t,
                                        C.c_int, C.c_int, C.c_float, C.c_float, vp]
    L.dfusion_raycast_march.argtypes = [DfVolume, C.POINTER(DfSlab), fp, fp, C.c_int, C.c_int, C.c_float, C.c_uint, vp, vp]
    L.dfusion_raycast_shade.argtypes = [DfVolume, C.POINTER(DfSlab), fp, fp, fp, vp, vp, C.c_size_t, vp, C.c_size_t,
                                        C.c_int, C.c_int, C.c_float, vp]
    L.dfusion_raycast_points_of_keys.argtypes = [fp, fp, fp, vp, vp, C.c_size_t, vp, C.c_size_t, C.c_int, C.c_int, vp]
    L.dfusion_raycast_points_of_keys_rows.argtypes = [fp, fp, fp, vp, vp, C.c_size_t, vp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.dfusion_extract_cloud.argtypes = [DfVolume, C.POINTER(DfSlab), fp, vp, C.c_ulonglong, vp, vp]
    L.dfusion_extract_normals.argtypes = [DfVolume, C.POINTER(DfSlab), fp, fp, vp, C.c_ulonglong, C.c_float, vp, vp]
    L.dfusion_warp_create.argtypes = [C.POINTER(vp)]
    L.dfusion_warp_destroy.argtypes = [vp]
    L.dfusion_warp_set_nodes.argtypes = [vp, vp, vp, vp, C.c_int, vp]
    L.dfusion_warp_set_transforms.argtypes = [vp, vp, vp]
    L.dfusion_warp_build_index.argtypes = [vp, DfVolume, C.POINTER(DfSlab), fp, C.c_int, C.c_uint, vp]
    L.dfusion_knn.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, vp]
    L.dfusion_warp_points.argtypes = [vp, C.c_int, vp, vp, C.c_int, fp, vp]
    L.dfusion_integrate_warped.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, DfVolume, C.POINTER(DfSlab), fp, fp, fp,
                                           vp, C.c_int, C.c_uint, vp, vp]
    sz = C.c_size_t
    L.dfusion_bilateral_filter.argtypes = [vp, sz, vp, sz, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, vp]
    L.dfusion_truncate_depth.argtypes = [vp, sz, C.c_int, C.c_int, C.c_float, vp]
    L.dfusion_cloud_to_depth.argtypes = [vp, sz, vp, sz, C.c_int, C.c_int, vp]
    L.dfusion_depth_pyramid.argtypes = [vp, sz, C.c_int, C.c_int, vp, sz, C.c_float, vp]
    L.dfusion_compute_normals_mask_depth.argtypes = [vp, sz, vp, sz, C.c_int, C.c_int, fp, vp]
    L.dfusion_compute_point_normals.argtypes = [vp, sz, vp, sz, vp, sz, C.c_int, C.c_int, fp, vp]
    L.dfusion_resize_depth_normals.argtypes = [vp, sz, vp, sz, C.c_int, C.c_int, vp, sz, vp, sz, vp]
    L.dfusion_resize_points_normals.argtypes = [vp, sz, vp, sz, C.c_int, C.c_int, vp, sz, vp, sz, vp]
    L.dfusion_transform_points.argtypes = [vp, sz, C.c_int, vp, sz, C.c_int, C.c_int, C.c_int, fp, vp]
    L.dfusion_warp_solve_data_term.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_float, vp, vp, vp]
    L.dfusion_warp_index_info.argtypes = [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_uint), C.POINTER(C.c_int)]
    L.dfusion_selftest_exact_forms.argtypes = [C.c_ulonglong, vp, vp]
    L.dfusion_raycast_sum_pieces.argtypes = [vp, C.c_int, C.c_ulonglong, vp, vp]
    L.dfusion_raycast_min_pieces.argtypes = [vp, C.c_int, C.c_ulonglong, vp, vp]
    L.dfusion_integrate_warped_prepare.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, DfVolume, C.POINTER(DfSlab), fp, fp, fp, vp, C.c_int, C.c_uint, vp]
    L.dfusion_integrate_warped_sweep.argtypes = [DfVolume, C.POINTER(DfSlab), vp, vp, vp]
    L.dfusion_warp_set_point_tiling.argtypes = [vp, C.c_int]
    L.dfusion_icp_estimate.argtypes = [C.POINTER(DfIcpLevel), C.c_int, C.c_int, fp, C.c_float, C.c_float, vp, vp, vp]
    L.dfusion_icp_workspace_floats.argtypes = [C.c_int, C.c_int]
    L.dfusion_icp_sums_points.argtypes = [vp, sz, vp, sz, vp, sz, vp, sz, C.c_int, C.c_int, fp, fp, C.c_float, C.c_float, vp, vp, vp, vp]
    L.dfusion_icp_sums_depth.argtypes = [vp, sz, vp, sz, vp, sz, vp, sz, C.c_int, C.c_int, fp, fp, C.c_float, C.c_float, vp, vp, vp, vp]
    L.dfusion_render_image_points.argtypes = [vp, sz, vp, sz, C.c_int, C.c_int, fp, vp, sz, vp]
    L.dfusion_render_image_depth.argtypes = [vp, sz, vp, sz, C.c_int, C.c_int, fp, fp, vp, sz, vp]
    L.dfusion_render_tangent_colors.argtypes = [vp, sz, C.c_int, C.c_int, vp, sz, vp]
    L.dfusion_integrate_ex.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, DfVolume, C.POINTER(DfSlab), fp, fp, C.c_uint, vp, vp, vp]
    L.dfusion_warp_debug_counters.argtypes = [vp, vp]
    L.dfusion_warp_alive_blocks.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp]
    L.dfusion_warp_coded_blocks.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp]
    L.dfusion_copy_bandwidth_probe.argtypes = [vp, vp, C.c_size_t, vp]
    L.dfusion_read_bandwidth_probe.argtypes = [vp, C.c_size_t, vp, vp]
    for s in SYMBOLS:
        if s not in ("dfusion_error_string",):
            getattr(L, s).restype = C.c_int
    return L


def check(rc, what=""):
    if rc != 0:
        msg = lib().dfusion_error_string(rc)
        raise DfusionError("%s failed: %s (code %d)" % (what or "dfusion call", msg.decode() if msg else "?", rc))


def floats(seq):
    """Host float block (affines, intrinsics) -> C float array."""
    seq = [float(x) for x in seq]
    return (C.c_float * len(seq))(*seq)
