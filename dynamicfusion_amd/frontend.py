"""Host-side mirror of the reference's depth front-end and projective ICP:
kfusion::cuda::{depthBilateralFilter, depthTruncation, depthBuildPyramid, computeNormalsAndMaskDepth, computePointNormals,
resizeDepthNormals, resizePointsNormals} (kfusion/src/imgproc.cpp:10-150) and kfusion::cuda::ProjectiveICP
(kfusion/src/projective_icp.cpp:64-213) over the C-ABI.  Images are device tensors: depth int16 [rows, cols] (u16 bits),
points / normals float32 [rows, cols, 4].
"""
import math

import numpy as np
import torch

from . import capi
from .synth import aff12
from .tsdf_volume import F32, Intr, _ptr, _stream


def intr_level(intr, level):
    """Intr::operator()(level), precomp.cpp:10-14."""
    div = 1 << level
    return Intr(F32(intr.fx) / F32(div), F32(intr.fy) / F32(div), F32(intr.cx) / F32(div), F32(intr.cy) / F32(div))


def _new_depth(rows, cols, device):
    return torch.empty((rows, cols), dtype=torch.int16, device=device)


def _new_image4(rows, cols, device):
    return torch.empty((rows, cols, 4), dtype=torch.float32, device=device)


def depthBilateralFilter(depth, kernel_size, sigma_spatial, sigma_depth, out=None):
    rows, cols = depth.shape
    if out is None:
        out = _new_depth(rows, cols, depth.device)
    capi.check(capi.lib().dfusion_bilateral_filter(_ptr(depth), cols * 2, _ptr(out), cols * 2, cols, rows, int(kernel_size),
                                                   float(sigma_spatial), float(sigma_depth), _stream()), "dfusion_bilateral_filter")
    return out


def depthTruncation(depth, threshold):
    rows, cols = depth.shape
    capi.check(capi.lib().dfusion_truncate_depth(_ptr(depth), cols * 2, cols, rows, float(threshold), _stream()), "dfusion_truncate_depth")
    return depth


def cloudToDepth(cloud, out=None):
    """cuda::cloudToDepth (imgproc.cpp:98-103): depth mm = points.z * 1000 (a NaN point -- a ray-cast miss -- gives 0)."""
    rows, cols = cloud.shape[:2]
    if out is None:
        out = torch.empty((rows, cols), dtype=torch.int16, device=cloud.device)
    capi.check(capi.lib().dfusion_cloud_to_depth(_ptr(cloud), cols * 16, _ptr(out), cols * 2, cols, rows, _stream()), "dfusion_cloud_to_depth")
    return out


def depthBuildPyramid(depth, sigma_depth, out=None):
    rows, cols = depth.shape
    if out is None:
        out = _new_depth(rows // 2, cols // 2, depth.device)
    capi.check(capi.lib().dfusion_depth_pyramid(_ptr(depth), cols * 2, cols, rows, _ptr(out), (cols // 2) * 2, float(sigma_depth),
                                                _stream()), "dfusion_depth_pyramid")
    return out


def computeNormalsAndMaskDepth(intr, depth, normals=None):
    rows, cols = depth.shape
    if normals is None:
        normals = _new_image4(rows, cols, depth.device)
    capi.check(capi.lib().dfusion_compute_normals_mask_depth(_ptr(depth), cols * 2, _ptr(normals), cols * 16, cols, rows, intr.as_proj(),
                                                             _stream()), "dfusion_compute_normals_mask_depth")
    return normals


def computePointNormals(intr, depth, points=None, normals=None):
    rows, cols = depth.shape
    if points is None:
        points = _new_image4(rows, cols, depth.device)
    if normals is None:
        normals = _new_image4(rows, cols, depth.device)
    capi.check(capi.lib().dfusion_compute_point_normals(_ptr(depth), cols * 2, _ptr(points), cols * 16, _ptr(normals), cols * 16, cols,
                                                        rows, intr.as_proj(), _stream()), "dfusion_compute_point_normals")
    return points, normals


def resizeDepthNormals(depth, normals):
    rows, cols = depth.shape
    d = _new_depth(rows // 2, cols // 2, depth.device)
    n = _new_image4(rows // 2, cols // 2, depth.device)
    capi.check(capi.lib().dfusion_resize_depth_normals(_ptr(depth), cols * 2, _ptr(normals), cols * 16, cols, rows, _ptr(d),
                                                       (cols // 2) * 2, _ptr(n), (cols // 2) * 16, _stream()), "dfusion_resize_depth_normals")
    return d, n


def resizePointsNormals(points, normals):
    rows, cols = points.shape[:2]
    p = _new_image4(rows // 2, cols // 2, points.device)
    n = _new_image4(rows // 2, cols // 2, points.device)
    capi.check(capi.lib().dfusion_resize_points_normals(_ptr(points), cols * 16, _ptr(normals), cols * 16, cols, rows, _ptr(p),
                                                        (cols // 2) * 16, _ptr(n), (cols // 2) * 16, _stream()),
               "dfusion_resize_points_normals")
    return p, n


def renderImage(src, normals, intr, light_pose, image=None):
    """kfusion::cuda::renderImage (imgproc.cpp:152-186): Phong view of a points image (float32 [rows, cols, 4]) or a depth image
    (int16 [rows, cols]) with its normals; image: uint8 [rows, cols, 4] BGRA (device)."""
    rows, cols = normals.shape[:2]
    if image is None:
        image = torch.empty((rows, cols, 4), dtype=torch.uint8, device=normals.device)
    light = capi.floats(light_pose)
    if src.dtype == torch.float32:
        capi.check(capi.lib().dfusion_render_image_points(_ptr(src), cols * 16, _ptr(normals), cols * 16, cols, rows, light, _ptr(image),
                                                          image.stride(0), _stream()), "dfusion_render_image_points")
    else:
        capi.check(capi.lib().dfusion_render_image_depth(_ptr(src), cols * 2, _ptr(normals), cols * 16, cols, rows, intr.as_proj(), light,
                                                         _ptr(image), image.stride(0), _stream()), "dfusion_render_image_depth")
    return image


def renderTangentColors(normals, image=None):
    """kfusion::cuda::renderTangentColors (imgproc.cpp:193-201)."""
    rows, cols = normals.shape[:2]
    if image is None:
        image = torch.empty((rows, cols, 4), dtype=torch.uint8, device=normals.device)
    capi.check(capi.lib().dfusion_render_tangent_colors(_ptr(normals), cols * 16, cols, rows, _ptr(image), image.stride(0), _stream()),
               "dfusion_render_tangent_colors")
    return image


def unpack_icp_sums(sums):
    """StreamHelper::get (projective_icp.cpp:43-61): 27 floats -> symmetric A (6x6) and b (6)."""
    A = np.zeros((6, 6), F32)
    b = np.zeros(6, F32)
    shift = 0
    for i in range(6):
        for j in range(i, 7):
            v = sums[shift]
            shift += 1
            if j == 6:
                b[i] = v
            else:
                A[i, j] = A[j, i] = v
    return A, b


def rodrigues_affine(r):
    """cv::Affine3f(rvec, t) (projective_icp.cpp:162 `Affine3f Tinc(Vec3f(r.val), Vec3f(r.val+3))`): Rodrigues rotation of
    r[0:3], translation r[3:6].  OpenCV is not in the reference tree; evaluated here in f64 and rounded to f32."""
    rvec = np.asarray(r[:3], np.float64)
    theta = float(np.linalg.norm(rvec))
    R = np.eye(3)
    if theta >= np.finfo(np.float64).eps:
        c, s = math.cos(theta), math.sin(theta)
        k = rvec / theta
        K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        R = c * np.eye(3) + (1 - c) * np.outer(k, k) + s * K
    T = np.eye(4, dtype=F32)
    T[:3, :3] = R.astype(F32)
    T[:3, 3] = np.asarray(r[3:6], F32)
    return T


class ProjectiveICP:
    """kfusion::cuda::ProjectiveICP (projective_icp.cpp:64-213).  The device part (correspondences + the 27 sums) is the
    HIP kernel pair behind dfusion_icp_sums_*; the 6x6 solve is host side as in the reference (cv::solve DECOMP_SVD there,
    numpy f64 here -- OpenCV's Jacobi SVD is a third-party algorithm outside the reference tree, so the pose is matched to
    tolerance, not bit for bit; the sums are)."""
    MAX_PYRAMID_LEVELS = 4

    def __init__(self):
        self.angle_thres_ = float(F32(20.0) * F32(0.017453293))      # deg2rad(20.f), projective_icp.cpp:66
        self.dist_thres_ = 0.1
        self.setIterationsNum([10, 5, 4, 0])
        self._ws = None
        self._sums = None
        self.last_accepted = None

    def setIterationsNum(self, iters):
        iters = list(iters)[:self.MAX_PYRAMID_LEVELS]
        self.iters_ = iters + [0] * (self.MAX_PYRAMID_LEVELS - len(iters))

    def setDistThreshold(self, d):
        self.dist_thres_ = float(d)

    def setAngleThreshold(self, a):
        self.angle_thres_ = float(a)

    def getUsedLevelsNum(self):
        i = self.MAX_PYRAMID_LEVELS - 1
        while i >= 0 and not self.iters_[i]:
            i -= 1
        return i + 1

    def thresholds(self):
        """ComputeIcpHelper ctor (projective_icp.cpp:11-15): (dist2_thres, min_cosine) as floats."""
        return float(F32(self.dist_thres_) * F32(self.dist_thres_)), float(F32(math.cos(F32(self.angle_thres_))))

    def sums(self, level_intr, curr, ncurr, prev, nprev, affine, depth_variant=False):
        """One ComputeIcpHelper::operator() call: returns the 27 sums (numpy f32)."""
        rows, cols = nprev.shape[:2]
        need = capi.lib().dfusion_icp_workspace_floats(cols, rows)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(max(need, 27 * 1200 + 27), dtype=torch.float32, device=nprev.device)
        if self._sums is None:
            self._sums = torch.empty(27, dtype=torch.float32, device=nprev.device)
        acc = torch.zeros(1, dtype=torch.int32, device=nprev.device)
        d2, mc = self.thresholds()
        L = capi.lib()
        if depth_variant:
            capi.check(L.dfusion_icp_sums_depth(_ptr(curr), cols * 2, _ptr(ncurr), cols * 16, _ptr(prev), cols * 2, _ptr(nprev), cols * 16,
                                                cols, rows, capi.floats(aff12(affine)), level_intr.as_proj(), d2, mc, _ptr(self._ws),
                                                _ptr(self._sums), _ptr(acc), _stream()), "dfusion_icp_sums_depth")
        else:
            capi.check(L.dfusion_icp_sums_points(_ptr(curr), cols * 16, _ptr(ncurr), cols * 16, _ptr(prev), cols * 16, _ptr(nprev),
                                                 cols * 16, cols, rows, capi.floats(aff12(affine)), level_intr.as_proj(), d2, mc,
                                                 _ptr(self._ws), _ptr(self._sums), _ptr(acc), _stream()), "dfusion_icp_sums_points")
        out = self._sums.cpu().numpy().copy()                      # cudaStreamSynchronize + pinned copy in the reference (:45)
        self.last_accepted = int(acc.item())
        return out

    def estimateTransformDevice(self, intr, curr_pyr, ncurr_pyr, prev_pyr, nprev_pyr, depth_variant=False):
        """The same loop as ONE enqueue (dfusion_icp_estimate): sums, 6x6 solve and pose update all on the GPU, one read-back at the
        end.  Returns (ok, affine 4x4 f32 curr -> prev)."""
        n = self.getUsedLevelsNum()
        levels = (capi.DfIcpLevel * n)()
        esz = 2 if depth_variant else 16
        for l in range(n):
            rows, cols = nprev_pyr[l].shape[:2]
            levels[l] = capi.DfIcpLevel(curr_pyr[l].data_ptr(), cols * esz, ncurr_pyr[l].data_ptr(), cols * 16, prev_pyr[l].data_ptr(), cols * esz,
                                        nprev_pyr[l].data_ptr(), cols * 16, cols, rows, self.iters_[l])
        rows0, cols0 = nprev_pyr[0].shape[:2]
        need = capi.lib().dfusion_icp_workspace_floats(cols0, rows0) + 27
        dev = nprev_pyr[0].device
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(max(need, 27 * 1200 + 27), dtype=torch.float32, device=dev)
        state = torch.empty(13, dtype=torch.float32, device=dev)
        d2, mc = self.thresholds()
        capi.check(capi.lib().dfusion_icp_estimate(levels, n, 1 if depth_variant else 0, intr.as_proj(), d2, mc, _ptr(self._ws), _ptr(state),
                                                   _stream()), "dfusion_icp_estimate")
        st = state.cpu().numpy()
        affine = np.eye(4, dtype=F32)
        affine[:3, :3] = st[:9].reshape(3, 3)
        affine[:3, 3] = st[9:12]
        return bool(st[12] != 0), affine

    def estimateTransform(self, intr, curr_pyr, ncurr_pyr, prev_pyr, nprev_pyr, depth_variant=False):
        """projective_icp.cpp:129-213 with the reference's control flow (one host solve per iteration).  Returns (ok, affine 4x4 f32
        curr -> prev)."""
        affine = np.eye(4, dtype=F32)
        for level in range(self.getUsedLevelsNum() - 1, -1, -1):
            li = intr_level(intr, level)                           # setLevelIntr, projective_icp.cpp:17-23
            for _ in range(self.iters_[level]):
                s = self.sums(li, curr_pyr[level], ncurr_pyr[level], prev_pyr[level], nprev_pyr[level], affine, depth_variant)
                A, b = unpack_icp_sums(s)
                det = float(np.linalg.det(A.astype(np.float64)))   # cv::determinant(A), :150
                if abs(det) < 1e-15 or math.isnan(det):
                    return False, affine
                r = np.linalg.solve(A.astype(np.float64), b.astype(np.float64))     # cv::solve(A, b, r, DECOMP_SVD), :159
                affine = (rodrigues_affine(r).astype(np.float64) @ affine.astype(np.float64)).astype(F32)   # Tinc * affine, :163
        return True, affine
