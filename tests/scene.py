"""Shared test scaffolding: one small synthetic scene driven through the oracle (CPU) and through the
HIP C-ABI (GPU) with identical inputs."""
import numpy as np

import oracle_lib as O
from dynamicfusion_amd import synth

F32 = np.float32


class Scene:
    """Inputs of n_frames frames for a Config, everything as numpy (device upload is the caller's)."""

    def __init__(self, cfg, n_frames=2, with_nodes=True, sigma_mode="spacing", identity_warp=False):
        self.cfg = cfg
        self.intr = np.array(cfg.intr, F32)
        self.reproj = np.array([F32(1) / F32(cfg.intr[0]), F32(1) / F32(cfg.intr[1]), cfg.intr[2], cfg.intr[3]], F32)
        self.vs = np.array([F32(cfg.size) / F32(d) for d in cfg.dims], F32)
        self.trunc = float(max(F32(cfg.trunc_dist), F32(2.1) * max(self.vs)))     # tsdf_volume.cpp:68-73
        self.pose = cfg.volume_pose
        self.depths = [synth.depth_frame(cfg, f) for f in range(n_frames)]
        self.dists = [O.compute_dists(d, self.intr) for d in self.depths]
        self.cam_poses = [synth.camera_pose(cfg, f) for f in range(n_frames)]
        if with_nodes and cfg.nodes:
            self.pos, self.sigma = synth.make_nodes(cfg)
            if sigma_mode == "reference":
                self.sigma = np.full_like(self.sigma, 3.0)                         # warp_field.cpp:84
            if identity_warp:
                self.dqs = [synth.identity_dq(cfg.nodes) for _ in range(n_frames)]
            else:
                self.dqs = [synth.node_transforms(cfg, f) for f in range(n_frames)]

    # ---- affines exactly as the host wrappers derive them
    def vol2cam(self, f):
        return synth.affine_mul(synth.affine_inv(self.cam_poses[f]), self.pose)

    def world2cam(self, f):
        return synth.affine_inv(self.cam_poses[f])

    def cam2vol(self, f):
        return synth.affine_mul(synth.affine_inv(self.pose), self.cam_poses[f])

    def rinv(self, f):
        return np.linalg.inv(self.cam2vol(f)[:3, :3].astype(np.float64)).astype(F32)

    def new_volume(self, z_store_n=None):
        X, Y, Z = self.cfg.dims
        return np.zeros((Z if z_store_n is None else z_store_n, Y, X), np.uint32)

    def ovol(self, vol):
        return O.make_volume(vol, self.cfg.dims, self.vs, self.trunc, self.cfg.max_weight)


def decode(vol_u32):
    """-> (tsdf float32, weight uint16)"""
    half = (vol_u32 & 0xffff).astype(np.uint16).view(np.float16).astype(np.float32)
    return half, (vol_u32 >> 16).astype(np.uint16)


def compare_volumes(a_u32, b_u32):
    """Parity summary between two packed volumes."""
    ta, wa = decode(a_u32)
    tb, wb = decode(b_u32)
    n = a_u32.size
    return {
        "n": n,
        "bits_mismatch": int((a_u32 != b_u32).sum()),
        "weight_mismatch": int((wa != wb).sum()),
        "max_abs_dtsdf": float(np.max(np.abs(ta - tb))) if n else 0.0,
        "n_dtsdf_gt_1e-4": int((np.abs(ta - tb) > 1e-4).sum()),
    }
