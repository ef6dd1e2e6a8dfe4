"""dynamicfusion_amd -- MI355X (gfx950) native DynamicFusion hot path.

Warp-field-deformed TSDF integration, surface ray-casting and the per-voxel dual-quaternion blend
(k-NN over warp nodes) as hand-written HIP kernels behind a C-ABI (include/dfusion.h), with
host-side mirrors of kfusion::cuda::TsdfVolume / kfusion::WarpField, the depth front-end + ProjectiveICP (frontend) and the Z-slab
collectives (sharded).
"""
from . import build, capi, frontend, sharded, synth  # noqa: F401
from .tsdf_volume import Intr, TsdfVolume, compute_dists, download_u16, upload_u16  # noqa: F401
from .warp_field import WarpField  # noqa: F401

__all__ = ["build", "capi", "frontend", "sharded", "synth", "Intr", "TsdfVolume", "WarpField", "compute_dists", "upload_u16", "download_u16"]
