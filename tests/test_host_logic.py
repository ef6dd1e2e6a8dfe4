"""Host-side mirror logic (no GPU): kfusion::cuda::TsdfVolume parameter semantics, slab bookkeeping,
synthetic-input determinism."""
import numpy as np
import pytest

from dynamicfusion_amd import TsdfVolume, capi, sharded, synth

F32 = np.float32


def test_ctor_defaults_match_reference():          # tsdf_volume.cpp:7-14
    v = TsdfVolume((512, 512, 512), allocate=False)
    assert v.getMaxWeight() == 128 and v.getSize().tolist() == [3, 3, 3]
    assert v.getRaycastStepFactor() == 0.75 and v.getGradientDeltaFactor() == 0.75
    assert abs(v.getTruncDist() - 0.03) < 1e-8 and np.array_equal(v.getPose(), np.eye(4, dtype=F32))
    assert np.allclose(v.getVoxelSize(), 3.0 / 512)


def test_trunc_dist_clamp():                        # tsdf_volume.cpp:68-73
    v = TsdfVolume((256, 256, 256), allocate=False)
    v.setSize([1, 1, 1])
    v.setTruncDist(0.001)
    assert v.getTruncDist() == float(F32(2.1) * (F32(1) / F32(256)))
    v.setTruncDist(0.04)
    assert v.getTruncDist() == float(F32(0.04))


def test_kinfu_setter_order_quirk_is_preserved():
    """KinFu::KinFu (kinfu.cpp:102-107) calls setTruncDist BEFORE setSize: the clamp is evaluated against the
    ctor's 3 m size and setSize only ever re-clamps upwards, so 128^3 @ 1 m ends with trunc 0.0492, not 0.04."""
    v = TsdfVolume((128, 128, 128), allocate=False)
    v.setTruncDist(0.04)
    v.setSize([1, 1, 1])
    assert v.getTruncDist() == float(F32(2.1) * (F32(3) / F32(128)))
    w = TsdfVolume((512, 512, 512), allocate=False)       # default_params(): 512^3 @ 3 m is unaffected
    w.setTruncDist(0.04); w.setSize([3, 3, 3])
    assert w.getTruncDist() == float(F32(0.04))


def test_apply_affine_premultiplies():              # tsdf_volume.cpp:88
    v = TsdfVolume((64, 64, 64), allocate=False)
    v.setPose(synth.translation(1, 2, 3))
    v.applyAffine(synth.translation(10, 0, 0))
    assert v.getPose()[:3, 3].tolist() == [11, 2, 3]


def test_compute_calls_without_blob_raise():
    v = TsdfVolume((64, 64, 64), allocate=False)
    with pytest.raises(capi.DfusionError):
        v.clear()


def test_dims_must_be_multiple_of_4():
    with pytest.raises(ValueError):
        TsdfVolume((66, 64, 64), allocate=False)


def test_slab_bookkeeping():
    v = TsdfVolume((64, 64, 64), slab=(16, 16, 5), allocate=False)
    assert (v.z_store0, v.z_store_n, v.z_own0, v.z_own_n) == (11, 26, 16, 16)
    lo = TsdfVolume((64, 64, 64), slab=(0, 16, 5), allocate=False)
    assert (lo.z_store0, lo.z_store_n) == (0, 21)
    hi = TsdfVolume((64, 64, 64), slab=(48, 16, 5), allocate=False)
    assert (hi.z_store0, hi.z_store_n) == (43, 21)


@pytest.mark.parametrize("Z,world", [(512, 1), (512, 2), (512, 4), (512, 8), (1024, 8), (64, 3), (100, 3)])
def test_slab_range_partitions_the_volume(Z, world):
    cover = []
    for r in range(world):
        z0, n = sharded.slab_range(Z, r, world)
        cover += list(range(z0, z0 + n))
    assert cover == list(range(Z))


def test_halo_planes_headline_config():
    # 512^3 @ 3 m: time_step = 0.03 m, voxel 5.86 mm -> ceil(5.12 + 0.5) + 2 = 8
    assert sharded.halo_planes(0.04, 0.75, 0.5, 3.0 / 512) == 8


def test_synth_is_deterministic_and_sane():
    cfg = synth.Config(64, 1.0, cols=160, rows=120, nodes=50, k=4)
    d0, d1 = synth.depth_frame(cfg, 0), synth.depth_frame(cfg, 0)
    assert np.array_equal(d0, d1) and d0.dtype == np.uint16
    assert 0.01 < (d0 == 0).mean() < 0.05                                  # the 2 % holes
    p0, s0 = synth.make_nodes(cfg)
    p1, s1 = synth.make_nodes(cfg)
    assert np.array_equal(p0, p1) and np.array_equal(s0, s1) and p0.shape == (50, 3)
    dq = synth.node_transforms(cfg, 3)
    assert dq.shape == (50, 8) and np.allclose(np.linalg.norm(dq[:, :4], axis=1), 1, atol=1e-6)
    ident = synth.identity_dq(4)
    assert ident[:, 0].tolist() == [1] * 4 and ident[:, 4].tolist() == [1] * 4   # dual_quaternion.hpp:25-29
