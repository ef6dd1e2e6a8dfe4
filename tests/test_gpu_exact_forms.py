"""The warped sweep replaces sqrtf, the f64 reciprocal inside Quaternion::normalize and the quaternion products by shorter
instruction sequences that are claimed to give THE SAME BITS (dynamicfusion_amd/csrc/dfusion_device.h).  The library checks those
claims on the device over the whole domain of each form (dfusion_selftest_exact_forms); this test runs the check."""
import pytest
import torch

from dynamicfusion_amd import capi

pytestmark = pytest.mark.gpu


def test_short_forms_equal_generic_forms_on_every_input():
    counts = torch.zeros(10, dtype=torch.int64, device="cuda")
    capi.check(capi.lib().dfusion_selftest_exact_forms(1 << 27, counts.data_ptr(), None))
    torch.cuda.synchronize()
    sqrt_bad, rcp_bad, qmul_bad, unit_bad, unit_seen, fuse_bad, sample_bad, sample_upd, div_bad, div_seen = [int(c) for c in counts.cpu()]
    assert sqrt_bad == 0          # every finite f32 >= 2^-96 (1.8e9 values)
    assert rcp_bad == 0           # every positive normal f32 as the f64 denominator (2.1e9 values)
    assert qmul_bad == 0          # 2 x 1.3e8 random quaternion products, zeros / infinities / NaNs / denormals included
    assert unit_bad == 0 and unit_seen > (1 << 24)      # second normalisation of already normalised quaternions
    assert fuse_bad == 0          # the fuse division: every finite stored half x 97 weights x 64 tsdf values (4e8 cases)
    # the projective sample: shared-reciprocal divisions, short sqrtf, one-compare pixel test, and the rigid sweep's saturation
    # decision on the approximate root -- verdicts and tsdf bits equal to the generic statements, domain edges included
    assert sample_bad == 0 and sample_upd > (1 << 20)
    # round 5: the blend's FIRST normalisation as an f32 division instead of the reference's f64 reciprocal-and-scale -- the same bits
    # on every quaternion the domain test lets through (zeros of both signs, components at 2^-100 and around it, norms at both ends)
    assert div_bad == 0 and div_seen > (1 << 25)
