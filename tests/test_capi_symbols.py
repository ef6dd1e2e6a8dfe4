"""The C-ABI library loads on a box without a GPU and exports every symbol include/dfusion.h declares;
without a device the entry points fail LOUDLY (error code), never fall back to a CPU path."""
import ctypes as C
import os
import re

import pytest

from dynamicfusion_amd import build, capi

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(REPO, "include", "dfusion.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dfusion_\w+)\s*\(", src)))


def test_header_and_binding_agree():
    assert declared_functions() == sorted(capi.SYMBOLS)


def test_library_builds_for_gfx950_and_exports_every_symbol():
    path = build.build_library()
    assert os.path.exists(path)
    L = C.CDLL(path)
    for name in declared_functions():
        assert hasattr(L, name), name
    assert capi.lib().dfusion_abi_version() == 7                      # DFUSION_ABI_VERSION


def test_struct_layout_matches_header():
    # DfVolume: void* + 3 int + 3 float + float + int = 8 + 12 + 12 + 4 + 4 = 40 ; DfSlab: 4 int
    assert C.sizeof(capi.DfVolume) == 40 and C.sizeof(capi.DfSlab) == 16


def test_no_device_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    assert capi.lib().dfusion_warp_create(C.byref(h)) == 100003            # DF_E_NO_DEVICE
    with pytest.raises(capi.DfusionError):
        capi.check(100003, "dfusion_warp_create")
    assert b"no HIP device" in capi.lib().dfusion_error_string(100003)


def test_invalid_arguments_are_rejected():
    L = capi.lib()
    v = capi.DfVolume()                                                     # null data, zero dims
    assert L.dfusion_clear(v, None, None) == 100001                          # DF_E_INVALID
    assert L.dfusion_integrate(None, 0, 0, 0, v, None, None, None, None, None) == 100001
    assert L.dfusion_raycast_points(v, None, None, None, None, None, 0, None, 0, 0, 0, 0.75, 0.5, None, None) == 100001


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(REPO, "dynamicfusion_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                text = open(os.path.join(root, f)).read()
                assert "oracle_lib" not in text and "liboracle" not in text and "libdfref" not in text, f
