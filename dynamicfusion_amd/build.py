"""In-tree build of the gfx950 HIP library (libdfusion_hip.so) with hipcc.

The .so is git-ignored but travels to the GPU box with the gpurun snapshot, so a box without
the sources' mtimes changing never rebuilds.  hipcc cross-compiles without a GPU.
"""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libdfusion_hip.so")

SOURCES = ["dfusion_volume.hip", "dfusion_warp.hip", "dfusion_raycast.hip", "dfusion_frontend.hip", "dfusion_solver.hip", "dfusion_selftest.hip"]
HEADERS = ["dfusion_device.h", "dfusion_internal.h", "dfusion_nanoflann.h", "dfusion_pyramid.h", "dfusion_warp_blocks.h", os.path.join(REPO_DIR, "include", "dfusion.h")]

# -ffp-contract=off: fused multiply-adds only where the reference writes __fmaf_rn (explicit fmaf);
# that is what makes the kernels bit-comparable with the IEEE CPU oracle.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-fno-fast-math", "-Wno-unused-result"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the dynamicfusion_amd HIP library cannot be built")


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def kernel_source_sha(kernel):
    """sha256 over the sources a kernel (by name) is built from: its .hip plus the device headers.  Stamps profiles/pmc_latest.json
    (tools/pmc_summary.py); bench.py drops counters whose stamp is not the tree's."""
    import hashlib
    k = kernel.replace("void ", "")
    f = ("dfusion_warp.hip" if k.startswith(("df_warp", "df_sweep", "df_block", "df_blocks", "df_brick", "df_scan", "df_pack", "df_node", "df_points"))
         else "dfusion_raycast.hip" if k.startswith(("df_raycast", "df_extract"))
         else "dfusion_solver.hip" if k.startswith("df_sv")
         else "dfusion_volume.hip")
    h = hashlib.sha256()
    for name in (f, "dfusion_warp_blocks.h", "dfusion_device.h", "dfusion_internal.h", "dfusion_pyramid.h"):
        h.update(open(os.path.join(CSRC, name), "rb").read())
    return h.hexdigest()


def build_library(force=False, verbose=False):
    """Compile every HIP source for gfx950 into dynamicfusion_amd/libdfusion_hip.so."""
    if not force and not _stale():
        return LIB_PATH
    cmd = [_hipcc()] + HIPCC_FLAGS + ["-I", os.path.join(REPO_DIR, "include"), "-I", CSRC]
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


HOST_DIR = os.path.join(PKG_DIR, "host")
HOST_LIB = os.path.join(HOST_DIR, "libkfusion_hip.so")
HOST_APP = os.path.join(HOST_DIR, "headless_frame")
HOST_KINFU_APP = os.path.join(HOST_DIR, "kinfu_headless")
HOST_WARP_TESTS = os.path.join(HOST_DIR, "warp_tests")
HOST_DEMO_CALLS = os.path.join(HOST_DIR, "demo_calls")
HOST_ZSLAB_LIB = os.path.join(HOST_DIR, "libkfusion_zslab.so")       # kfusion::cuda::ZSlabComm: the RCCL side of the Z-slab sharding
HOST_ZSLAB_APP = os.path.join(HOST_DIR, "zslab_frame")


def build_host(force=False, verbose=False):
    """C++ host mirror (kfusion::cuda::TsdfVolume / WarpField over the C-ABI) + the headless harness, with g++."""
    build_library(force=False)
    src = os.path.join(HOST_DIR, "src", "kfusion_hip.cpp")
    app = os.path.join(HOST_DIR, "apps", "headless_frame.cpp")
    app2 = os.path.join(HOST_DIR, "apps", "kinfu_headless.cpp")
    app3 = os.path.join(HOST_DIR, "apps", "warp_tests.cpp")
    app4 = os.path.join(HOST_DIR, "apps", "demo_calls.cpp")
    zsrc = os.path.join(HOST_DIR, "src", "zslab_rccl.cpp")
    app5 = os.path.join(HOST_DIR, "apps", "zslab_frame.cpp")
    deps = [src, app, app2, app3, app4, zsrc, app5, LIB_PATH] + [os.path.join(r, f) for r, _, fs in os.walk(os.path.join(HOST_DIR, "include")) for f in fs]
    outs = (HOST_LIB, HOST_APP, HOST_KINFU_APP, HOST_WARP_TESTS, HOST_DEMO_CALLS, HOST_ZSLAB_LIB, HOST_ZSLAB_APP)
    if not force and all(os.path.exists(f) for f in outs) and min(os.path.getmtime(f) for f in outs) >= max(os.path.getmtime(d) for d in deps):
        return HOST_LIB, HOST_APP
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    common = ["g++", "-std=c++17", "-O2", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(HOST_DIR, "include"),
              "-I", os.path.join(REPO_DIR, "include"), "-I", os.path.join(rocm, "include")]
    link = ["-L", PKG_DIR, "-ldfusion_hip", "-L", os.path.join(rocm, "lib"), "-lamdhip64"]
    cmds = [common + ["-fPIC", "-shared", src, "-o", HOST_LIB] + link + ["-Wl,-rpath,$ORIGIN/.."],
            common + [app, "-o", HOST_APP, "-L", HOST_DIR, "-lkfusion_hip"] + link + ["-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/.."],
            common + [app2, "-o", HOST_KINFU_APP, "-L", HOST_DIR, "-lkfusion_hip"] + link + ["-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/.."],
            common + [app3, "-o", HOST_WARP_TESTS, "-L", HOST_DIR, "-lkfusion_hip"] + link + ["-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/.."],
            common + [app4, "-o", HOST_DEMO_CALLS, "-L", HOST_DIR, "-lkfusion_hip"] + link + ["-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/.."],
            common + ["-fPIC", "-shared", zsrc, "-o", HOST_ZSLAB_LIB, "-L", HOST_DIR, "-lkfusion_hip"] + link + ["-lrccl", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/.."],
            common + [app5, "-o", HOST_ZSLAB_APP, "-L", HOST_DIR, "-lkfusion_zslab", "-lkfusion_hip"] + link + ["-lrccl", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/.."]]
    for c in cmds:
        if verbose:
            print(" ".join(c))
        subprocess.check_call(c)
    return HOST_LIB, HOST_APP


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
    print(build_host(force=True, verbose=True))
