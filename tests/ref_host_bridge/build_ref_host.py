"""TEST INFRASTRUCTURE: INTEGRATION.md section A, compiled.  The reference's OWN host layer

    /root/reference/kfusion/src/{tsdf_volume.cpp, imgproc.cpp, precomp.cpp, device_memory.cpp}      (unmodified, read where they lie)
  + tests/ref_host_bridge/hip_bridge.cpp     (kfusion::device::*  ->  the C-ABI of libdfusion_hip.so: what a maintainer adds)
    ->  tests/ref_host_bridge/_build/libkfusion_refhost.so
  + tests/ref_host_bridge/ref_host_frame.cpp (driver)  ->  tests/ref_host_bridge/_build/ref_host_frame

with g++ against the reference's own headers, tests/opencv_stub (OpenCV is not installed in this image) and tests/ref_host_bridge/shim
(a dozen cudaX -> hipX names for the reference's safe_call.hpp / device_memory.cpp; a two-line stand-in for the absent third-party
Opt solver header that kfusion/kinfu.hpp pulls in).  Nothing is copied out of /root/reference; the outputs are git-ignored but travel to the
GPU box with the gpurun snapshot, where tests/test_gpu_ref_host_bridge.py runs the binary (/root/reference does not exist there)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "_build")
REF = "/root/reference/kfusion"
LIB = os.path.join(OUT, "libkfusion_refhost.so")
APP = os.path.join(OUT, "ref_host_frame")
REF_SOURCES = ["tsdf_volume.cpp", "imgproc.cpp", "precomp.cpp", "device_memory.cpp"]


def have_reference():
    return os.path.exists(os.path.join(REF, "src", "tsdf_volume.cpp"))


def flags():
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    inc = ["-I", os.path.join(HERE, "shim"), "-I", os.path.join(REPO, "tests", "opencv_stub"), "-I", os.path.join(REF, "include"),
           "-I", os.path.join(REF, "src", "utils"), "-I", os.path.join(REF, "src"), "-I", os.path.join(REPO, "include"), "-I", os.path.join(rocm, "include")]
    return ["g++", "-std=c++17", "-O2", "-w", "-D__HIP_PLATFORM_AMD__"] + inc, rocm


def build(force=False):
    """Returns the driver's path (built here when the reference is present, else whatever travelled with the snapshot)."""
    if not have_reference():
        return APP if os.path.exists(APP) else None
    import sys
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    from dynamicfusion_amd import build as B
    B.build_library()
    srcs = [os.path.join(REF, "src", f) for f in REF_SOURCES] + [os.path.join(HERE, "hip_bridge.cpp")]
    drv = os.path.join(HERE, "ref_host_frame.cpp")
    deps = srcs + [drv, B.LIB_PATH, os.path.join(REPO, "include", "dfusion.h")] + \
           [os.path.join(r, f) for d in (os.path.join(HERE, "shim"), os.path.join(REPO, "tests", "opencv_stub")) for r, _, fs in os.walk(d) for f in fs]
    if not force and os.path.exists(APP) and os.path.exists(LIB) and min(os.path.getmtime(APP), os.path.getmtime(LIB)) >= max(os.path.getmtime(d) for d in deps):
        return APP
    os.makedirs(OUT, exist_ok=True)
    common, rocm = flags()
    link = ["-L", os.path.join(REPO, "dynamicfusion_amd"), "-ldfusion_hip", "-L", os.path.join(rocm, "lib"), "-lamdhip64"]
    rpath = ["-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/../../../dynamicfusion_amd"]
    subprocess.check_call(common + ["-fPIC", "-shared"] + srcs + ["-o", LIB] + link + rpath)
    subprocess.check_call(common + [drv, "-o", APP, "-L", OUT, "-lkfusion_refhost"] + link + rpath)
    return APP


def undefined_device_symbols():
    """kfusion::device::* symbols the reference's host objects need that the library does not define (must be empty but for the ones the
    bridge's header comment lists)."""
    r = subprocess.run(["nm", "-D", "--undefined-only", "-C", LIB], capture_output=True, text=True)
    return [l.split(" U ")[-1].strip() for l in r.stdout.splitlines() if "kfusion::device::" in l]


if __name__ == "__main__":
    print(build(force=True))
    print("undefined kfusion::device symbols:", undefined_device_symbols())
