"""TEST INFRASTRUCTURE: builds the REFERENCE'S OWN apps/demo.cpp, unmodified, against this repository's kfusion mirror.

    /root/reference/apps/demo.cpp  +  dynamicfusion_amd/host (mirror, compiled with -DKFUSION_USE_OPENCV)  +  tests/opencv_stub
    ->  tests/_demo_ref/libkfusion_hip_cv.so, tests/_demo_ref/demo_ref

OpenCV is not installed in this image; tests/opencv_stub supplies the OpenCV names demo.cpp uses (windows become files).  The source
is read where it lies under /root/reference -- nothing is copied -- and the outputs are git-ignored but travel to the GPU box with the
gpurun snapshot, where tests/test_demo_ref.py runs the binary (/root/reference does not exist there).
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
OUT = os.path.join(HERE, "_demo_ref")
DEMO_SRC = "/root/reference/apps/demo.cpp"
LIB = os.path.join(OUT, "libkfusion_hip_cv.so")
APP = os.path.join(OUT, "demo_ref")
STUB = os.path.join(HERE, "opencv_stub")
HOST = os.path.join(REPO, "dynamicfusion_amd", "host")


def flags():
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    return ["g++", "-std=c++17", "-O2", "-D__HIP_PLATFORM_AMD__", "-DKFUSION_USE_OPENCV", "-I", STUB, "-I", os.path.join(HOST, "include"),
            "-I", os.path.join(REPO, "include"), "-I", os.path.join(rocm, "include")], rocm


def have_reference():
    return os.path.exists(DEMO_SRC)


def syntax_check():
    """g++ -fsyntax-only of the reference file against mirror + stub; returns (ok, compiler output)."""
    common, _ = flags()
    r = subprocess.run(common + ["-fsyntax-only", DEMO_SRC], capture_output=True, text=True)
    return r.returncode == 0, r.stdout + r.stderr


def build(force=False):
    """Returns the path of the demo binary (built here when the reference is present, else whatever travelled with the snapshot)."""
    if not have_reference():
        return APP if os.path.exists(APP) else None
    from dynamicfusion_amd import build as B
    B.build_library()
    src = os.path.join(HOST, "src", "kfusion_hip.cpp")
    stub_cpp = os.path.join(STUB, "opencv_stub.cpp")
    deps = [DEMO_SRC, src, stub_cpp, B.LIB_PATH] + [os.path.join(r, f) for d in (os.path.join(HOST, "include"), STUB) for r, _, fs in os.walk(d) for f in fs]
    if not force and os.path.exists(APP) and os.path.exists(LIB) and min(os.path.getmtime(APP), os.path.getmtime(LIB)) >= max(os.path.getmtime(d) for d in deps):
        return APP
    os.makedirs(OUT, exist_ok=True)
    common, rocm = flags()
    link = ["-L", os.path.join(REPO, "dynamicfusion_amd"), "-ldfusion_hip", "-L", os.path.join(rocm, "lib"), "-lamdhip64"]
    rpath = ["-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/../../dynamicfusion_amd"]
    subprocess.check_call(common + ["-fPIC", "-shared", src, stub_cpp, "-o", LIB] + link + rpath)
    subprocess.check_call(common + [DEMO_SRC, "-o", APP, "-L", OUT, "-lkfusion_hip_cv"] + link + rpath)
    return APP


if __name__ == "__main__":
    import sys
    sys.path.insert(0, REPO)
    print(syntax_check())
    print(build(force=True))
