"""The reference's OWN apps/demo.cpp against this repository's kfusion mirror (north star: "apps/demo.cpp links unchanged").

OpenCV is absent from this image, so the file is compiled against tests/opencv_stub (the OpenCV names it uses; windows are files) and
the mirror built with -DKFUSION_USE_OPENCV, where getNodesAsMat() / get_cloud_host() return cv::Mat and KinFu::Ptr is cv::Ptr<KinFu>
exactly as in the reference's headers (kfusion/include/kfusion/kinfu.hpp:52, warp_field.hpp:78, cuda/tsdf_volume.hpp:24-28).
  * CPU (where /root/reference exists): the file compiles UNMODIFIED and links into tests/_demo_ref/demo_ref;
  * GPU: the binary -- built in the container, travelled with the snapshot -- runs its own frame loop on a synthetic sequence; every
    "Scene" window it shows and the "warp_field" cloud equal, byte for byte, what host/apps/demo_calls.cpp (the look-alike of round 2,
    same calls through the non-OpenCV build of the mirror) produces on the same frames."""
import os
import struct
import subprocess

import numpy as np
import pytest

import build_demo_ref as D
from dynamicfusion_amd import build, synth

F32 = np.float32


@pytest.mark.skipif(not D.have_reference(), reason="/root/reference not present")
def test_reference_demo_cpp_compiles_unmodified_and_links():
    ok, log = D.syntax_check()                                  # g++ -fsyntax-only /root/reference/apps/demo.cpp
    assert ok, log
    app = D.build()
    assert app and os.path.exists(app) and os.path.exists(D.LIB)
    # the binary's undefined kfusion symbols resolve against the mirror: KinFu, both renderImage overloads, getNodesAsMat -> cv::Mat
    syms = subprocess.run(["nm", "-C", "--undefined-only", app], capture_output=True, text=True).stdout
    for s in ("kfusion::KinFu::KinFu(kfusion::KinFuParams const&)", "kfusion::KinFuParams::default_params_dynamicfusion()",
              "kfusion::KinFu::renderImage(kfusion::cuda::DeviceArray2D<kfusion::cuda::RGB>&, int)",
              "kfusion::KinFu::renderImage(kfusion::cuda::DeviceArray2D<kfusion::cuda::RGB>&, cv::Affine3<float> const&, int)",
              "kfusion::WarpField::getNodesAsMat() const", "kfusion::KinFu::getCameraPose(int) const"):
        assert s in syms, s
    defined = subprocess.run(["nm", "-C", "--defined-only", "-D", D.LIB], capture_output=True, text=True).stdout
    assert "kfusion::WarpField::getNodesAsMat() const" in defined and "kfusion::cuda::TsdfVolume::get_cloud_host() const" in defined


def write_raw(path, arr, cv_type):
    with open(path, "wb") as f:
        f.write(b"DFRW" + struct.pack("<iii", arr.shape[0], arr.shape[1], cv_type))
        f.write(np.ascontiguousarray(arr).tobytes())


@pytest.mark.gpu
def test_reference_demo_binary_runs_and_matches_demo_calls(tmp_path):
    app = D.build()
    if not app:
        pytest.skip("tests/_demo_ref/demo_ref was not built (needs /root/reference in the build container)")
    # demo.cpp takes its parameters from default_params_dynamicfusion(): 640 x 480, 256^3 / 1 m, fx = fy = 570.342 (kinfu.cpp:15-50)
    cfg = synth.Config(256, 1.0, cols=640, rows=480, nodes=0, k=8)
    frames = 4
    depths = [synth.depth_frame(cfg, 2 * f) for f in range(frames)]
    os.makedirs(tmp_path / "data" / "depth"); os.makedirs(tmp_path / "data" / "color"); os.makedirs(tmp_path / "out")
    for i, d in enumerate(depths):
        write_raw(tmp_path / "data" / "depth" / ("%04d.png" % i), d, 2)                                  # CV_16UC1
        write_raw(tmp_path / "data" / "color" / ("%04d.png" % i), np.zeros((cfg.rows, cfg.cols * 3), np.uint8).reshape(cfg.rows, -1), 16)   # CV_8UC3
    env = dict(os.environ, DFUSION_CVSTUB_OUT=str(tmp_path / "out"))
    r = subprocess.run([app, str(tmp_path / "data")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    shown = frames - 1                                          # operator() returns false on frame 0 (kinfu.cpp:250)
    scene = np.fromfile(tmp_path / "out" / "Scene.bin", np.uint8).reshape(shown, cfg.rows, 2 * cfg.cols, 4)
    warp = np.fromfile(tmp_path / "out" / "warp_field.bin", F32).reshape(-1, 3)
    # the look-alike through the plain build of the mirror, same frames, same parameters
    build.build_host()
    fin, prefix = str(tmp_path / "demo_in.bin"), str(tmp_path / "calls")
    with open(fin, "wb") as f:
        f.write(np.asarray(cfg.intr, F32).tobytes())
        for d in depths:
            f.write(d.tobytes())
    r2 = subprocess.run([build.HOST_DEMO_CALLS, str(cfg.cols), str(cfg.rows), str(frames), str(cfg.dims[0]), str(cfg.size), fin, prefix],
                        capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0 and "demo_calls ok" in r2.stdout, r2.stdout + r2.stderr
    views = np.fromfile(prefix + ".views.bin", np.uint8).reshape(shown, 2, cfg.rows, 2 * cfg.cols, 4)
    nodes = np.fromfile(prefix + ".nodes.bin", F32).reshape(-1, 3)
    assert np.array_equal(scene, views[:, 0])                   # renderImage(view_device_, 3), demo.cpp:50
    assert len(warp) > 50 and np.array_equal(warp.view(np.uint32), nodes.view(np.uint32))     # getNodesAsMat(), demo.cpp:67
    assert len(np.unique(scene[-1][..., 0])) > 30               # a shaded surface, not a blank window
