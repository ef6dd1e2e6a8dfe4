#!/usr/bin/env python3
"""Run the C++ KinFu mirror (kinfu_headless) on the synthetic 640x480 sequence at 256^3 / 1 m and 512^3 / 3 m and print its per-frame wall clock."""
import os, subprocess, sys, tempfile
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from dynamicfusion_amd import build, synth
build.build_host()
for name in ("256", "512"):
    cfg = synth.CONFIGS[name]; frames = 12
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            f.write(np.asarray(cfg.intr, np.float32).tobytes())
            for i in range(frames):
                f.write(synth.depth_frame(cfg, i).tobytes())
        for extra in ([], ["nosolver"], ["host"], ["warped"], ["warped-host"]):
            r = subprocess.run([build.HOST_KINFU_APP, str(cfg.cols), str(cfg.rows), str(frames), str(cfg.dims[0]), str(cfg.size), fin, fout] + extra,
                               capture_output=True, text=True, timeout=600)
            print(name, extra[0] if extra else "default", r.stdout.strip(), r.stderr.strip()[-400:])
