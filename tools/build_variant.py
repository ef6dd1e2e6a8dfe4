#!/usr/bin/env python3
"""Build an experimental variant of the HIP library next to the product one:  tools/build_variant.py TAG -DNAME=1 ...
-> build/libdfusion_hip_TAG.so (git-ignored, ships with gpurun).  Used with tools/ab_libs.py for same-box A/B timing."""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from dynamicfusion_amd import build as B
tag, defs = sys.argv[1], sys.argv[2:]
out = os.path.join(REPO, "build", "libdfusion_hip_%s.so" % tag)
os.makedirs(os.path.dirname(out), exist_ok=True)
cmd = [B._hipcc()] + B.HIPCC_FLAGS + defs + ["-I", os.path.join(REPO, "include"), "-I", B.CSRC]
cmd += [os.path.join(B.CSRC, s) for s in B.SOURCES] + ["-o", out]
subprocess.check_call(cmd)
print(out)
