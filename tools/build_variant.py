#!/usr/bin/env python3
"""Build an experimental variant of the HIP library next to the product one:
    tools/build_variant.py TAG [--only dfusion_volume.hip[,dfusion_warp.hip]] -DNAME=1 ...
-> build/libdfusion_hip_TAG.so (git-ignored, ships with gpurun).  Used with tools/ab_libs.py / tools/ab_rigid_libs.py for same-box
A/B timing.  With --only, the -D flags apply to the listed sources and every other translation unit is taken from a cached plain
object under build/obj/ (rebuilt when its source or a header is newer), so a variant costs one file's compile time."""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from dynamicfusion_amd import build as B
args = sys.argv[1:]
tag = args.pop(0)
only = None
if args and args[0] == "--only":
    args.pop(0); only = args.pop(0).split(",")
defs = args
out = os.path.join(REPO, "build", "libdfusion_hip_%s.so" % tag)
obj_dir = os.path.join(REPO, "build", "obj")
os.makedirs(obj_dir, exist_ok=True)
cflags = [f for f in B.HIPCC_FLAGS if f != "-shared"] + ["-I", os.path.join(REPO, "include"), "-I", B.CSRC]
hdrs = [h if os.path.isabs(h) else os.path.join(B.CSRC, h) for h in B.HEADERS]


def obj_for(src, extra, name):
    o = os.path.join(obj_dir, name)
    srcp = os.path.join(B.CSRC, src)
    if extra or not os.path.exists(o) or any(os.path.getmtime(d) > os.path.getmtime(o) for d in [srcp] + hdrs):
        subprocess.check_call([B._hipcc()] + cflags + extra + ["-c", srcp, "-o", o])
    return o


if only is None:
    cmd = [B._hipcc()] + B.HIPCC_FLAGS + defs + ["-I", os.path.join(REPO, "include"), "-I", B.CSRC]
    cmd += [os.path.join(B.CSRC, s) for s in B.SOURCES] + ["-o", out]
    subprocess.check_call(cmd)
else:
    objs = []
    for s in B.SOURCES:
        if s in only:
            objs.append(obj_for(s, defs, "%s.%s.o" % (s, tag)))
        else:
            objs.append(obj_for(s, [], "%s.o" % s))
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
print(out)
