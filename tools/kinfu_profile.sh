#!/bin/bash
# rocprofv3 kernel stats of the C++ KinFu mirror (kinfu_headless, 512^3 / 3 m, 640x480, 12 synthetic frames).
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; export TMPDIR=/tmp
mkdir -p $R/gpurun_out
python - <<PY
import os, sys, numpy as np
sys.path.insert(0, "$R")
from dynamicfusion_amd import build, synth
build.build_host()
cfg = synth.CONFIGS["512"]
with open("/tmp/kin512.bin", "wb") as f:
    f.write(np.asarray(cfg.intr, np.float32).tobytes())
    for i in range(12): f.write(synth.depth_frame(cfg, i).tobytes())
PY
rm -rf $R/gpurun_out/kinfu_prof
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kinfu_prof -o kinfu -- $R/dynamicfusion_amd/host/kinfu_headless 640 480 12 512 3.0 /tmp/kin512.bin /tmp/kout.bin ${1:-} 2>&1 | tail -2)
head -25 $R/gpurun_out/kinfu_prof/kinfu_kernel_stats.csv | cut -c1-170
