#!/bin/bash
# round 5, session 2: full parity suite (new: the reference's host class over hip_bridge.cpp, selftest [8]); the f32-division variant once more
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s2_pytest_gpu.txt | tail -15
echo "== A/B warped"; (timeout 600 python tools/ab_libs.py 512 w_div w_base 2>&1 | grep -v amdgpu.ids) | tee gpurun_out/s2_ab_warp.txt
