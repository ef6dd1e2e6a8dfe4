#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 600 python tools/ab_libs.py 512 noidspf nocodes 2>&1 | grep -v amdgpu.ids) | tee gpurun_out/s20_ab.txt
echo "== parity"; timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cull_fuzz.py tests/test_gpu_refcu.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s20_pytest.txt | tail -6
