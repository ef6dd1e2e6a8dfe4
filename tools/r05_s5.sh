#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s5_pytest_gpu.txt | tail -15
