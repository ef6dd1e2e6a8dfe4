#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== A/B warped table layout"; (timeout 600 python tools/ab_libs.py 512 tabrow tabpatch 2>&1 | grep -v amdgpu.ids) | tee gpurun_out/s6_ab_tab.txt
(timeout 600 python tools/ab_libs.py 256 tabrow tabpatch 2>&1 | grep -v amdgpu.ids) | tee -a gpurun_out/s6_ab_tab.txt
