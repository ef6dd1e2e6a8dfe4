#!/bin/bash
# round 5, session 1: parity suite on the new kernels, same-box A/Bs of the rigid pair/row plan and of the warped sweep's variants
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s1_pytest_gpu.txt | tail -15
echo "== A/B rigid"; (timeout 400 python tools/ab_rigid_libs.py 512 pair0 slack0 slack2 slack4 2>&1 | grep -v amdgpu.ids) | tee gpurun_out/s1_ab_rigid.txt
echo "== A/B warped"; (timeout 600 python tools/ab_libs.py 512 w_base w_nodiv w_nofuse w_nohoist w_split w_noslp 2>&1 | grep -v amdgpu.ids) | tee gpurun_out/s1_ab_warp.txt
echo "== bench 512"; timeout 600 python bench.py --steps 20 --warmup 5 --no-kinfu 2>gpurun_out/s1_bench.err | grep -v amdgpu.ids | tail -1 > gpurun_out/s1_bench_512.json; cut -c1-1500 gpurun_out/s1_bench_512.json; tail -3 gpurun_out/s1_bench.err
