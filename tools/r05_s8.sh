#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== new parity test + neighbours"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_gpu_cxx_host.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s8_pytest.txt | tail -12
echo "== A/B: frames pipelined over two streams (A) against one stream (B), same box, interleaved"
rm -f gpurun_out/s8_ab_bench.txt; bash tools/ab_bench.sh s8 "" "--no-pipeline" 3 > /dev/null 2>&1; cat gpurun_out/s8_ab_bench.txt
echo "== the driver's command"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/s8_bench.err | tail -1 > gpurun_out/s8_bench_512.json; cut -c1-900 gpurun_out/s8_bench_512.json; tail -2 gpurun_out/s8_bench.err
