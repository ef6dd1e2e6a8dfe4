#!/bin/bash
# round 5, session 3: the new parity tests, the default bench line (long sweep, stage times), the N = 2 / 8 launcher smokes with variants
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest gpu (new tests first)"; timeout 1500 python -m pytest tests/test_gpu_refcu.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_sharded.py -m gpu -q -x -p no:cacheprovider --durations=8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s3_pytest_gpu.txt | tail -25
echo "== bench 512 (the driver's command)"; /usr/bin/time -v timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/s3_bench.err | grep -v amdgpu.ids | tail -1 > gpurun_out/s3_bench_512.json; cut -c1-2500 gpurun_out/s3_bench_512.json; grep -E "Elapsed|Maximum resident" gpurun_out/s3_bench.err; tail -3 gpurun_out/s3_bench.err
echo "== bench --gpus 2 on one GPU (oversubscribed smoke)"; (timeout 900 python bench.py --gpus 2 --steps 4 --warmup 1 --no-extras --no-cpu-baseline --long-frames 0 2>gpurun_out/s3_n2.err | tail -1) > gpurun_out/s3_bench_n2_one_gpu_smoke.txt; cut -c1-3000 gpurun_out/s3_bench_n2_one_gpu_smoke.txt; tail -5 gpurun_out/s3_n2.err
echo "== bench --gpus 8 on one GPU (oversubscribed smoke)"; (timeout 1200 python bench.py --gpus 8 --steps 3 --warmup 1 --no-extras --no-cpu-baseline --long-frames 0 2>gpurun_out/s3_n8.err | tail -1) > gpurun_out/s3_bench_n8_one_gpu_smoke.txt; cut -c1-1500 gpurun_out/s3_bench_n8_one_gpu_smoke.txt; tail -5 gpurun_out/s3_n8.err
