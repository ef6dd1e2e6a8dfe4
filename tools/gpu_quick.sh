#!/bin/bash
# Quick GPU check: tests, A/B of variant libraries, the default bench line, a kernel trace of the bench command.  Tag $1; env AB_RIGID / AB_WARP = variant tags.
set -u
T=${1:-q}
mkdir -p gpurun_out build; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
if [ "${SKIP_TESTS:-0}" != "1" ]; then echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ${PYTEST_ARGS:-} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${T}_pytest_gpu.txt | tail -25; fi
if [ -n "${AB_RIGID:-}" ]; then echo "== A/B rigid: $AB_RIGID"; (timeout 600 python tools/ab_rigid_libs.py 512 $AB_RIGID 2>&1 | grep -v amdgpu.ids) | tee gpurun_out/${T}_ab_rigid.txt; fi
if [ -n "${AB_WARP:-}" ]; then echo "== A/B warped: $AB_WARP"; (timeout 600 python tools/ab_libs.py 512 $AB_WARP 2>&1 | grep -v amdgpu.ids) | tee gpurun_out/${T}_ab_warp.txt; fi
echo "== bench 512"; timeout 900 python bench.py --steps 20 --warmup 5 ${BENCH_ARGS:-} 2>gpurun_out/${T}_bench_512.err | grep -v amdgpu.ids | tail -1 > gpurun_out/${T}_bench_512.json; cut -c1-1300 gpurun_out/${T}_bench_512.json; tail -3 gpurun_out/${T}_bench_512.err
echo "== rocprof kernel trace of the bench command"
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o trace -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kinfu --no-extras > $R/gpurun_out/rocprof.log 2>&1)
tail -1 gpurun_out/rocprof.log | cut -c1-300
cp $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_kernel_stats.csv
head -12 gpurun_out/${T}_kernel_stats.csv | cut -c1-160
