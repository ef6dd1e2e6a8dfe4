#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== parity"; timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cxx_host.py tests/test_gpu_solver.py tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s10_pytest.txt | tail -12
echo "== A/B: frames pipelined over two streams (A) against one stream (B), same box, interleaved"
rm -f gpurun_out/s10_ab_bench.txt; bash tools/ab_bench.sh s10 "" "--no-pipeline" 3 > /dev/null 2>&1; cat gpurun_out/s10_ab_bench.txt
echo "== frame trace of the pipelined run"
rm -rf gpurun_out/prof10; R=${GRAFT_REPO_ROOT:-$PWD}
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof10 -o trace -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kinfu --no-extras --long-frames 0 > $R/gpurun_out/rocprof10.log 2>&1)
tail -1 gpurun_out/rocprof10.log | cut -c1-600
python tools/frame_trace.py $(find gpurun_out/prof10 -name "*kernel_trace.csv" | head -1) > gpurun_out/s10_frame_trace.txt 2>&1; sed -n 1,2p gpurun_out/s10_frame_trace.txt | cut -c1-400; sed -n 14,20p gpurun_out/s10_frame_trace.txt | cut -c1-420
