#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== A/B: frames pipelined over two streams (A) against one stream (B), same box, interleaved"
rm -f gpurun_out/s9_ab_bench.txt; bash tools/ab_bench.sh s9 "" "--no-pipeline" 3 > /dev/null 2>&1; cat gpurun_out/s9_ab_bench.txt
echo "== frame trace of the pipelined run"
rm -rf gpurun_out/prof9; R=${GRAFT_REPO_ROOT:-$PWD}
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof9 -o trace -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kinfu --no-extras --long-frames 0 > $R/gpurun_out/rocprof9.log 2>&1)
python tools/frame_trace.py $(find gpurun_out/prof9 -name "*kernel_trace.csv" | head -1) > gpurun_out/s9_frame_trace.txt 2>&1; sed -n 1,2p gpurun_out/s9_frame_trace.txt | cut -c1-400; sed -n 14,22p gpurun_out/s9_frame_trace.txt | cut -c1-420
