#!/bin/bash
# One gpurun call: bench + rocprofv3 kernel-trace summary.  Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
echo "== bench"; timeout 900 python bench.py --steps 40 --warmup 5 --rigid 2>&1 | tee gpurun_out/bench.log | tail -3
echo "== rocprof kernel trace"
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r01 -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/rocprof.log 2>&1)
tail -2 gpurun_out/rocprof.log
find gpurun_out/prof -type f | head -20
