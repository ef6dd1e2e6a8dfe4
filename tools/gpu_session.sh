#!/bin/bash
# One gpurun call: smoke -> parity tests -> probe sweep -> bench -> rocprof.  Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tee gpurun_out/pytest_gpu.log | tail -40
echo "== probe 256"; timeout 300 python tools/gpu_probe.py 256 2>&1 | tee gpurun_out/probe_256.log | tail -40
echo "== probe 512"; timeout 600 python tools/gpu_probe.py 512 2>&1 | tee gpurun_out/probe_512.log | tail -40
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 3 --rigid 2>&1 | tee gpurun_out/bench.log | tail -5
echo "== rocprof"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1); tail -3 gpurun_out/rocprof.log
find gpurun_out/prof -name "*stats*" | head; 
