#!/bin/bash
# Round-1 measurement session: tests, bench, rocprofv3 kernel trace, PMC passes.  Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tee gpurun_out/pytest_gpu.log | tail -3
echo "== bench 512"; timeout 900 python bench.py --steps 40 --warmup 5 2>&1 | tee gpurun_out/bench.log | tail -2
echo "== bench 256"; timeout 300 python bench.py --steps 40 --warmup 5 --config 256 2>&1 | tee gpurun_out/bench_256.log | tail -2
echo "== bench N=2 code path on one GPU (gloo stand-in for RCCL)"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/bench_multi_smoke.py --gpus 2 --steps 3 --warmup 1 --no-extras 2>&1 | tail -1 | cut -c1-300 | tee gpurun_out/bench_multi_smoke.log
echo "== rocprof kernel trace"
rm -rf gpurun_out/prof gpurun_out/pmc
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o trace -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kinfu > $R/gpurun_out/rocprof.log 2>&1)
tail -1 gpurun_out/rocprof.log
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $pass | cut -d' ' -f1)
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/pmc/$tag -o p -- python $R/tools/pmc_run.py 512 3 tables > $R/gpurun_out/pmc_$tag.log 2>&1)
  tail -1 gpurun_out/pmc_$tag.log
done
find gpurun_out/prof gpurun_out/pmc -name "*.csv" | head -20
echo "== kinfu frame profile"
bash tools/kinfu_profile.sh > gpurun_out/kinfu_profile.log 2>&1; grep "ms/frame" gpurun_out/kinfu_profile.log | cut -c1-160
python tools/kinfu_probe.py > gpurun_out/kinfu_probe.log 2>&1; cat gpurun_out/kinfu_probe.log | cut -c1-170
