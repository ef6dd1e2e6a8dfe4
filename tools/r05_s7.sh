#!/bin/bash
# round 5, final measurement pass: smoke(), the driver's bench command (timed), rocprofv3 stats + frame trace, PMC passes over the bench poses
set -u
T=r05
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
echo "== rocprof kernel trace of the bench command"
rm -rf gpurun_out/prof gpurun_out/pmc
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o trace -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kinfu > $R/gpurun_out/rocprof.log 2>&1)
cp $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_kernel_stats.csv
python tools/frame_trace.py $(find gpurun_out/prof -name "*kernel_trace.csv" | head -1) > gpurun_out/${T}_frame_trace.txt 2>&1; head -1 gpurun_out/${T}_frame_trace.txt | cut -c1-400
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum" "TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_BUSY_sum TCC_EA0_WRREQ_STALL_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/pmc/$tag -o p -- python $R/tools/pmc_run.py 512 20 bench > $R/gpurun_out/pmc_$tag.log 2>&1)
done
python tools/pmc_summary.py gpurun_out/pmc --last 20 --json gpurun_out/pmc_latest.json --config 512 --tag "round 5" > gpurun_out/${T}_pmc_512.txt 2>&1; tail -3 gpurun_out/${T}_pmc_512.txt
cp gpurun_out/pmc_latest.json profiles/pmc_latest.json
echo "== bench 512 (the driver's command), wall clock"; SECONDS=0; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/bench.err | grep -v amdgpu.ids | tail -1 > gpurun_out/${T}_bench_512.json; echo "bench wall clock: $SECONDS s"; cut -c1-300 gpurun_out/${T}_bench_512.json
