#!/bin/bash
# Measurement session of a round: tests, bench, rocprofv3 kernel trace, PMC passes.  Everything lands in gpurun_out/ under names
# prefixed with the round tag (default r02); copy what is to be judged into profiles/.
set -u
T=${1:-r02}
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${T}_pytest_gpu.txt | tail -3
echo "== rocprof kernel trace of the bench command"
rm -rf gpurun_out/prof gpurun_out/pmc
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o trace -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kinfu > $R/gpurun_out/rocprof.log 2>&1)
tail -1 gpurun_out/rocprof.log | cut -c1-200
cp $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_kernel_stats.csv
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $pass | cut -d' ' -f1)
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/pmc/$tag -o p -- python $R/tools/pmc_run.py 512 3 tables > $R/gpurun_out/pmc_$tag.log 2>&1)
  tail -1 gpurun_out/pmc_$tag.log | cut -c1-120
done
python tools/pmc_summary.py gpurun_out/pmc --json gpurun_out/pmc_latest.json --config 512 --tag "round ${T#r0}" > gpurun_out/${T}_pmc_512.txt 2>&1; tail -3 gpurun_out/${T}_pmc_512.txt
cp gpurun_out/pmc_latest.json profiles/pmc_latest.json      # bench.py reads roofline.traffic from here (this run's counters)
echo "== bench 512"; timeout 900 python bench.py --steps 40 --warmup 5 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/${T}_bench_512.json; cut -c1-400 gpurun_out/${T}_bench_512.json
echo "== bench 256"; timeout 300 python bench.py --steps 40 --warmup 5 --config 256 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/${T}_bench_256.json; cut -c1-300 gpurun_out/${T}_bench_256.json
echo "== bench 1024 (the 8-GPU stress config on ONE GPU)"; timeout 600 python bench.py --steps 10 --warmup 2 --config 1024 --no-cpu-baseline --no-kinfu 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/${T}_bench_1024.json; cut -c1-300 gpurun_out/${T}_bench_1024.json
echo "== kinfu frame profile"
bash tools/kinfu_profile.sh > gpurun_out/kinfu_profile.log 2>&1; grep "ms/frame" gpurun_out/kinfu_profile.log | cut -c1-160
cp $(find gpurun_out -path "*kinfu*" -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_kinfu_kernel_stats.csv 2>/dev/null
python tools/kinfu_probe.py > gpurun_out/${T}_kinfu_frame_ms.txt 2>&1; cut -c1-170 gpurun_out/${T}_kinfu_frame_ms.txt
