#!/bin/bash
# Measurement session of a round: tests, bench, rocprofv3 kernel trace, PMC passes, probes.  Everything lands in gpurun_out/ under
# names prefixed with the round tag (default r03); copy what is to be judged into profiles/.
set -u
T=${1:-r06}
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${T}_pytest_gpu.txt | tail -3
echo "== rocprof kernel trace of the bench command"
rm -rf gpurun_out/prof gpurun_out/pmc gpurun_out/pmc_rigid
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o trace -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kinfu --no-other-configs > $R/gpurun_out/rocprof.log 2>&1)
tail -1 gpurun_out/rocprof.log | cut -c1-200
cp $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_kernel_stats.csv
python tools/frame_trace.py $(find gpurun_out/prof -name "*kernel_trace.csv" | head -1) > gpurun_out/${T}_frame_trace.txt 2>&1; head -3 gpurun_out/${T}_frame_trace.txt
# PMC: separate --pmc passes (MI355X_MICROARCH.md), --kernel-trace only beside them
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum" "TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_BUSY_sum TCC_EA0_WRREQ_STALL_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/pmc/$tag -o p -- python $R/tools/pmc_run.py 512 20 bench > $R/gpurun_out/pmc_$tag.log 2>&1)
  tail -1 gpurun_out/pmc_$tag.log | cut -c1-120
done
python tools/pmc_summary.py gpurun_out/pmc --last 20 --json gpurun_out/pmc_latest.json --config 512 --tag "round ${T#r0}" > gpurun_out/${T}_pmc_512.txt 2>&1; tail -3 gpurun_out/${T}_pmc_512.txt
cp gpurun_out/pmc_latest.json profiles/pmc_latest.json      # bench.py reads roofline.traffic from here (this run's counters, stamped with the source's sha256)
echo "== bench 512 (the driver's command)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/${T}_bench_512.json; cut -c1-400 gpurun_out/${T}_bench_512.json
echo "== A/B against round 5's library (build/libdfusion_hip_r05.so: git archive f5aaea7 | hipcc), same box, interleaved"
if [ -f build/libdfusion_hip_r05.so ]; then
  (timeout 300 python tools/ab_rounds.py 512 r05 2>&1 | grep -v amdgpu.ids) | tee gpurun_out/${T}_ab_r05_512.txt
  (timeout 300 python tools/ab_rounds.py 512 --nodes 8000 r05 2>&1 | grep -v amdgpu.ids) | tee gpurun_out/${T}_ab_r05_512_nodes8000.txt
  (timeout 600 python tools/ab_rounds.py 1024 r05 2>&1 | grep -v amdgpu.ids) | tee gpurun_out/${T}_ab_r05_1024.txt
fi
echo "== A/B: look-ahead builds on / off, same box"; bash tools/ab_bench.sh ${T} "" "--no-prefetch" 2 > /dev/null; cat gpurun_out/${T}_ab_bench.txt
echo "== bench 256"; timeout 300 python bench.py --steps 40 --warmup 5 --config 256 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/${T}_bench_256.json; cut -c1-300 gpurun_out/${T}_bench_256.json
echo "== bench 1024 (the 8-GPU stress config on ONE GPU)"; timeout 900 python bench.py --steps 10 --warmup 2 --config 1024 --no-kinfu 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/${T}_bench_1024.json; cut -c1-300 gpurun_out/${T}_bench_1024.json
echo "== kinfu frame profile"
bash tools/kinfu_profile.sh > gpurun_out/kinfu_profile.log 2>&1; grep "ms/frame" gpurun_out/kinfu_profile.log | cut -c1-160
cp $(find gpurun_out -path "*kinfu*" -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_kinfu_kernel_stats.csv 2>/dev/null
python tools/kinfu_probe.py > gpurun_out/${T}_kinfu_frame_ms.txt 2>&1; cut -c1-170 gpurun_out/${T}_kinfu_frame_ms.txt
echo "== probes: copy forms, RMW access patterns"
hipcc --offload-arch=gfx950 -O3 tools/copy_probe.hip -o build/copy_probe 2>/dev/null; timeout 200 build/copy_probe > gpurun_out/${T}_copy_probe.txt 2>&1; grep "1 GiB" -A12 gpurun_out/${T}_copy_probe.txt | grep "chunk x4 nt\|stride x1 " | head -4
hipcc --offload-arch=gfx950 -O3 tools/rmw_probe.hip -o build/rmw_probe 2>/dev/null; timeout 100 build/rmw_probe > gpurun_out/${T}_rmw_probe.txt 2>&1; head -13 gpurun_out/${T}_rmw_probe.txt
echo "== per-wave timelines"
python tools/build_variant.py trace --only dfusion_volume.hip,dfusion_warp.hip -DDF_TRACE_WG=1 > /dev/null
(timeout 300 python tools/trace_sweep.py 512 2>&1 | grep -v amdgpu.ids) > gpurun_out/${T}_trace_sweep_512.txt; head -2 gpurun_out/${T}_trace_sweep_512.txt
(timeout 300 python tools/trace_sweep.py 512 rigid 2>&1 | grep -v amdgpu.ids) > gpurun_out/${T}_trace_rigid_512.txt; head -1 gpurun_out/${T}_trace_rigid_512.txt
python tools/build_variant.py vtrace --only dfusion_warp.hip -DDF_TRACE_VERDICT=1 > /dev/null
(timeout 300 python tools/trace_verdict.py 512 2>&1 | grep -v amdgpu.ids) > gpurun_out/${T}_trace_verdict_512.txt; head -2 gpurun_out/${T}_trace_verdict_512.txt
python tools/build_variant.py rtrace --only dfusion_raycast.hip -DDF_TRACE_RAYCAST=1 > /dev/null
(timeout 300 python tools/trace_raycast.py 512 2>&1 | grep -v amdgpu.ids) > gpurun_out/${T}_trace_raycast_512.txt; head -3 gpurun_out/${T}_trace_raycast_512.txt
echo "== predicted Z-slab scaling (measured per-slab kernels + collective model)"
for kind in measured balanced uniform; do
  (timeout 600 python tools/scale_model.py 512 $kind 2>&1 | grep -v amdgpu.ids) | tee gpurun_out/${T}_scale_model_512_$kind.txt | head -7
  cp gpurun_out/scale_model_512_$kind.json gpurun_out/${T}_scale_model_512_$kind.json
done
echo "== the frame after a node-set change, per kernel"
rm -rf gpurun_out/prof_nc
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_nc -o trace -- python $R/tools/nodes_changed.py 512 3 > $R/gpurun_out/${T}_nodes_changed.txt 2>&1)
cp $(find gpurun_out/prof_nc -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_nodes_changed_kernel_stats.csv; grep "set_nodes" gpurun_out/${T}_nodes_changed.txt
echo "== python bench.py --gpus 8 on ONE GPU: the launcher + the N = 8 code path (oversubscribed: gloo, host-staged collectives; not a timing)"
(timeout 900 python bench.py --gpus 8 --steps 3 --warmup 1 --config 512 --no-extras --no-cpu-baseline 2>/dev/null | tail -1) > gpurun_out/${T}_bench_n8_one_gpu_smoke.txt; cut -c1-300 gpurun_out/${T}_bench_n8_one_gpu_smoke.txt
echo "== hipMallocAsync repro"
hipcc --offload-arch=gfx950 -O2 tools/async_scratch_repro.hip -o build/async_scratch_repro 2>/dev/null; timeout 300 build/async_scratch_repro 2000 > gpurun_out/${T}_async_scratch_repro.txt 2>&1; tail -1 gpurun_out/${T}_async_scratch_repro.txt
