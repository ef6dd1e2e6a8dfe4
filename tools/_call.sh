export TMPDIR=/tmp; mkdir -p gpurun_out; R=$PWD
(timeout 600 python tools/ab_rigid_libs.py 512 head sat2 sat2/nosat sat4w6 sat4w6/nosat pat64 pat16 sat2noload sat2nostore sat2nodists sat2nomem 2>&1 | grep -v amdgpu.ids) > gpurun_out/c2_ab_rigid.txt
cat gpurun_out/c2_ab_rigid.txt
rm -rf gpurun_out/pmc_rigid
for t in head sat2 "sat2 nosat"; do
  tag=$(echo $t | tr ' ' '_')
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_rigid/$tag/a -o p -- python $R/tools/pmc_rigid.py 512 $t > $R/gpurun_out/pmc_rigid_$tag.log 2>&1)
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_rigid/$tag/b -o p -- python $R/tools/pmc_rigid.py 512 $t >> $R/gpurun_out/pmc_rigid_$tag.log 2>&1)
  echo "== $t"; python tools/pmc_summary.py gpurun_out/pmc_rigid/$tag 2>&1 | grep -A16 "df_integrate_rigid" | head -40
done > gpurun_out/c2_pmc_rigid.txt 2>&1
cat gpurun_out/c2_pmc_rigid.txt
