#!/bin/bash
# tools/clock_probe.sh CONFIG TAG [TAG ...]   ("product" = the product library): cycles per launch, duration and clock of the warped sweep
# through each build, one rocprofv3 --pmc pass each (SQ_BUSY_CYCLES is summed over the 32 SQs, GRBM_GUI_ACTIVE over the 8 XCDs).
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$PWD}; C=$1; shift
for t in "$@"; do
  tag=$t; [ "$t" = product ] && tag=""
  rm -rf $R/gpurun_out/clk_$t
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/clk_$t -o p -- python $R/tools/clock_probe.py $C $tag > $R/gpurun_out/clk_$t.log 2>&1)
  python - <<PY
import csv, glob, collections, statistics
d="$R/gpurun_out/clk_$t"
kt=glob.glob(d+"/**/*kernel_trace.csv", recursive=True)[0]; cc=glob.glob(d+"/**/*counter_collection.csv", recursive=True)[0]
dur={}
for r in csv.DictReader(open(kt)):
    if "rows_pipe" in r["Kernel_Name"]: dur[r["Dispatch_Id"]]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
cnt=collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    if "rows_pipe" in r["Kernel_Name"]: cnt[r["Dispatch_Id"]][r["Counter_Name"]]=float(r["Counter_Value"])
ids=sorted(dur, key=int)[-30:]
ns=statistics.mean(dur[i] for i in ids)
g=lambda c: statistics.mean(cnt[i].get(c,0) for i in ids)
print("%-14s %.1f us | SQ_BUSY_CYCLES/32 %.4g cycles -> %.3f GHz | GRBM_GUI_ACTIVE/8 %.4g -> %.3f GHz | VALU insts %.4g, active %.4g (x4/1024 = %.4g cycles per SIMD)" % (
      "$t", ns/1e3, g("SQ_BUSY_CYCLES")/32, g("SQ_BUSY_CYCLES")/32/ns, g("GRBM_GUI_ACTIVE")/8, g("GRBM_GUI_ACTIVE")/8/ns, g("SQ_INSTS_VALU"), g("SQ_ACTIVE_INST_VALU"), g("SQ_ACTIVE_INST_VALU")*4/1024))
PY
done
