#!/bin/bash
# same-box A/B of bench.py variants, interleaved: tools/ab_bench.sh TAG "args A" "args B" [rounds]
T=$1; A=$2; B=$3; N=${4:-3}
mkdir -p gpurun_out
for i in $(seq $N); do
  for v in A B; do
    if [ $v = A ]; then X="$A"; else X="$B"; fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-verify-cull $X 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); fs = d['frame_stats']
print('$v [$X]: %.1f fps  integrate %.4f ms  raycast %.4f  frame p10/med/p90 %.3f/%.3f/%.3f  loop %.4f  swept/upd %.3f' % (d['value'], d['kernel_ms']['integrate_warped'], d['kernel_ms']['raycast(+merge)'],
      fs['integrate+raycast_ms']['p10'], fs['integrate+raycast_ms']['median'], fs['integrate+raycast_ms']['p90'], fs.get('steady_state_loop_ms', 0), d['roofline']['swept_over_updated']))" | tee -a gpurun_out/${T}_ab_bench.txt
  done
done
