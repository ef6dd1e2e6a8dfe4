#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== A/B/C/D: prepare stream on 16 CUs (A), unmasked (B), 8 CUs (C), 32 CUs (D), one stream (E); same box, interleaved"
rm -f gpurun_out/s11_ab.txt
for i in 1 2 3; do
  for X in "" "--prep-cus 0" "--prep-cus 8" "--prep-cus 32" "--no-pipeline"; do
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-verify-cull --long-frames 0 $X 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); fs = d['frame_stats']
print('[%-14s]: %.1f fps  integrate %.4f ms (prepare %s)  raycast %.4f  loop %.4f' % ('$X', d['value'], d['kernel_ms']['integrate_warped'], d['kernel_ms'].get('integrate_prepare_on_side_stream'), d['kernel_ms']['raycast(+merge)'], fs.get('steady_state_loop_ms', 0)))" | tee -a gpurun_out/s11_ab.txt
  done
done
