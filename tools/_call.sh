export TMPDIR=/tmp; mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 2>&1 | grep -v amdgpu.ids | tail -30) | tee gpurun_out/c13_pytest.txt
(timeout 900 python bench.py --steps 20 --warmup 3 2>&1 | grep -v amdgpu.ids | tail -1) > gpurun_out/c13_bench.json; cut -c1-1800 gpurun_out/c13_bench.json
