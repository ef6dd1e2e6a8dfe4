"""bench.py's launcher: `python bench.py --gpus N` with no rank environment must start its own ranks, print ONE JSON
line last and hand back the ranks' status -- the driver's multi-GPU run cannot be allowed to die on the launch.  On a box without a
GPU the ranks run the rendezvous, the slab partition and the frame's collectives over gloo (a dry run, value null); the GPU variants
of this test are in tests/test_gpu_sharded.py."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)


def have_gpu():
    import torch
    return torch.cuda.is_available()


def test_gpus_2_without_rank_environment_launches_itself():
    if have_gpu():
        import pytest
        pytest.skip("GPU box: covered by tests/test_gpu_sharded.py (oversubscribed / RCCL variants)")
    r = run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = r.stdout.strip().splitlines()
    assert len([ln for ln in lines if ln.startswith('{"metric"')]) == 1 and lines[-1].startswith('{"metric"'), r.stdout[-800:]
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1
    assert d["value"] is None and "dry_run" in d                       # no GPU here: nothing measured, and the line says so
    b = d["config"]["slab_bounds"]
    assert b[0] == 0 and b[-1] == 512 and len(b) == 3 and all(x % 8 == 0 for x in b) and b[1] > 256      # work-balanced: the near slab is the thick one


def test_a_failing_rank_fails_the_launcher():
    if have_gpu():
        import pytest
        pytest.skip("CPU variant")
    r = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"DFUSION_BENCH_TEST_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not any(ln.startswith('{"metric"') for ln in r.stdout.splitlines())
    assert "fails on purpose" in r.stderr


def test_three_ranks_uniform_slabs_dry_run():
    if have_gpu():
        import pytest
        pytest.skip("CPU variant")
    r = run_bench(["--gpus", "3", "--steps", "1", "--warmup", "0", "--slabs", "uniform", "--config", "256"])
    assert r.returncode == 0, r.stderr[-1500:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 3 and d["config"]["slab_bounds"] == [0, 88, 176, 256]
