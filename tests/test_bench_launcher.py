"""bench.py --gpus N must start N ranks itself when no launcher did (the driver runs `python bench.py --gpus N`),
and must refuse a WORLD_SIZE that disagrees with --gpus.  CPU-only: nothing here reaches the GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(kw)
    return env


def test_plain_invocation_spawns_n_ranks():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "3", "--warmup", "1"],
                         env=_env(AMMSB_BENCH_SPAWN_DRYRUN="1"), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    cmd = json.loads(out.stdout.strip().splitlines()[-1])["spawn"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(BENCH) + 1:]
    assert tail == ["--gpus", "4", "--steps", "3", "--warmup", "1"]


def test_single_gpu_does_not_spawn():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "1"], env=_env(AMMSB_BENCH_SPAWN_DRYRUN="1"),
                         capture_output=True, text=True, timeout=300)
    assert "spawn" not in out.stdout
    # on the CPU box the run then stops at the device check; on a GPU box it would run the benchmark
    assert out.returncode != 0 or '"metric"' in out.stdout


def test_world_size_mismatch_is_an_error():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8"], env=_env(RANK="0", WORLD_SIZE="2", LOCAL_RANK="0"),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0
    assert "WORLD_SIZE=2" in out.stderr


def test_spawned_children_failure_is_relayed():
    """Without a GPU every rank exits non-zero ("needs a HIP device"); the parent must relay that, not print a
    one-rank result."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-box check")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_env(),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert '"metric"' not in out.stdout
