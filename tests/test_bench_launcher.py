"""bench.py --gpus N must start N ranks itself when no launcher did (the driver runs `python bench.py --gpus N`),
and must refuse a WORLD_SIZE that disagrees with --gpus.  CPU-only: nothing here reaches the GPU."""
import json
import os
import subprocess
import sys

import pytest

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


def test_watchdog_names_the_phase_a_rank_is_stuck_in():
    """A rank blocked in a collective its peers never join says nothing for RCCL's ten minutes; the bench's watchdog
    ends the process with the phase's name (rank 0 also as a JSON line on stdout) and exit code 124."""
    code = ("import sys, time; sys.path.insert(0, %r); import bench; wd = bench.Watchdog(0.5, 0, 8); "
            "wd.phase('Learner() incl. the split calibration'); time.sleep(30)" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert out.returncode == 124
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["error"] == "watchdog" and rec["phase"].startswith("Learner()") and rec["rank"] == 0 and rec["world"] == 8
    assert "gave up" in out.stderr
    # a phase change in time keeps the run alive; "done" stops the watching
    code = ("import sys, time; sys.path.insert(0, %r); import bench; wd = bench.Watchdog(1.0, 1, 2)\n"
            "for i in range(4):\n    wd.phase('p%%d' %% i); time.sleep(0.4)\n"
            "wd.phase('done'); time.sleep(1.5); print('alive')" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and out.stdout.strip() == "alive", out.stderr


@pytest.mark.gpu
def test_plain_bench_two_ranks_end_to_end():
    """`python bench.py --gpus 2 --workload C1` exactly as the driver types it (no launcher): the parent starts the two
    ranks itself, they rendezvous on 127.0.0.1, run the sharded learner and rank 0 prints ONE JSON line.  On a one-GPU
    box the ranks share the device and the exchange is staged through host memory (AMMSB_BENCH_BACKEND=gloo -- RCCL
    refuses two ranks on one device): a rehearsal of the code path, labelled as such in the line, never a number."""
    import json
    env = dict(os.environ, AMMSB_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--workload", "C1", "--steps", "30", "--warmup", "5",
                          "--no-cpu-baseline", "--cpp-dropin", "0", "--sustained-s", "0.5"], env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 0 and "REHEARSAL" in rec["data"]
    assert rec["steps"] == 30 and rec["value"] > 0 and rec["scaling"] == "strong"
    assert rec["sustained"]["value"] > 0 and rec["sustained"]["untimed_steps"] > 0  # the second window, on every rank
    split = rec["config"]["phi_split"]
    assert split is not None and split["exchange"] in ("collective", "p2p") and 0.0 <= split["rho"] <= 1.0
    assert split["predicted_phi_speedup"] is None or split["predicted_phi_speedup"] > 0
    assert "node-sharded phi x2" in rec["config"]["parallelism"]
    assert split["beta_gradient"].startswith(("replicated", "sharded")) and split["beta_gradient"] in rec["config"]["parallelism"]
    assert "gradient" in split["trace"] or "error" in split["trace"]
    # the step trace that makes a multi-GPU line diagnosable by itself
    tr = split["trace"]
    assert "error" not in tr, tr
    if tr["steps"]:
        for key in ("step_ms", "phi_phase_ms", "update_pi_ms", "grads_local_ms", "grad_allgather_ms", "update_theta_ms",
                    "exchange_ms", "exchange_chunks"):
            assert key in tr, key
        assert tr["step_ms"] > 0


@pytest.mark.gpu
def test_single_gpu_line_carries_device_state_and_settle():
    """One JSON line; `roofline.device_state` says what the line ran at (sysfs clocks / power before and after the timed
    window, partitions, the shader clock held under load per XCD) and `settle` what ran before the warm-up steps."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, BENCH, "--workload", "C1", "--steps", "200", "--warmup", "20", "--settle-s", "0.2",
                          "--sustained-s", "0.3",
                          "--no-cpu-baseline", "--cpp-dropin", "0", "--extras", "0"], env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["steps"] == 200 and rec["value"] > 0 and rec["dtype"] == "f32"
    assert rec["settle"]["steps"] > 0 and rec["settle"]["seconds"] == 0.2
    su = rec["sustained"]  # the window after seconds of load, beside the contract's
    assert su["after_s"] == 0.3 and su["untimed_steps"] > 0 and su["value"] > 0 and su["steps"] == 200
    assert su["device_state"]["power_w"][0] > 0
    ds = rec["roofline"]["device_state"]
    assert ds["compute_partition"] and ds["memory_partition"]
    assert len(ds["sclk_mhz"]) == 2 and len(ds["power_w"]) == 2 and ds["power_cap_w"][0] > 0
    assert 300 < ds["shader_clock_under_load_mhz"] < 2600 and len(ds["shader_clock_per_xcd_mhz"]) == 8
    assert rec["large_configs"] is None and rec["small_configs"] is None  # (--extras 0)
