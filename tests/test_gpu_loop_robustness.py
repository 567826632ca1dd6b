"""The descriptor loop's hand-over under adverse conditions (csrc/ammsb_loop.hip, include/ammsb.h ammsb_loop_check).

The two chains of an iteration are ordered on the device by polling kernels.  These tests force the situations in
which such a wait cannot be satisfied and require what the library promises: nothing hangs, nothing runs on a
mini-batch that is not there, the run is finished on the stream-event hand-over and the trajectory is bit-identical
to the eager loop's (the reference's launch-by-launch order, learner.cc:222-247).
  * a sampler chain whose wait gives up (test hook AMMSB_LOOP_TEST_FAIL_AT + a short AMMSB_LOOP_WAIT_MS), at the
    ramp-up, in mid run, across the 1024-step descriptor chunk and across several enqueued runs;
  * one hardware queue for the whole process (GPU_MAX_HW_QUEUES=1): the create-time probe picks the event hand-over;
  * two learners in one process driven from two threads at once (shared hardware queues);
  * the host-sampling path with the device held back, so that the enqueue side runs iterations ahead of the copies
    out of its pinned staging buffers (the race fixed in ac9dc4d).
"""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (no fallback path exists)")
    import ammsb_pkg
    ammsb_pkg.load()
    from mcmc_ammsb_gpu_amd import hostlib, learner
    return hostlib, learner, torch


def _dataset(hostlib, N=3000, deg=12, seed=3):
    edges = hostlib.generate_graph(N, 8, deg, seed=seed)
    return hostlib.Dataset.robust(N, edges, heldout_ratio=0.02, rand_seed=1)


def _cfg(learner, graph, K=64, m=256, **kw):
    return learner.Config.from_cli_defaults(K=K, mini_batch_size=m, num_node_sample=16, strategy="Node",
                                            phi_wg_size=64, beta_wg_size=64, ppx_wg_size=64, device_sampling=True,
                                            graph_launch=graph, **kw)


def _state(lrn):
    lrn.drain()
    return (lrn.pi.host().copy(), lrn.theta.cpu().numpy().copy(), lrn.beta.cpu().numpy().copy(),
            lrn.phi.cpu().numpy().copy(), lrn.phiUpdater.rand.host().copy(), lrn.betaUpdater.rand.host().copy())


def _same(a, b):
    return all(np.array_equal(np.ascontiguousarray(x).view(np.uint8), np.ascontiguousarray(y).view(np.uint8))
               for x, y in zip(a, b))


@pytest.mark.parametrize("fail_at,steps,host_chunk", [(2, 40, 512), (3, 40, 512), (9, 40, 512),
                                                      (1030, 1300, 4096),   # one run, second 1024-step descriptor chunk
                                                      (600, 1500, 512),     # three enqueued runs: second poisoned, third skipped
                                                      (700, 1500, 700),     # the give-up at the very end of a run
                                                      (701, 1500, 700),     # ... of the run's PENDING mini-batch, a run follows
                                                      (701, 700, 700)])     # ... and nothing follows: the next Run() needs it
def test_given_up_wait_is_recovered_bit_identically(env, fail_at, steps, host_chunk):
    hostlib, learner, torch = env
    ds = _dataset(hostlib)
    eager = learner.Learner(_cfg(learner, False), ds)
    eager.Run(steps)
    want = _state(eager)
    want_ppx = eager.HeldoutPerplexity()
    eager.close()
    old = {k: os.environ.get(k) for k in ("AMMSB_LOOP_TEST_FAIL_AT", "AMMSB_LOOP_WAIT_MS", "AMMSB_LOOP_HANDSHAKE")}
    os.environ.update(AMMSB_LOOP_TEST_FAIL_AT=str(fail_at), AMMSB_LOOP_WAIT_MS="300", AMMSB_LOOP_HANDSHAKE="flag")
    try:
        lrn = learner.Learner(_cfg(learner, True), ds)   # (the hooks are read when the loop is created)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert lrn.loop is not None and lrn.loop.status() == (False, 0)
    lrn.GRAPH_CHUNK = host_chunk
    lrn.Run(steps)
    # the perplexity call must not see a state some iterations short: it recovers first
    got_ppx = lrn.HeldoutPerplexity()
    assert lrn.loop.status() == (True, 1), "the run was expected to fall back to the event hand-over exactly once"
    assert lrn.loop_fallbacks == 1
    assert _same(_state(lrn), want)
    assert got_ppx == want_ppx
    lrn.Run(25)   # and goes on, on events
    eager2 = learner.Learner(_cfg(learner, False), ds)
    eager2.Run(steps + 25)
    assert _same(_state(lrn), _state(eager2))
    assert lrn.loop.status() == (True, 1)
    eager2.close()
    lrn.close()


CHILD_ONE_QUEUE = r"""
import sys
import numpy as np
sys.path[:0] = [%(root)r, %(tests)r]
import ammsb_pkg
ammsb_pkg.load()
from mcmc_ammsb_gpu_amd import hostlib, learner
from test_gpu_loop_robustness import _dataset, _cfg, _state, _same
ds = _dataset(hostlib)
a = learner.Learner(_cfg(learner, False), ds)
a.Run(120)
b = learner.Learner(_cfg(learner, True), ds)
b.Run(120)
assert _same(_state(a), _state(b)), "trajectories differ"
print("handover", b.loop.status())
assert b.loop.status()[1] == 0
print("one queue ok")
"""


def test_single_hardware_queue(env):
    """GPU_MAX_HW_QUEUES=1: both of the loop's streams share one hardware queue.  No wait may give up (the probe at
    loop creation sees that the streams cannot overlap and takes the event hand-over); trajectory = eager."""
    e = dict(os.environ, GPU_MAX_HW_QUEUES="1")
    e.pop("AMMSB_LOOP_HANDSHAKE", None)
    script = CHILD_ONE_QUEUE % {"root": ROOT, "tests": os.path.join(ROOT, "tests")}
    out = subprocess.run([sys.executable, "-c", script], env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "one queue ok" in out.stdout


def test_two_loops_from_two_threads(env):
    """Two descriptor-loop learners in one process, their runs issued concurrently from two host threads (streams of
    different loops can share a hardware queue): no wait gives up, both trajectories equal their eager twins."""
    hostlib, learner, torch = env
    ds1, ds2 = _dataset(hostlib, seed=3), _dataset(hostlib, N=2500, seed=5)
    want = []
    for ds in (ds1, ds2):
        e = learner.Learner(_cfg(learner, False), ds)
        e.Run(600)
        want.append(_state(e))
        e.close()
    loops = [learner.Learner(_cfg(learner, True), ds) for ds in (ds1, ds2)]
    errs = []

    def work(l):
        try:
            torch.cuda.set_device(0)
            for _ in range(6):
                l.Run(100)
        except Exception as ex:  # noqa: BLE001
            errs.append(ex)
    ts = [threading.Thread(target=work, args=(l,)) for l in loops]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for l, w in zip(loops, want):
        assert _same(_state(l), w)
        assert l.loop.status()[1] == 0, "a device-side wait gave up"
        l.close()


def test_host_sampling_staging_buffers_survive_a_stalled_device(env):
    """Host sampling copies each mini-batch out of pinned staging buffers asynchronously.  With the device held back the
    enqueue side gets iterations ahead; the staging buffer of a sample must not be refilled before the copy issued two
    iterations earlier has executed (ac9dc4d).  Control: with that wait removed the same run diverges."""
    hostlib, learner, torch = env
    from mcmc_ammsb_gpu_amd import ops
    ds = _dataset(hostlib, N=2000)

    def cfg():
        return learner.Config.from_cli_defaults(K=32, mini_batch_size=64, num_node_sample=8, strategy="Node",
                                                phi_wg_size=32, beta_wg_size=32, ppx_wg_size=32, device_sampling=False)

    def run(stall, patched=False):
        lrn = learner.Learner(cfg(), ds)
        saved = ops.sync_event
        if patched:
            ops.sync_event = lambda ev: None
        try:
            if stall:
                for s in lrn.samples:   # ~0.2 s of spinning ahead of the first copies on every sample stream
                    with torch.cuda.stream(s.stream):
                        torch.cuda._sleep(int(4e8))
            lrn.Run(8)
            st = _state(lrn)
        finally:
            ops.sync_event = saved
        lrn.close()
        return st
    want = run(False)
    assert _same(run(True), want)
    assert not _same(run(True, patched=True), want), "the control run did not reproduce the race: the test has no teeth"
