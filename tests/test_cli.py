"""Command-line driver (reference main.cc:26-172): flag table, error exits, and -- on the GPU -- a full
run from a SNAP-style text file, the gzip data-set dump/load path, and checkpoint resume."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.environ.get("AMMSB_MAIN_EXE") or os.path.join(ROOT, "mcmc-ammsb-gpu_amd", "ammsb_main")  # (tools/run_asan.sh)

# every option of the reference's CLI (main.cc:43-81) with its short form and default
REFERENCE_FLAGS = [
    ("file", "f", None), ("heldout-ratio", "r", "0.01"), ("alpha", None, "0"), ("a", "a", "0.0315"),
    ("b", "b", "1024"), ("c", "c", "0.5"), ("epsilon", "e", "1e-07"), ("eta0", None, "1"), ("eta1", None, "1"),
    ("k", "k", "32"), ("mini_batch", "m", "32"), ("neighbors", "n", "32"), ("ppx-wg", None, "32"),
    ("ppx-interval", "i", "100"), ("phi-wg", None, "32"), ("beta-wg", None, "32"), ("max-iters", "x", "100"),
    ("sample", "s", "Node"), ("sampler-wg", None, "32"), ("phi-seed", None, "42,43"), ("beta-seed", None, "44,45"),
    ("neighbor-seed", None, "56,57"), ("phi-mode", None, "PHI_NODE_PER_WORKGROUP_NAIVE"),
    ("phi-probs-shared", None, "1"), ("phi-grads-shared", None, "1"), ("phi-pi-shared", None, "1"),
    ("phi-vwidth", None, "1"), ("beta-sum-grads-vwidth", None, "1"), ("dump-data", None, "0"),
    ("dump-file", None, None), ("load-data", None, "0"), ("load-file", None, None),
]


@pytest.fixture(scope="module")
def exe():
    import __graft_entry__ as ge
    ge.build()
    assert os.path.exists(EXE)
    return EXE


def test_help_lists_reference_flags(exe):
    out = subprocess.run([exe, "--help"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1  # main.cc:86-89
    for name, short, default in REFERENCE_FLAGS:
        pat = (r"-%s \[ --%s \] arg" % (short, re.escape(name))) if short else (r"--%s arg" % re.escape(name))
        if default is not None:
            pat += r" \(=%s\)" % re.escape(default)
        assert re.search(pat, out.stdout), name


def test_error_exits(exe, tmp_path):
    r = subprocess.run([exe, "-f", str(tmp_path / "missing.txt")], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "Failed to detect file" in r.stderr          # main.cc:91-94
    r = subprocess.run([exe, "--load-data", "1"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "load-file is required with load-data" in r.stderr
    g = tmp_path / "g.txt"
    g.write_text("#\n#\n#\n#\n0\t1\n")
    r = subprocess.run([exe, "-f", str(g), "--dump-data", "1"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "dump-file is required with dump-data" in r.stderr
    r = subprocess.run([exe, "--no-such-flag", "1"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "unrecognised option" in r.stderr
    r = subprocess.run([exe, "-k", "many"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "invalid" in r.stderr


def _snap_file(path, N=3000, deg=12, seed=4):
    rng = np.random.default_rng(seed)
    comm = rng.integers(0, 8, N)
    u = rng.integers(0, N, N * deg)
    v = rng.integers(0, N, N * deg)
    keep = (u != v) & ((comm[u] == comm[v]) | (rng.random(u.size) < 0.05))
    with open(path, "w") as f:
        f.write("# Directed graph\n# generated\n# Nodes: %d\n# FromNodeId\tToNodeId\n" % N)
        for a, b in zip(u[keep], v[keep]):
            f.write("%d\t%d\n" % (a * 7 + 3, b * 7 + 3))  # sparse ids: the loader renumbers (data.cc:55-66)


def test_dump_and_load_without_gpu(exe, tmp_path):
    """--dump-data stops after writing the data set (main.cc:110-127); no device is touched."""
    g, d = str(tmp_path / "g.txt"), str(tmp_path / "g.bin.gz")
    _snap_file(g)
    r = subprocess.run([exe, "-f", g, "--dump-data", "1", "--dump-file", d], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    import ammsb_pkg
    ammsb_pkg.load()
    from mcmc_ammsb_gpu_amd import hostlib
    N, ratio, edges = hostlib.load_dataset(d)
    assert N <= 3000 and abs(ratio - 0.01) < 1e-9 and edges.size > 1000
    # the loader renumbers AFTER ordering the end points (data.cc:47-71), so u < v need not survive; ids are dense
    lo, hi = edges >> np.uint64(32), edges & np.uint64(0xFFFFFFFF)
    assert (lo != hi).all() and np.unique(edges).size == edges.size and max(lo.max(), hi.max()) == N - 1


def _ppx_lines(stderr):
    return [(int(m.group(1)), float(m.group(2))) for m in re.finditer(r"ppx\[(\d+)\] = ([0-9.eE+-]+)", stderr)]


@pytest.mark.gpu
def test_cli_run_load_and_resume(exe, tmp_path):
    g, d, ck = str(tmp_path / "g.txt"), str(tmp_path / "g.bin.gz"), str(tmp_path / "state.ckpt")
    _snap_file(g)
    common = ["-k", "32", "-m", "64", "-n", "16", "-r", "0.05", "--phi-wg", "64", "--beta-wg", "64", "--ppx-wg", "64"]
    # the SNAP loader shuffles with the process-global generator; go through the dump so that every run sees the
    # same edge order (that is what --dump-data / --load-data are for)
    r = subprocess.run([exe, "-f", g, "--dump-data", "1", "--dump-file", d] + common, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stderr
    load = ["--load-data", "1", "--load-file", d]
    full = subprocess.run([exe] + load + common + ["-x", "300", "-i", "100"], capture_output=True, text=True, timeout=600)
    assert full.returncode == 0, full.stderr
    ppx = _ppx_lines(full.stderr)
    assert [s for s, _ in ppx] == [0, 100, 200, 300]
    assert ppx[-1][1] < ppx[0][1] and all(np.isfinite(p) for _, p in ppx)
    for cat in ("TOTAL", "PPX CALC", "PPX ACCUM", "SAMPLING", "PHI", "PI", "THETA SUM", "GRADS PAR", "GRADS SUM",
                "UPDATE THETA", "NORM THETA"):
        assert re.search(r"^%s *:" % cat, full.stderr, re.M), cat
    # 200 iterations + checkpoint, then resume for 100: same trajectory as the 300-iteration run.  Held-out
    # perplexity is a running mean over calls (perplexity.cc:51-52), so the calls must line up: 0,100,200 | 200,300
    a = subprocess.run([exe] + load + common + ["-x", "200", "-i", "100", "--checkpoint-out", ck], capture_output=True,
                       text=True, timeout=600)
    assert a.returncode == 0, a.stderr
    assert _ppx_lines(a.stderr) == ppx[:3]
    b = subprocess.run([exe] + load + common + ["-x", "100", "-i", "100", "--checkpoint-in", ck], capture_output=True,
                       text=True, timeout=600)
    assert b.returncode == 0, b.stderr
    # the resumed process evaluates once at start (call 4) and once after 100 more iterations (call 5); the
    # uninterrupted run made call 4 at iteration 300 -- so only the model state is comparable, through a second resume
    c = subprocess.run([exe] + load + common + ["-x", "100", "-i", "100", "--checkpoint-in", ck], capture_output=True,
                       text=True, timeout=600)
    assert _ppx_lines(b.stderr) == _ppx_lines(c.stderr) and len(_ppx_lines(b.stderr)) == 2


@pytest.mark.gpu
@pytest.mark.parametrize("loop", [[], ["--async", "1", "--device-sampling", "1"], ["--async", "1", "--graph", "1", "--device-sampling", "1"]])
def test_cli_reference_default_work_groups_at_k1024(exe, tmp_path, loop):
    """`ammsb_main -k 1024 -m 4096` with NO work-group flags: the reference's defaults are 32 for phi, beta, perplexity
    and the neighbour sampler (main.cc:61-64), i.e. 32 columns per work-item at K = 1024 -- the generic gradient
    kernel and the 32-column register form of update_phi / perplexity.  Must run, learn and print every category."""
    g, d = str(tmp_path / "g.txt"), str(tmp_path / "g.bin.gz")
    _snap_file(g, N=12000, deg=16)  # (device sampling wants N >= 2 m plus the largest degree)
    r = subprocess.run([exe, "-f", g, "--dump-data", "1", "--dump-file", d], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    run = subprocess.run([exe, "--load-data", "1", "--load-file", d, "-k", "1024", "-m", "4096", "-x", "40", "-i", "20"]
                         + loop, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    ppx = _ppx_lines(run.stderr)
    assert [s for s, _ in ppx] == [0, 20, 40]
    assert all(np.isfinite(p) and p > 1.0 for _, p in ppx)
    assert re.search(r"^TOTAL *:", run.stderr, re.M)


@pytest.mark.gpu
@pytest.mark.parametrize("loop", [["--async", "1"], ["--async", "1", "--graph", "1"]])
def test_print_stats_categories_in_the_enqueue_only_loops(exe, tmp_path, loop):
    """learner.cc:252-299: the per-kernel categories of PrintStats must carry device time under --async (event pairs
    read back at the drain) and --graph (time stamps the kernels write themselves), not zeros; --loop-timers 0 turns
    them off."""
    g, d = str(tmp_path / "g.txt"), str(tmp_path / "g.bin.gz")
    _snap_file(g)
    common = ["-k", "64", "-m", "128", "-n", "16", "-r", "0.05", "--phi-wg", "64", "--beta-wg", "64", "--ppx-wg", "64",
              "--device-sampling", "1", "-x", "300", "-i", "300"]
    r = subprocess.run([exe, "-f", g, "--dump-data", "1", "--dump-file", d], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    run = subprocess.run([exe, "--load-data", "1", "--load-file", d] + common + loop, capture_output=True, text=True,
                         timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]

    def cat(text, name):
        m = re.search(r"^%s *: ([0-9.eE+-]+)" % re.escape(name), text, re.M)
        assert m, name
        return float(m.group(1))
    total = cat(run.stderr, "TOTAL")
    parts = {n: cat(run.stderr, n) for n in ("PHI", "PI", "GRADS PAR", "UPDATE THETA")}
    assert parts["PHI"] > 0 and parts["GRADS PAR"] > 0 and parts["UPDATE THETA"] > 0, parts
    assert sum(parts.values()) <= 1.5 * total, (parts, total)   # device time of the chain, not more than the wall time
    off = subprocess.run([exe, "--load-data", "1", "--load-file", d] + common + loop + ["--loop-timers", "0"],
                         capture_output=True, text=True, timeout=600)
    assert off.returncode == 0, off.stderr[-3000:]
    assert cat(off.stderr, "PHI") == 0 and cat(off.stderr, "GRADS PAR") == 0
    # the trajectory does not depend on the timers
    assert _ppx_lines(off.stderr) == _ppx_lines(run.stderr)


@pytest.mark.gpu
def test_kernel_variant_flags_are_not_ignored_silently(exe, tmp_path):
    """main.cc:71-76 / phi.cc:608-700: --phi-mode PHI_NODE_PER_THREAD (another stream map) is refused; --phi-vwidth > 1
    (another column ownership) runs the width-1 form and SAYS so; the SHARED / CODE_GEN modes and
    --beta-sum-grads-vwidth are the same arithmetic and give the same trajectory, with a line naming what ran."""
    g, d = str(tmp_path / "g.txt"), str(tmp_path / "g.bin.gz")
    _snap_file(g)
    r = subprocess.run([exe, "-f", g, "--dump-data", "1", "--dump-file", d], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    common = [exe, "--load-data", "1", "--load-file", d, "-k", "64", "-m", "128", "-n", "16", "-r", "0.05", "-x", "20", "-i", "20"]
    base = subprocess.run(common, capture_output=True, text=True, timeout=600)
    assert base.returncode == 0 and "\nW " not in base.stderr, base.stderr[-2000:]
    thr = subprocess.run(common + ["--phi-mode", "THREAD"], capture_output=True, text=True, timeout=600)
    assert thr.returncode == 2 and "PHI_NODE_PER_THREAD" in thr.stderr and "F " in thr.stderr, thr.stderr[-2000:]
    vw = subprocess.run(common + ["--phi-vwidth", "4"], capture_output=True, text=True, timeout=600)
    assert vw.returncode == 0 and re.search(r"^W phi_vector_width 4: NOT reproduced", vw.stderr, re.M), vw.stderr[-2000:]
    assert _ppx_lines(vw.stderr) == _ppx_lines(base.stderr)  # it is the width-1 trajectory, as the line says
    sh = subprocess.run(common + ["--phi-mode", "WG-SHARED", "--beta-sum-grads-vwidth", "4"],
                        capture_output=True, text=True, timeout=600)
    assert sh.returncode == 0 and re.search(r"^W phi_mode WG-SHARED", sh.stderr, re.M), sh.stderr[-2000:]
    assert re.search(r"^W sum_grads_vector_width 4", sh.stderr, re.M) and _ppx_lines(sh.stderr) == _ppx_lines(base.stderr)
    bad = subprocess.run(common + ["--phi-mode", "PHI_NODE_PER_THREAD"], capture_output=True, text=True, timeout=600)
    assert bad.returncode == 2 and "invalid" in bad.stderr  # (the CLI tokens are THREAD / WG-NAIVE / WG-SHARED / WG-GEN, config.cc:118-131)


def test_python_config_refuses_per_thread_mode_and_warns_on_vector_width():
    import warnings
    import ammsb_pkg
    ammsb_pkg.load()
    from mcmc_ammsb_gpu_amd import learner
    from mcmc_ammsb_gpu_amd._capi import AmmsbError
    with pytest.raises(AmmsbError, match="PHI_NODE_PER_THREAD"):
        learner._check_kernel_variant_knobs(learner.Config(phi_mode="PHI_NODE_PER_THREAD"))
    with pytest.raises(AmmsbError, match="Invalid phi mode"):
        learner._check_kernel_variant_knobs(learner.Config(phi_mode="WHATEVER"))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        learner._check_kernel_variant_knobs(learner.Config(phi_vector_width=2, phi_mode="PHI_NODE_PER_WORKGROUP_CODE_GEN"))
    assert any("NOT reproduced" in str(x.message) for x in w) and any("CODE_GEN" in str(x.message) for x in w)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        learner._check_kernel_variant_knobs(learner.Config())
    assert not w
