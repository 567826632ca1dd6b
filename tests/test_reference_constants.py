"""Constants of the hot path checked against the TEXT of the reference (read-only study of /root/reference; nothing
is imported, built or copied).  Runs only where the reference tree is mounted (the build container); skipped on the
GPU box.  What it pins: the 128-strip ziggurat tables the oracle and the kernels derive from the recurrence
(tools/gen_ziggurat_tables.py) equal the literals of mcmc/random.cl.inc as binary32 / integers, and the scalar
constants (tail start R, cuckoo prime pairs, neighbour-sampler hash constant) are the reference's."""
import importlib.util
import os
import re
import struct

import numpy as np
import pytest

REF = "/root/reference/mcmc"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted here")


def _array(text, name):
    m = re.search(name + r"\[128\]\s*=\s*\{(.*?)\}", text, re.S)
    assert m, name
    body = re.sub(r'["\\n]', " ", m.group(1))          # the tables sit inside a C string literal
    return [t for t in re.split(r"[,\s]+", body) if t]


def test_ziggurat_tables_equal_the_reference_literals():
    text = open(os.path.join(REF, "random.cl.inc")).read()
    spec = importlib.util.spec_from_file_location("zig", os.path.join(ROOT, "tools", "gen_ziggurat_tables.py"))
    zig = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(zig)
    ytab, ktab, wtab = zig.tables()
    ref_y = np.array([float(t) for t in _array(text, "gsl_ytab")], dtype=np.float64).astype(np.float32)
    ref_w = np.array([float(t.rstrip("fF")) for t in _array(text, "gsl_wtab")], dtype=np.float64).astype(np.float32)
    ref_k = np.array([int(t.rstrip("uUlL")) for t in _array(text, "gsl_ktab")], dtype=np.uint64)
    assert np.array_equal(np.array(ytab, dtype=np.float32).view(np.uint32), ref_y.view(np.uint32))
    assert np.array_equal(np.array(wtab, dtype=np.float32).view(np.uint32), ref_w.view(np.uint32))
    assert np.array_equal(np.array(ktab, dtype=np.uint64), ref_k)
    # the committed fragments are what the generator emits today
    for path, prefix, qual in ((os.path.join(ROOT, "oracle", "zig_tables.inc"), "orc", "static const"),
                               (os.path.join(ROOT, "mcmc-ammsb-gpu_amd", "csrc", "zig_tables.inc"), "zig",
                                "__device__ const")):
        import io
        buf = io.StringIO()
        zig.emit(prefix, qual, buf)
        assert open(path).read() == buf.getvalue(), path
    m = re.search(r"#define PARAM_R FL\(([0-9.]+)\)", text)
    assert m and float(m.group(1)) == zig.R
    assert struct.pack("<f", zig.R) == struct.pack("<f", 3.44428647676)


def _mine(path, pattern):
    return re.findall(pattern, open(os.path.join(ROOT, path)).read())


def test_scalar_constants():
    cuckoo = open(os.path.join(REF, "cuckoo.cc")).read()
    ref_primes = [int(x) for pair in re.findall(r"make_pair<uint64_t, uint64_t>\((\d+), (\d+)\)", cuckoo) for x in pair]
    assert len(ref_primes) == 8
    for path in ("oracle/ammsb_oracle.c", "mcmc-ammsb-gpu_amd/csrc/ammsb_dev.h", "mcmc-ammsb-gpu_amd/host/cuckoo.cc"):
        text = open(os.path.join(ROOT, path)).read()
        pos = -1
        for p in ref_primes:       # the eight numbers appear in the reference's order
            nxt = text.find("%dull" % p, pos + 1)
            assert nxt > pos, (path, p)
            pos = nxt
    sample = open(os.path.join(REF, "sample.cc")).read()
    h = re.search(r"\(k \^ (\d+)\) % capacity", sample).group(1)
    for path in ("oracle/ammsb_oracle.c", "mcmc-ammsb-gpu_amd/csrc/ammsb_core.hip"):
        assert ("^ %su" % h) in open(os.path.join(ROOT, path)).read(), path
