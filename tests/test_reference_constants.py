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


def test_cli_flags_and_defaults_equal_main_cc():
    """Every option main.cc registers (name, short form, default) appears in `ammsb_main --help` with the same
    short form and an equivalent default."""
    import subprocess
    import __graft_entry__ as ge
    ge.build()
    main_cc = open("/root/reference/main.cc").read()
    block = main_cc[main_cc.index("options.add_options()"):main_cc.index("po::variables_map")]
    block = re.sub(r"#ifdef MCMC_CALC_TRAIN_PPX.*?#endif", "", block, flags=re.S)   # compiled out by default
    opts = re.findall(r'\("([A-Za-z0-9_-]+)(?:,([a-z]))?",\s*(?:po::value\(&[\w.]+\)(?:->default_value\((.*?)\))?|"[^"]*")\s*(?:,\s*"[^"]*")?\)',
                      block, re.S)
    assert len(opts) >= 33, len(opts)
    exe = os.path.join(ROOT, "mcmc-ammsb-gpu_amd", "ammsb_main")
    text = subprocess.run([exe, "--help"], capture_output=True, text=True, timeout=60).stdout

    def norm(v):
        v = v.strip().replace("mcmc::", "")
        v = {"true": "1", "false": "0"}.get(v, v)
        m = re.fullmatch(r"\{(\d+),\s*(\d+)\}", v)
        if m:
            return "%s,%s" % m.groups()
        try:
            return repr(float(v))
        except ValueError:
            return v
    for name, short, default in opts:
        if name == "help":
            continue
        pat = (r"-%s \[ --%s \] arg" % (short, re.escape(name))) if short else (r"--%s arg" % re.escape(name))
        m = re.search(pat + r"(?: \(=([^)]*)\))?", text)
        assert m, name
        if default:
            assert m.group(1) is not None, name
            assert norm(m.group(1).split(" ")[0]) == norm(default), (name, m.group(1), default)


def test_config_defaults_equal_config_h():
    """`mcmc::Config()` (reference config.h:68-101): every default assignment has the same value in this build's
    host/config.cc and in the Python Config."""
    ref = open(os.path.join(REF, "config.h")).read()
    body = ref[ref.index("Config() {"):]
    body = re.sub(r"#ifdef MCMC_CALC_TRAIN_PPX.*?#endif", "", body[:body.index("\n  }\n")], flags=re.S)
    ref_defaults = dict(re.findall(r"^\s*(\w+) = ([^;]+);", body, re.M))
    assert len(ref_defaults) >= 27
    mine = open(os.path.join(ROOT, "mcmc-ammsb-gpu_amd", "host", "config.cc")).read()
    mine = mine[mine.index("Config::Config()"):]
    mine_defaults = dict(re.findall(r"^\s*(\w+) = ([^;]+);", mine[:mine.index("\n}")], re.M))
    for k, v in ref_defaults.items():
        assert k in mine_defaults, k
        a, b = v.strip(), mine_defaults[k].strip()
        try:
            assert float(a.rstrip("f")) == float(b.rstrip("f")), (k, a, b)
        except ValueError:
            assert a.replace(" ", "") == b.replace(" ", ""), (k, a, b)
    import ammsb_pkg
    ammsb_pkg.load()
    from mcmc_ammsb_gpu_amd.learner import Config
    py = Config()
    for k, v in ref_defaults.items():
        got = getattr(py, k)
        v = v.strip()
        m = re.fullmatch(r"\{(\d+),\s*(\d+)\}", v)
        if m:
            assert tuple(got) == (int(m.group(1)), int(m.group(2))), k
        elif v in ("true", "false"):
            assert got == (v == "true"), k
        elif re.fullmatch(r"[A-Za-z_]+", v):
            assert str(got) in (v, {"PHI_NODE_PER_WORKGROUP_NAIVE": "PHI_NODE_PER_WORKGROUP_NAIVE"}.get(v, v)), k
        else:
            assert float(got) == float(v), k


def test_checkpoint_records_follow_protos_proto():
    """Every record of a checkpoint written by this build decodes to exactly the field numbers and wire types that
    mcmc/protos.proto declares for the message expected at that position (record order: learner.cc:316-329,
    sample.h:62-75, phi.cc:765-771, beta.cc:386-397, perplexity.cc:276-283)."""
    import io
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    proto = open(os.path.join(REF, "protos.proto")).read()
    wt = {"bytes": 2, "uint32": 0, "uint64": 0, "int32": 0, "double": 1}
    messages = {}
    for name, body in re.findall(r"message (\w+) \{(.*?)\}", proto, re.S):
        fields = re.findall(r"required (\w+) \w+ = (\d+);", body)
        messages[name] = [(int(num), wt.get(ty, 2)) for ty, num in fields]   # message-typed fields are length-delimited
    assert {"VectorStorage", "RpmProperties", "BetaProperties", "PhiProperties", "PerplexityProperties", "SampleStorage",
            "LearnerProperties"} <= set(messages)
    import ammsb_pkg
    ammsb_pkg.load()
    import numpy as np
    import oracle_ops
    from mcmc_ammsb_gpu_amd import checkpoint as ck, hostlib
    from mcmc_ammsb_gpu_amd.learner import Config, Learner
    rng = np.random.default_rng(3)
    u, v = rng.integers(0, 1024, 1500, dtype=np.uint64), rng.integers(0, 1024, 1500, dtype=np.uint64)
    e = np.unique((np.minimum(u, v)[u != v] << np.uint64(32)) | np.maximum(u, v)[u != v])
    ds = hostlib.Dataset.robust(1024, e, 0.1)
    lrn = Learner(Config(heldout_ratio=0.1), ds, ops=oracle_ops)
    lrn.Run(3)
    out = io.BytesIO()
    lrn.Serialize(out)
    lrn.close()
    sample = ["SampleStorage"] + ["VectorStorage"] * 4
    order = (["VectorStorage", "VectorStorage", "RpmProperties", "VectorStorage", "VectorStorage",   # beta theta pi phi
              "VectorStorage", "PhiProperties", "VectorStorage", "VectorStorage", "BetaProperties",  # updaters
              "PerplexityProperties", "VectorStorage", "LearnerProperties"] + sample + sample)
    raw, pos = out.getvalue(), 0
    for expect in order:
        (size,) = struct.unpack_from("<Q", raw, pos)
        msg = raw[pos + 8:pos + 8 + size]
        pos += 8 + size
        tags, p = [], 0
        while p < len(msg):   # walk the top-level fields without materialising the payloads
            key, p = ck._read_varint(msg, p)
            num, w = key >> 3, key & 7
            tags.append((num, w))
            if w == 0:
                _, p = ck._read_varint(msg, p)
            elif w == 1:
                p += 8
            else:
                n, p = ck._read_varint(msg, p)
                p += n
        assert tags == messages[expect], (expect, tags)
    assert pos == len(raw)
