"""Checkpoint stream (reference mcmc/serialize.h, protos.proto; tests mirror mcmc/serialize-test.cc):
wire-format known answers, buffer round trips, and the EndToEnd contract -- run, checkpoint, run ==
restore into a fresh learner, run -- on the CPU stand-in operators (the GPU version of the same test
lives in test_gpu_learner.py)."""
import io
import struct

import numpy as np
import pytest
import torch

import ammsb_pkg

ammsb_pkg.load()
from mcmc_ammsb_gpu_amd import checkpoint as ck


def test_wire_format_known_answers():
    # protoc --encode equivalents, written out by hand from the proto2 wire rules
    assert ck.encode([(1, ck.VARINT, 1024), (2, ck.VARINT, 32), (3, ck.VARINT, 300)]) == \
        bytes([0x08, 0x80, 0x08, 0x10, 0x20, 0x18, 0xAC, 0x02])
    assert ck.encode([(1, ck.BYTES, b"\x01\x02\x03")]) == bytes([0x0A, 0x03, 1, 2, 3])
    assert ck.encode([(1, ck.VARINT, 7), (2, ck.FIXED64, 1.5), (3, ck.FIXED64, 0.0)]) == \
        bytes([0x08, 0x07, 0x11]) + struct.pack("<d", 1.5) + bytes([0x19]) + bytes(8)
    # int32 phase = -1 is a ten-byte varint
    m = ck.encode([(4, ck.VARINT, -1)])
    assert m == bytes([0x20] + [0xFF] * 9 + [0x01]) and ck.decode(m)[4] == (1 << 64) - 1
    lp = ck.encode([(1, ck.VARINT, 21), (2, ck.VARINT, 123456789012), (3, ck.VARINT, 5), (4, ck.VARINT, 1),
                    (5, ck.FIXED64, 2.75)])
    assert ck.decode(lp) == {1: 21, 2: 123456789012, 3: 5, 4: 1, 5: 2.75}
    # decoder accepts any field order and skips nothing silently
    assert ck.decode(bytes([0x18, 0x05, 0x08, 0x01])) == {3: 5, 1: 1}
    with pytest.raises(ck.CheckpointError):
        ck.decode(bytes([0x0A, 0x05, 1, 2]))


def test_record_framing_and_buffers():
    # serialize-test.cc:56-88 (Buffers): several vectors through one stream, sizes must match on the way back
    out = io.BytesIO()
    rng = np.random.default_rng(0)
    vecs = [torch.from_numpy(rng.integers(-2**62, 2**62, 1000 + 37 * i)) for i in range(5)]
    vecs.append(torch.from_numpy(rng.random(300, dtype=np.float32)))
    vecs.append(torch.zeros(0, dtype=torch.float32))
    for v in vecs:
        ck.write_buffer(out, v)
    raw = out.getvalue()
    (sz,) = struct.unpack_from("<Q", raw, 0)
    assert sz == 1 + 2 + 8000 and raw[8] == 0x0A  # key, 2-byte varint length 8000, payload
    inp = io.BytesIO(raw)
    for v in vecs:
        back = torch.full_like(v, 7)
        ck.read_buffer(inp, back)
        assert torch.equal(back, v)
    assert inp.read() == b""
    inp = io.BytesIO(raw)
    with pytest.raises(ck.CheckpointError):
        ck.read_buffer(inp, torch.zeros(999, dtype=torch.int64))
    with pytest.raises(ck.CheckpointError):
        ck.read_buffer(io.BytesIO(raw[:100]), torch.zeros(1000, dtype=torch.int64))


def test_row_partitioned_matrix_round_trip():
    # serialize-test.cc:41-54 (RowPartitionedMatrix): 11 full blocks + 1 short one, shape checked on the way back
    class Rpm:
        def __init__(self, rows, cols, rib):
            self.rows, self.cols, self.rib = rows, cols, rib
            self.blocks = [torch.zeros((min(rib, rows - r), cols)) for r in range(0, rows, rib)]

        def Rows(self):
            return self.rows

        def Cols(self):
            return self.cols

        def RowsPerBlock(self):
            return self.rib

        def Blocks(self):
            return self.blocks
    a = Rpm(1000, 33, 91)
    assert len(a.blocks) == 11
    rng = np.random.default_rng(1)
    for b in a.blocks:
        b.copy_(torch.from_numpy(rng.random(tuple(b.shape), dtype=np.float32)))
    out = io.BytesIO()
    ck.write_rpm(out, a)
    b = Rpm(1000, 33, 91)
    ck.read_rpm(io.BytesIO(out.getvalue()), b)
    assert all(torch.equal(x, y) for x, y in zip(a.blocks, b.blocks))
    first = ck.decode(out.getvalue()[8:8 + struct.unpack_from("<Q", out.getvalue(), 0)[0]])
    assert first == {1: 1000, 2: 33, 3: 91}
    with pytest.raises(ck.CheckpointError):
        ck.read_rpm(io.BytesIO(out.getvalue()), Rpm(1000, 33, 100))


def test_large_buffer_is_streamed(monkeypatch):
    monkeypatch.setattr(ck, "_PIECE", 1000)
    v = torch.arange(5000, dtype=torch.int32)
    out = io.BytesIO()
    ck.write_buffer(out, v)
    back = torch.zeros_like(v)
    ck.read_buffer(io.BytesIO(out.getvalue()), back)
    assert torch.equal(back, v)


def _dataset(N=1024, E=1024, seed=3):
    from mcmc_ammsb_gpu_amd import hostlib
    rng = np.random.default_rng(seed)
    u = rng.integers(0, N, E, dtype=np.uint64)
    v = rng.integers(0, N, E, dtype=np.uint64)
    keep = u != v
    e = np.unique((np.minimum(u, v)[keep] << np.uint64(32)) | np.maximum(u, v)[keep])
    return hostlib.Dataset.robust(N, e, 0.1)


@pytest.mark.parametrize("parallel", [True, False])
def test_end_to_end(parallel):
    """serialize-test.cc:90-134: N = 1024, 1024 random edges, heldout_ratio 0.1, 10 + 10 iterations."""
    import oracle_ops
    from mcmc_ammsb_gpu_amd.learner import Config, Learner
    ds = _dataset()
    iters = 10

    def cfg():
        return Config(heldout_ratio=0.1, ppx_interval=2 * iters - 1, sample_parallel=parallel)
    out = io.BytesIO()
    l1 = Learner(cfg(), ds, ops=oracle_ops)
    l1.Run(iters)
    assert l1.Serialize(out)
    l1.Run(iters)
    ppx = l1.HeldoutPerplexity()
    pi1 = l1.pi.host()
    l1.close()
    l2 = Learner(cfg(), ds, ops=oracle_ops)
    assert l2.Parse(io.BytesIO(out.getvalue()))
    assert l2.stepCount == iters + 1 and l2.phiUpdater.count_calls == iters
    l2.Run(iters)
    assert l2.HeldoutPerplexity() == ppx
    assert np.array_equal(l2.pi.host(), pi1)
    # a second checkpoint of the restored learner, taken at the same point, is byte-identical
    l3 = Learner(cfg(), ds, ops=oracle_ops)
    l3.Parse(io.BytesIO(out.getvalue()))
    again = io.BytesIO()
    l3.Serialize(again)
    a, b = out.getvalue(), again.getvalue()
    assert len(a) == len(b)
    # the only bytes allowed to differ are the wall-clock fields of LearnerProperties (ns resolution round trip)
    assert sum(x != y for x, y in zip(a, b)) <= 16
    l2.close(), l3.close()


def test_parse_rejects_other_shape():
    import oracle_ops
    from mcmc_ammsb_gpu_amd.learner import Config, Learner
    ds = _dataset()
    out = io.BytesIO()
    l1 = Learner(Config(heldout_ratio=0.1, K=32), ds, ops=oracle_ops)
    l1.Serialize(out)
    l1.close()
    l2 = Learner(Config(heldout_ratio=0.1, K=64), ds, ops=oracle_ops)
    with pytest.raises(ck.CheckpointError):
        l2.Parse(io.BytesIO(out.getvalue()))
    l2.close()
