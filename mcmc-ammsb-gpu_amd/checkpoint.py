"""Checkpoint stream of the reference (mcmc/serialize.h:13-38, mcmc/protos.proto): a sequence of
records, each `[u64 little-endian byte size][proto2 message]`.  The messages are tiny and fixed, so the
wire format is written by hand here (there is no protobuf dependency on the device side):

    VectorStorage         1: bytes storage
    RpmProperties         1: uint32 rows, 2: uint32 cols, 3: uint32 rows_in_block
    PhiProperties         1: uint32 count_calls, 2: double update_phi_time, 3: double update_pi_time
    BetaProperties        1: uint32 count_calls, 2..6: double timers
    PerplexityProperties  1: uint32 count_calls, 2: double ppx_time, 3: double accumulate_time
    SampleStorage         1: bytes edges, 2: bytes nodes_vec, 3: uint32 seed
    LearnerProperties     1: uint32 stepCount, 2: uint64 time, 3: uint64 samplingTime, 4: int32 phase,
                          5: double weight

Fields are emitted in field-number order with every required field present, which is what protobuf's
C++ serializer produces, so files are byte-compatible in both directions.  Unlike protobuf (whose
`bytes` fields and ByteSize() are limited to 2 GiB) buffers of any size are accepted: lengths are
64-bit varints and large device buffers are streamed in pieces.
"""
import struct

import numpy as np
import torch

VARINT, FIXED64, BYTES = 0, 1, 2
_PIECE = 64 << 20  # bytes per host<->device staging piece


class CheckpointError(RuntimeError):
    pass


def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, pos):
    shift = v = 0
    while True:
        if pos >= len(buf) or shift > 63:
            raise CheckpointError("truncated varint")
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, pos
        shift += 7


def encode(fields):
    """fields: [(number, wire type, value)] in field order -> message bytes."""
    out = bytearray()
    for num, wt, val in fields:
        out += _varint((num << 3) | wt)
        if wt == VARINT:
            out += _varint(int(val))
        elif wt == FIXED64:
            out += struct.pack("<d", float(val))
        else:
            out += _varint(len(val))
            out += val
    return bytes(out)


def decode(buf):
    """message bytes -> {number: value}; unknown wire types are rejected, unknown fields kept."""
    pos, res = 0, {}
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == VARINT:
            res[num], pos = _read_varint(buf, pos)
        elif wt == FIXED64:
            if pos + 8 > len(buf):
                raise CheckpointError("truncated double")
            res[num] = struct.unpack_from("<d", buf, pos)[0]
            pos += 8
        elif wt == BYTES:
            n, pos = _read_varint(buf, pos)
            if pos + n > len(buf):
                raise CheckpointError("truncated bytes field")
            res[num] = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            pos += 4
        else:
            raise CheckpointError("unsupported wire type %d" % wt)
    return res


def write_message(out, fields):
    """SerializeMessage (serialize.h:13-24)."""
    msg = encode(fields)
    out.write(struct.pack("<Q", len(msg)))
    out.write(msg)


def _read_exact(inp, n):
    b = inp.read(n)
    if b is None or len(b) != n:
        raise CheckpointError("unexpected end of checkpoint")
    return b


def read_message(inp, required=()):
    """ParseMessage (serialize.h:26-38)."""
    (size,) = struct.unpack("<Q", _read_exact(inp, 8))
    if size > (1 << 31):
        raise CheckpointError("property record of %d bytes" % size)
    m = decode(_read_exact(inp, size))
    for f in required:
        if f not in m:
            raise CheckpointError("missing required field %d" % f)
    return m


def _as_bytes_view(t):
    """1-D uint8 view of a contiguous tensor (no copy)."""
    if not t.is_contiguous():
        raise CheckpointError("buffer is not contiguous")
    return t.reshape(-1).view(torch.uint8)


def write_buffer(out, tensor):
    """Serialize(out, Buffer<T>*) (serialize.h:40-53): one VectorStorage with the whole buffer."""
    raw = _as_bytes_view(tensor)
    n = raw.numel()
    head = _varint((1 << 3) | BYTES) + _varint(n)
    out.write(struct.pack("<Q", len(head) + n))
    out.write(head)
    for lo in range(0, n, _PIECE):
        out.write(raw[lo:lo + _PIECE].cpu().numpy().tobytes())


def read_buffer(inp, tensor):
    """Parse(in, Buffer<T>*) (serialize.h:55-70): sizes must agree exactly."""
    raw = _as_bytes_view(tensor)
    n = raw.numel()
    (size,) = struct.unpack("<Q", _read_exact(inp, 8))
    head = _read_exact(inp, min(size, 11))  # key byte + at most 10 varint bytes
    key, pos = _read_varint(head, 0)
    if key != ((1 << 3) | BYTES):
        raise CheckpointError("expected a VectorStorage record")
    length, pos = _read_varint(head, pos)
    if length != n or size != pos + n:
        raise CheckpointError("buffer size mismatch: checkpoint has %d bytes, buffer %d" % (length, n))
    carry = head[pos:]
    done = 0
    while done < n:
        take = min(_PIECE, n - done)
        piece = carry[:take] + _read_exact(inp, take - min(len(carry), take))
        carry = carry[take:]
        raw[done:done + take].copy_(torch.frombuffer(bytearray(piece), dtype=torch.uint8))
        done += take
    if carry:
        raise CheckpointError("trailing bytes in VectorStorage record")


def write_rpm(out, rpm):
    """Serialize(out, RowPartitionedMatrix*) (serialize.h:72-90)."""
    write_message(out, [(1, VARINT, rpm.Rows()), (2, VARINT, rpm.Cols()), (3, VARINT, rpm.RowsPerBlock())])
    for b in rpm.Blocks():
        write_buffer(out, b)


def read_rpm(inp, rpm):
    """Parse(in, RowPartitionedMatrix*) (serialize.h:92-115)."""
    m = read_message(inp, (1, 2, 3))
    if (m[1], m[2], m[3]) != (rpm.Rows(), rpm.Cols(), rpm.RowsPerBlock()):
        raise CheckpointError("matrix shape mismatch: checkpoint %s, learner %s" %
                              ((m[1], m[2], m[3]), (rpm.Rows(), rpm.Cols(), rpm.RowsPerBlock())))
    for b in rpm.Blocks():
        read_buffer(inp, b)


def write_host_bytes(out, arr):
    data = np.ascontiguousarray(arr).tobytes()
    write_message(out, [(1, BYTES, data)])
