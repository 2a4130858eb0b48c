"""TFRecord files of tf.train.Example protos without TensorFlow -- the on-disk format of the
reference's datasets (multi_mnist.py:186-212 writer, :228-296 readers), so that files written by
the reference load here and files written here load in the reference.

Record framing: uint64 length | uint32 masked-crc32c(length) | data | uint32 masked-crc32c(data).
Example: features(1) -> map<string, Feature>(1); Feature: bytes_list(1) / float_list(2) /
int64_list(3), each with repeated value(1).  CRC-32C (Castagnoli) is computed for many records at
once (numpy, one table step per byte position across all records of equal length).
"""
import struct

import numpy as np

_POLY = 0x82F63B78
_TABLE = np.zeros(256, np.uint32)
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ (_POLY if _c & 1 else 0)
    _TABLE[_i] = _c


def crc32c_many(data):
    """data uint8 [n, L] -> uint32 [n] CRC-32C of every row."""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    crc = np.full(data.shape[0], 0xFFFFFFFF, np.uint32)
    for j in range(data.shape[1]):
        crc = _TABLE[(crc ^ data[:, j]) & 0xFF] ^ (crc >> np.uint32(8))
    return crc ^ np.uint32(0xFFFFFFFF)


def _raw_update(state, data):
    """CRC register update (no init / final xor): state uint32 [n], data uint8 [n, L]."""
    for j in range(data.shape[1]):
        state = _TABLE[(state ^ data[:, j]) & 0xFF] ^ (state >> np.uint32(8))
    return state


def crc32c(b):
    """CRC-32C of one buffer.  Large buffers are cut into K chunks of L bytes whose register
    updates run side by side (numpy lanes); chunk results are chained with the linear operator
    "advance the register over L zero bytes", obtained from 32 extra lanes fed with zeros."""
    a = np.frombuffer(bytes(b), np.uint8)
    n = a.size
    if n < 4096:
        return int(crc32c_many(a[None, :])[0])
    L = 1 << max(8, int(np.log2(n) / 2))
    K = n // L
    lanes = np.zeros((K + 32, L), np.uint8)
    lanes[:K] = a[:K * L].reshape(K, L)
    init = np.zeros(K + 32, np.uint32)
    init[K:] = np.uint32(1) << np.arange(32, dtype=np.uint32)        # basis states over zero bytes
    out = _raw_update(init, lanes)
    chunk, basis = out[:K], out[K:]
    s = np.uint32(0xFFFFFFFF)
    bits = np.arange(32, dtype=np.uint32)
    for k in range(K):
        sel = ((s >> bits) & np.uint32(1)).astype(bool)
        s = np.bitwise_xor.reduce(basis[sel]) ^ chunk[k] if sel.any() else chunk[k]
    tail = a[K * L:]
    if tail.size:
        s = _raw_update(np.array([s], np.uint32), tail[None, :])[0]
    return int(s ^ np.uint32(0xFFFFFFFF))


def masked(crc):
    crc = np.asarray(crc, dtype=np.uint64)
    return ((((crc >> np.uint64(15)) | (crc << np.uint64(17))) + np.uint64(0xA282EAD8)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)


# ----------------------------------------------------------------------------- protobuf wire format
def _varint(n):
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _read_varint(b, i):
    r, s = 0, 0
    while True:
        c = b[i]
        i += 1
        r |= (c & 0x7F) << s
        if not c & 0x80:
            return r, i
        s += 7


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def encode_example(features):
    """features: {name: ("int64", [ints]) | ("bytes", [bytes]) | ("float", [floats])} -> serialized Example."""
    entries = b""
    for name in sorted(features):
        kind, values = features[name]
        if kind == "int64":
            feat = _ld(3, _ld(1, b"".join(_varint(int(v)) for v in values)))          # packed
        elif kind == "bytes":
            feat = _ld(1, b"".join(_ld(1, bytes(v)) for v in values))
        elif kind == "float":
            feat = _ld(2, _ld(1, struct.pack("<%df" % len(values), *values)))
        else:
            raise ValueError(kind)
        entries += _ld(1, _ld(1, name.encode()) + _ld(2, feat))
    return _ld(1, entries)


def _fields(b):
    i, n = 0, len(b)
    while i < n:
        key, i = _read_varint(b, i)
        f, w = key >> 3, key & 7
        if w == 0:
            v, i = _read_varint(b, i)
        elif w == 1:
            v = b[i:i + 8]; i += 8
        elif w == 2:
            ln, i = _read_varint(b, i)
            v = b[i:i + ln]; i += ln
        elif w == 5:
            v = b[i:i + 4]; i += 4
        else:
            raise ValueError("unsupported wire type %d" % w)
        yield f, w, v


def parse_example(b):
    """serialized Example -> {name: list of ints / bytes / floats}."""
    out = {}
    for f, _, feats in _fields(b):
        if f != 1:
            continue
        for f2, _, entry in _fields(feats):
            if f2 != 1:
                continue
            name, feat = None, b""
            for f3, _, v in _fields(entry):
                if f3 == 1:
                    name = v.decode()
                elif f3 == 2:
                    feat = v
            values = []
            for kind, _, lst in _fields(feat):
                for f5, w5, v in _fields(lst):
                    if f5 != 1:
                        continue
                    if kind == 1:
                        values.append(bytes(v))
                    elif kind == 3:
                        if w5 == 2:                                   # packed
                            i = 0
                            while i < len(v):
                                x, i = _read_varint(v, i)
                                values.append(x - (1 << 64) if x >> 63 else x)
                        else:
                            values.append(v - (1 << 64) if v >> 63 else v)
                    elif kind == 2:
                        values.extend(struct.unpack("<%df" % (len(v) // 4), v) if w5 == 2 else struct.unpack("<f", v))
            out[name] = values
    return out


# ----------------------------------------------------------------------------- record files
def write_records(path, records):
    """records: list of bytes.  CRCs are computed per group of equal-length records."""
    records = [bytes(r) for r in records]
    lens = np.array([len(r) for r in records], np.uint64)
    len_bytes = lens.astype("<u8").view(np.uint8).reshape(-1, 8)
    len_crc = masked(crc32c_many(len_bytes))
    data_crc = np.zeros(len(records), np.uint32)
    for L in np.unique(lens):
        idx = np.nonzero(lens == L)[0]
        block = np.frombuffer(b"".join(records[i] for i in idx), np.uint8).reshape(len(idx), int(L))
        data_crc[idx] = masked(crc32c_many(block))
    with open(path, "wb") as f:
        for i, r in enumerate(records):
            f.write(len_bytes[i].tobytes())
            f.write(struct.pack("<I", int(len_crc[i])))
            f.write(r)
            f.write(struct.pack("<I", int(data_crc[i])))
    return path


def read_records(path, verify=False):
    """-> list of record payloads (tf.python_io.tf_record_iterator)."""
    out = []
    with open(path, "rb") as f:
        buf = f.read()
    i, n = 0, len(buf)
    while i < n:
        (L,) = struct.unpack_from("<Q", buf, i)
        (lc,) = struct.unpack_from("<I", buf, i + 8)
        data = buf[i + 12:i + 12 + L]
        (dc,) = struct.unpack_from("<I", buf, i + 12 + L)
        if len(data) != L:
            raise IOError("truncated TFRecord file %s" % path)
        if verify:
            if int(masked(crc32c(buf[i:i + 8]))) != lc or int(masked(crc32c(data))) != dc:
                raise IOError("corrupted record at byte %d of %s" % (i, path))
        out.append(data)
        i += 16 + L
    return out
