"""TensorFlow checkpoint ("tensor bundle") import / export without TensorFlow.

The reference restores `model/air-model` with tf.train.Saver (demo.py:33) and saves
`air-model-<step>` every 10 000 iterations (training.py:203-207).  A bundle is
  <prefix>.index                 LevelDB-style SSTable: key "" -> BundleHeaderProto, tensor name ->
                                 BundleEntryProto {dtype, shape, shard_id, offset, size, masked crc32c}
  <prefix>.data-00000-of-00001   the raw little-endian tensor bytes, in key order
(layout decoded from the reference's own model/air-model.index).  Variable names
follow the reference graph: trainables under `air/rnn/`, Adam slots `air/training/<var>/Adam`
(m) and `/Adam_1` (v), `air/training/beta{1,2}_power`, `air/global_step`.
"""
import struct
from collections import OrderedDict

import numpy as np

from tfrecord import _fields, _ld, _read_varint, _varint, crc32c, masked

MAGIC = 0xDB4775248B80FB57
DT_FLOAT, DT_INT32 = 1, 3
_NP = {DT_FLOAT: np.float32, DT_INT32: np.int32}
_DT = {np.dtype(np.float32): DT_FLOAT, np.dtype(np.int32): DT_INT32}


# ----------------------------------------------------------------------------- SSTable reading
def _block_entries(block):
    """Prefix-compressed entries of one block (restart array ignored: sequential scan)."""
    (nrestarts,) = struct.unpack_from("<I", block, len(block) - 4)
    end = len(block) - 4 - 4 * nrestarts
    i, key = 0, b""
    while i < end:
        shared, i = _read_varint(block, i)
        non_shared, i = _read_varint(block, i)
        vlen, i = _read_varint(block, i)
        key = key[:shared] + block[i:i + non_shared]
        i += non_shared
        yield key, block[i:i + vlen]
        i += vlen


def _read_block(buf, offset, size, verify):
    data, ctype = buf[offset:offset + size], buf[offset + size]
    (crc,) = struct.unpack_from("<I", buf, offset + size + 1)
    if ctype != 0:
        raise NotImplementedError("compressed SSTable block (type %d)" % ctype)
    if verify and int(masked(crc32c(buf[offset:offset + size + 1]))) != crc:
        raise IOError("SSTable block checksum mismatch at %d" % offset)
    return data


def _handle(b, i=0):
    off, i = _read_varint(b, i)
    size, i = _read_varint(b, i)
    return off, size, i


def read_index(path, verify=True):
    """-> (header dict, OrderedDict name -> entry dict(dtype, shape, shard_id, offset, size, crc32c))."""
    buf = open(path, "rb").read()
    footer = buf[-48:]
    if struct.unpack("<Q", footer[40:])[0] != MAGIC:
        raise IOError("%s is not an SSTable (bad magic)" % path)
    _, _, i = _handle(footer)                              # metaindex handle (unused)
    ioff, isize, _ = _handle(footer, i)
    header, entries = {}, OrderedDict()
    for _, hv in _block_entries(_read_block(buf, ioff, isize, verify)):
        boff, bsize, _ = _handle(hv)
        for key, val in _block_entries(_read_block(buf, boff, bsize, verify)):
            if key == b"":
                for f, _, v in _fields(val):
                    if f == 1: header["num_shards"] = v
                    elif f == 2: header["endianness"] = v
                    elif f == 3: header["version"] = {f2: v2 for f2, _, v2 in _fields(v)}
                continue
            e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=0)
            for f, w, v in _fields(val):
                if f == 1: e["dtype"] = v
                elif f == 2:
                    for f2, _, dim in _fields(v):
                        if f2 == 2:
                            e["shape"].append(next((x for f3, _, x in _fields(dim) if f3 == 1), 0))
                elif f == 3: e["shard_id"] = v
                elif f == 4: e["offset"] = v
                elif f == 5: e["size"] = v
                elif f == 6: e["crc32c"] = struct.unpack("<I", v)[0]
            entries[key.decode()] = e
    return header, entries


def load_checkpoint(prefix, verify=True):
    """-> OrderedDict tensor name -> np.ndarray (tf.train.Saver.restore's view of the files)."""
    header, entries = read_index(prefix + ".index", verify)
    shards = {}
    out = OrderedDict()
    for name, e in entries.items():
        if e["shard_id"] not in shards:
            shards[e["shard_id"]] = open("%s.data-%05d-of-%05d" % (prefix, e["shard_id"], header.get("num_shards", 1)), "rb").read()
        raw = shards[e["shard_id"]][e["offset"]:e["offset"] + e["size"]]
        if len(raw) != e["size"]:
            raise IOError("tensor %s exceeds the data file" % name)
        if verify and int(masked(crc32c(raw))) != e["crc32c"]:
            raise IOError("tensor %s: checksum mismatch" % name)
        out[name] = np.frombuffer(raw, _NP[e["dtype"]]).reshape(e["shape"]).copy()
    return out


# ----------------------------------------------------------------------------- SSTable writing
def _build_block(items, restart_interval=16):
    out, restarts, last = bytearray(), [], b""
    for n, (key, val) in enumerate(items):
        shared = 0
        if n % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(key), len(last)) and key[shared] == last[shared]:
                shared += 1
        out += _varint(shared) + _varint(len(key) - shared) + _varint(len(val)) + key[shared:] + val
        last = key
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _emit_block(f, block):
    off = f.tell()
    f.write(block)
    f.write(b"\x00")                                        # no compression
    f.write(struct.pack("<I", int(masked(crc32c(block + b"\x00")))))
    return off, len(block)


def save_checkpoint(prefix, tensors, block_size=4096):
    """tensors: {name: ndarray (float32 / int32)} -> <prefix>.index + <prefix>.data-00000-of-00001."""
    names = sorted(tensors)
    # BundleHeaderProto: num_shards = 1 (field 1), endianness LITTLE = 0 (default, omitted), version {producer: 1}
    items = [(b"", _varint((1 << 3) | 0) + _varint(1) + _ld(3, _varint((1 << 3) | 0) + _varint(1)))]
    offset = 0
    with open(prefix + ".data-00000-of-00001", "wb") as df:
        for name in names:
            a = np.asarray(tensors[name])            # (np.ascontiguousarray would turn scalars into [1])
            a = a if a.flags.c_contiguous else a.copy()
            if a.dtype not in _DT:
                a = a.astype(np.float32)
            raw = a.tobytes()
            shape = b"".join(_ld(2, _varint((1 << 3) | 0) + _varint(int(d))) for d in a.shape)
            entry = _varint((1 << 3) | 0) + _varint(_DT[a.dtype]) + _ld(2, shape)
            if offset:
                entry += _varint((4 << 3) | 0) + _varint(offset)
            entry += _varint((5 << 3) | 0) + _varint(len(raw))
            entry += _varint((6 << 3) | 5) + struct.pack("<I", int(masked(crc32c(raw))))
            items.append((name.encode(), entry))
            df.write(raw)
            offset += len(raw)
    with open(prefix + ".index", "wb") as f:
        index_items, cur, size = [], [], 0
        for key, val in items:
            cur.append((key, val))
            size += len(key) + len(val) + 3
            if size >= block_size:
                off, ln = _emit_block(f, _build_block(cur))
                index_items.append((cur[-1][0], _varint(off) + _varint(ln)))
                cur, size = [], 0
        if cur:
            off, ln = _emit_block(f, _build_block(cur))
            index_items.append((cur[-1][0], _varint(off) + _varint(ln)))
        moff, mlen = _emit_block(f, _build_block([]))
        ioff, ilen = _emit_block(f, _build_block(index_items, restart_interval=1))
        footer = _varint(moff) + _varint(mlen) + _varint(ioff) + _varint(ilen)
        f.write(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", MAGIC))
    return prefix


# ----------------------------------------------------------------------------- AIRModel <-> TF names
SCOPE = "air/rnn/"


def model_to_tensors(state_dict, variables_shapes, beta1=0.9, beta2=0.999):
    """state_dict of AIRModel (TF-named variables + global_step + flat Adam slots are NOT needed:
    per-variable slots are passed as `<name>/Adam`, `<name>/Adam_1` keys if present)."""
    out = {}
    step = int(state_dict.get("global_step", 0))
    for name in variables_shapes:
        out[SCOPE + name] = np.asarray(state_dict[name], np.float32)
        for slot in ("Adam", "Adam_1"):
            key = name + "/" + slot
            a = np.asarray(state_dict[key], np.float32) if key in state_dict else np.zeros(variables_shapes[name], np.float32)
            out["air/training/" + SCOPE + name + "/" + slot] = a
    out["air/global_step"] = np.asarray(step, np.int32)
    out["air/training/beta1_power"] = np.asarray(beta1 ** (step + 1), np.float32)    # TF keeps beta^(t+1)
    out["air/training/beta2_power"] = np.asarray(beta2 ** (step + 1), np.float32)
    return out


def tensors_to_state_dict(tensors):
    """Inverse of model_to_tensors: TF bundle -> dict loadable by AIRModel.load_state_dict (+ slots)."""
    sd = {}
    for name, a in tensors.items():
        if name.startswith(SCOPE):
            sd[name[len(SCOPE):]] = a
        elif name.startswith("air/training/" + SCOPE):
            sd[name[len("air/training/" + SCOPE):]] = a
        elif name == "air/global_step":
            sd["global_step"] = int(a)
    return sd
