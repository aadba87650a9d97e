"""TensorFlow-1.x checkpoint interop (tensor-bundle "V2" files) without TensorFlow.

The reference restores ``model.ckpt-<step>`` through ``slim.assign_from_checkpoint(..., ignore_missing_vars=True)``
(cub/code/SB_model48i/model.py:592-602); the variable names (``encoder_0/conv2d_3/V``, ``.../b``: cub/code/nn.py:40-46, 644-652)
are the ones this package uses, so reading the bundle is all that is needed to load a published checkpoint
(cub/train/checkpoints/model.ckpt-60000.{index,data-00000-of-00001}; Git-LFS stubs in the reference tree).

A bundle is ``<prefix>.index`` -- a LevelDB-format table (tensorflow/core/lib/io/table*.cc) mapping "" to a BundleHeaderProto
and every tensor name to a BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c} -- plus ``<prefix>.data-SSSSS-of-NNNNN``
shards holding the raw little-endian tensor bytes (tensorflow/core/util/tensor_bundle/tensor_bundle.cc).  ``read_bundle`` parses
both; ``write_bundle`` emits the same format (uncompressed blocks, as TF's BundleWriter does) so that weights trained here can be
handed back to the reference's TF graph.  NumPy only -- outside the hot path.
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
# tensorflow/core/framework/types.proto
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
          17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
DTYPE_CODES = {np.dtype(v): k for k, v in DTYPES.items()}


# ----------------------------------------------------------------------------- varints / protobuf wire format
def _varint(buf, pos):
    out, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if b < 0x80:
            return out, pos
        shift += 7


def _put_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _fields(buf):
    """protobuf message -> list of (field number, wire type, value); value = int (varint / fixed) or bytes."""
    pos, out = 0, []
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        num, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]; pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = bytes(buf[pos:pos + n]); pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]; pos += 4
        else:
            raise ValueError("unsupported protobuf wire type {}".format(wt))
        out.append((num, wt, v))
    return out


def _parse_entry(buf):
    """BundleEntryProto (tensorflow/core/protobuf/tensor_bundle.proto)."""
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": 0, "slices": False}
    for num, _wt, v in _fields(buf):
        if num == 1:
            e["dtype"] = v
        elif num == 2:      # TensorShapeProto: repeated Dim dim = 2 { int64 size = 1 }
            for n2, _w2, v2 in _fields(v):
                if n2 == 2:
                    size = 0
                    for n3, _w3, v3 in _fields(v2):
                        if n3 == 1:
                            size = v3
                    e["shape"].append(size)
        elif num == 3:
            e["shard_id"] = v
        elif num == 4:
            e["offset"] = v
        elif num == 5:
            e["size"] = v
        elif num == 6:
            e["crc32c"] = v
        elif num == 7:
            e["slices"] = True
    return e


def _parse_header(buf):
    h = {"num_shards": 1, "endianness": 0}
    for num, _wt, v in _fields(buf):
        if num == 1:
            h["num_shards"] = v
        elif num == 2:
            h["endianness"] = v
    return h


# ----------------------------------------------------------------------------- LevelDB-format table
def _block_entries(block):
    """One table block (without its 5-byte trailer) -> [(key, value)] (prefix-compressed keys, restart array ignored)."""
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared]); pos += non_shared
        out.append((key, bytes(block[pos:pos + vlen]))); pos += vlen
    return out


def _read_block(data, offset, size):
    ctype = data[offset + size]
    if ctype != 0:
        raise NotImplementedError("compressed table block (type {}): TF's BundleWriter writes uncompressed blocks".format(ctype))
    return data[offset:offset + size]


def read_index(path):
    """``<prefix>.index`` -> (header dict, {tensor name: entry dict})."""
    data = open(path, "rb").read()
    if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != TABLE_MAGIC:
        raise ValueError("{} is not a TensorFlow checkpoint index (bad table magic; a Git-LFS stub?)".format(path))
    footer = data[-48:]
    _mo, p = _varint(footer, 0)
    _ms, p = _varint(footer, p)
    io_, p = _varint(footer, p)
    isz, p = _varint(footer, p)
    header, entries = None, {}
    for _k, handle in _block_entries(_read_block(data, io_, isz)):
        off, q = _varint(handle, 0)
        size, q = _varint(handle, q)
        for key, val in _block_entries(_read_block(data, off, size)):
            if key == b"":
                header = _parse_header(val)
            else:
                entries[key.decode()] = _parse_entry(val)
    if header is None:
        raise ValueError("{}: no bundle header".format(path))
    if header["endianness"] != 0:
        raise NotImplementedError("big-endian bundle")
    return header, entries


def read_bundle(prefix, names=None):
    """``model.ckpt-60000`` -> {variable name: np.ndarray}.  ``names``: optional iterable restricting what is read."""
    header, entries = read_index(prefix + ".index")
    want = set(names) if names is not None else None
    shards = {}
    out = {}
    for name, e in entries.items():
        if want is not None and name not in want:
            continue
        if e["slices"]:
            raise NotImplementedError("{}: partitioned variable".format(name))
        if e["dtype"] not in DTYPES:
            continue                                  # strings etc.: nothing the model restores
        sid = e["shard_id"]
        if sid not in shards:
            shards[sid] = np.memmap("{}.data-{:05d}-of-{:05d}".format(prefix, sid, header["num_shards"]), dtype=np.uint8, mode="r")
        dt = np.dtype(DTYPES[e["dtype"]])
        count = int(np.prod(e["shape"])) if e["shape"] else 1
        if count * dt.itemsize != e["size"]:
            raise ValueError("{}: entry size {} does not match shape {} of {}".format(name, e["size"], e["shape"], dt))
        raw = np.asarray(shards[sid][e["offset"]:e["offset"] + e["size"]])
        out[name] = raw.view(dt).reshape(tuple(e["shape"])).copy()
    return out


def is_bundle(prefix):
    return os.path.exists(prefix + ".index")


# ----------------------------------------------------------------------------- writer (export / test fixtures)
_CRC_TABLE = None


def crc32c(data, crc=0):
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t.append(c)
        _CRC_TABLE = t
    c = crc ^ 0xFFFFFFFF
    t = _CRC_TABLE
    for b in bytes(data):
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _mask(crc):
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _pb_varint_field(num, v):
    return _put_varint((num << 3) | 0) + _put_varint(v)


def _pb_bytes_field(num, b):
    return _put_varint((num << 3) | 2) + _put_varint(len(b)) + b


def _entry_proto(dtype_code, shape, offset, size, crc):
    shape_pb = b"".join(_pb_bytes_field(2, _pb_varint_field(1, int(d))) for d in shape)
    out = _pb_varint_field(1, dtype_code) + _pb_bytes_field(2, shape_pb)
    if offset:
        out += _pb_varint_field(4, offset)
    out += _pb_varint_field(5, size)
    out += _put_varint((6 << 3) | 5) + struct.pack("<I", _mask(crc))
    return out


def _build_block(items, restart_interval=16):
    out, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def write_bundle(prefix, tensors, block_bytes=4096, with_crc=True):
    """{name: np.ndarray} -> ``prefix.index`` + ``prefix.data-00000-of-00001`` (one shard, uncompressed table blocks)."""
    names = sorted(tensors)
    entries, offset = [], 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for n in names:
            a = np.asarray(tensors[n])
            a = a if a.flags.c_contiguous else a.copy()      # (np.ascontiguousarray would turn a scalar into shape (1,))
            raw = a.tobytes()
            f.write(raw)
            entries.append((n.encode(), _entry_proto(DTYPE_CODES[a.dtype], a.shape, offset, len(raw), crc32c(raw) if with_crc else 0)))
            offset += len(raw)
    header = _pb_varint_field(1, 1) + _pb_bytes_field(3, _pb_varint_field(1, 1))        # num_shards = 1, version.producer = 1
    items = [(b"", header)] + entries
    blocks, cur, size = [], [], 0
    for kv in items:
        cur.append(kv); size += len(kv[0]) + len(kv[1]) + 3
        if size >= block_bytes:
            blocks.append(cur); cur, size = [], 0
    if cur:
        blocks.append(cur)
    out = bytearray()

    def emit(block):
        off = len(out)
        out.extend(block)
        out.append(0)                                                        # kNoCompression
        out.extend(struct.pack("<I", _mask(crc32c(block + b"\x00"))))
        return off, len(block)

    index_items = []
    for blk in blocks:
        off, sz = emit(_build_block(blk))
        index_items.append((blk[-1][0], _put_varint(off) + _put_varint(sz)))      # separator >= last key of the block
    moff, msz = emit(_build_block([]))
    ioff, isz = emit(_build_block(index_items, restart_interval=1))
    footer = _put_varint(moff) + _put_varint(msz) + _put_varint(ioff) + _put_varint(isz)
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    out.extend(footer)
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))


# ----------------------------------------------------------------------------- model <-> bundle naming
def to_trainer_state(bundle):
    """Split a TF checkpoint's variables into this package's checkpoint pieces: parameters by name, Adam slots
    (``<var>/Adam`` = m, ``<var>/Adam_1`` = v: tf.train.AdamOptimizer slot names; one optimizer per loss key adds ``_k``
    suffixes to the later optimizers' slots of shared names only, which this model does not have) and everything else
    (``other``: the global step, the optimizers' ``beta1_power`` / ``beta2_power`` scalars and the reference's UNNAMED
    non-trainable variables -- ``Variable``, ``Variable_1``, ...: the Lagrangian multipliers lon / loa / lor and the seven EMAs,
    model.py:503, 829-834, 862-865, 890, 921 -- which carry no name of their own: Trainer._initialize_from_tf maps them by creation
    order when exactly those ten are present and reports them as not restored otherwise)."""
    params, m, v, other = {}, {}, {}, {}
    for name, arr in bundle.items():
        if name.endswith("/Adam"):
            m[name[:-5]] = arr
        elif name.endswith("/Adam_1"):
            v[name[:-7]] = arr
        elif name.endswith("/V") or name.endswith("/b"):
            params[name] = arr
        else:
            other[name] = arr
    return params, m, v, other
