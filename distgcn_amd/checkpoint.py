"""TensorFlow V2 checkpoint-bundle reader/writer, written from the published format.

The reference restores weights with ``tf.compat.v1.train.Saver`` (reference
``mwis_dqn_call.py:188-195``, ``mwis_gdpg_call.py:109-118``).  TensorFlow is not a
dependency of this package, so the two files of a bundle are parsed directly:

* ``<prefix>.index`` - a LevelDB-format sorted string table (prefix-compressed
  keys, restart array, 5-byte block trailer, 48-byte footer).  Key ``""`` holds a
  ``BundleHeaderProto``; every other key is a variable name whose value is a
  ``BundleEntryProto{dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6}``.
* ``<prefix>.data-00000-of-00001`` - raw little-endian tensor bytes.

Only what the GCN path needs is supported: uncompressed blocks, one shard,
DT_FLOAT / DT_DOUBLE / DT_INT32 / DT_INT64 tensors.  A missing checkpoint is a
hard error (the reference silently keeps random weights, ``mwis_dqn_test.py:215-219``).
"""
from __future__ import annotations

import os
import struct
from typing import Dict, Tuple

import numpy as np

_TABLE_MAGIC = 0xDB4775248B80FB57
_DTYPES = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 9: np.dtype("<i8")}
_DTYPE_IDS = {np.dtype("float32"): 1, np.dtype("float64"): 2, np.dtype("int32"): 3, np.dtype("int64"): 9}


class CheckpointError(RuntimeError):
    pass


# ----------------------------------------------------------------------------- varint / protobuf
def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    out = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if b < 0x80:
            return out, pos
        shift += 7


def _put_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _proto_fields(buf: bytes):
    """Yield (field_number, wire_type, value) of one protobuf message."""
    pos = 0
    n = len(buf)
    while pos < n:
        tag, pos = _varint(buf, pos)
        field, wire = tag >> 3, tag & 7
        if wire == 0:
            val, pos = _varint(buf, pos)
        elif wire == 1:
            val = buf[pos:pos + 8]
            pos += 8
        elif wire == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wire == 5:
            val = buf[pos:pos + 4]
            pos += 4
        else:
            raise CheckpointError("unsupported protobuf wire type %d" % wire)
        yield field, wire, val


def _parse_shape(buf: bytes) -> Tuple[int, ...]:
    dims = []
    for field, _, val in _proto_fields(buf):
        if field == 2:  # repeated Dim
            size = 0
            for f2, _, v2 in _proto_fields(val):
                if f2 == 1:
                    size = v2
            dims.append(size)
    return tuple(dims)


def _parse_entry(buf: bytes) -> dict:
    ent = {"dtype": 0, "shape": (), "shard_id": 0, "offset": 0, "size": 0, "crc32c": None}
    for field, wire, val in _proto_fields(buf):
        if field == 1:
            ent["dtype"] = val
        elif field == 2:
            ent["shape"] = _parse_shape(val)
        elif field == 3:
            ent["shard_id"] = val
        elif field == 4:
            ent["offset"] = val
        elif field == 5:
            ent["size"] = val
        elif field == 6 and wire == 5:
            ent["crc32c"] = struct.unpack("<I", val)[0]
    return ent


# ----------------------------------------------------------------------------- crc32c (Castagnoli)
def _make_crc_table():
    tbl = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tbl.append(c)
    return tbl


_CRC_TABLE = _make_crc_table()


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    tbl = _CRC_TABLE
    for b in data:
        c = tbl[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _mask_crc(c: int) -> int:
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ----------------------------------------------------------------------------- table reader
def _read_block(data: bytes, offset: int, size: int) -> bytes:
    block = data[offset:offset + size]
    ctype = data[offset + size]
    if ctype != 0:
        raise CheckpointError("compressed checkpoint index blocks are not supported")
    return block


def _block_entries(block: bytes):
    num_restarts = struct.unpack("<I", block[-4:])[0]
    limit = len(block) - 4 - 4 * num_restarts
    pos = 0
    key = b""
    while pos < limit:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_index(index_path: str) -> Dict[str, dict]:
    with open(index_path, "rb") as fh:
        data = fh.read()
    if len(data) < 48:
        raise CheckpointError("%s: too short for a table footer" % index_path)
    footer = data[-48:]
    if struct.unpack("<Q", footer[40:])[0] != _TABLE_MAGIC:
        raise CheckpointError("%s: bad table magic" % index_path)
    pos = 0
    _, pos = _varint(footer, pos)  # metaindex offset
    _, pos = _varint(footer, pos)  # metaindex size
    idx_off, pos = _varint(footer, pos)
    idx_size, pos = _varint(footer, pos)
    entries: Dict[str, dict] = {}
    for _, handle in _block_entries(_read_block(data, idx_off, idx_size)):
        off, p = _varint(handle, 0)
        size, p = _varint(handle, p)
        for key, val in _block_entries(_read_block(data, off, size)):
            if key == b"":
                continue  # BundleHeaderProto
            entries[key.decode("utf-8")] = _parse_entry(val)
    return entries


def resolve_prefix(path: str) -> str:
    """Accept a model directory (with a ``checkpoint`` text file, as
    ``tf.train.get_checkpoint_state`` does) or a bundle prefix."""
    if os.path.isdir(path):
        state = os.path.join(path, "checkpoint")
        if os.path.isfile(state):
            with open(state, "r") as fh:
                for line in fh:
                    if line.startswith("model_checkpoint_path:"):
                        name = line.split(":", 1)[1].strip().strip('"')
                        cand = name if os.path.isabs(name) else os.path.join(path, name)
                        if os.path.isfile(cand + ".index"):
                            return cand
                        # the recorded path may be stale (trained elsewhere): fall back on basename
                        cand = os.path.join(path, os.path.basename(name))
                        if os.path.isfile(cand + ".index"):
                            return cand
        cand = os.path.join(path, "model.ckpt")
        if os.path.isfile(cand + ".index"):
            return cand
        raise CheckpointError("no checkpoint bundle found in directory %r" % path)
    if os.path.isfile(path + ".index"):
        return path
    raise CheckpointError("no checkpoint bundle at %r" % path)


def load_bundle(path: str, verify_crc: bool = True) -> Dict[str, np.ndarray]:
    """Return {variable name: ndarray} for every tensor in the bundle."""
    prefix = resolve_prefix(path)
    entries = read_index(prefix + ".index")
    shards: Dict[int, bytes] = {}
    out: Dict[str, np.ndarray] = {}
    for name, ent in entries.items():
        sid = ent["shard_id"]
        if sid not in shards:
            # shard file names carry the shard count; only single-shard bundles are written by Saver
            cands = [f for f in os.listdir(os.path.dirname(prefix) or ".")
                     if f.startswith(os.path.basename(prefix) + ".data-%05d-of-" % sid)]
            if not cands:
                raise CheckpointError("missing data shard %d for %r" % (sid, prefix))
            with open(os.path.join(os.path.dirname(prefix) or ".", cands[0]), "rb") as fh:
                shards[sid] = fh.read()
        if ent["dtype"] not in _DTYPES:
            continue  # strings etc.: nothing on the GCN path uses them
        raw = shards[sid][ent["offset"]:ent["offset"] + ent["size"]]
        if len(raw) != ent["size"]:
            raise CheckpointError("tensor %r truncated" % name)
        if verify_crc and ent["crc32c"] is not None and ent["size"] <= (1 << 20):
            if _mask_crc(crc32c(raw)) != ent["crc32c"]:
                raise CheckpointError("tensor %r fails its crc32c" % name)
        arr = np.frombuffer(raw, dtype=_DTYPES[ent["dtype"]]).reshape(ent["shape"]).copy()
        out[name] = arr
    return out


# ----------------------------------------------------------------------------- table writer
def _block_build(items) -> bytes:
    """One table block with a restart point at every entry (no prefix sharing)."""
    body = bytearray()
    restarts = []
    for key, val in items:
        restarts.append(len(body))
        body += _put_varint(0) + _put_varint(len(key)) + _put_varint(len(val)) + key + val
    if not restarts:
        restarts = [0]
    for r in restarts:
        body += struct.pack("<I", r)
    body += struct.pack("<I", len(restarts))
    return bytes(body)


def _block_emit(out: bytearray, block: bytes) -> bytes:
    handle = _put_varint(len(out)) + _put_varint(len(block))
    out += block
    out += b"\x00" + struct.pack("<I", _mask_crc(crc32c(block + b"\x00")))
    return handle


def _shape_proto(shape) -> bytes:
    out = bytearray()
    for d in shape:
        dim = b"\x08" + _put_varint(int(d))
        out += b"\x12" + _put_varint(len(dim)) + dim
    return bytes(out)


def save_bundle(prefix: str, tensors: Dict[str, np.ndarray]) -> None:
    """Write ``tensors`` as a single-shard V2 bundle plus the ``checkpoint`` state
    file that ``resolve_prefix`` / TF's ``get_checkpoint_state`` look for."""
    os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
    data = bytearray()
    items = []
    for name in sorted(tensors, key=lambda s: s.encode("utf-8")):
        arr = np.asarray(tensors[name], order="C")
        if arr.dtype not in _DTYPE_IDS:
            raise CheckpointError("unsupported dtype %s for %r" % (arr.dtype, name))
        raw = arr.astype(arr.dtype.newbyteorder("<")).tobytes()
        ent = bytearray()
        ent += b"\x08" + _put_varint(_DTYPE_IDS[arr.dtype])
        shp = _shape_proto(arr.shape)
        ent += b"\x12" + _put_varint(len(shp)) + shp
        if len(data):
            ent += b"\x20" + _put_varint(len(data))
        ent += b"\x28" + _put_varint(len(raw))
        ent += b"\x35" + struct.pack("<I", _mask_crc(crc32c(raw)))
        items.append((name.encode("utf-8"), bytes(ent)))
        data += raw
    header = b"\x08\x01\x1a\x02\x08\x01"  # num_shards=1, little endian, version{producer=1}
    items.insert(0, (b"", header))

    out = bytearray()
    data_handle = _block_emit(out, _block_build(items))
    meta_handle = _block_emit(out, _block_build([]))
    last_key = items[-1][0] + b"\x00"
    index_handle = _block_emit(out, _block_build([(last_key, data_handle)]))
    footer = meta_handle + index_handle
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", _TABLE_MAGIC)
    out += footer
    with open(prefix + ".index", "wb") as fh:
        fh.write(bytes(out))
    with open(prefix + ".data-00000-of-00001", "wb") as fh:
        fh.write(bytes(data))
    state = os.path.join(os.path.dirname(prefix) or ".", "checkpoint")
    base = os.path.basename(prefix)
    with open(state, "w") as fh:
        fh.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))
