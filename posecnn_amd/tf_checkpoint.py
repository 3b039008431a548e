"""Reader for TensorFlow "tensor bundle" checkpoints (`<prefix>.index` + `<prefix>.data-NNNNN-of-MMMMM`),
the format `tf.train.Saver` has written since TF 1.0 and the one the reference restores with
`saver.restore(sess, ckpt)` (lib/fcn/test.py:1809-1811, lib/fcn/train.py:58-91). No TensorFlow needed.

    variables = read_checkpoint("output/lov/vgg16_fcn_color_single_frame_2d_pose_add_lov_iter_160000.ckpt")
    net.load({...})            # or Network.load_file(prefix), which calls this

Format (tensorflow/core/util/tensor_bundle + tensorflow/core/lib/io/table, itself LevelDB's table):
  * the index file is a sorted string table: data blocks, a metaindex block, an index block and a 48-byte
    footer (two BlockHandles as varint64 pairs, zero padding, magic 0xdb4775248b80fb57);
  * a block is a run of entries `varint32 shared | varint32 non_shared | varint32 value_len | key tail |
    value` followed by uint32 restart offsets and their count; on disk each block carries a 5-byte trailer
    (compression type 0 = none / 1 = snappy, masked crc32c);
  * key "" maps to a BundleHeaderProto (num_shards = 1, endianness = 2, version = 3), every other key is a
    variable name mapping to a BundleEntryProto (dtype = 1, shape = 2, shard_id = 3, offset = 4, size = 5,
    crc32c = 6, slices = 7);
  * tensor bytes sit raw (little endian, row major) at [offset, offset + size) of the named data shard.

Integrity: every table block carries a masked CRC32C of (contents + compression byte) and every
BundleEntryProto the masked CRC32C of its tensor bytes (tensorflow/core/lib/hash/crc32c.h: Mask(crc) =
rotr(crc, 15) + 0xa282ead8). Both are verified on read (`verify=True`, the default) through the native
`pcnn_crc32c` of libposecnn_hip.so; a mismatch raises instead of handing wrong weights to the network.

VALIDATION STATUS: no TensorFlow and no checkpoint file exist in the build environment, so this reader is
tested against an independent WRITER of the documented format (tests/test_tf_checkpoint.py: prefix-
compressed keys, several data blocks, restarts, snappy blocks, several shards, correct and corrupted
checksums), not against files written by TensorFlow itself: **parity unpinned** for TF-written files.
The checksum itself is pinned by the published CRC32C known answers (RFC 3720 B.4).
Partitioned variables (`slices`) are reported as an error, not guessed.
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64,
           10: np.bool_, 17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}


def _varint(buf, pos):
    shift, out = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _snappy_decompress(data):
    """Raw snappy (no framing): varint uncompressed length, then literal / copy elements."""
    n, pos = _varint(data, 0)
    out = bytearray()
    while pos < len(data):
        tag = data[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                   # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(data[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += data[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:                                   # copy, 1-byte offset
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | data[pos]
            pos += 1
        elif kind == 2:                                 # copy, 2-byte offset
            ln = (tag >> 2) + 1
            off = int.from_bytes(data[pos:pos + 2], "little")
            pos += 2
        else:                                           # copy, 4-byte offset
            ln = (tag >> 2) + 1
            off = int.from_bytes(data[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError("corrupt snappy block")
        for _ in range(ln):                             # copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("snappy length mismatch")
    return bytes(out)


_CRC_MASK_DELTA = 0xA282EAD8


def crc32c(data, seed=0):
    """CRC32C of a bytes-like object (native, SSE4.2)."""
    import ctypes
    from . import _lib
    buf = bytes(data) if not isinstance(data, (bytes, bytearray)) else data
    arr = (ctypes.c_char * len(buf)).from_buffer_copy(buf) if len(buf) else None
    return int(_lib.lib().pcnn_crc32c(ctypes.cast(arr, ctypes.c_void_p) if arr is not None else None, len(buf), seed))


def crc32c_array(a):
    """CRC32C of a numpy array's bytes without a copy."""
    import ctypes
    from . import _lib
    a = np.ascontiguousarray(a)
    return int(_lib.lib().pcnn_crc32c(ctypes.c_void_p(a.ctypes.data), a.nbytes, 0))


def mask_crc(crc):
    return (((crc >> 15) | (crc << 17)) + _CRC_MASK_DELTA) & 0xFFFFFFFF


def _read_block(f, offset, size, verify=True):
    f.seek(offset)
    raw = f.read(size + 5)
    if len(raw) < size + 5:
        raise ValueError("truncated table block")
    body, ctype = raw[:size], raw[size]
    if verify:
        stored = struct.unpack_from("<I", raw, size + 1)[0]
        if mask_crc(crc32c(raw[:size + 1])) != stored:
            raise ValueError("table block at offset %d: crc32c mismatch (corrupt index file)" % offset)
    if ctype == 1:
        body = _snappy_decompress(body)
    elif ctype != 0:
        raise ValueError("unknown block compression type %d" % ctype)
    return body


def _block_entries(block):
    """Yields (key, value) of one table block (restart array ignored: a linear scan needs no seeks)."""
    if len(block) < 4:
        raise ValueError("table block too small")
    num_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * num_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def _proto_fields(buf):
    """Minimal protobuf wire decoder: yields (field number, wire type, value)."""
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            val = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield field, wt, val


def _parse_shape(buf):
    dims = []
    for field, wt, val in _proto_fields(buf):
        if field == 2 and wt == 2:                      # TensorShapeProto.dim
            size = 0
            for f2, w2, v2 in _proto_fields(val):
                if f2 == 1 and w2 == 0:
                    size = v2 if v2 < (1 << 63) else v2 - (1 << 64)
            dims.append(size)
        elif field == 3 and wt == 0 and val:            # unknown_rank
            raise ValueError("tensor of unknown rank in checkpoint")
    return tuple(dims)


def _parse_entry(buf):
    e = {"dtype": 0, "shape": (), "shard_id": 0, "offset": 0, "size": 0, "slices": 0, "crc32c": None}
    for field, wt, val in _proto_fields(buf):
        if field == 1 and wt == 0:
            e["dtype"] = val
        elif field == 2 and wt == 2:
            e["shape"] = _parse_shape(val)
        elif field == 3 and wt == 0:
            e["shard_id"] = val
        elif field == 4 and wt == 0:
            e["offset"] = val
        elif field == 5 and wt == 0:
            e["size"] = val
        elif field == 6 and wt == 5:
            e["crc32c"] = struct.unpack("<I", val)[0]
        elif field == 7:
            e["slices"] += 1
    return e


def read_index(prefix, verify=True):
    """-> (header dict, {variable name: entry dict}) from `<prefix>.index`."""
    path = prefix + ".index"
    with open(path, "rb") as f:
        f.seek(0, os.SEEK_END)
        fsize = f.tell()
        if fsize < 48:
            raise ValueError("%s is too small to be a tensor bundle index" % path)
        f.seek(fsize - 48)
        footer = f.read(48)
        if struct.unpack_from("<Q", footer, 40)[0] != TABLE_MAGIC:
            raise ValueError("%s: bad table magic (not a TF tensor bundle index)" % path)
        pos = 0
        _, pos = _varint(footer, pos)                   # metaindex handle
        _, pos = _varint(footer, pos)
        idx_off, pos = _varint(footer, pos)
        idx_size, pos = _varint(footer, pos)
        header, entries = {}, {}
        for _, handle in _block_entries(_read_block(f, idx_off, idx_size, verify)):
            boff, p = _varint(handle, 0)
            bsize, p = _varint(handle, p)
            for key, value in _block_entries(_read_block(f, boff, bsize, verify)):
                if key == b"":
                    for field, wt, val in _proto_fields(value):
                        if field == 1 and wt == 0:
                            header["num_shards"] = val
                        elif field == 2 and wt == 0:
                            header["endianness"] = val
                else:
                    entries[key.decode("utf-8")] = _parse_entry(value)
    if header.get("endianness", 0) != 0:
        raise ValueError("big-endian checkpoints are not supported")
    return header, entries


def read_checkpoint(prefix, names=None, verify=True):
    """Reads variables of a TF checkpoint `prefix` into numpy arrays. `names` restricts the set
    (default: everything whose dtype is numeric). `verify` checks the block and tensor checksums."""
    header, entries = read_index(prefix, verify)
    num_shards = max(int(header.get("num_shards", 1)), 1)
    out, files = {}, {}
    try:
        for name in sorted(entries):
            if names is not None and name not in names:
                continue
            e = entries[name]
            if e["slices"]:
                raise ValueError("variable %s is partitioned (slices); not supported" % name)
            if e["dtype"] not in _DTYPES:
                if names is None:
                    continue                            # strings / resources: not network weights
                raise ValueError("variable %s has unsupported dtype %d" % (name, e["dtype"]))
            dt = np.dtype(_DTYPES[e["dtype"]])
            count = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
            if count * dt.itemsize != e["size"]:
                raise ValueError("variable %s: %d bytes on disk, shape %s needs %d" % (name, e["size"], e["shape"], count * dt.itemsize))
            sid = e["shard_id"]
            if sid not in files:
                files[sid] = open("%s.data-%05d-of-%05d" % (prefix, sid, num_shards), "rb")
            files[sid].seek(e["offset"])
            raw = files[sid].read(e["size"])
            if len(raw) != e["size"]:
                raise ValueError("variable %s: data shard truncated" % name)
            arr = np.frombuffer(raw, dtype=dt.newbyteorder("<")).reshape(e["shape"])
            if verify and e["crc32c"] is not None and mask_crc(crc32c_array(arr)) != e["crc32c"]:
                raise ValueError("variable %s: crc32c mismatch (corrupt data shard)" % name)
            out[name] = arr.astype(dt, copy=True)
    finally:
        for fh in files.values():
            fh.close()
    return out


def to_layer_dict(variables):
    """{'conv1_1/weights': a, 'conv1_1/biases': b, ...} -> {layer: {'weights': a, 'biases': b}} — the layout
    `Network.load` takes (network.py:71-107). Optimizer slots ('<var>/Momentum', 'global_step', ...) are dropped."""
    data = {}
    for key, arr in variables.items():
        if key.count("/") != 1:
            continue
        layer_name, pname = key.split("/")
        if pname in ("weights", "biases"):
            data.setdefault(layer_name, {})[pname] = arr
    return data
