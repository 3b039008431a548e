"""posecnn_amd.tf_checkpoint against an independent WRITER of the documented tensor-bundle format
(no TensorFlow and no TF-written checkpoint exist offline — see the module's validation note)."""
import struct

import numpy as np
import pytest

from posecnn_amd import tf_checkpoint as tfc

F = np.float32


def varint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def pb_varint(field, v):
    return varint(field << 3) + varint(v)


def pb_bytes(field, b):
    return varint((field << 3) | 2) + varint(len(b)) + b


def shape_proto(shape):
    return b"".join(pb_bytes(2, pb_varint(1, d)) for d in shape)


def crc32c_bitwise(data, crc=0):
    """Independent restatement (bit by bit, reflected polynomial 0x82F63B78) of the checksum the
    reader computes natively."""
    crc ^= 0xFFFFFFFF
    for b in bytes(data):
        crc ^= b
        for _ in range(8):
            crc = (crc >> 1) ^ 0x82F63B78 if crc & 1 else crc >> 1
    return crc ^ 0xFFFFFFFF


def masked(crc):
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def entry_proto(dtype, shape, shard, offset, size, crc=0xDEADBEEF):
    out = pb_varint(1, dtype) + pb_bytes(2, shape_proto(shape))
    if shard:
        out += pb_varint(3, shard)
    if offset:
        out += pb_varint(4, offset)
    out += pb_varint(5, size) + varint((6 << 3) | 5) + struct.pack("<I", crc)
    return out


def snappy_compress(data):
    """A deliberately simple encoder: literals, plus 2-byte-offset copies for 8-byte repeats."""
    out = bytearray(varint(len(data)))
    i, lit = 0, bytearray()

    def flush():
        nonlocal lit
        while lit:
            chunk, lit = lit[:60], lit[60:]
            out.append((len(chunk) - 1) << 2)
            out.extend(chunk)
    while i < len(data):
        found = 0
        if i >= 8:
            for off in (8, 16, 4):
                if i - off >= 0 and data[i:i + 8] == data[i - off:i - off + 8] and len(data[i:i + 8]) == 8:
                    found = off
                    break
        if found:
            flush()
            out.append(((8 - 1) << 2) | 2)
            out.extend(struct.pack("<H", found))
            i += 8
        else:
            lit.append(data[i])
            i += 1
    flush()
    return bytes(out)


def build_block(items, restart_interval=4):
    buf, restarts, prev = bytearray(), [], b""
    for n, (k, v) in enumerate(items):
        shared = 0
        if n % restart_interval == 0:
            restarts.append(len(buf))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        buf += varint(shared) + varint(len(k) - shared) + varint(len(v)) + k[shared:] + v
        prev = k
    for r in restarts:
        buf += struct.pack("<I", r)
    buf += struct.pack("<I", len(restarts))
    return bytes(buf)


def write_bundle(prefix, tensors, shards=1, per_block=3, compress=False, checksums=True):
    names = sorted(tensors)
    data = [bytearray() for _ in range(shards)]
    items = [(b"", pb_varint(1, shards) + pb_varint(2, 0) + pb_bytes(3, pb_varint(1, 1)))]
    for n, name in enumerate(names):
        a = np.array(tensors[name], order="C")   # (ascontiguousarray would turn a scalar into shape (1,))
        sid = n % shards
        dtype = {np.dtype(F): 1, np.dtype(np.int32): 3, np.dtype(np.int64): 9, np.dtype(np.float64): 2}[a.dtype]
        crc = masked(crc32c_bitwise(a.tobytes())) if checksums else 0xDEADBEEF
        items.append((name.encode(), entry_proto(dtype, a.shape, sid, len(data[sid]), a.nbytes, crc)))
        data[sid] += a.tobytes()
    for sid in range(shards):
        with open("%s.data-%05d-of-%05d" % (prefix, sid, shards), "wb") as f:
            f.write(bytes(data[sid]))
    out, index_items = bytearray(), []

    def emit(block):
        body, ctype = (snappy_compress(block), 1) if compress else (block, 0)
        handle = varint(len(out)) + varint(len(body))
        trailer = struct.pack("<I", masked(crc32c_bitwise(body + bytes([ctype])))) if checksums else b"\0\0\0\0"
        out.extend(body + bytes([ctype]) + trailer)
        return handle
    for i in range(0, len(items), per_block):
        chunk = items[i:i + per_block]
        index_items.append((chunk[-1][0] + b"\xff", emit(build_block(chunk))))
    meta = emit(build_block([]))
    index = emit(build_block(index_items, restart_interval=1))
    footer = meta + index
    footer += b"\0" * (40 - len(footer)) + struct.pack("<Q", tfc.TABLE_MAGIC)
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out) + footer)


def sample_tensors(rng):
    t = {"conv1_1/weights": rng.standard_normal((3, 3, 3, 8)).astype(F), "conv1_1/biases": rng.standard_normal(8).astype(F),
         "conv1_1/weights/Momentum": rng.standard_normal((3, 3, 3, 8)).astype(F), "conv1_2/weights": rng.standard_normal((3, 3, 8, 8)).astype(F),
         "conv1_2/biases": np.zeros(8, F), "fc6/weights": rng.standard_normal((32, 16)).astype(F), "fc6/biases": rng.standard_normal(16).astype(F),
         "global_step": np.array(160000, np.int64), "Variable": np.arange(6, dtype=np.int32).reshape(2, 3)}
    return t


@pytest.mark.parametrize("shards,per_block,compress", [(1, 3, False), (2, 2, False), (1, 100, True), (3, 1, True)])
def test_reader_round_trip(tmp_path, shards, per_block, compress):
    rng = np.random.default_rng(shards * 10 + per_block)
    tensors = sample_tensors(rng)
    prefix = str(tmp_path / "model.ckpt")
    write_bundle(prefix, tensors, shards, per_block, compress)
    header, entries = tfc.read_index(prefix)
    assert header["num_shards"] == shards and set(entries) == set(tensors)
    got = tfc.read_checkpoint(prefix)
    for k, v in tensors.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k
    only = tfc.read_checkpoint(prefix, names={"fc6/biases"})
    assert list(only) == ["fc6/biases"]
    layers = tfc.to_layer_dict(got)
    assert sorted(layers) == ["conv1_1", "conv1_2", "fc6"] and set(layers["conv1_1"]) == {"weights", "biases"}


def test_snappy_decoder_handles_overlapping_copies():
    data = b"abcdabcdabcdabcdabcdabcdXYZ" + bytes(range(200)) + b"abcdabcd" * 9
    assert tfc._snappy_decompress(snappy_compress(data)) == data
    # hand-built: literal "ab" then a copy of length 6 from offset 2 (overlaps its own output) -> "abababab"
    stream = bytes([8, (2 - 1) << 2]) + b"ab" + bytes([((6 - 4) << 2) | 1, 2])
    assert tfc._snappy_decompress(stream) == b"abababab"


def test_errors(tmp_path):
    prefix = str(tmp_path / "bad")
    with open(prefix + ".index", "wb") as f:
        f.write(b"\0" * 64)
    with pytest.raises(ValueError):
        tfc.read_index(prefix)
    rng = np.random.default_rng(0)
    write_bundle(prefix, sample_tensors(rng))
    with open(prefix + ".data-00000-of-00001", "r+b") as f:
        f.truncate(10)
    with pytest.raises(ValueError):
        tfc.read_checkpoint(prefix)


def test_network_load_file_takes_a_checkpoint_prefix(tmp_path):
    import torch
    from cpu_reference import vgg16_convs_cpu
    rng = np.random.default_rng(5)
    w = rng.standard_normal((3, 3, 5, 4)).astype(F); b = rng.standard_normal(4).astype(F)
    prefix = str(tmp_path / "net.ckpt")
    write_bundle(prefix, {"c/weights": w, "c/biases": b, "c/weights/Momentum": w * 0, "global_step": np.array(7, np.int64)}, compress=True)
    net = vgg16_convs_cpu("COLOR", 22, 64, (1.0,), 1.0, -1.0)
    assert net.load_file(prefix) == ["c"]
    assert torch.equal(net.vars["c/biases"], torch.from_numpy(b))
    assert torch.equal(net.vars["c/weights"], torch.from_numpy(w).permute(3, 2, 0, 1))


def test_crc32c_known_answers():
    """RFC 3720 appendix B.4 test vectors + the classic check value; pins the native checksum
    independently of this file's writer."""
    assert tfc.crc32c(b"123456789") == 0xE3069283
    assert tfc.crc32c(bytes(32)) == 0x8A9136AA
    assert tfc.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert tfc.crc32c(bytes(range(32))) == 0x46DD794E
    assert tfc.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    assert tfc.crc32c(b"") == 0
    a, b = b"hello wor", b"ld, crc32c in pieces"
    assert tfc.crc32c(b, tfc.crc32c(a)) == tfc.crc32c(a + b) == crc32c_bitwise(a + b)
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, 100003, dtype=np.uint8)
    assert tfc.crc32c_array(x[3:]) == crc32c_bitwise(x[3:].tobytes())     # unaligned start, odd length
    assert tfc.mask_crc(0) == 0xA282EAD8


def test_checksums_are_verified(tmp_path):
    rng = np.random.default_rng(1)
    tensors = sample_tensors(rng)
    prefix = str(tmp_path / "m.ckpt")
    write_bundle(prefix, tensors, shards=2, per_block=2, compress=True)
    assert set(tfc.read_checkpoint(prefix)) == set(tensors)
    # one flipped bit in a tensor payload
    shard = prefix + ".data-00001-of-00002"
    raw = bytearray(open(shard, "rb").read()); raw[5] ^= 0x10
    open(shard, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="crc32c"):
        tfc.read_checkpoint(prefix)
    assert set(tfc.read_checkpoint(prefix, verify=False)) == set(tensors)   # explicit opt-out still reads
    # one flipped bit in an index block
    write_bundle(prefix, tensors, shards=1, per_block=3)
    raw = bytearray(open(prefix + ".index", "rb").read()); raw[12] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="crc32c"):
        tfc.read_index(prefix)
    # files whose writer left the checksums blank are refused unless verification is turned off
    write_bundle(prefix, tensors, checksums=False)
    with pytest.raises(ValueError, match="crc32c"):
        tfc.read_checkpoint(prefix)
    assert set(tfc.read_checkpoint(prefix, verify=False)) == set(tensors)
