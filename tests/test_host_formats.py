"""Host C++ runtime: crc32c, TFRecord, tf.Example, SSTable / tensor bundle, event files.

No TensorFlow is available, so the oracles are (a) published CRC32-C test vectors,
(b) an independent pure-Python implementation of each container written from
the format description, and (c) hand-assembled protobuf bytes.
"""
import os
import struct

import numpy as np
import pytest

from distributed_tensorflow_amd.io import bundle, events, hostlib, tfrecord


# ------------------------------------------------------------------ pure-python oracles
def _crc_table():
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t.append(c)
    return t


_T = _crc_table()


def py_crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c = _T[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def py_mask(c: int) -> int:
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def py_tfrecord(payload: bytes) -> bytes:
    head = struct.pack("<Q", len(payload))
    return head + struct.pack("<I", py_mask(py_crc32c(head))) + payload + struct.pack("<I", py_mask(py_crc32c(payload)))


def _varint(b, i):
    v = s = 0
    while True:
        c = b[i]
        i += 1
        v |= (c & 0x7F) << s
        if not c & 0x80:
            return v, i
        s += 7


def py_read_table(blob: bytes):
    """Independent LevelDB-table reader; verifies every block checksum."""
    assert struct.unpack("<Q", blob[-8:])[0] == 0xDB4775248B80FB57
    footer = blob[-48:-8]
    i = 0
    _, i = _varint(footer, i)
    _, i = _varint(footer, i)
    ioff, i = _varint(footer, i)
    isize, i = _varint(footer, i)

    def block(off, size):
        body = blob[off:off + size]
        assert blob[off + size] == 0  # no compression
        stored = struct.unpack("<I", blob[off + size + 1:off + size + 5])[0]
        assert stored == py_mask(py_crc32c(blob[off:off + size + 1]))
        nrest = struct.unpack("<I", body[-4:])[0]
        limit = len(body) - 4 - 4 * nrest
        restarts = struct.unpack("<%dI" % nrest, body[limit:limit + 4 * nrest])
        out, key, p = [], b"", 0
        starts = []
        while p < limit:
            starts.append(p)
            shared, p = _varint(body, p)
            unshared, p = _varint(body, p)
            vlen, p = _varint(body, p)
            key = key[:shared] + body[p:p + unshared]
            p += unshared
            out.append((key, body[p:p + vlen]))
            p += vlen
        return out, restarts, starts

    idx, _, _ = block(ioff, isize)
    entries, all_restarts = [], []
    for _, handle in idx:
        off, j = _varint(handle, 0)
        size, j = _varint(handle, j)
        e, r, s = block(off, size)
        entries += e
        all_restarts.append((r, s))
    return entries, all_restarts


# ------------------------------------------------------------------ crc32c
def test_crc32c_known_vectors():
    assert tfrecord.crc32c(b"123456789") == 0xE3069283
    assert tfrecord.crc32c(bytes(32)) == 0x8A9136AA
    assert tfrecord.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert tfrecord.crc32c(bytes(range(32))) == 0x46DD794E
    assert tfrecord.crc32c(b"") == 0


def test_crc32c_hw_sw_python_agree():
    L = hostlib.lib()
    rng = np.random.default_rng(0)
    for n in (1, 7, 8, 9, 63, 64, 65, 1000, 4097):
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert tfrecord.crc32c(data) == py_crc32c(data) == L.dtfa_crc32c_sw_for_test(data, n)
        assert tfrecord.masked_crc32c(data) == py_mask(py_crc32c(data))


# ------------------------------------------------------------------ TFRecord
def test_tfrecord_bytes_match_format(tmp_path):
    p = str(tmp_path / "a.tfrecord")
    payloads = [b"", b"hello", os.urandom(1000), b"x" * 70000]
    with tfrecord.TFRecordWriter(p) as w:
        for x in payloads:
            w.write(x)
    blob = open(p, "rb").read()
    assert blob == b"".join(py_tfrecord(x) for x in payloads)
    assert list(tfrecord.tfrecord_iterator(p)) == payloads
    off, ln = tfrecord.tfrecord_index(p)
    assert list(ln) == [len(x) for x in payloads]
    assert off[0] == 0 and off[1] == 16 and off[2] == 16 + 16 + 5


def test_tfrecord_detects_corruption(tmp_path):
    p = str(tmp_path / "a.tfrecord")
    with tfrecord.TFRecordWriter(p) as w:
        w.write(b"hello world")
    blob = bytearray(open(p, "rb").read())
    blob[14] ^= 1
    open(p, "wb").write(bytes(blob))
    with pytest.raises(IOError):
        list(tfrecord.tfrecord_iterator(p))
    assert len(list(tfrecord.tfrecord_iterator(p, verify_crc=False))) == 1
    open(p, "wb").write(bytes(blob[:-3]))
    with pytest.raises(IOError):
        list(tfrecord.tfrecord_iterator(p))


# ------------------------------------------------------------------ tf.Example
def test_example_golden_bytes():
    # Example{features{feature{key:"label" value{int64_list{value:[3]}}}}}
    got = tfrecord.make_example({"label": 3})
    entry = b"\x0a\x05label" + b"\x12\x05" + (b"\x1a\x03" + b"\x0a\x01\x03")
    features = b"\x0a" + bytes([len(entry)]) + entry
    want = b"\x0a" + bytes([len(features)]) + features
    assert got == want


def test_example_roundtrip_reference_schema():
    img = np.random.default_rng(1).integers(0, 256, (5, 7, 3), dtype=np.uint8)
    rec = tfrecord.make_example({"image_raw": img.tobytes(), "label": 4, "height": 5, "width": 7, "channel": 3,
                                 "floats": [0.5, -1.25], "image/encoded": [b"\xff\xd8abc", b""],
                                 "neg": [-1, 1 << 40]})
    ex = tfrecord.parse_example(rec)
    assert ex["image_raw"] == [img.tobytes()]
    assert ex["label"] == [4] and ex["height"] == [5] and ex["width"] == [7] and ex["channel"] == [3]
    assert ex["floats"] == [0.5, -1.25]
    assert ex["image/encoded"] == [b"\xff\xd8abc", b""]
    assert ex["neg"] == [-1, 1 << 40]


def test_example_parser_accepts_unpacked_scalars():
    # int64_list with two *unpacked* varints (field 1, wire type 0)
    feat = b"\x1a\x04" + b"\x08\x05\x08\x07"
    entry = b"\x0a\x01k" + b"\x12" + bytes([len(feat)]) + feat
    features = b"\x0a" + bytes([len(entry)]) + entry
    rec = b"\x0a" + bytes([len(features)]) + features
    assert tfrecord.parse_example(rec) == {"k": [5, 7]}
    with pytest.raises(ValueError):
        tfrecord.parse_example(b"\x0a\x7f\x01")


# ------------------------------------------------------------------ table / bundle
def test_table_format_independent_reader(tmp_path):
    import ctypes as C

    L = hostlib.lib()
    keys = [b"", b"a", b"a/b/c", b"a/b/d", b"b"] + [("key%05d" % i).encode() for i in range(300)]
    keys = sorted(set(keys))
    vals = [os.urandom((i * 7) % 50) for i in range(len(keys))]
    n = len(keys)
    path = str(tmp_path / "t.index")
    rc = L.dtfa_table_write(path.encode(), n, (C.c_char_p * n)(*keys), (C.c_uint64 * n)(*[len(k) for k in keys]),
                            (C.c_char_p * n)(*vals), (C.c_uint64 * n)(*[len(v) for v in vals]), 1024)
    assert rc == 0
    entries, restarts = py_read_table(open(path, "rb").read())
    assert entries == list(zip(keys, vals))
    assert len(restarts) > 1  # small block size -> several data blocks
    for r, starts in restarts:  # restart interval 16
        assert list(r) == [starts[i] for i in range(0, len(starts), 16)]
    # C++ reader agrees
    h = L.dtfa_table_read(path.encode())
    assert L.dtfa_table_error(h) is None and L.dtfa_table_count(h) == n
    L.dtfa_table_free(h)


def test_bundle_golden_protos_and_roundtrip(tmp_path):
    prefix = str(tmp_path / "model.ckpt-7")
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    step = np.array(7, dtype=np.int64)
    idx = np.array([1, 2, 3], dtype=np.int32)
    bundle.write_bundle(prefix, {"w/kernel": a, "global_step": step, "idx": idx})
    assert sorted(os.listdir(tmp_path)) == ["model.ckpt-7.data-00000-of-00001", "model.ckpt-7.index"]
    entries, _ = py_read_table(open(prefix + ".index", "rb").read())
    assert [k for k, _ in entries] == [b"", b"global_step", b"idx", b"w/kernel"]
    assert entries[0][1] == b"\x08\x01\x1a\x02\x08\x01"  # num_shards=1, version{producer=1}
    data = open(prefix + ".data-00000-of-00001", "rb").read()
    assert data == step.tobytes() + idx.tobytes() + a.tobytes()  # key order, no padding
    crc = struct.pack("<I", py_mask(py_crc32c(a.tobytes())))
    # dtype=DT_FLOAT, shape{dim{size:2} dim{size:3}}, offset=20, size=24, crc32c
    assert entries[3][1] == b"\x08\x01\x12\x08\x12\x02\x08\x02\x12\x02\x08\x03\x20\x14\x28\x18\x35" + crc
    # scalar int64: dtype=9, empty shape, (offset 0 omitted), size=8
    assert entries[1][1][:6] == b"\x08\x09\x12\x00\x28\x08"
    back = bundle.read_bundle(prefix)
    assert set(back) == {"w/kernel", "global_step", "idx"}
    np.testing.assert_array_equal(back["w/kernel"], a)
    assert back["global_step"].shape == () and back["global_step"] == 7
    listed = bundle.list_bundle(prefix)
    assert [e[0] for e in listed] == ["global_step", "idx", "w/kernel"]
    assert listed[2][1:3] == ("float32", (2, 3))


def test_bundle_detects_data_corruption(tmp_path):
    prefix = str(tmp_path / "c")
    bundle.write_bundle(prefix, {"x": np.ones(100, dtype=np.float32)})
    p = prefix + ".data-00000-of-00001"
    blob = bytearray(open(p, "rb").read())
    blob[10] ^= 0xFF
    open(p, "wb").write(bytes(blob))
    with pytest.raises(IOError):
        bundle.read_bundle(prefix)


def test_bundle_many_random_tensors(tmp_path):
    rng = np.random.default_rng(3)
    tensors = {}
    for i in range(200):
        shape = tuple(rng.integers(1, 6, rng.integers(0, 4)))
        name = "InceptionV3/scope_%d/%s" % (rng.integers(0, 50), ["weights", "BatchNorm/beta", "x%d" % i][i % 3])
        tensors[name + str(i)] = rng.standard_normal(shape).astype(np.float32)
    prefix = str(tmp_path / "m")
    bundle.write_bundle(prefix, tensors)
    back = bundle.read_bundle(prefix)
    assert set(back) == set(tensors)
    for k in tensors:
        np.testing.assert_array_equal(back[k], tensors[k])


def test_checkpoint_state_file(tmp_path):
    d = str(tmp_path)
    assert bundle.latest_checkpoint(d) is None
    for step in range(1, 8):
        base = "model.ckpt-%d" % step
        bundle.write_bundle(os.path.join(d, base), {"s": np.array(step, dtype=np.int64)})
        bundle.update_checkpoint_state(d, base, max_to_keep=5)
    latest, allp = bundle.read_checkpoint_state(d)
    assert latest == "model.ckpt-7" and allp == ["model.ckpt-%d" % s for s in range(3, 8)]
    assert bundle.latest_checkpoint(d) == os.path.join(d, "model.ckpt-7")
    assert not os.path.exists(os.path.join(d, "model.ckpt-1.index"))
    text = open(os.path.join(d, "checkpoint")).read().splitlines()
    assert text[0] == 'model_checkpoint_path: "model.ckpt-7"'
    assert text[1] == 'all_model_checkpoint_paths: "model.ckpt-3"'


# ------------------------------------------------------------------ events
def test_event_file(tmp_path):
    w = events.EventWriter(str(tmp_path))
    vals = np.array([-2.0, -1e-13, 0.0, 1e-13, 0.5, 0.5, 3.0, 1e25], dtype=np.float32)
    w.add_summary(3, {"loss": 1.5, "learning_rate": 1e-3}, {"logits histogram": vals}, wall_time=123.0)
    w.add_summary(4, {"loss": 1.25})
    w.close()
    name = os.path.basename(w.path)
    assert name.startswith("events.out.tfevents.")
    evs = list(events.read_events(w.path))
    assert evs[0]["file_version"] == "brain.Event:2" and evs[0]["step"] == 0
    assert evs[1]["step"] == 3 and evs[1]["wall_time"] == 123.0
    assert evs[1]["scalars"]["loss"] == 1.5 and abs(evs[1]["scalars"]["learning_rate"] - 1e-3) < 1e-9
    h = evs[1]["histograms"]["logits_histogram"]  # tag sanitised like TF1
    assert h["num"] == 8 and h["min"] == -2.0 and h["max"] == float(np.float32(1e25))
    assert abs(h["sum"] - float(vals.astype(np.float64).sum())) < 1e18
    assert sum(h["bucket"]) == 8 and len(h["bucket"]) == len(h["bucket_limit"])
    lim = h["bucket_limit"]
    assert lim == sorted(lim) and lim[-1] == np.finfo(np.float64).max
    # every value lies in the bucket whose limit is the first one > value
    for v in vals.astype(np.float64):
        i = next(k for k, l in enumerate(lim) if l > v or l == np.finfo(np.float64).max)
        assert h["bucket"][i] >= 1
    assert evs[2]["step"] == 4 and evs[2]["scalars"] == {"loss": 1.25}
