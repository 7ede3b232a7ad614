"""Malformed JPEG input against the host entropy decoder (csrc/host/jpeg_huffman.cc): every case must be
rejected with an error code, never crash or write out of bounds.  The last test replays the same inputs
through a copy of the decoder built with AddressSanitizer (host code only)."""
import io
import os
import shutil
import struct
import subprocess
import sys

import numpy as np
import pytest

from distributed_tensorflow_amd.data import jpeg
from distributed_tensorflow_amd.serving.batcher import DynamicBatcher

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _good_jpeg(h=40, w=56, **kw):
    from PIL import Image

    rng = np.random.default_rng(0)
    img = rng.integers(0, 255, (h, w, 3), dtype=np.uint8)
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, format="JPEG", quality=90, subsampling=0, **kw)
    return buf.getvalue()


def _segments(data):
    """[(marker, start, end)] of the header segments up to and including SOS."""
    out, p = [], 2
    while p + 4 <= len(data):
        assert data[p] == 0xFF
        m = data[p + 1]
        ln = (data[p + 2] << 8) | data[p + 3]
        out.append((m, p, p + 2 + ln))
        p += 2 + ln
        if m == 0xDA:
            break
    return out


def _with_dht_counts(data, counts):
    """Replace the first Huffman table segment by one table whose code-length histogram is ``counts`` (16 entries)
    and that stores exactly ``sum(counts)`` values, so the segment is well-formed and only the histogram is not."""
    for m, s, e in _segments(data):
        if m == 0xC4:
            n = sum(counts)
            body = bytes([data[s + 4]]) + bytes(counts) + bytes(i % 12 for i in range(n))
            seg = b"\xff\xc4" + struct.pack(">H", 2 + len(body)) + body
            return data[:s] + seg + data[e:]
    raise AssertionError("no DHT segment")


def _with_size(data, h, w):
    for m, s, e in _segments(data):
        if m in (0xC0, 0xC1):
            b = bytearray(data)
            b[s + 5:s + 9] = struct.pack(">HH", h, w)
            return bytes(b)
    raise AssertionError("no SOF segment")


def hostile_inputs():
    good = _good_jpeg()
    cases = {
        # 3 codes of length 1 (only 2 exist): the over-subscribed table that indexed past the 9-bit lookup array
        "dht_oversubscribed_len1": _with_dht_counts(good, [3] + [0] * 15),
        # 255 codes of length 1
        "dht_255_len1": _with_dht_counts(good, [255] + [0] * 15),
        # Kraft sum > 1 spread over lengths
        "dht_oversubscribed_mixed": _with_dht_counts(good, [1, 2, 4, 8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]),
        # over-subscribed beyond the 9-bit lookup range (length 10 ..)
        "dht_oversubscribed_long": _with_dht_counts(good, [0, 1, 5, 1, 1, 1, 1, 1, 1, 250, 0, 0, 0, 0, 0, 0][:16]),
        "huge_header": _with_size(good, 65535, 65535),
        "truncated_scan": good[: len(good) * 2 // 3],
        "truncated_header": good[:30],
        "header_only": good[: _segments(good)[-1][2]],
        "garbage_scan": good[: _segments(good)[-1][2]] + bytes(range(256)) * 4,
        "empty": b"",
        "soi_only": b"\xff\xd8",
    }
    # a claimed size larger than the data can hold, still under the pixel cap: decode must stop, not crawl
    cases["inflated_size"] = _with_size(good, 4000, 4000)
    return good, cases


def test_hostile_jpegs_are_rejected_not_crashing():
    good, cases = hostile_inputs()
    assert jpeg.entropy_decode(good) is not None
    for name, data in cases.items():
        assert jpeg.entropy_decode(data) is None, name
        assert jpeg.parse_header(data) is None or name in ("truncated_scan", "header_only", "garbage_scan",
                                                           "inflated_size"), name
    assert jpeg.parse_header(cases["huge_header"]) is None


def test_batch_decode_names_the_bad_image():
    good, cases = hostile_inputs()
    with pytest.raises(jpeg.JpegDecodeError) as ei:
        _decode_or_raise([good, cases["dht_oversubscribed_len1"], good])
    assert ei.value.index == 1


def _decode_or_raise(jpegs):
    """The host half of ``decode_batch_to_device`` (header pass + PIL fallback), which runs without a GPU."""
    for i, d in enumerate(jpegs):
        if jpeg.entropy_decode(d) is None:
            try:
                jpeg.decode_pil(d)
            except Exception as e:  # noqa: BLE001
                raise jpeg.JpegDecodeError(i, str(e)) from None


def test_batcher_fails_only_the_offending_request():
    def fn(items):
        if any(x < 0 for x in items):
            raise ValueError("bad item")
        return {"y": np.asarray(items, dtype=np.int64) * 2}

    b = DynamicBatcher(fn, max_batch=16, timeout_ms=200.0)
    try:
        futs = [b.submit([1, 2]), b.submit([-1]), b.submit([3])]
        assert futs[0].result(10)["y"].tolist() == [2, 4]
        with pytest.raises(ValueError):
            futs[1].result(10)
        assert futs[2].result(10)["y"].tolist() == [6]
        assert b.split_batches >= 1
    finally:
        b.close()


ASAN_DRIVER = r"""
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
extern "C" {
struct DtfaJpegInfo;
int dtfa_jpeg_parse_header(const uint8_t*, uint64_t, DtfaJpegInfo*);
int dtfa_jpeg_decode_coefs(const uint8_t*, uint64_t, DtfaJpegInfo*, int16_t*, long long);
int dtfa_jpeg_info_size();
}
int main(int argc, char** argv) {
  for (int i = 1; i < argc; ++i) {
    FILE* f = fopen(argv[i], "rb");
    if (!f) return 2;
    std::vector<uint8_t> d;
    uint8_t buf[4096];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
    fclose(f);
    std::vector<uint8_t> info(dtfa_jpeg_info_size() + 64);
    int rc = dtfa_jpeg_parse_header(d.data(), d.size(), (DtfaJpegInfo*)info.data());
    long long total = 0;
    if (rc == 0) total = *(long long*)(info.data() + 4 * (3 + 4 + 4 + 4 + 4) + 4 + 8 * 4);  // total_coefs
    std::vector<int16_t> coefs(rc == 0 ? (size_t)total : 1);
    int rc2 = dtfa_jpeg_decode_coefs(d.data(), d.size(), (DtfaJpegInfo*)info.data(), coefs.data(), (long long)coefs.size());
    printf("%s %d %d\n", argv[i], rc, rc2);
  }
  return 0;
}
"""


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_hostile_jpegs_under_address_sanitizer(tmp_path):
    good, cases = hostile_inputs()
    cases["good"] = good
    paths = []
    for name, data in cases.items():
        p = tmp_path / (name + ".jpg")
        p.write_bytes(data)
        paths.append(str(p))
    drv = tmp_path / "drv.cc"
    drv.write_text(ASAN_DRIVER)
    exe = tmp_path / "drv"
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-static-libasan", "-fno-omit-frame-pointer",
           str(drv), os.path.join(ROOT, "csrc", "host", "jpeg_huffman.cc"), "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "asan" in (r.stderr + r.stdout).lower():
        pytest.skip("AddressSanitizer runtime not available: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr
    env = dict(os.environ)
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0"
    r = subprocess.run([str(exe)] + paths, capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr
    lines = dict((ln.split()[0], ln.split()[1:]) for ln in r.stdout.strip().splitlines())
    assert lines[str(tmp_path / "good.jpg")] == ["0", "0"]
    for name in cases:
        if name != "good":
            assert lines[str(tmp_path / (name + ".jpg"))][1] != "0", name
