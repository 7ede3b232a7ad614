"""GPU input pipeline kernels: resize/normalise, JPEG IDCT + colour conversion, record loader upload."""
import glob
import io
import os

import numpy as np
import pytest
import torch

from distributed_tensorflow_amd.data import jpeg, preprocess
from distributed_tensorflow_amd.data.records import RecordInput
from distributed_tensorflow_amd.io import tfrecord

pytestmark = pytest.mark.gpu
# the bundled sample dataset (drawn by tools/make_flower_fixture.py; same class / file counts and size range as
# the photographs the reference ships in its tree)
REF_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "recognition_flower",
                        "flower_photos_labeled")


@pytest.mark.parametrize("method", ["bilinear_legacy", "bilinear", "bicubic_legacy"])
@pytest.mark.parametrize("bgr", [False, True])
def test_resize_kernel_matches_numpy_reference(method, bgr):
    rng = np.random.default_rng(0)
    imgs = [rng.integers(0, 256, (int(rng.integers(20, 90)), int(rng.integers(20, 90)), 3), dtype=np.uint8)
            for _ in range(5)]
    imgs.append(rng.integers(0, 256, (33, 47), dtype=np.uint8))  # grayscale
    crops = [None, preprocess.CropSpec(3, 5, 12, 10, True), preprocess.random_crop_spec(*imgs[2].shape[:2], 31, 31, 0.2, rng),
             None, preprocess.CropSpec(0, 0, imgs[4].shape[0], imgs[4].shape[1], True), None]
    out = preprocess.preprocess_batch(imgs, (31, 37), "cuda", method, crops, bgr)
    assert out.shape == (6, 31, 37, 8) and out.dtype == torch.bfloat16
    assert float(out[..., 3:].float().abs().max()) == 0.0
    got = out[..., :3].float().cpu().numpy()
    for i, im in enumerate(imgs):
        want = preprocess.resize_reference(im, (31, 37), method, crops[i], bgr)
        np.testing.assert_allclose(got[i], want, atol=0.02, err_msg="image %d" % i)  # bf16 output in [-1, 1]


def _jpeg_bytes(img, **kw):
    from PIL import Image

    b = io.BytesIO()
    Image.fromarray(img).save(b, format="JPEG", **kw)
    return b.getvalue()


def test_jpeg_device_decode_matches_numpy_and_libjpeg():
    rng = np.random.default_rng(1)
    jpegs = []
    for i, kw in enumerate([dict(quality=90, subsampling=0), dict(quality=80, subsampling=2), dict(quality=95, subsampling=0),
                            dict(quality=85, subsampling=0, progressive=True)]):
        h, w = int(rng.integers(30, 120)), int(rng.integers(30, 120))
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([(yy * 3 + i * 20) % 256, (xx * 2) % 256, (yy + xx) % 256], axis=2).astype(np.uint8)
        jpegs.append(_jpeg_bytes(img, **kw))
    jpegs.append(_jpeg_bytes((np.arange(40 * 56) % 256).astype(np.uint8).reshape(40, 56), quality=90))
    packed, shapes, offs = jpeg.decode_batch_to_device(jpegs, "cuda")
    host = packed.cpu().numpy()
    for i, data in enumerate(jpegs):
        h, w, c = shapes[i]
        got = host[offs[i]: offs[i] + h * w * 3].reshape(h, w, 3).astype(np.int32)
        res = jpeg.entropy_decode(data)
        if res is not None:
            want = jpeg.reconstruct_reference(*res).astype(np.int32)
            assert np.abs(got - want).max() <= 1, i  # same algorithm, fp32 rounding only
        else:
            want = jpeg.decode_pil(data).astype(np.int32)  # progressive file: PIL fallback uploaded as pixels
            assert np.array_equal(got, want)
    for k in (0, 1):   # 4:4:4 and 4:2:0 (triangle-filtered chroma) against libjpeg itself
        pil = jpeg.decode_pil(jpegs[k]).astype(np.int32)
        got_k = host[offs[k]: offs[k] + pil.size].reshape(pil.shape).astype(np.int32)
        assert np.abs(got_k - pil).max() <= 3, k


@pytest.mark.skipif(not os.path.isdir(REF_DATA), reason="bundled sample dataset missing (run tools/make_flower_fixture.py)")
def test_reference_photos_decode_augment_resize_on_device():
    files = sorted(glob.glob(os.path.join(REF_DATA, "*", "*.jpg")))[:16]
    jpegs = [open(f, "rb").read() for f in files]
    packed, shapes, offs = jpeg.decode_batch_to_device(jpegs, "cuda")
    descs = torch.from_numpy(preprocess.make_descs(shapes, offs)).cuda()
    x = preprocess.resize_on_device(packed, descs, len(files), (299, 299), "bicubic_legacy")
    assert x.shape == (16, 299, 299, 8)
    for i in (0, 7, 15):
        want = preprocess.resize_reference(jpeg.decode_pil(jpegs[i]), (299, 299), "bicubic_legacy")
        diff = np.abs(x[i, :, :, :3].float().cpu().numpy() - want)
        assert diff.mean() < 0.01 and diff.max() < 0.12  # +-3/255*2 from the decoder, amplified by bicubic overshoot


def test_record_input_on_device(tmp_path):
    path = str(tmp_path / "t.tfrecord")
    rng = np.random.default_rng(0)
    imgs = []
    with tfrecord.TFRecordWriter(path) as w:
        for i in range(10):
            h, wd = int(rng.integers(30, 60)), int(rng.integers(30, 60))
            img = rng.integers(0, 256, (h, wd, 3), dtype=np.uint8)
            imgs.append(img)
            w.write(tfrecord.make_example({"image_raw": img.tobytes(), "label": i % 5, "height": h, "width": wd,
                                           "channel": 3}))
    inp = RecordInput(path, 4, "cuda", image_size=32, shuffle=False, max_image_bytes=1 << 16)
    batches = list(inp)
    inp.close()
    assert [b[0].shape[0] for b in batches] == [4, 4, 2]
    x, y = batches[0]
    assert x.is_cuda and x.dtype == torch.bfloat16 and y.is_cuda and y.tolist() == [0, 1, 2, 3]
    want = preprocess.resize_reference(imgs[1], (32, 32), "bilinear_legacy", bgr=True)
    np.testing.assert_allclose(x[1, :, :, :3].float().cpu().numpy(), want, atol=0.02)


def test_record_input_prefetch_stream_matches_cpu_reader(tmp_path):
    """Batches prepared one step ahead on the copy stream (several epochs, ragged last batch, busy compute stream)
    equal the CPU reader's; the epoch end surfaces as StopIteration exactly after the last batch."""
    path = str(tmp_path / "t.tfrecord")
    rng = np.random.default_rng(1)
    with tfrecord.TFRecordWriter(path) as w:
        for i in range(37):
            h, wd = int(rng.integers(20, 70)), int(rng.integers(20, 70))
            img = rng.integers(0, 256, (h, wd, 3), dtype=np.uint8)
            w.write(tfrecord.make_example({"image_raw": img.tobytes(), "label": i % 5, "height": h, "width": wd,
                                           "channel": 3}))
    kw = dict(image_size=24, num_epochs=3, shuffle=True, seed=7, max_image_bytes=1 << 16, prefetch=3)
    cpu = list(RecordInput(path, 8, "cpu", **kw))
    inp = RecordInput(path, 8, "cuda", **kw)
    assert inp._stream is not None
    busy = torch.randn(2048, 2048, device="cuda")
    got = []
    for x, y in inp:
        busy = busy @ busy * 1e-3          # compute-stream work the copy stream overlaps with
        got.append((x.float().cpu(), y.cpu()))
    inp.close()
    assert len(got) == len(cpu) == 14 and got[-1][0].shape[0] == 37 * 3 - 13 * 8
    for (gx, gy), (cx, cy) in zip(got, cpu):
        assert gy.tolist() == cy.tolist()
        np.testing.assert_allclose(gx[..., :3].numpy(), cx[..., :3].float().numpy(), atol=0.02)


def test_device_prefetcher_hands_over_batches_in_order():
    """Batches produced one ahead on the helper thread + copy stream arrive in production order, complete (the
    consumer's stream waits for the producer's event), while the compute stream is kept busy."""
    from distributed_tensorflow_amd.data.prefetch import DevicePrefetcher

    state = {"k": 0}

    def produce():
        k = state["k"]
        state["k"] += 1
        host = torch.full((1 << 20,), float(k)).pin_memory()
        x = host.to("cuda", non_blocking=True)
        return x * 2 + 1, torch.tensor([k], device="cuda")

    pf = DevicePrefetcher(produce, "cuda")
    busy = torch.randn(2048, 2048, device="cuda")
    for k in range(12):
        busy = busy @ busy * 1e-3
        x, y = pf.next()
        assert int(y.item()) == k
        assert float(x.min().item()) == float(x.max().item()) == 2.0 * k + 1
    pf.close()
    assert state["k"] >= 12
    with pytest.raises(ValueError):
        DevicePrefetcher(produce, "cpu")


def test_device_prefetcher_switches_between_ahead_and_in_line_without_losing_order():
    """``set_ahead(False)`` drains the queued batches and then runs the producer in line (no batch produced that is
    not handed out); ``set_ahead(True)`` resumes; the sequence of batches is the producer's throughout."""
    from distributed_tensorflow_amd.data.prefetch import DevicePrefetcher

    state = {"k": 0}

    def produce():
        k = state["k"]
        state["k"] += 1
        return (torch.full((1024,), float(k), device="cuda"),)

    pf = DevicePrefetcher(produce, "cuda", depth=3)
    seen = []
    for step in range(20):
        if step == 5:
            pf.set_ahead(False)
        if step == 12:
            assert state["k"] == 12   # in line: exactly as many produced as consumed
            pf.set_ahead(True)
        (x,) = pf.next()
        seen.append(int(x[0].item()))
    pf.close()
    assert seen == list(range(20))


def test_decoded_image_cache_serves_pixels_from_the_device(tmp_path):
    """Files are read and entropy-decoded once; later batches point the resize kernel at the cached pixels
    (offsets relative to another buffer, modulo 2^64) and produce identical network inputs."""
    import torch

    from distributed_tensorflow_amd.data.preprocess import NORM_INCEPTION, full_crop, make_descs, resize_on_device

    rng = np.random.default_rng(3)
    paths = []
    for i in range(6):
        h, w = int(rng.integers(40, 90)), int(rng.integers(40, 90))
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        p = tmp_path / ("im%d.jpg" % i)
        p.write_bytes(_jpeg_bytes(img, quality=90, subsampling=2 if i % 2 else 0))
        paths.append(str(p))
    cache = jpeg.DecodedImageCache("cuda", budget_bytes=1 << 20, arena_bytes=1 << 16)

    def batch(sel):
        base, shapes, offs = jpeg.decode_files_to_device([paths[i] for i in sel], "cuda", cache)
        descs = torch.from_numpy(make_descs(shapes, offs, [full_crop(h, w) for h, w, _ in shapes])).cuda()
        return resize_on_device(base, descs, len(sel), (32, 32), "bilinear", False, NORM_INCEPTION), base

    x1, _ = batch([0, 1, 2, 3])
    assert (cache.hits, cache.misses, len(cache)) == (0, 4, 4)
    x2, base2 = batch([3, 2, 4, 0])       # three hits in two arenas and one miss in the same batch
    assert (cache.hits, cache.misses, len(cache)) == (3, 5, 5)
    assert torch.equal(x2[0], x1[3]) and torch.equal(x2[1], x1[2]) and torch.equal(x2[3], x1[0])
    x3, base3 = batch([4, 0])             # all hits: nothing decoded, the base buffer is a dummy
    assert base3.numel() == 16 and torch.equal(x3[0], x2[2]) and torch.equal(x3[1], x1[0])
    # without a cache the same call decodes every time
    base, shapes, offs = jpeg.decode_files_to_device(paths[:2], "cuda", None)
    assert all(0 <= o < base.numel() for o in offs)
    # a full cache stops storing but keeps serving
    small = jpeg.DecodedImageCache("cuda", budget_bytes=20000, arena_bytes=1 << 16)
    jpeg.decode_files_to_device(paths, "cuda", small)
    assert 0 < len(small) < len(paths) and small.bytes_used <= 20000
