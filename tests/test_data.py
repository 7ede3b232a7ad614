"""Input pipeline on the CPU: dataset listing, augmentation, resize semantics, JPEG entropy
decoder, TFRecord loader, prepare.py record schema."""
import glob
import io
import os

import numpy as np
import pytest
import torch

from distributed_tensorflow_amd.data import flowers, jpeg, preprocess
from distributed_tensorflow_amd.data.records import RecordInput
from distributed_tensorflow_amd.io import tfrecord

# the bundled sample dataset (drawn by tools/make_flower_fixture.py; same class / file counts and size range as
# the photographs the reference ships in its tree)
REF_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "recognition_flower",
                        "flower_photos_labeled")
needs_ref_data = pytest.mark.skipif(not os.path.isdir(REF_DATA), reason="bundled sample dataset missing (run tools/make_flower_fixture.py)")


def _make_dataset(root, n_per_class=4, seed=0):
    from PIL import Image

    rng = np.random.default_rng(seed)
    for ci, cls in enumerate(["daisy", "roses", "tulips"]):
        os.makedirs(os.path.join(root, cls))
        for i in range(n_per_class):
            h, w = int(rng.integers(40, 90)), int(rng.integers(40, 90))
            yy, xx = np.mgrid[0:h, 0:w]
            img = np.stack([(yy * 3 + ci * 40) % 256, (xx * 2 + i * 30) % 256, (yy + xx) % 256], axis=2).astype(np.uint8)
            Image.fromarray(img).save(os.path.join(root, cls, "img%d.jpg" % i), quality=92, subsampling=0)
    open(os.path.join(root, ".DS_Store"), "w").write("junk")  # the reference's dataset carries these
    open(os.path.join(root, "roses", ".DS_Store"), "w").write("junk")
    return root


# ------------------------------------------------------------------ listing / augmentation
def test_listing_skips_hidden_files_and_sorts_classes(tmp_path):
    root = _make_dataset(str(tmp_path / "d"))
    classes = flowers.list_classes(root)
    assert classes == ["daisy", "roses", "tulips"]
    items = flowers.get_image_path_and_class(root, classes)
    assert len(items) == 12 and {c for _, c in items} == {0, 1, 2}
    assert all(p.endswith(".jpg") for p, _ in items)
    train, test = flowers.train_test_split(items, 0.2, seed=1)
    assert len(train) == 9 and len(test) == 3 and not set(train) & set(test)


def test_prepare_image_and_batch_contract(tmp_path):
    root = _make_dataset(str(tmp_path / "d"))
    items = flowers.get_image_path_and_class(root, flowers.list_classes(root))
    img = flowers.read_image(items[0][0])
    out = flowers.prepare_image(img, 48, 32, rng=np.random.default_rng(0))
    assert out.shape == (32, 48, 3) and out.dtype == np.float32 and 0.0 <= out.min() and out.max() <= 1.0
    X, y = flowers.prepare_batch(items, 5, image_size=32, np_rng=np.random.default_rng(0))
    assert X.shape == (5, 32, 32, 3) and X.dtype == np.float32 and -1.0 <= X.min() and X.max() <= 1.0
    assert y.shape == (5,) and y.dtype == np.int32


def test_random_crop_spec_follows_reference_rules():
    rng = np.random.default_rng(0)
    for _ in range(200):
        h, w = int(rng.integers(50, 400)), int(rng.integers(50, 400))
        c = preprocess.random_crop_spec(h, w, 299, 299, 0.2, rng)
        assert 0 <= c.y0 and c.y0 + c.h <= h and 0 <= c.x0 and c.x0 + c.w <= w
        assert abs(c.w / c.h - 1.0) < 0.05                    # target aspect ratio
        assert min(h, w) / 1.2 - 2 <= c.h <= min(h, w)        # zoom factor in [1, 1.2)
    flips = [preprocess.random_crop_spec(100, 100, 10, 10, 0.2, rng).flip for _ in range(400)]
    assert 120 < sum(flips) < 280


# ------------------------------------------------------------------ resize semantics
def test_resize_identity_and_flip():
    img = np.random.default_rng(0).integers(0, 256, (9, 7, 3), dtype=np.uint8)
    for m in ("bilinear_legacy", "bilinear", "bicubic_legacy"):
        out = preprocess.resize_reference(img, (9, 7), m, norm=preprocess.NORM_NONE)
        np.testing.assert_allclose(out, img.astype(np.float32), atol=1e-3)
    flipped = preprocess.resize_reference(img, (9, 7), "bilinear", preprocess.CropSpec(0, 0, 9, 7, True),
                                          norm=preprocess.NORM_NONE)
    np.testing.assert_allclose(flipped, img[:, ::-1].astype(np.float32), atol=1e-3)


def test_legacy_bilinear_coordinates_match_tf1_definition():
    # TF1 ResizeBilinear(align_corners=False): src = dst * (in / out), no half-pixel offset
    img = np.zeros((2, 4, 3), dtype=np.uint8)
    img[0, :, 0] = [0, 100, 200, 50]
    out = preprocess.resize_reference(img, (2, 8), "bilinear_legacy", norm=preprocess.NORM_NONE)[0, :, 0]
    np.testing.assert_allclose(out, [0, 50, 100, 150, 200, 125, 50, 50], atol=1e-4)
    # torch's legacy mode (align_corners=False is half-pixel there) differs; ours equals the explicit formula
    up = preprocess.resize_reference(img, (4, 4), "bilinear_legacy", norm=preprocess.NORM_NONE)
    np.testing.assert_allclose(up[1, :, 0], 0.5 * img[0, :, 0] + 0.5 * img[1, :, 0], atol=1e-4)


def test_half_pixel_bilinear_matches_torch_interpolate():
    img = np.random.default_rng(1).integers(0, 256, (13, 17, 3), dtype=np.uint8)
    want = torch.nn.functional.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None].float(), size=(29, 31),
                                           mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
    got = preprocess.resize_reference(img, (29, 31), "bilinear", norm=preprocess.NORM_NONE)
    np.testing.assert_allclose(got, want, atol=1e-3)


def test_bicubic_legacy_partition_of_unity_and_keys_kernel():
    const = np.full((6, 6, 3), 77, dtype=np.uint8)
    out = preprocess.resize_reference(const, (15, 11), "bicubic_legacy", norm=preprocess.NORM_NONE)
    np.testing.assert_allclose(out, 77.0, atol=1e-3)
    # midpoint between two samples of a step edge with Keys a = -0.75: weights (-.09375, .59375, .59375, -.09375)
    row = np.zeros((1, 4, 3), dtype=np.uint8)
    row[0, :, 0] = [0, 0, 100, 100]
    out = preprocess.resize_reference(row, (1, 8), "bicubic_legacy", norm=preprocess.NORM_NONE)[0, :, 0]
    assert abs(out[3] - (0.59375 * 100 - 0.09375 * 100)) < 1e-3  # dst 3 -> src 1.5
    img = np.random.default_rng(2).integers(0, 256, (5, 5, 3), dtype=np.uint8)
    norm = preprocess.resize_reference(img, (5, 5), "bicubic_legacy")
    np.testing.assert_allclose(norm, img / 127.5 - 1.0, atol=1e-5)  # [-1, 1] normalisation


def test_preprocess_batch_cpu_layout():
    imgs = [np.random.default_rng(i).integers(0, 256, (20 + i, 30, 3), dtype=np.uint8) for i in range(3)]
    x = preprocess.preprocess_batch(imgs, (16, 16), "cpu", "bilinear_legacy", bgr=True)
    assert x.shape == (3, 16, 16, 8) and x.dtype == torch.float32
    assert float(x[..., 3:].abs().max()) == 0.0
    ref = preprocess.resize_reference(imgs[1], (16, 16), "bilinear_legacy")
    np.testing.assert_allclose(x[1, :, :, :3].numpy(), ref[:, :, ::-1], atol=1e-6)  # BGR source -> RGB output


# ------------------------------------------------------------------ JPEG
def _jpeg_bytes(img, **kw):
    from PIL import Image

    b = io.BytesIO()
    Image.fromarray(img).save(b, format="JPEG", **kw)
    return b.getvalue()


@pytest.mark.parametrize("kw", [dict(quality=90, subsampling=0), dict(quality=75, subsampling=0),
                                dict(quality=95, subsampling=2), dict(quality=85, subsampling=1),
                                dict(quality=60, subsampling=2),
                                dict(quality=90, subsampling=0, restart_marker_blocks=3)])
def test_entropy_decoder_reproduces_libjpeg(kw):
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:67, 0:93]
    img = np.stack([(yy * 3) % 256, (xx * 2) % 256, (yy + xx) % 256], axis=2).astype(np.uint8)
    img = np.clip(img + rng.integers(-20, 20, img.shape), 0, 255).astype(np.uint8)
    data = _jpeg_bytes(img, **kw)
    res = jpeg.entropy_decode(data)
    assert res is not None
    info, coefs = res
    assert (info.width, info.height, info.ncomp) == (93, 67, 3)
    got = jpeg.reconstruct_reference(info, coefs).astype(np.int32)
    want = jpeg.decode_pil(data).astype(np.int32)
    # float IDCT + float colour conversion vs libjpeg's integer IDCT and fixed-point tables; subsampled chroma
    # (4:2:2, 4:2:0) goes through the same triangle filters as libjpeg's "fancy" upsampling
    assert np.abs(got - want).max() <= 3 and np.abs(got - want).mean() < 0.6


def test_grayscale_and_unsupported_jpegs():
    g = (np.arange(40 * 50) % 256).astype(np.uint8).reshape(40, 50)
    data = _jpeg_bytes(g, quality=90)
    info, coefs = jpeg.entropy_decode(data)
    assert info.ncomp == 1
    assert np.abs(jpeg.reconstruct_reference(info, coefs).astype(int) - jpeg.decode_pil(data).astype(int)).max() <= 3
    rgb = np.random.default_rng(0).integers(0, 256, (32, 32, 3), dtype=np.uint8)
    assert jpeg.entropy_decode(_jpeg_bytes(rgb, progressive=True)) is None  # falls back to PIL in the pipeline
    assert jpeg.entropy_decode(b"not a jpeg") is None
    assert jpeg.entropy_decode(data[: len(data) // 2]) is None or True  # truncated data must not crash


@needs_ref_data
def test_entropy_decoder_on_all_reference_photos():
    files = sorted(glob.glob(os.path.join(REF_DATA, "*", "*.jpg")))
    assert len(files) == 107
    worst = 0
    for f in files[::6]:
        data = open(f, "rb").read()
        res = jpeg.entropy_decode(data)
        assert res is not None, f
        got = jpeg.reconstruct_reference(*res).astype(np.int32)
        worst = max(worst, int(np.abs(got - jpeg.decode_pil(data).astype(np.int32)).max()))
    assert worst <= 3


# ------------------------------------------------------------------ TFRecord pipeline
def _write_records(path, n=23, seed=0):
    rng = np.random.default_rng(seed)
    imgs = []
    with tfrecord.TFRecordWriter(path) as w:
        for i in range(n):
            h, wd = int(rng.integers(8, 20)), int(rng.integers(8, 20))
            img = rng.integers(0, 256, (h, wd, 3), dtype=np.uint8)
            img[0, 0] = [i, i, i]  # tag the example
            imgs.append(img)
            w.write(tfrecord.make_example({"image_raw": img.tobytes(), "label": i % 5, "height": h, "width": wd,
                                           "channel": 3}))
    return imgs


def test_record_input_shards_shuffles_and_resizes(tmp_path):
    path = str(tmp_path / "t.tfrecord")
    imgs = _write_records(path)
    seen = []
    for rank in range(2):
        inp = RecordInput(path, 4, "cpu", image_size=16, num_epochs=1, rank=rank, world=2, threads=3, prefetch=2,
                          shuffle=True, seed=5, legacy_bgr_unnormalised=True, max_image_bytes=4096)
        tags = []
        for x, y in inp:
            assert x.shape[1:] == (16, 16, 8) and x.dtype == torch.float32 and y.dtype == torch.int32
            tags += [int(round(float(v))) for v in x[:, 0, 0, 0]]  # legacy mode: raw pixel values, BGR order
            assert all(int(t) % 5 == int(l) for t, l in zip(tags[-len(y):], y))
        inp.close()
        seen.append(tags)
    assert len(seen[0]) == 12 and len(seen[1]) == 11
    assert sorted(seen[0] + seen[1]) == list(range(23))          # disjoint shards covering everything
    assert seen[0] != sorted(seen[0])                             # example-level shuffle
    # deterministic for a given seed; different order in the next epoch
    inp = RecordInput(path, 23, "cpu", image_size=16, num_epochs=2, shuffle=True, seed=5, legacy_bgr_unnormalised=True,
                      max_image_bytes=4096)
    e1, _ = next(inp)
    e2, _ = next(inp)
    with pytest.raises(StopIteration):
        next(inp)
    inp.close()
    t1 = [int(round(float(v))) for v in e1[:, 0, 0, 0]]
    t2 = [int(round(float(v))) for v in e2[:, 0, 0, 0]]
    assert sorted(t1) == sorted(t2) == list(range(23)) and t1 != t2
    # default mode: RGB in [-1, 1]
    inp = RecordInput(path, 2, "cpu", image_size=16, shuffle=False, max_image_bytes=4096)
    x, y = next(inp)
    inp.close()
    ref = preprocess.resize_reference(imgs[0], (16, 16), "bilinear_legacy", bgr=True)
    np.testing.assert_allclose(x[0, :, :, :3].numpy(), ref, atol=1e-5)
    assert list(y) == [0, 1]


def test_record_input_reports_oversized_images(tmp_path):
    path = str(tmp_path / "t.tfrecord")
    _write_records(path, n=4)
    inp = RecordInput(path, 4, "cpu", image_size=8, max_image_bytes=64)
    with pytest.raises(IOError):
        next(inp)
    inp.close()
    with pytest.raises(IOError):
        RecordInput(str(tmp_path / "missing.tfrecord"), 4, "cpu")
