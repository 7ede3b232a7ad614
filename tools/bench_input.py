"""Throughput of the two input paths on flower-photo-shaped data (synthetic images of the dataset's typical
500x375 size), measured without a model behind them:

  jpeg   JPEG bytes -> host Huffman decode (threads) -> device IDCT + colour -> device crop/flip/resize -> bf16 NHWC8
  raw    TFRecord of raw pixels (the reference's prepare.py format) -> C++ loader -> device resize -> bf16 NHWC8

    python tools/bench_input.py [batch] [threads]
"""
import io
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.data import jpeg as J  # noqa: E402
from distributed_tensorflow_amd.data.preprocess import NORM_INCEPTION, make_descs, resize_on_device  # noqa: E402
from distributed_tensorflow_amd.data.records import RecordInput  # noqa: E402
from distributed_tensorflow_amd.io import tfrecord  # noqa: E402


def make_jpegs(n, seed=0):
    from PIL import Image

    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        h, w = (375, 500) if i % 3 else (333, 500)
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([(np.sin(xx / (7 + i % 5)) * 60 + 120), (np.cos(yy / (9 + i % 7)) * 60 + 120),
                        ((xx + yy) % 255)], -1) + rng.normal(0, 12, (h, w, 3))
        buf = io.BytesIO()
        Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(buf, format="JPEG", quality=90, subsampling=0)
        out.append(buf.getvalue())
    return out


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    dev = torch.device("cuda", 0)
    uniq = make_jpegs(64)
    jpegs = [uniq[i % len(uniq)] for i in range(batch)]
    print("mean JPEG size %.1f KB" % (sum(map(len, uniq)) / len(uniq) / 1024))

    def jpeg_batch():
        packed, shapes, offs = J.decode_batch_to_device(jpegs, dev, threads=threads)
        descs = torch.from_numpy(make_descs(shapes, offs)).to(dev)
        return resize_on_device(packed, descs, batch, (299, 299), "bilinear_legacy", False, NORM_INCEPTION)

    for _ in range(2):
        jpeg_batch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_it = 6
    for _ in range(n_it):
        jpeg_batch()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("jpeg path: %.0f images/s (batch %d, %d host threads)" % (batch * n_it / dt, batch, threads))

    from PIL import Image

    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "raw.tfrecord")
        with tfrecord.TFRecordWriter(path) as w:
            for i in range(512):
                img = np.asarray(Image.open(io.BytesIO(uniq[i % len(uniq)])))[:, :, ::-1].copy()  # BGR like cv2
                w.write(tfrecord.make_example({"image_raw": img.tobytes(), "label": [i % 5], "height": [img.shape[0]],
                                               "width": [img.shape[1]], "channel": [3]}))
        inp = RecordInput(path, batch, dev, image_size=299, num_epochs=8, threads=threads, prefetch=4)
        next(inp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for x, y in inp:
            n += x.shape[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        inp.close()
        print("raw-record path: %.0f images/s (batch %d, %d loader threads)" % (n / dt, batch, threads))

        # the same reader feeding a training step (Inception-v3 head, the reference's model): with the upload +
        # resize of the next batch on the copy stream, and with everything on the compute stream
        from distributed_tensorflow_amd.models import build_model
        from distributed_tensorflow_amd.parallel.engine import DataParallelEngine
        from distributed_tensorflow_amd.train.session import TrainingSession

        model = build_model("inception_v3_head", num_classes=5, device=dev)
        eng = DataParallelEngine(model, mode="sync_dp", optimizer="adam", lr=1e-3)
        sess = TrainingSession(model, eng, num_classes=5, hip_graph=True)
        for overlap in (True, False):
            inp = RecordInput(path, batch, dev, image_size=299, num_epochs=14, threads=threads, prefetch=4,
                              drop_remainder=True)
            if not overlap:
                inp._stream = None
            for _ in range(4):
                sess.run_step(*next(inp))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 0
            for x, y in inp:
                sess.run_step(x, y)
                n += x.shape[0]
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            inp.close()
            print("training from raw records, copy stream %s: %.0f images/s (%.2f ms/step)"
                  % ("on " if overlap else "off", n / dt, dt / (n / batch) * 1e3))


if __name__ == "__main__":
    main()
