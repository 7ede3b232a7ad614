"""The single-process trainer end to end on a GPU (``recognition_flower/single/flower_train.py``: JPEG files ->
host entropy decode -> device IDCT / crop / flip / resize -> Inception-v3 trunk + head, batch 32), with the next
batch prepared during the step (``--prefetch``) and in line with it, decoded images cached on the device
(``--image_cache_gb``) and decoded afresh every time.  The dataset is synthetic: 120 JPEGs of the
flower set's common sizes (500x375 / 500x333, quality 90) in five class directories.

    python tools/bench_single.py [--epochs 600]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=600)
    args = ap.parse_args()
    from bench_input import make_jpegs

    with tempfile.TemporaryDirectory() as d:
        data = os.path.join(d, "flowers")
        for i, blob in enumerate(make_jpegs(120)):
            cls = os.path.join(data, ["daisy", "dandelion", "roses", "sunflower", "tulips"][i % 5])
            os.makedirs(cls, exist_ok=True)
            with open(os.path.join(cls, "%03d.jpg" % i), "wb") as f:
                f.write(blob)
        for prefetch, cache_gb in (("true", 8), ("true", 0), ("false", 8), ("false", 0)):
            out = os.path.join(d, "out_%s_%d" % (prefetch, cache_gb))
            r = subprocess.run([sys.executable, "flower_train.py", "--dataset_path=" + data, "--n_epochs=%d" % args.epochs,
                                "--device=cuda", "--prefetch=" + prefetch, "--export=false",
                                "--image_cache_gb=%d" % cache_gb,
                                "--output_dir=%s/models" % out, "--summaries_dir=%s/events/" % out],
                               cwd=os.path.join(ROOT, "recognition_flower", "single"), stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, text=True)
            m = re.search(r"trained (\d+) steps in ([0-9.]+)s \(([0-9.]+) images/sec\)", r.stdout)
            if r.returncode != 0 or not m:
                print(r.stdout[-3000:])
                raise SystemExit("flower_train.py failed")
            steps, wall = int(m.group(1)), float(m.group(2))
            print("prefetch=%-5s image_cache_gb=%d %d steps in %.2f s wall = %.0f images/sec end to end (start-up and graph "
                  "capture included); steps alone %s images/sec" % (prefetch, cache_gb, steps, wall, steps * 32 / wall,
                                                                    m.group(3)), flush=True)


if __name__ == "__main__":
    main()
