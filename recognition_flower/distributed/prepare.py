#!/usr/bin/env python
"""JPEG directory -> TFRecord of raw decoded pixels.

Same flags and record schema as the reference's
``recognition_flower/distributed/prepare.py`` (flags: lines 10-14; writer loop:
26-53): one ``tf.train.Example`` per image with ``image_raw`` (H*W*C uint8
bytes in OpenCV's BGR channel order), ``label``, ``height``, ``width``,
``channel``; files are sorted, labelled by parent directory and shuffled once.

Differences: the class-id map uses *sorted* directory names (the reference's
depends on ``os.listdir`` order) and is written next to the records as
``<tfrecords_path>.labels.json`` so later stages do not have to re-list the
dataset; ``--seed`` makes the shuffle reproducible.
"""
import glob
import json
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

import numpy as np  # noqa: E402

from distributed_tensorflow_amd.utils import flags  # noqa: E402

flags.DEFINE_string("dataset_path", "../flower_photos_labeled", "Needs to provide the dataset path for training.")
flags.DEFINE_string("tfrecords_path", "./train_v1.tfrecord",
                    "Needs to provide the tf records file for training.(train_v2~1.15G ,train_v1~38M)")
flags.DEFINE_integer("seed", -1, "Shuffle seed (-1: non-deterministic, like the reference).")

FLAGS = flags.FLAGS


def shuffle_data_and_label(files, char_dict, rng=random):
    files = sorted(files)
    labels = [char_dict[os.path.basename(os.path.dirname(f))] for f in files]
    pairs = list(zip(files, labels))
    rng.shuffle(pairs)
    return pairs


def encode_to_tfrecord(file_path, tfrecord_path, col=None, row=None, seed=None):
    from distributed_tensorflow_amd.data.flowers import list_classes, read_image
    from distributed_tensorflow_amd.io.tfrecord import TFRecordWriter, make_example

    classes = list_classes(file_path)
    char_dict = dict(zip(classes, range(len(classes))))
    print(char_dict)
    files = glob.glob(os.path.join(file_path, "*", "*.jpg"))
    rng = random.Random(seed) if seed is not None and seed >= 0 else random
    pairs = shuffle_data_and_label(files, char_dict, rng)
    os.makedirs(os.path.dirname(os.path.abspath(tfrecord_path)), exist_ok=True)
    n = 0
    with TFRecordWriter(tfrecord_path) as writer:
        for path, label in pairs:
            rgb = read_image(path)
            if col is not None and row is not None:
                from PIL import Image

                rgb = np.asarray(Image.fromarray(rgb).resize((col, row), Image.BILINEAR))
            bgr = np.ascontiguousarray(rgb[:, :, ::-1])  # cv2.imread order, as the reference stores it
            h, w, c = bgr.shape  # recorded *after* any resize (the reference records the pre-resize size)
            writer.write(make_example({"image_raw": bgr.tobytes(), "label": int(label), "height": int(h),
                                       "width": int(w), "channel": int(c)}))
            n += 1
    with open(tfrecord_path + ".labels.json", "w") as f:
        json.dump({"classes": classes}, f)
    return n, classes


def main(argv):
    n, classes = encode_to_tfrecord(FLAGS.dataset_path, FLAGS.tfrecords_path, seed=FLAGS.seed)
    print("wrote %d examples of %d classes to %s" % (n, len(classes), FLAGS.tfrecords_path))
    return 0


if __name__ == "__main__":
    print("prepare tf records start")
    flags.run(main)
