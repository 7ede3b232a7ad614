#!/usr/bin/env python
"""Checkpoint -> servable converter.

Flags and behaviour of the reference's
``recognition_flower/distributed/save_model.py`` (flags: lines 24-34 there; main:
87-171): rebuild the model, find the newest checkpoint through the ``checkpoint``
state file in ``--checkpoint_dir``, restore it by variable name, run a smoke
forward pass on a zero image, and export ``<output_dir>/<model_version>`` with
the ``serving_default`` (classify, tf.Example in) and ``predict_images`` (JPEG
bytes in) signatures.  The class vocabulary comes from ``--dataset_path``'s
sub-directories (sorted) or, when that is absent, from the ``labels.json``
written by ``prepare.py``.

Unlike the reference, a missing checkpoint is an error (the reference silently
exports random weights) and the servable runs in inference mode (the
reference's exported graph has dropout and batch statistics switched on).
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for peer-mapped GPU memory (RCCL, xGMI mailboxes)

import torch  # noqa: E402

from distributed_tensorflow_amd.utils import flags  # noqa: E402

flags.DEFINE_string("output_dir", "./output/models", "Directory where to export the model.")
flags.DEFINE_integer("model_version", 1, "Version number of the model.")
flags.DEFINE_string("dataset_path", "../flower_photos_labeled", "Needs to provide the dataset path as in training.")
flags.DEFINE_string("checkpoint_dir", "./output/checkpoint_dir/", "Needs to provide the model path (checkpoint,meta) in training.")
# additions
flags.DEFINE_string("model", "auto", "inception_v3_head | flower_cnn | resnet50 | auto")
flags.DEFINE_integer("image_size", 299, "Network input size.")
flags.DEFINE_string("labels_json", "", "labels.json written by prepare.py (used when dataset_path does not exist).")
flags.DEFINE_string("device", "auto", "cuda | cpu | auto")
flags.DEFINE_boolean("allow_missing_checkpoint", False, "Export freshly initialised weights when no checkpoint exists.")

FLAGS = flags.FLAGS


def load_classes():
    from distributed_tensorflow_amd.data.flowers import list_classes

    if os.path.isdir(FLAGS.dataset_path):
        return list_classes(FLAGS.dataset_path)
    if FLAGS.labels_json and os.path.exists(FLAGS.labels_json):
        with open(FLAGS.labels_json) as f:
            return json.load(f)["classes"]
    raise SystemExit("neither --dataset_path nor --labels_json provides the class names")


def main(argv):
    from distributed_tensorflow_amd.io import bundle
    from distributed_tensorflow_amd.io.export import export_servable
    from distributed_tensorflow_amd.io.saver import Saver
    from distributed_tensorflow_amd.models import build_model

    device = FLAGS.device
    if device == "auto":
        device = "cuda" if torch.cuda.is_available() else "cpu"
    device = torch.device(device)
    flower_classes = load_classes()
    model_name = FLAGS.model if FLAGS.model != "auto" else ("inception_v3_head" if device.type == "cuda" else "flower_cnn")
    model = build_model(model_name, num_classes=len(flower_classes), device=device)
    saver = Saver(model)
    ckpt = bundle.latest_checkpoint(FLAGS.checkpoint_dir)
    step = 0
    if ckpt is not None:
        step = saver.restore(ckpt, restore_slots=False)
        print("restored", ckpt, "(global step %d)" % step)
    elif not FLAGS.allow_missing_checkpoint:
        raise SystemExit("no checkpoint found in %s" % FLAGS.checkpoint_dir)
    # smoke forward on a zero image (recognition_flower/distributed/save_model.py:114 in the reference)
    model.eval()
    with torch.no_grad():
        dtype = torch.bfloat16 if device.type == "cuda" else torch.float32
        out = model(torch.zeros(1, FLAGS.image_size, FLAGS.image_size, 8, device=device, dtype=dtype))
        logits = out[0] if isinstance(out, tuple) else out
        print("smoke logits:", logits[0, :len(flower_classes)].float().tolist())
    path = export_servable(model, flower_classes, FLAGS.output_dir, FLAGS.model_version, model_name=model_name,
                           image_size=FLAGS.image_size, global_step=step)
    print("Exported servable to", path)
    print("Done exporting!")
    return 0


if __name__ == "__main__":
    flags.run(main)
