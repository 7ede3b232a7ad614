#!/usr/bin/env python
"""Single-process flower classifier training + servable export.

Entry point with the flags of the reference's
``recognition_flower/single/flower_train.py`` (flags: lines 10-31 there; loop:
170-193; export: 195-251):

    python flower_train.py [--output_dir=../output/models] [--model_version=1] [--image_size=299]
                           [--n_epochs=10] [--dataset_path=../flower_photos_labeled]
                           [--model_path=] [--summaries_dir=../output/events/]

Behaviour kept: class-per-directory dataset listing with an 80/20 split, batches
of 32 augmented crops in [-1, 1], Inception-v3 trunk (frozen) + 2048-1024-5 head
trained with Adam under ``exponential_decay(0.001, 200, 0.95)``, loss = mean
softmax cross-entropy + L2 activity terms, per-step console log, TensorBoard
scalars (learning_rate / loss / accuracy) and ``logits histogram``, optional
restore from ``--model_path``, export of a servable with the ``serving_default``
and ``predict_images`` signatures under ``<output_dir>/<model_version>``.

Deliberate differences: accuracy is taken from the training forward pass (the
reference runs a second full forward per step for it), summaries are indexed by
global step (the reference uses the epoch index), the held-out 20 % is actually
evaluated at the end (every image once, un-augmented; ``accuracy`` / ``loss``
written under ``<summaries_dir><model_version>/eval``), and on a GPU the JPEG decode / augmentation / resize run
on the device.  ``--model=flower_cnn`` trains the small end-to-end CNN instead
(the default on a machine without a GPU).
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for peer-mapped GPU memory (RCCL, xGMI mailboxes)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from distributed_tensorflow_amd.utils import flags  # noqa: E402

flags.DEFINE_string("output_dir", "../output/models", "Directory where to export the model.")
flags.DEFINE_integer("model_version", 1, "Version number of the model.")
flags.DEFINE_integer("image_size", 299, "Needs to provide same value as in training.")
flags.DEFINE_integer("n_epochs", 10, "Needs to provide the number of epochs as in training.")
flags.DEFINE_string("dataset_path", "../flower_photos_labeled", "Needs to provide the dataset path as in training.")
flags.DEFINE_string("model_path", "", "Checkpoint prefix to restore before training (optional).")
flags.DEFINE_string("summaries_dir", "../output/events/", "Needs to provide the summary output dir in training.")
# additions
flags.DEFINE_string("model", "auto", "inception_v3_head | inception_v3_finetune | flower_cnn | resnet50 | auto (inception on GPU, flower_cnn on CPU)")
flags.DEFINE_integer("batch_size", 32, "Training batch size (the reference hard-codes 32).")
flags.DEFINE_float("learning_rate", 0.001, "Initial learning rate of the exponential decay schedule.")
flags.DEFINE_float("head_dropout_rate", 0.5, "Drop probability of the head dropout (the reference feeds 0.5).")
flags.DEFINE_string("trunk_bn", "batch", "Inception trunk batch-norm while training: batch (statistics of the batch, as the "
                    "reference's is_training=True) | moving (frozen moving statistics, folded into the convolutions).")
flags.DEFINE_float("bn_decay", 0.0, "Decay of the Inception trunk's batch-norm moving statistics (0 = slim's 0.9997, which "
                   "needs ~10k steps to leave the initial 0 / 1; a short run that is to be evaluated or served wants e.g. 0.9).")
flags.DEFINE_string("device", "auto", "cuda | cpu | auto")
flags.DEFINE_integer("max_steps", 0, "Stop after this many steps (0 = n_epochs * iterations per epoch).")
flags.DEFINE_integer("seed", 42, "Random seed.")
flags.DEFINE_boolean("prefetch", True, "GPU: prepare the next batch on a helper thread and a copy stream during the step.")
flags.DEFINE_integer("prefetch_depth", 2, "GPU: batches prepared ahead of the step.")
flags.DEFINE_float("image_cache_gb", 8.0, "GPU: device memory for decoded images (a file is read and entropy-decoded the "
                   "first time it is drawn; afterwards its pixels are served from the device). 0 disables the cache.")
flags.DEFINE_string("checkpoint_dir", "", "Also write a training checkpoint here at the end (optional).")
flags.DEFINE_string("profile_dir", "", "Write a torch.profiler Chrome trace of a few steps here (empty = off).")
flags.DEFINE_boolean("hip_graph", True, "replay the training step from a HIP graph (GPU only; steps of a different batch size run eagerly)")
flags.DEFINE_boolean("export", True, "Export the servable at the end.")

FLAGS = flags.FLAGS


def main(argv):
    from distributed_tensorflow_amd.data import flowers
    from distributed_tensorflow_amd.io.export import export_servable
    from distributed_tensorflow_amd.io.saver import Saver
    from distributed_tensorflow_amd.models import build_model
    from distributed_tensorflow_amd.optim import exponential_decay
    from distributed_tensorflow_amd.parallel.engine import DataParallelEngine
    from distributed_tensorflow_amd.train.session import (LoggingHook, SummarySaverHook, ThroughputMeter,
                                                          TrainingSession)
    import flower_input_data

    device = FLAGS.device
    if device == "auto":
        device = "cuda" if torch.cuda.is_available() else "cpu"
    device = torch.device(device)
    model_name = FLAGS.model
    if model_name == "auto":
        model_name = "inception_v3_head" if device.type == "cuda" else "flower_cnn"
    # one seed for every generator the run draws from (the reference's reset_graph(seed) seeds TensorFlow and
    # numpy, recognition_flower/single/flower_train.py:35-38; batches are drawn with Python's random.sample)
    import random

    random.seed(FLAGS.seed)
    torch.manual_seed(FLAGS.seed)
    np_rng = np.random.default_rng(FLAGS.seed)

    flowers_root_path = FLAGS.dataset_path
    flower_classes = flowers.list_classes(flowers_root_path)
    items = flowers.get_image_path_and_class(flowers_root_path, flower_classes)
    train, test = flowers.train_test_split(items, test_ratio=0.2, seed=FLAGS.seed)
    print("classes:", flower_classes, "| train", len(train), "test", len(test))

    batch_size = min(FLAGS.batch_size, len(train))
    n_iterations_per_epoch = max(1, len(train) // batch_size)
    if model_name == "inception_v3_head":
        kw = {"head_dropout_rate": FLAGS.head_dropout_rate, "trunk_bn": FLAGS.trunk_bn}
    elif model_name == "inception_v3_finetune":  # the trunk is trained as well (beyond the reference's var_list=fc_vars)
        kw = {"head_dropout_rate": FLAGS.head_dropout_rate}
    else:
        kw = {"dropout_rate": FLAGS.head_dropout_rate} if model_name == "flower_cnn" else {}
    model = build_model(model_name, num_classes=len(flower_classes), device=device, **kw)
    if FLAGS.bn_decay > 0 and hasattr(model, "trunk"):
        from distributed_tensorflow_amd.models.layers import ConvBnAct

        for m in model.trunk.modules():
            if isinstance(m, ConvBnAct):
                m.momentum = 1.0 - FLAGS.bn_decay
    engine = DataParallelEngine(model, mode="sync_dp", optimizer="adam",
                                lr=exponential_decay(FLAGS.learning_rate, decay_steps=200, decay_rate=0.95))
    saver = Saver(model, engine)
    global_step = 0
    if FLAGS.model_path:
        global_step = saver.restore(FLAGS.model_path)
        print("restored", FLAGS.model_path, "at global step", global_step)

    meter = ThroughputMeter()
    hooks = [SummarySaverHook(1, os.path.join(FLAGS.summaries_dir + str(FLAGS.model_version), "train")), meter,
             LoggingHook(1)]
    if FLAGS.profile_dir:
        from distributed_tensorflow_amd.train.session import ProfilerHook

        hooks.append(ProfilerHook(FLAGS.profile_dir))
    head_l2 = getattr(getattr(model, "head", None), "act_l2", 0.0)
    sess = TrainingSession(model, engine, num_classes=len(flower_classes), hooks=hooks, global_step=global_step,
                           logits_act_l2=head_l2, hip_graph=FLAGS.hip_graph and device.type == "cuda")

    image_cache = None
    if device.type == "cuda":
        from distributed_tensorflow_amd.data.jpeg import default_cache

        image_cache = default_cache(device, FLAGS.image_cache_gb)

    def next_batch(pool, bs):
        if device.type == "cuda":
            return flower_input_data.prepare_batch_on_device(pool, bs, device, FLAGS.image_size, np_rng=np_rng,
                                                             cache=image_cache)
        X, y = flowers.prepare_batch(pool, bs, FLAGS.image_size, np_rng)
        Xp = np.zeros(X.shape[:3] + (8,), dtype=np.float32)
        Xp[..., :3] = X
        return torch.from_numpy(Xp), torch.from_numpy(y)

    # on a GPU the next training batch is read, decoded and augmented on a helper thread and a copy stream while
    # the current step runs
    prefetcher = None
    if device.type == "cuda" and FLAGS.prefetch:
        from distributed_tensorflow_amd.data.prefetch import DevicePrefetcher

        prefetcher = DevicePrefetcher(lambda: next_batch(train, batch_size), device, depth=FLAGS.prefetch_depth)
    t0 = time.time()
    done = False
    misses_seen = 0
    with sess:
        for epoch in range(FLAGS.n_epochs):
            print("Epoch", epoch, end="\n")
            for _ in range(n_iterations_per_epoch):
                X, y = prefetcher.next() if prefetcher is not None else next_batch(train, batch_size)
                sess.run_step(X, y)
                if FLAGS.max_steps and sess.global_step - global_step >= FLAGS.max_steps:
                    done = True
                    break
            if done:
                break
            if prefetcher is not None and image_cache is not None:
                # an epoch served entirely from the device-resident image cache leaves the helper thread nothing to
                # overlap (a batch is then a few kernel launches): prepare batches in line until files miss again
                missed, misses_seen = image_cache.misses - misses_seen, image_cache.misses
                prefetcher.set_ahead(missed > 0)
        if prefetcher is not None:
            prefetcher.close()
        if test:
            # the whole held-out split, each image once, no augmentation; example-weighted means
            n = 0
            acc_sum = loss_sum = 0.0
            for X, y in flower_input_data.evaluation_batches(test, batch_size, device, FLAGS.image_size, cache=image_cache):
                ev = sess.evaluate(X, y)
                k = int(y.shape[0])
                acc_sum, loss_sum, n = acc_sum + ev["accuracy"] * k, loss_sum + ev["loss"] * k, n + k
            held_out = {"accuracy": acc_sum / n, "loss": loss_sum / n}
            print("held-out set (%d images): accuracy %.4f loss %.4f" % (n, held_out["accuracy"], held_out["loss"]))
            if hasattr(model, "trunk") and getattr(model, "trunk_bn", "") == "batch" and FLAGS.bn_decay <= 0 \
                    and sess.global_step < 5000:
                # the evaluation normalises with the moving statistics, the training step with the batch's
                print("  (evaluated with moving statistics that %d steps at decay 0.9997 have barely moved from their "
                      "initial 0 / 1, as in the reference, which never updates them at all: --bn_decay=0.9 tracks the "
                      "batch statistics, --trunk_bn=moving trains on the moving ones)" % sess.global_step)
            from distributed_tensorflow_amd.io.events import EventWriter

            ew = EventWriter(os.path.join(FLAGS.summaries_dir + str(FLAGS.model_version), "eval"))
            ew.add_summary(sess.global_step, held_out, {})
            ew.close()
        if FLAGS.checkpoint_dir:
            print("checkpoint:", sess.save_checkpoint(saver, FLAGS.checkpoint_dir, sess.global_step))
    print("trained %d steps in %.1fs (%.1f images/sec)" % (sess.global_step - global_step, time.time() - t0,
                                                           meter.images_per_sec))

    if FLAGS.export:
        path = export_servable(model, flower_classes, FLAGS.output_dir, FLAGS.model_version, model_name=model_name,
                               image_size=FLAGS.image_size, model_kwargs=kw, engine=engine,
                               global_step=sess.global_step)
        print("Exported servable to", path)
    print("Done exporting!")
    return 0


if __name__ == "__main__":
    flags.run(main)
