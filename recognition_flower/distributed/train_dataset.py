#!/usr/bin/env python
"""Distributed flower classifier training driven by ``TF_CONFIG``.

Entry point with the flags and roles of the reference's
``recognition_flower/distributed/train_dataset.py`` (flags: lines 10-31 there;
role dispatch and training loop: 122-192):

    TF_CONFIG='{"cluster": {"ps": [...], "worker": [...]}, "task": {"type": "ps"|"worker", "index": N}}' \
        python train_dataset.py [--tfrecords_path=./train_v1.tfrecord] [--checkpoint_dir=./output/checkpoint_dir/]
                                [--summaries_dir=./output/events/] [--train_steps=10] [--batch_size=32]
                                [--learning_rate=0.001] [--n_epochs=1000] [--model_version=1]

``--mode`` selects how the cluster is realised:

``async_ps`` (default)  the reference's semantics -- ps tasks own the trainable variables and the
    Adam slots, workers pull / compute / push asynchronously over TCP, ``--train_steps`` is a budget of
    *global* steps shared by all workers, checkpoints and summaries every 2 global steps.
    ``--staleness=s`` bounds how far a worker may run ahead (``-1`` = unbounded, as in the reference).
``sync_dp`` / ``sharded_ps`` / ``async_ssp``  collective modes over ``torch.distributed`` (RCCL on GPUs,
    gloo on CPUs); only worker tasks take part, ps tasks exit immediately ("every GPU is the ps for
    1/N of the parameters").  Also usable under ``torchrun`` without ``TF_CONFIG``.

Differences from the reference, on purpose: variables are initialised once (by the chief), file-writing
hooks run on the chief only, a ps does not wipe the checkpoint directory unless ``--wipe_checkpoint_dir``
is given, ``--resume`` restores the latest checkpoint, input is RGB in [-1, 1] sharded per worker with
example-level shuffling, and head dropout defaults to rate 0.2 (the reference's ``keep_prob=0.8`` passed
as a *rate* drops 80 %; ``--head_dropout_rate=0.8`` reproduces that).
"""
import json
import os
import shutil
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for peer-mapped GPU memory (RCCL, xGMI mailboxes)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for peer-mapped GPU memory (RCCL, xGMI mailboxes)

import torch  # noqa: E402

from distributed_tensorflow_amd.utils import flags  # noqa: E402

flags.DEFINE_integer("model_version", 1, "Version number of the flags.")
flags.DEFINE_integer("n_epochs", 1000, "Needs to provide the number of epochs as in training.")
flags.DEFINE_string("tfrecords_path", "./train_v1.tfrecord",
                    "Needs to provide the tf records file for training.(train_v2~1.15G ,train_v1~38M)")
flags.DEFINE_string("checkpoint_dir", "./output/checkpoint_dir/",
                    "Needs to provide the model path (checkpoint,meta) in training.")
flags.DEFINE_string("summaries_dir", "./output/events/", "Needs to provide the summary output dir in training.")
flags.DEFINE_integer("train_steps", 10, "Number of training steps to perform")
flags.DEFINE_integer("batch_size", 32, "Training batch size ")
flags.DEFINE_float("learning_rate", 0.001, "Learning rate")
# additions
flags.DEFINE_string("mode", "async_ps", "async_ps | sync_dp | sharded_ps | async_ssp")
flags.DEFINE_float("bn_decay", 0.0, "decay of the Inception trunk's batch-norm moving statistics (0 = slim's 0.9997)")
flags.DEFINE_string("model", "auto", "inception_v3_head | inception_v3_finetune (the trunk is trained too) | flower_cnn | resnet50 | auto")
flags.DEFINE_integer("image_size", 299, "Network input size.")
flags.DEFINE_integer("staleness", -1, "Staleness bound (async_ps: -1 = unbounded; async_ssp: delay in steps).")
flags.DEFINE_float("bucket_mb", 32.0, "Gradient bucket size for the collective modes.")
flags.DEFINE_float("head_dropout_rate", 0.2, "Drop probability of the head dropout.")
flags.DEFINE_string("trunk_bn", "batch", "Inception trunk batch-norm while training: batch (statistics of the batch, as the "
                    "reference's is_training=True) | moving (frozen moving statistics, folded into the convolutions).")
flags.DEFINE_integer("save_steps", 2, "Checkpoint / summary period in global steps (the reference uses 2).")
flags.DEFINE_boolean("resume", False, "Restore the latest checkpoint from checkpoint_dir before training.")
flags.DEFINE_boolean("wipe_checkpoint_dir", False, "ps task 0 deletes checkpoint_dir at start (reference behaviour).")
flags.DEFINE_boolean("synthetic", False, "Use synthetic input instead of the TFRecord file.")
flags.DEFINE_boolean("legacy_bgr_input", False, "Feed BGR [0,255] pixels like the reference's reader.")
flags.DEFINE_string("device", "auto", "cuda | cpu | auto")
flags.DEFINE_integer("p2p_small_kb", 0, "sync_dp on GPUs: gradient buckets up to this size use the one-shot xGMI mailbox all-reduce instead of RCCL (0 = off)")
flags.DEFINE_string("profile_dir", "", "Write a torch.profiler Chrome trace of a few steps here (empty = off).")
flags.DEFINE_string("hip_graph", "auto", "on a GPU: replay the device part of the step from a HIP graph -- true | false | auto "
                    "(auto: on for async_ps, where the graph holds forward + backward and the pull / push stay on the host around "
                    "it; off for the collective modes, whose graph would hold the collectives as well)")
flags.DEFINE_string("fault_inject", "", "Testing hook: 'exit_at_step:N' makes this worker die abruptly at global step N.")
flags.DEFINE_boolean("shutdown_ps", True, "The chief stops the ps tasks when it finishes (the reference's ps never exits).")
flags.DEFINE_integer("eval_batches", 8, "evaluator: batches of the record file evaluated per checkpoint.")
flags.DEFINE_float("eval_poll_seconds", 0.5, "evaluator: pause between looks at checkpoint_dir.")
flags.DEFINE_float("eval_timeout", 120.0, "evaluator: stop when no new checkpoint appeared for this many seconds.")
flags.DEFINE_float("collective_timeout", 600.0, "Collective modes: seconds a rank waits inside a collective before the job "
                   "aborts (a lost rank must end the job so that a restart with --resume can take over).")
flags.DEFINE_string("ps_backend", "tcp", "async_ps: tcp (ps processes over TCP, CPUs or GPUs, elastic worker set) | xgmi (one node "
                    "of GPUs: every worker GPU owns 1/N of the variables and their optimizer slots, gradients are pushed into "
                    "the owner's memory and variables pulled from it over xGMI peer access; ps tasks have nothing to do)")
flags.DEFINE_string("ps_apply", "auto", "ps_backend=xgmi: per_push (one optimizer step per arrived gradient, the reference's ps) | "
                    "aggregate (staleness 0 only: one step per round on the sum of the workers' gradients) | auto")
flags.DEFINE_string("ps_pipeline", "auto", "ps_backend=tcp: push and pull on a helper thread while the device computes the next step (one "
                    "more step of staleness on the worker's own gradient) -- true | false | auto (on unless --staleness=0)")
flags.DEFINE_string("optimizer", "adam", "adam (the reference) | momentum")
flags.DEFINE_float("momentum", 0.9, "momentum optimizer: heavy-ball coefficient.")
flags.DEFINE_float("weight_decay", 0.0, "momentum optimizer: L2 weight decay on convolution / dense kernels.")
flags.DEFINE_integer("lr_decay_steps", 0, "Exponential learning-rate decay period in global steps (0 = constant rate, as "
                     "the reference's distributed script; its single-process script uses 200).")
flags.DEFINE_float("lr_decay_rate", 0.95, "Exponential learning-rate decay factor per lr_decay_steps.")
flags.DEFINE_string("grad_dtype", "fp32", "Collective modes: dtype of the gradient buckets on the wire -- fp32 | bf16.")
flags.DEFINE_float("worker_timeout", 30.0, "Seconds of silence after which a worker no longer counts for the staleness bound.")

FLAGS = flags.FLAGS

params = {
    "learning_rate": FLAGS.learning_rate,
    "train_steps": FLAGS.train_steps,
    "batch_size": FLAGS.batch_size,
    "num_epochs": FLAGS.n_epochs,
    "threads": 4,
}


def _device():
    d = FLAGS.device
    if d == "auto":
        d = "cuda" if torch.cuda.is_available() else "cpu"
    return torch.device(d)


def _model_name(device):
    return FLAGS.model if FLAGS.model != "auto" else ("inception_v3_head" if device.type == "cuda" else "flower_cnn")


def _num_classes():
    meta = FLAGS.tfrecords_path + ".labels.json"
    if os.path.exists(meta):
        with open(meta) as f:
            return len(json.load(f)["classes"])
    return 5  # the reference hard-codes n_outputs=5


def train_input_fn(tfrecords_path, params, device, rank, world, channels=8):
    from distributed_tensorflow_amd.data.records import RecordInput, SyntheticInput

    if FLAGS.synthetic:
        return SyntheticInput(params["batch_size"], FLAGS.image_size, device, seed=rank, channels=channels)
    return RecordInput(tfrecords_path, params["batch_size"], device, image_size=FLAGS.image_size,
                       num_epochs=params["num_epochs"], rank=rank, world=world, threads=params["threads"],
                       prefetch=4, shuffle=True, seed=1234, legacy_bgr_unnormalised=FLAGS.legacy_bgr_input,
                       channels=channels)


def _optimizer_config():
    """(optimizer name, its keyword arguments, learning rate or schedule) from the flags; the same on every task."""
    name = FLAGS.optimizer.lower()
    kwargs = {"momentum": FLAGS.momentum, "weight_decay": FLAGS.weight_decay} if name == "momentum" else {}
    rate = params["learning_rate"]
    if FLAGS.lr_decay_steps > 0:
        from distributed_tensorflow_amd.optim.schedules import exponential_decay

        rate = exponential_decay(rate, FLAGS.lr_decay_steps, FLAGS.lr_decay_rate)
    return name, kwargs, rate


def _xps_cluster(spec):
    """ps tasks own the variables in GPU memory: ps task 0's address is where the rendezvous store lives."""
    from distributed_tensorflow_amd.parallel.xps import XpsCluster

    host, port = spec.address("ps", 0)
    return XpsCluster("127.0.0.1" if host == "localhost" else host, port, len(spec.ps_hosts), spec.num_workers)


def run_ps(spec):
    from distributed_tensorflow_amd.parallel.ps import ParameterServer

    if FLAGS.mode != "async_ps":
        print("ps task %d: mode %s keeps parameters on the workers; nothing to serve" % (spec.task_index, FLAGS.mode))
        return 0
    if FLAGS.wipe_checkpoint_dir and spec.task_index == 0 and os.path.exists(FLAGS.checkpoint_dir):
        shutil.rmtree(FLAGS.checkpoint_dir)
    if FLAGS.ps_backend == "xgmi":
        # the ps task owns its shard of the variables and their optimizer slots in GPU memory and applies the
        # gradients the workers push into its mailboxes (the reference's ps: train_dataset.py:141-145)
        from distributed_tensorflow_amd.parallel.xps import XgmiPSServer
        from distributed_tensorflow_amd.utils.cluster import local_device_index

        if not torch.cuda.is_available():
            raise ValueError("--ps_backend=xgmi keeps the variables in GPU memory")
        idx = local_device_index(spec.num_workers + spec.task_index, torch.cuda.device_count()) or 0
        torch.cuda.set_device(idx)
        opt_name, opt_kwargs, lr = _optimizer_config()
        server = XgmiPSServer(_xps_cluster(spec), spec.task_index, optimizer=opt_name, lr=lr, opt_kwargs=opt_kwargs,
                              timeout=FLAGS.worker_timeout, device=torch.device("cuda", idx))
        print("ps %d owning shard %d of %d on GPU %d" % (spec.task_index, spec.task_index, len(spec.ps_hosts), idx), flush=True)
        applied = server.serve_forever()
        print("ps %d stopped after %d optimizer steps" % (spec.task_index, applied))
        return 0
    host, port = spec.my_address()
    server = ParameterServer("0.0.0.0" if host not in ("localhost", "127.0.0.1") else "127.0.0.1", port,
                             ps_index=spec.task_index, staleness=FLAGS.staleness, worker_timeout=FLAGS.worker_timeout)
    print("ps %d serving on port %d" % (spec.task_index, port), flush=True)
    server.serve_forever()  # the counterpart of server.join()
    print("ps %d stopped at global step %d" % (spec.task_index, server.global_step))
    return 0


def run_worker(spec):
    from distributed_tensorflow_amd.io.saver import Saver
    from distributed_tensorflow_amd.models import build_model
    from distributed_tensorflow_amd.parallel.ps import PSStopped
    from distributed_tensorflow_amd.train.session import (CheckpointSaverHook, LoggingHook, StopAtStepHook,
                                                          SummarySaverHook, ThroughputMeter, TrainingSession)
    from distributed_tensorflow_amd.utils.cluster import rendezvous_from_cluster, torchrun_env

    device = _device()
    task_index = spec.trainer_index
    world = spec.num_workers
    tr = torchrun_env()
    if spec.task_type is None and tr is not None:
        task_index, world = tr[0], tr[1]
        if device.type == "cuda":
            torch.cuda.set_device(tr[2])
            device = torch.device("cuda", tr[2])
    elif spec.task_type is not None and device.type == "cuda" and device.index is None and world > 1:
        # TF_CONFIG roles on one machine (run.sh, a TFJob pod with several GPUs): worker i takes GPU i modulo the
        # number of visible devices, as the reference's one-worker-per-GPU deployments do with CUDA_VISIBLE_DEVICES;
        # --device=cuda:N pins a device instead.  (Counting devices does not initialise the GPU.)
        from distributed_tensorflow_amd.utils.cluster import local_device_index

        idx = local_device_index(task_index, torch.cuda.device_count())
        if idx is not None:
            torch.cuda.set_device(idx)
            device = torch.device("cuda", idx)
    is_chief = task_index == 0
    torch.manual_seed(1234)  # identical initial values on every worker
    model_name = _model_name(device)
    n_classes = _num_classes()
    if model_name == "inception_v3_head":
        kw = {"head_dropout_rate": FLAGS.head_dropout_rate, "trunk_bn": FLAGS.trunk_bn}
    elif model_name == "inception_v3_finetune":
        kw = {"head_dropout_rate": FLAGS.head_dropout_rate, "trunk_bn": FLAGS.trunk_bn}
    else:
        kw = {"dropout_rate": FLAGS.head_dropout_rate} if model_name == "flower_cnn" else {}
    model = build_model(model_name, num_classes=n_classes, device=device, **kw)
    if FLAGS.bn_decay > 0 and hasattr(model, "trunk"):
        from distributed_tensorflow_amd.models.layers import ConvBnAct

        for m in model.trunk.modules():
            if isinstance(m, ConvBnAct):
                m.momentum = 1.0 - FLAGS.bn_decay

    opt_name, opt_kwargs, lr_schedule = _optimizer_config()
    lr = params["learning_rate"]
    xgmi = FLAGS.mode == "async_ps" and FLAGS.ps_backend == "xgmi"
    if xgmi:
        import datetime

        import torch.distributed as dist
        from distributed_tensorflow_amd.parallel.xps import XgmiPSEngine

        if device.type != "cuda":
            raise ValueError("--ps_backend=xgmi keeps the variables in GPU memory")
        grad_dtype = {"fp32": torch.float32, "bf16": torch.bfloat16}[FLAGS.grad_dtype]
        if spec.ps_hosts:
            # the reference's topology: the ps tasks own the variables, this worker is a pure client of their
            # GPU memory and may start late or die without taking anything with it
            engine = XgmiPSEngine(model, optimizer=opt_name, lr=lr_schedule, opt_kwargs=opt_kwargs, staleness=FLAGS.staleness,
                                  apply=FLAGS.ps_apply, grad_dtype=grad_dtype, timeout=FLAGS.worker_timeout,
                                  cluster=_xps_cluster(spec), worker_index=task_index,
                                  defer_init=FLAGS.resume and is_chief)
        else:
            if world > 1 and not dist.is_initialized():
                if tr is None:
                    os.environ.update(rendezvous_from_cluster(spec))
                # the process group only carries the IPC handles and the chief's initial values at start-up
                dist.init_process_group("gloo", rank=task_index, world_size=world,
                                        timeout=datetime.timedelta(seconds=FLAGS.collective_timeout))
            # no ps tasks: every worker GPU owns 1/N of the variables
            engine = XgmiPSEngine(model, optimizer=opt_name, lr=lr_schedule, opt_kwargs=opt_kwargs, staleness=FLAGS.staleness,
                                  apply=FLAGS.ps_apply, grad_dtype=grad_dtype, timeout=FLAGS.worker_timeout)
    elif FLAGS.mode == "async_ps" and spec.ps_hosts:
        from distributed_tensorflow_amd.parallel.ps import PSWorkerEngine

        ps_addr = [spec.address("ps", i) for i in range(len(spec.ps_hosts))]
        ps_addr = [("127.0.0.1" if h == "localhost" else h, p) for h, p in ps_addr]
        # a resuming chief restores the checkpoint first and initialises the ps tasks from it
        engine = PSWorkerEngine(model, ps_addr, task_index, is_chief, optimizer=opt_name, lr=lr, opt_kwargs=opt_kwargs,
                                defer_init=FLAGS.resume and is_chief,
                                lr_decay=(FLAGS.lr_decay_steps, FLAGS.lr_decay_rate) if FLAGS.lr_decay_steps > 0 else None,
                                pipeline=_flag_on(FLAGS.ps_pipeline, FLAGS.staleness != 0, "--ps_pipeline"))
    else:
        import torch.distributed as dist
        from distributed_tensorflow_amd.parallel.engine import DataParallelEngine

        if world > 1 and not dist.is_initialized():
            if tr is None:
                os.environ.update(rendezvous_from_cluster(spec))
            import datetime

            dist.init_process_group("nccl" if device.type == "cuda" else "gloo", rank=task_index, world_size=world,
                                    timeout=datetime.timedelta(seconds=FLAGS.collective_timeout))
        mode = "sync_dp" if FLAGS.mode == "async_ps" else FLAGS.mode
        grad_dtype = {"fp32": torch.float32, "bf16": torch.bfloat16}[FLAGS.grad_dtype]
        engine = DataParallelEngine(model, mode=mode, optimizer=opt_name, lr=lr_schedule, opt_kwargs=opt_kwargs,
                                    bucket_mb=FLAGS.bucket_mb, staleness=max(FLAGS.staleness, 0),
                                    p2p_max_bytes=FLAGS.p2p_small_kb * 1024, grad_dtype=grad_dtype)
    saver = Saver(model, engine)
    start_step = 0
    # async_ps: the chief restores and uploads to the ps tasks; collective modes: every rank loads the file
    collective = not (FLAGS.mode == "async_ps" and spec.ps_hosts)
    if FLAGS.resume and (is_chief or collective):
        restored = saver.restore_latest(FLAGS.checkpoint_dir)
        if restored is not None:
            start_step = restored
            print("resumed from global step", restored)
    if hasattr(engine, "initialize"):
        engine.initialize()  # (deferred on a resuming chief; a no-op otherwise)
        start_step = max(start_step, int(engine.global_step))

    hooks = [StopAtStepHook(params["train_steps"]),
             CheckpointSaverHook(FLAGS.checkpoint_dir, FLAGS.save_steps, saver),
             SummarySaverHook(FLAGS.save_steps, os.path.join(FLAGS.summaries_dir + str(FLAGS.model_version), "train")),
             ThroughputMeter(world_size=world),
             LoggingHook(2, worker_index=task_index)]
    if FLAGS.profile_dir:
        from distributed_tensorflow_amd.train.session import ProfilerHook

        hooks.append(ProfilerHook(os.path.join(FLAGS.profile_dir, "task%d" % task_index)))
    head_l2 = getattr(getattr(model, "head", None), "act_l2", 0.0)
    # the reference scales the activity regulariser by 0.01 in this script (train_dataset.py:159-160)
    sess = TrainingSession(model, engine, num_classes=n_classes, hooks=hooks, is_chief=is_chief, reg_weight=0.01,
                           logits_act_l2=0.01 * head_l2, global_step=start_step,
                           hip_graph=_want_hip_graph() and device.type == "cuda")
    # (image layout the model's stem prefers on this device: 8 channels, or 4 for the pixel-pair stem)
    data = train_input_fn(FLAGS.tfrecords_path, params, device, task_index, world,
                          channels=getattr(model, "input_channels", 8) if FLAGS.image_size % 2 == 0 else 8)
    fault_step = int(FLAGS.fault_inject.split(":")[1]) if FLAGS.fault_inject.startswith("exit_at_step:") else None
    t0 = time.time()
    with sess:
        step = getattr(engine, "global_step", start_step)
        while not sess.should_stop() and step < params["train_steps"]:
            try:
                x, y = next(data)
            except StopIteration:  # input exhausted after n_epochs, like OutOfRangeError
                break
            try:
                r = sess.run_step(x, y)
            except PSStopped:
                print("Worker %d: parameter server finished the step budget" % task_index)
                break
            gs = getattr(engine, "global_step", None)
            if gs is not None and FLAGS.mode == "async_ps":
                sess.global_step = step = int(gs)  # the step budget is global, shared by all workers
            else:
                step = r.global_step
            if fault_step is not None and step >= fault_step:
                print("Worker %d: injected fault at step %d" % (task_index, step), flush=True)
                os._exit(17)
    print("finished training (worker %d, %.1fs, global step %d)" % (task_index, time.time() - t0, step))
    sess.release_graph()   # (before any communicator goes away: a captured step holds its kernels)
    if hasattr(engine, "close"):
        engine.close(shutdown_ps=FLAGS.shutdown_ps and is_chief)
    if hasattr(data, "close"):
        data.close()
    return 0


def _flag_on(value, auto: bool, name: str) -> bool:
    v = str(value).strip().lower()
    if v in ("auto", ""):
        return bool(auto)
    if v in ("1", "true", "yes", "on"):
        return True
    if v in ("0", "false", "no", "off"):
        return False
    raise ValueError("%s must be true, false or auto (got %r)" % (name, value))


def _want_hip_graph() -> bool:
    return _flag_on(FLAGS.hip_graph, FLAGS.mode == "async_ps", "--hip_graph")


def run_evaluator(spec):
    """``evaluator`` replica of a TFJob: follows ``checkpoint_dir``, loads every new checkpoint into an
    inference-mode model, measures loss / top-1 accuracy over ``--eval_batches`` batches of the record file and
    writes them as summaries under ``<summaries_dir><model_version>/eval`` at the checkpoint's global step.
    (The reference computes a held-out split and never evaluates it, recognition_flower/single/flower_train.py:105-106;
    its distributed script has no evaluation at all.)  Ends when a checkpoint at or beyond ``--train_steps`` has
    been evaluated, or when no new checkpoint appears for ``--eval_timeout`` seconds."""
    from distributed_tensorflow_amd.io import bundle
    from distributed_tensorflow_amd.io.events import EventWriter
    from distributed_tensorflow_amd.io.saver import Saver
    from distributed_tensorflow_amd.models import build_model
    from distributed_tensorflow_amd.ops import functional as F

    device = _device()
    model_name = _model_name(device)
    n_classes = _num_classes()
    model = build_model(model_name, num_classes=n_classes, device=device).eval()
    saver = Saver(model)
    writer = EventWriter(os.path.join(FLAGS.summaries_dir + str(FLAGS.model_version), "eval"))
    eval_params = dict(params, num_epochs=1)
    seen, last_new = None, time.time()
    while True:
        prefix = bundle.latest_checkpoint(FLAGS.checkpoint_dir) if os.path.isdir(FLAGS.checkpoint_dir) else None
        if prefix is None or prefix == seen:
            if time.time() - last_new > FLAGS.eval_timeout:
                print("Evaluator: no new checkpoint for %.0f s, stopping" % FLAGS.eval_timeout)
                break
            time.sleep(FLAGS.eval_poll_seconds)
            continue
        try:
            step = saver.restore(prefix, restore_slots=False)
        except Exception as e:  # noqa: BLE001 -- e.g. pruned by max_to_keep between listing and reading
            print("Evaluator: skipping %s (%s: %s)" % (os.path.basename(prefix), type(e).__name__, e))
            seen, last_new = prefix, time.time()
            continue
        seen, last_new = prefix, time.time()
        data = train_input_fn(FLAGS.tfrecords_path, eval_params, device, 0, 1,
                              channels=getattr(model, "input_channels", 8) if FLAGS.image_size % 2 == 0 else 8)
        n = correct = 0
        loss_sum = 0.0
        with torch.no_grad():
            for i, (x, y) in enumerate(data):
                if i >= FLAGS.eval_batches:
                    break
                out = model(x)
                logits = out[0] if isinstance(out, tuple) else out
                xent, _, _, ok = F.softmax_xent(logits, y, n_classes)
                loss_sum += float(xent.float().item()) * int(y.shape[0])
                correct += int(ok.float().sum().item())
                n += int(y.shape[0])
        if hasattr(data, "close"):
            data.close()
        acc, loss = correct / max(n, 1), loss_sum / max(n, 1)
        writer.add_summary(step, {"accuracy": acc, "loss": loss})
        writer.flush()
        print("Evaluator: global step %d accuracy %.4f loss %.4f (%d examples)" % (step, acc, loss, n), flush=True)
        if step >= params["train_steps"]:
            break
    writer.close()
    return 0


def main(unused_argv):
    from distributed_tensorflow_amd.utils.cluster import parse_tf_config

    print(os.environ.get("TF_CONFIG"))
    spec = parse_tf_config()
    if spec.task_type == "ps":
        return run_ps(spec)
    if spec.task_type == "evaluator":
        return run_evaluator(spec)
    if not spec.is_trainer:
        print("task type %r has no role in this trainer; exiting" % spec.task_type)
        return 0
    return run_worker(spec)


if __name__ == "__main__":
    flags.run(main)
