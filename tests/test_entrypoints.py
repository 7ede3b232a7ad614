"""End-to-end runs of the reference entry points on the CPU (small CNN, tiny synthetic flower dataset):
single-process train + export + serve, and prepare -> 1 ps + 2 workers (async, one worker killed) ->
checkpoint -> save_model -> serve."""
import glob
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from distributed_tensorflow_amd.io import bundle, events
from distributed_tensorflow_amd.io.export import Servable, example_from_jpeg
from test_data import _make_dataset

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SINGLE = os.path.join(ROOT, "recognition_flower", "single")
DIST = os.path.join(ROOT, "recognition_flower", "distributed")


def _run(cmd, cwd, env=None, timeout=300):
    e = dict(os.environ)
    e.pop("TF_CONFIG", None)
    e.update(env or {})
    return subprocess.run([sys.executable] + cmd, cwd=cwd, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                          text=True, timeout=timeout)


def test_single_trainer_trains_exports_and_serves(tmp_path):
    data = _make_dataset(str(tmp_path / "flowers"), n_per_class=8)
    out = tmp_path / "out"
    r = _run(["flower_train.py", "--dataset_path=" + data, "--image_size=32", "--n_epochs=6", "--batch_size=8",
              "--learning_rate=0.01", "--output_dir=%s/models" % out, "--summaries_dir=%s/events/" % out,
              "--checkpoint_dir=%s/ckpt" % out, "--model_version=3", "--profile_dir=%s/prof" % out], SINGLE)
    assert r.returncode == 0, r.stdout[-3000:]
    assert glob.glob(str(out / "prof" / "trace_steps_*.json"))  # torch.profiler window
    losses = [float(m) for m in re.findall(r"step \d+: .* loss (\d+\.\d+)", r.stdout)]
    assert len(losses) >= 12 and np.mean(losses[-3:]) < np.mean(losses[:3])  # loss decreases
    assert "Done exporting!" in r.stdout
    # the whole held-out 20 % (24 images -> 5) is evaluated, un-augmented, and written as eval summaries
    assert re.search(r"held-out set \(5 images\): accuracy \d\.\d+ loss \d+\.\d+", r.stdout), r.stdout[-1500:]
    ev_eval = glob.glob(str(out / "events" / "3" / "eval" / "events.out.tfevents.*"))
    assert len(ev_eval) == 1
    ee = [e for e in events.read_events(ev_eval[0]) if e["scalars"]]
    assert len(ee) == 1 and {"accuracy", "loss"} <= set(ee[0]["scalars"]) and ee[0]["step"] == len(losses)
    # summaries: scalars + logits histogram, indexed by global step
    ev_files = glob.glob(str(out / "events" / "3" / "train" / "events.out.tfevents.*"))
    assert len(ev_files) == 1
    evs = list(events.read_events(ev_files[0]))
    assert [e["step"] for e in evs[1:4]] == [1, 2, 3]
    assert {"loss", "accuracy", "learning_rate"} <= set(evs[1]["scalars"]) and "logits_histogram" in evs[1]["histograms"]
    assert abs(evs[1]["scalars"]["learning_rate"] - 0.01) < 1e-6 and evs[-1]["scalars"]["learning_rate"] < 0.01
    # checkpoint in Saver-V2 layout with the head's reference names
    assert bundle.latest_checkpoint(str(out / "ckpt")) is not None
    keys = [e[0] for e in bundle.list_bundle(bundle.latest_checkpoint(str(out / "ckpt")))]
    assert "fc_trainable/dense_prelogit1/kernel" in keys and "fc_trainable/flower_logits/bias/Adam_1" in keys
    # servable: versioned directory, two signatures, class strings + raw-logit scores
    export = out / "models" / "3"
    assert sorted(os.listdir(export)) == ["saved_model.json", "saved_model.pb", "variables"]
    assert sorted(os.listdir(export / "variables")) == ["variables.data-00000-of-00001", "variables.index"]
    manifest = json.load(open(export / "saved_model.json"))
    assert manifest["signature_def"]["serving_default"]["method_name"] == "tensorflow/serving/classify"
    assert manifest["signature_def"]["predict_images"]["method_name"] == "tensorflow/serving/predict"
    assert manifest["classes"] == ["daisy", "roses", "tulips"]
    s = Servable(str(export), device="cpu")
    assert s.signature_def["serving_default"]["inputs"]["inputs"]["tensor"] == "tf_example:0"
    jpegs = [open(f, "rb").read() for f in sorted(glob.glob(os.path.join(data, "*", "*.jpg")))[:4]]
    a = s.predict_images(jpegs)
    b = s.serving_default([example_from_jpeg(j) for j in jpegs])
    assert a["classes"].shape == (4, 1) and a["scores"].shape == (4, 1) and a["scores"].dtype == np.float32
    assert a["classes"].tolist() == b["classes"].tolist() and np.allclose(a["scores"], b["scores"])
    assert all(c[0] in manifest["classes"] for c in a["classes"].tolist())
    # resume from the checkpoint through --model_path
    r2 = _run(["flower_train.py", "--dataset_path=" + data, "--image_size=32", "--n_epochs=1", "--batch_size=8",
               "--model_path=" + bundle.latest_checkpoint(str(out / "ckpt")), "--output_dir=%s/models2" % out,
               "--summaries_dir=%s/events2/" % out], SINGLE)
    assert r2.returncode == 0 and "restored" in r2.stdout, r2.stdout[-2000:]
    assert re.search(r"step %d:" % (len(losses) + 1), r2.stdout)  # global step continues


def test_default_dataset_path_works_from_a_clean_checkout(tmp_path):
    """``--dataset_path`` defaults to ``../flower_photos_labeled`` relative to the script directory, as in the
    reference (recognition_flower/single/flower_train.py:22-23, recognition_flower/distributed/prepare.py:10-11);
    the repository ships a drawn stand-in for the reference's 107 sample photographs there."""
    out = tmp_path / "out"
    r = _run(["flower_train.py", "--image_size=32", "--n_epochs=1", "--max_steps=2", "--batch_size=16",
              "--output_dir=%s/models" % out, "--summaries_dir=%s/events/" % out], SINGLE)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "classes: ['daisy', 'dandelion', 'roses', 'sunflower', 'tulips'] | train 85 test 22" in r.stdout
    assert re.search(r"held-out set \(22 images\)", r.stdout)
    rec = str(tmp_path / "train_v1.tfrecord")
    r = _run(["prepare.py", "--tfrecords_path=" + rec], DIST)
    assert r.returncode == 0, r.stdout[-3000:]
    assert json.load(open(rec + ".labels.json"))["classes"] == ["daisy", "dandelion", "roses", "sunflower", "tulips"]
    assert 30e6 < os.path.getsize(rec) < 45e6  # raw pixels of the 107 images: the reference's "train_v1~38M"


def test_distributed_ps_workers_checkpoint_and_export(tmp_path):
    data = _make_dataset(str(tmp_path / "flowers"), n_per_class=8)
    rec = str(tmp_path / "train.tfrecord")
    r = _run(["prepare.py", "--dataset_path=" + data, "--tfrecords_path=" + rec, "--seed=3"], DIST)
    assert r.returncode == 0, r.stdout
    from distributed_tensorflow_amd.io import tfrecord

    recs = list(tfrecord.tfrecord_iterator(rec))
    assert len(recs) == 24
    ex = tfrecord.parse_example(recs[0])
    assert set(ex) == {"image_raw", "label", "height", "width", "channel"}  # reference schema
    assert len(ex["image_raw"][0]) == ex["height"][0] * ex["width"][0] * ex["channel"][0] and ex["channel"] == [3]
    assert json.load(open(rec + ".labels.json"))["classes"] == ["daisy", "roses", "tulips"]

    import socket

    def port():
        s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p

    cluster = {"ps": ["localhost:%d" % port()], "worker": ["localhost:%d" % port() for _ in range(3)]}
    common = ["train_dataset.py", "--tfrecords_path=" + rec, "--checkpoint_dir=%s/ckpt/" % tmp_path,
              "--summaries_dir=%s/events/" % tmp_path, "--train_steps=16", "--batch_size=8", "--image_size=32",
              "--learning_rate=0.01", "--staleness=3", "--worker_timeout=3"]

    def launch(task, extra=()):
        env = dict(os.environ, TF_CONFIG=json.dumps({"cluster": cluster, "task": task}))
        return subprocess.Popen([sys.executable] + common + list(extra), cwd=DIST, env=env, stdout=subprocess.PIPE,
                                stderr=subprocess.STDOUT, text=True)

    procs = {"ps": launch({"type": "ps", "index": 0}),
             "w0": launch({"type": "worker", "index": 0}),
             "w1": launch({"type": "worker", "index": 1}),
             "w2": launch({"type": "worker", "index": 2}, ["--fault_inject=exit_at_step:5"])}
    outs = {}
    try:
        for k, p in procs.items():
            outs[k], _ = p.communicate(timeout=240)
    finally:
        for p in procs.values():
            if p.poll() is None:
                p.kill()
    assert procs["w2"].returncode == 17, outs["w2"][-1500:]          # the injected crash
    assert procs["w0"].returncode == 0, outs["w0"][-3000:]            # ... did not hang or kill the others
    assert procs["w1"].returncode == 0, outs["w1"][-3000:]
    assert procs["ps"].returncode == 0, outs["ps"][-3000:]
    assert "finished training" in outs["w0"] and "ps 0 stopped at global step" in outs["ps"]
    assert re.search(r"Worker \d: training step \d+ done , loss is", outs["w0"] + outs["w1"])
    gstep = int(re.search(r"ps 0 stopped at global step (\d+)", outs["ps"]).group(1))
    assert 16 <= gstep <= 18  # the budget is global; workers in flight may add a step each
    # chief-only checkpoints every 2 global steps, TF state file, at most 5 kept
    latest, allp = bundle.read_checkpoint_state(str(tmp_path / "ckpt"))
    assert latest and len(allp) <= 5 and latest == allp[-1]
    data_ck = bundle.read_bundle(bundle.latest_checkpoint(str(tmp_path / "ckpt")))
    assert int(data_ck["global_step"]) >= 16 and "fc_trainable/flower_logits/kernel/Adam" in data_ck
    assert len(glob.glob(str(tmp_path / "events" / "1" / "train" / "events.out.tfevents.*"))) == 1  # chief only

    r = _run(["save_model.py", "--checkpoint_dir=%s/ckpt/" % tmp_path, "--output_dir=%s/models" % tmp_path,
              "--dataset_path=" + data, "--image_size=32", "--model_version=2"], DIST)
    assert r.returncode == 0 and "restored" in r.stdout and "Done exporting!" in r.stdout, r.stdout[-2000:]
    s = Servable(str(tmp_path / "models" / "2"), device="cpu")
    jpegs = [open(f, "rb").read() for f in sorted(glob.glob(os.path.join(data, "*", "*.jpg")))[:3]]
    res = s.predict_images(jpegs)
    assert res["classes"].shape == (3, 1) and np.isfinite(res["scores"]).all()
    # the servable's variables are the checkpoint's variables
    sv = bundle.read_bundle(str(tmp_path / "models" / "2" / "variables" / "variables"))
    np.testing.assert_array_equal(sv["fc_trainable/flower_logits/kernel"], data_ck["fc_trainable/flower_logits/kernel"])
    # save_model refuses to export when there is no checkpoint
    r = _run(["save_model.py", "--checkpoint_dir=%s/none/" % tmp_path, "--output_dir=%s/m3" % tmp_path,
              "--dataset_path=" + data, "--image_size=32"], DIST)
    assert r.returncode != 0


def test_collective_mode_via_tf_config_workers(tmp_path):
    """--mode=sharded_ps: the worker list of TF_CONFIG becomes a gloo process group; ps tasks have nothing to do."""
    import socket

    def port():
        s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p

    cluster = {"ps": ["localhost:%d" % port()], "worker": ["localhost:%d" % port(), "localhost:%d" % port()]}
    common = ["train_dataset.py", "--synthetic", "--mode=sharded_ps", "--checkpoint_dir=%s/ckpt/" % tmp_path,
              "--summaries_dir=%s/events/" % tmp_path, "--train_steps=4", "--batch_size=4", "--image_size=32"]
    procs = []
    for task in ({"type": "ps", "index": 0}, {"type": "worker", "index": 0}, {"type": "worker", "index": 1}):
        env = dict(os.environ, TF_CONFIG=json.dumps({"cluster": cluster, "task": task}))
        procs.append(subprocess.Popen([sys.executable] + common, cwd=DIST, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert [p.returncode for p in procs] == [0, 0, 0], outs
    assert "nothing to serve" in outs[0]
    assert int(bundle.read_bundle(bundle.latest_checkpoint(str(tmp_path / "ckpt")))["global_step"]) == 4


def test_collective_mode_lost_rank_ends_the_job_and_resume_continues(tmp_path):
    """sync_dp over two workers: one rank dies at step 3 (fault hook); the survivor must fail instead of waiting
    for ever, and a relaunch with --resume continues from the last checkpoint to the step budget."""
    import socket

    def port():
        s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p

    def launch(extra):
        cluster = {"worker": ["localhost:%d" % port(), "localhost:%d" % port()]}
        common = ["train_dataset.py", "--synthetic", "--mode=sync_dp", "--checkpoint_dir=%s/ckpt/" % tmp_path,
                  "--summaries_dir=%s/events/" % tmp_path, "--train_steps=8", "--batch_size=4", "--image_size=32",
                  "--collective_timeout=20"]
        procs = []
        for i in range(2):
            env = dict(os.environ, TF_CONFIG=json.dumps({"cluster": cluster, "task": {"type": "worker", "index": i}}))
            procs.append(subprocess.Popen([sys.executable] + common + extra[i], cwd=DIST, env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        outs = [p.communicate(timeout=240)[0] for p in procs]
        return [p.returncode for p in procs], outs

    codes, outs = launch([[], ["--fault_inject=exit_at_step:3"]])
    assert codes[1] == 17 and codes[0] != 0, outs
    saved = int(bundle.read_bundle(bundle.latest_checkpoint(str(tmp_path / "ckpt")))["global_step"])
    assert 2 <= saved <= 4
    codes, outs = launch([["--resume"], ["--resume"]])
    assert codes == [0, 0], outs
    assert "resumed from global step %d" % saved in outs[0]
    assert int(bundle.read_bundle(bundle.latest_checkpoint(str(tmp_path / "ckpt")))["global_step"]) == 8


def test_run_sh_launches_ps_and_workers(tmp_path):
    """The local launcher starts one ps and N workers (the reference's run.sh started worker 0 twice and no ps,
    recognition_flower/distributed/run.sh:3-6) and returns when training is done."""
    import socket

    def port():
        s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p

    env = dict(os.environ, PS_PORT=str(port()), WORKER_PORT0=str(port()))
    r = subprocess.run(["bash", "run.sh", "2", "--synthetic", "--model=flower_cnn", "--image_size=32", "--batch_size=4",
                        "--train_steps=6", "--checkpoint_dir=%s/ckpt/" % tmp_path, "--summaries_dir=%s/events/" % tmp_path],
                       cwd=DIST, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
    assert r.returncode == 0, r.stdout
    assert r.stdout.count("finished training") == 2
    assert int(bundle.read_bundle(bundle.latest_checkpoint(str(tmp_path / "ckpt")))["global_step"]) >= 6


def test_evaluator_replica_follows_checkpoints(tmp_path):
    """A TFJob-style ``evaluator`` task next to a ps and a worker: it evaluates checkpoints as they appear, writes
    eval summaries at their global steps and ends once the final checkpoint (train_steps) has been evaluated."""
    import socket

    from distributed_tensorflow_amd.io import events

    def port():
        s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p

    cluster = {"ps": ["localhost:%d" % port()], "worker": ["localhost:%d" % port()], "evaluator": ["localhost:%d" % port()]}
    common = ["train_dataset.py", "--synthetic", "--model=flower_cnn", "--image_size=32", "--batch_size=4",
              "--train_steps=6", "--checkpoint_dir=%s/ckpt/" % tmp_path, "--summaries_dir=%s/events/" % tmp_path,
              "--eval_batches=2", "--eval_timeout=60", "--eval_poll_seconds=0.1"]
    procs = []
    for task in ({"type": "ps", "index": 0}, {"type": "worker", "index": 0}, {"type": "evaluator", "index": 0}):
        env = dict(os.environ, TF_CONFIG=json.dumps({"cluster": cluster, "task": task}))
        procs.append(subprocess.Popen([sys.executable] + common, cwd=DIST, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert [p.returncode for p in procs] == [0, 0, 0], outs
    steps = [int(m) for m in re.findall(r"Evaluator: global step (\d+) accuracy", outs[2])]
    assert steps and steps[-1] == 6 and steps == sorted(steps), outs[2]
    ev_files = glob.glob(str(tmp_path / "events" / "1" / "eval" / "events.out.tfevents.*"))
    assert len(ev_files) == 1
    evs = [e for e in events.read_events(ev_files[0]) if e.get("scalars")]
    assert [e["step"] for e in evs] == steps and {"accuracy", "loss"} <= set(evs[-1]["scalars"])
