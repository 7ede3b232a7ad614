"""The reference entry points on an MI355X: device JPEG decode + resize, HIP kernels, step graph,
checkpoint, export and serving on the GPU."""
import glob
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from distributed_tensorflow_amd.io import bundle
from distributed_tensorflow_amd.io.export import Servable, example_from_jpeg
from test_data import _make_dataset

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SINGLE = os.path.join(ROOT, "recognition_flower", "single")
DIST = os.path.join(ROOT, "recognition_flower", "distributed")


def _run(cmd, cwd, timeout=300):
    e = dict(os.environ)
    e.pop("TF_CONFIG", None)
    return subprocess.run([sys.executable] + cmd, cwd=cwd, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                          text=True, timeout=timeout)


def test_single_trainer_inception_head_on_gpu(tmp_path):
    """The reference's own configuration: Inception-v3 trunk (frozen) + flower head, Adam, exponential
    decay, 299x299 -- trained from JPEG files, exported, and served from the GPU."""
    data = _make_dataset(str(tmp_path / "flowers"), n_per_class=12)
    out = tmp_path / "out"
    r = _run(["flower_train.py", "--dataset_path=" + data, "--n_epochs=6", "--batch_size=8", "--device=cuda",
              "--output_dir=%s/models" % out, "--summaries_dir=%s/events/" % out, "--checkpoint_dir=%s/ckpt" % out], SINGLE)
    assert r.returncode == 0, r.stdout[-3000:]
    losses = [float(m) for m in re.findall(r"loss (\d+\.\d+)", r.stdout)]
    assert len(losses) >= 12 and all(np.isfinite(losses))
    assert np.mean(losses[-4:]) < np.mean(losses[:4])
    keys = {e[0] for e in bundle.list_bundle(bundle.latest_checkpoint(str(out / "ckpt")))}
    assert {"InceptionV3/Conv2d_1a_3x3/weights", "InceptionV3/Mixed_7c/Branch_1/Conv2d_0c_3x1/BatchNorm/beta",
            "InceptionV3/AuxLogits/Conv2d_2b_1x1/biases", "InceptionV3/Logits/Conv2d_1c_1x1/weights",
            "fc_trainable/dense_prelogit1/kernel", "fc_trainable/flower_logits/kernel/Adam", "global_step"} <= keys
    export = out / "models" / "1"
    s = Servable(str(export), device="cuda")
    jpegs = [open(f, "rb").read() for f in sorted(glob.glob(os.path.join(data, "*", "*.jpg")))[:5]]
    a = s.predict_images(jpegs)
    b = s.serving_default([example_from_jpeg(j) for j in jpegs])
    assert a["classes"].shape == (5, 1) and a["scores"].dtype == np.float32
    assert a["classes"].tolist() == b["classes"].tolist() and np.allclose(a["scores"], b["scores"])
    # the same servable on the CPU reference path agrees on the scores up to bf16 noise
    c = Servable(str(export), device="cpu").predict_images(jpegs)
    assert np.abs(c["scores"] - a["scores"]).max() < 0.15 * max(1.0, np.abs(c["scores"]).max())


def test_collective_trainer_checkpoint_resume_export_on_gpu(tmp_path):
    data = _make_dataset(str(tmp_path / "flowers"), n_per_class=8)
    rec = str(tmp_path / "train.tfrecord")
    r = _run(["prepare.py", "--dataset_path=" + data, "--tfrecords_path=" + rec, "--seed=3"], DIST)
    assert r.returncode == 0, r.stdout
    common = ["train_dataset.py", "--tfrecords_path=" + rec, "--checkpoint_dir=%s/ckpt/" % tmp_path,
              "--summaries_dir=%s/events/" % tmp_path, "--batch_size=8", "--image_size=64", "--model=flower_cnn",
              "--device=cuda", "--learning_rate=0.003"]
    r1 = _run(common + ["--mode=sharded_ps", "--train_steps=12", "--hip_graph=true"], DIST)
    assert r1.returncode == 0, r1.stdout[-3000:]
    latest, _ = bundle.read_checkpoint_state(str(tmp_path / "ckpt"))
    assert latest == "model.ckpt-12"
    # resume in a different mode from the same files; the global step continues
    r2 = _run(common + ["--mode=sync_dp", "--train_steps=18", "--resume=true"], DIST)
    assert r2.returncode == 0 and "resumed from global step 12" in r2.stdout, r2.stdout[-3000:]
    assert bundle.read_checkpoint_state(str(tmp_path / "ckpt"))[0] == "model.ckpt-18"
    r3 = _run(["save_model.py", "--checkpoint_dir=%s/ckpt/" % tmp_path, "--output_dir=%s/models" % tmp_path,
               "--dataset_path=" + data, "--model=flower_cnn", "--image_size=64", "--device=cuda"], DIST)
    assert r3.returncode == 0 and "Done exporting!" in r3.stdout, r3.stdout[-3000:]
    manifest = json.load(open(tmp_path / "models" / "1" / "saved_model.json"))
    assert manifest["classes"] == ["daisy", "roses", "tulips"] and manifest["global_step"] == 18
    s = Servable(str(tmp_path / "models" / "1"), device="cuda")
    jpegs = [open(f, "rb").read() for f in sorted(glob.glob(os.path.join(data, "*", "*.jpg")))[:3]]
    out = s.predict_images(jpegs)
    assert out["classes"].shape == (3, 1) and np.isfinite(out["scores"]).all()


def test_ps_worker_on_gpu_via_tf_config(tmp_path):
    """TF_CONFIG roles with the worker on the GPU: one ps task (CPU) + one worker whose device step is replayed
    from a HIP graph (--hip_graph=auto), checkpoint written by the chief, ps stopped at the end."""
    import socket

    def port():
        s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p

    cluster = {"ps": ["localhost:%d" % port()], "worker": ["localhost:%d" % port()]}
    common = ["train_dataset.py", "--synthetic", "--model=flower_cnn", "--image_size=64", "--batch_size=8",
              "--train_steps=10", "--checkpoint_dir=%s/ckpt/" % tmp_path, "--summaries_dir=%s/events/" % tmp_path]
    procs = []
    for task, dev in (({"type": "ps", "index": 0}, "cpu"), ({"type": "worker", "index": 0}, "cuda")):
        env = dict(os.environ, TF_CONFIG=json.dumps({"cluster": cluster, "task": task}))
        procs.append(subprocess.Popen([sys.executable] + common + ["--device=" + dev], cwd=DIST, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    assert "finished training (worker 0" in outs[1]
    assert int(bundle.read_bundle(bundle.latest_checkpoint(str(tmp_path / "ckpt")))["global_step"]) == 10


@pytest.mark.gpu
def test_model_server_on_gpu_matches_servable(tmp_path):
    """REST server over the HIP inference path: batched answers equal the servable called directly."""
    import base64
    import io
    import json
    import urllib.request

    import numpy as np
    import torch
    from PIL import Image

    from distributed_tensorflow_amd.io.export import Servable, export_servable
    from distributed_tensorflow_amd.models import build_model
    from distributed_tensorflow_amd.serving import ModelManager, ModelServer

    classes = ["daisy", "dandelion", "roses", "sunflowers", "tulips"]
    torch.manual_seed(0)
    model = build_model("flower_cnn", num_classes=5, device=torch.device("cuda"))
    export_servable(model, classes, str(tmp_path / "m"), 1, model_name="flower_cnn", image_size=64)
    rng = np.random.default_rng(0)
    jpegs = []
    for i in range(10):
        b = io.BytesIO()
        Image.fromarray(rng.integers(0, 255, (80 + 8 * i, 96, 3), dtype=np.uint8)).save(b, format="JPEG", quality=90)
        jpegs.append(b.getvalue())
    ref = Servable(str(tmp_path / "m" / "1"), device="cuda", hip_graphs=False).predict_images(jpegs)
    # the graph-replayed forward (bucket 16 for 10 images, then bucket 4 twice) equals the eager one
    graphed = Servable(str(tmp_path / "m" / "1"), device="cuda", hip_graphs=True)
    for chunk in (jpegs, jpegs[:3], jpegs[3:6]):
        a, b = graphed.predict_images(chunk), Servable.predict_images(graphed, chunk)
        assert a["classes"].tolist() == b["classes"].tolist() and np.array_equal(a["scores"], b["scores"])
    got = graphed.predict_images(jpegs)
    assert sorted(graphed._graphs) == [4, 16]
    assert got["classes"].tolist() == ref["classes"].tolist() and np.allclose(got["scores"], ref["scores"], atol=2e-2)
    manager = ModelManager(str(tmp_path / "m"), "flowers", device="cuda", poll_seconds=0).start()
    server = ModelServer(manager, port=0, max_batch=4, batch_timeout_ms=5.0).start()
    try:
        body = json.dumps({"instances": [{"b64": base64.b64encode(j).decode()} for j in jpegs]}).encode()
        with urllib.request.urlopen("http://127.0.0.1:%d/v1/models/flowers:predict" % server.port, body, timeout=120) as r:
            out = json.loads(r.read())
        assert [p["classes"] for p in out["predictions"]] == ref["classes"].tolist()
        assert np.allclose([p["scores"] for p in out["predictions"]], ref["scores"], atol=2e-2)
        assert server.batcher.stats()["batches"] == 3  # 10 images in chunks of max_batch = 4
    finally:
        server.stop()

def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.gpu
def test_xgmi_parameter_server_via_tf_config_roles(tmp_path):
    """The reference's launch (TF_CONFIG roles: one ps task, two workers; run.sh) with --ps_backend=xgmi: the ps task
    owns the variables and their Adam slots in GPU memory and applies what the workers push into its mailboxes, the
    workers are clients of that memory (here all on the one GPU of the box), the learning rate decays on the ps
    task's own step count, the chief writes the checkpoint and stops the ps task when the step budget is used up."""
    cluster = {"ps": ["localhost:%d" % _free_port()], "worker": ["localhost:%d" % _free_port(), "localhost:%d" % _free_port()]}
    common = ["train_dataset.py", "--synthetic", "--model=flower_cnn", "--image_size=64", "--batch_size=8",
              "--train_steps=12", "--mode=async_ps", "--ps_backend=xgmi", "--staleness=1", "--device=cuda",
              "--lr_decay_steps=4", "--checkpoint_dir=%s/ckpt/" % tmp_path, "--summaries_dir=%s/events/" % tmp_path]
    procs = []
    for task in ({"type": "ps", "index": 0}, {"type": "worker", "index": 0}, {"type": "worker", "index": 1}):
        env = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}   # the entry point sets it
        env["TF_CONFIG"] = json.dumps({"cluster": cluster, "task": task})
        procs.append(subprocess.Popen([sys.executable] + common, cwd=DIST, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert [p.returncode for p in procs] == [0, 0, 0], outs
    assert "owning shard 0 of 1" in outs[0] and "stopped after" in outs[0]
    applied = int(outs[0].split("stopped after")[1].split()[0])
    assert applied >= 12
    assert "finished training (worker 0" in outs[1] and "finished training (worker 1" in outs[2]
    assert int(bundle.read_bundle(bundle.latest_checkpoint(str(tmp_path / "ckpt")))["global_step"]) >= 12


@pytest.mark.gpu
def test_xgmi_parameter_server_without_ps_tasks_resumes(tmp_path):
    """TF_CONFIG with workers only and --ps_backend=xgmi: every worker GPU owns half of the variables; a second run
    with --resume continues from the first run's checkpoint."""
    def run(steps, resume):
        cluster = {"worker": ["localhost:%d" % _free_port(), "localhost:%d" % _free_port()]}
        common = ["train_dataset.py", "--synthetic", "--model=flower_cnn", "--image_size=64", "--batch_size=8",
                  "--train_steps=%d" % steps, "--mode=async_ps", "--ps_backend=xgmi", "--staleness=0", "--device=cuda",
                  "--checkpoint_dir=%s/ckpt/" % tmp_path, "--summaries_dir=%s/events/" % tmp_path] + (["--resume"] if resume else [])
        procs = []
        for i in range(2):
            env = dict(os.environ, TF_CONFIG=json.dumps({"cluster": cluster, "task": {"type": "worker", "index": i}}))
            procs.append(subprocess.Popen([sys.executable] + common, cwd=DIST, env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        outs = [p.communicate(timeout=240)[0] for p in procs]
        assert [p.returncode for p in procs] == [0, 0], outs
        return outs

    run(6, False)
    assert int(bundle.read_bundle(bundle.latest_checkpoint(str(tmp_path / "ckpt")))["global_step"]) >= 6
    outs = run(10, True)
    assert "resumed from global step" in outs[0]
    assert int(bundle.read_bundle(bundle.latest_checkpoint(str(tmp_path / "ckpt")))["global_step"]) >= 10
