"""Flags (tf.app.flags work-alike) and TF_CONFIG cluster specification."""
import json

import pytest

from distributed_tensorflow_amd.utils import cluster, flags


def _fv():
    fv = flags.FlagValues()
    flags.DEFINE_string("output_dir", "../output/models", "d", fv)
    flags.DEFINE_integer("model_version", 1, "v", fv)
    flags.DEFINE_float("learning_rate", 0.001, "lr", fv)
    flags.DEFINE_boolean("resume", False, "r", fv)
    return fv


def test_flag_defaults_and_syntaxes():
    fv = _fv()
    rest = fv.parse(["--output_dir=/x", "--model_version", "7", "--learning_rate=0.5", "--resume", "pos", "--unknown=1"])
    assert fv.output_dir == "/x" and fv.model_version == 7 and fv.learning_rate == 0.5 and fv.resume is True
    assert rest == ["pos", "--unknown=1"]
    fv2 = _fv()
    fv2.parse(["--noresume", "-model_version=3"])
    assert fv2.resume is False and fv2.model_version == 3 and fv2.output_dir == "../output/models"
    fv3 = _fv()
    fv3.parse(["--resume=false"])
    assert fv3.resume is False


def test_flag_errors():
    fv = _fv()
    with pytest.raises(ValueError):
        fv.parse(["--model_version=abc"])
    with pytest.raises(ValueError):
        fv.parse(["--model_version"])
    with pytest.raises(AttributeError):
        _ = fv.nonexistent
    with pytest.raises(ValueError):
        flags.DEFINE_integer("model_version", 2, "dup", fv)


def test_lazy_parse_on_first_access(monkeypatch):
    # the reference reads FLAGS at import time, before tf.app.run() parses argv
    fv = _fv()
    monkeypatch.setattr("sys.argv", ["prog", "--model_version=9"])
    assert fv.model_version == 9


def test_reference_flag_names_and_defaults():
    """Flag surface of the reference scripts (recognition_flower/single/flower_train.py:10-31,
    distributed/train_dataset.py:10-31, distributed/save_model.py:24-34, distributed/prepare.py:10-14)."""
    import importlib
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    expected = {
        "recognition_flower/single/flower_train.py": {
            "output_dir": "../output/models", "model_version": 1, "image_size": 299, "n_epochs": 10,
            "dataset_path": "../flower_photos_labeled", "model_path": "", "summaries_dir": "../output/events/"},
        "recognition_flower/distributed/train_dataset.py": {
            "model_version": 1, "n_epochs": 1000, "tfrecords_path": "./train_v1.tfrecord",
            "checkpoint_dir": "./output/checkpoint_dir/", "summaries_dir": "./output/events/", "train_steps": 10,
            "batch_size": 32, "learning_rate": 0.001},
        "recognition_flower/distributed/save_model.py": {
            "output_dir": "./output/models", "model_version": 1, "dataset_path": "../flower_photos_labeled",
            "checkpoint_dir": "./output/checkpoint_dir/"},
        "recognition_flower/distributed/prepare.py": {
            "dataset_path": "../flower_photos_labeled", "tfrecords_path": "./train_v1.tfrecord"},
    }
    import re

    for rel, want in expected.items():
        src = open(os.path.join(root, rel)).read()
        for name, default in want.items():
            m = re.search(r'flags\.DEFINE_\w+\(\s*"%s",\s*([^,]+),' % name, src)
            assert m, (rel, name)
            assert eval(m.group(1)) == default, (rel, name, m.group(1))


def test_tf_config_parsing():
    cfg = {"cluster": {"ps": ["localhost:2223"], "worker": ["localhost:2222", "10.0.0.2:2224"]},
           "task": {"type": "worker", "index": 1}}
    spec = cluster.parse_tf_config(json.dumps(cfg))
    assert spec.task_type == "worker" and spec.task_index == 1 and not spec.is_chief
    assert spec.num_workers == 2 and spec.ps_hosts == ["localhost:2223"]
    assert spec.address("worker", 1) == ("10.0.0.2", 2224) and spec.my_address() == ("10.0.0.2", 2224)
    chief = cluster.parse_tf_config(json.dumps({**cfg, "task": {"type": "worker", "index": 0}}))
    assert chief.is_chief
    ps = cluster.parse_tf_config(json.dumps({**cfg, "task": {"type": "ps", "index": 0}}))
    assert ps.task_type == "ps" and not ps.is_chief  # unlike the reference, ps 0 is not a chief
    empty = cluster.parse_tf_config("")
    assert empty.task_type is None and empty.is_chief and empty.num_workers == 1
    with pytest.raises(ValueError):
        cluster.parse_tf_config(json.dumps({**cfg, "task": {"type": "worker", "index": 5}}))
    env = cluster.rendezvous_from_cluster(spec)
    assert env["RANK"] == "1" and env["WORLD_SIZE"] == "2" and env["MASTER_ADDR"] == "127.0.0.1"


def test_tf_config_from_env(monkeypatch):
    monkeypatch.setenv("TF_CONFIG", '{"cluster":{"ps":["a:1"],"worker":["b:2"]},"task":{"index":0,"type":"ps"}}')
    assert cluster.parse_tf_config().task_type == "ps"
    monkeypatch.delenv("TF_CONFIG")
    assert cluster.parse_tf_config().task_type is None


def test_tfjob_style_cluster_with_chief_replica():
    """Kubeflow TFJob writes a separate chief replica: it is trainer 0, the workers follow."""
    from distributed_tensorflow_amd.utils.cluster import parse_tf_config, rendezvous_from_cluster

    cluster = {"chief": ["c0:2222"], "worker": ["w0:2222", "w1:2222"], "ps": ["p0:2222"]}
    chief = parse_tf_config(json.dumps({"cluster": cluster, "task": {"type": "chief", "index": 0}}))
    w1 = parse_tf_config(json.dumps({"cluster": cluster, "task": {"type": "worker", "index": 1}}))
    ps = parse_tf_config(json.dumps({"cluster": cluster, "task": {"type": "ps", "index": 0}}))
    assert chief.is_chief and chief.trainer_index == 0 and chief.num_workers == 3
    assert not w1.is_chief and w1.trainer_index == 2 and w1.num_workers == 3
    assert not ps.is_chief and not ps.is_trainer
    env = rendezvous_from_cluster(w1)
    assert env == {"MASTER_ADDR": "c0", "MASTER_PORT": "3222", "RANK": "2", "WORLD_SIZE": "3"}
    # the reference's own form: no chief job, worker 0 is the chief
    plain = parse_tf_config(json.dumps({"cluster": {"worker": ["a:1", "b:2"], "ps": ["p:3"]},
                                        "task": {"type": "worker", "index": 0}}))
    assert plain.is_chief and plain.trainer_index == 0 and plain.num_workers == 2
    assert rendezvous_from_cluster(plain)["MASTER_PORT"] == "1001"


def test_tf_config_workers_spread_over_the_visible_gpus():
    """Several TF_CONFIG workers on one machine: worker i trains on GPU i modulo the device count; with one (or no)
    visible device there is nothing to choose."""
    from distributed_tensorflow_amd.utils.cluster import local_device_index

    assert [local_device_index(i, 8) for i in range(10)] == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1]
    assert local_device_index(3, 1) is None and local_device_index(3, 0) is None
