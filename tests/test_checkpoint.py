"""Saver: TensorFlow variable names, layouts, round trips, cross-mode restore."""
import os

import numpy as np
import torch

from distributed_tensorflow_amd.io import bundle
from distributed_tensorflow_amd.io.saver import Saver, model_variables
from distributed_tensorflow_amd.models import build_model
from distributed_tensorflow_amd.ops import functional as F
from distributed_tensorflow_amd.parallel.engine import DataParallelEngine


def _train(model, engine, steps=3, seed=0, size=32):
    g = torch.Generator().manual_seed(seed)
    x = torch.zeros(8, size, size, 8)
    x[..., :3] = torch.randn(8, size, size, 3, generator=g)
    y = torch.randint(0, 5, (8,), generator=g)
    for _ in range(steps):
        engine.begin_step()
        out = model(x)
        logits, reg = out if isinstance(out, tuple) else (out, None)
        loss, _, _, _ = F.softmax_xent(logits, y, 5)
        (loss if reg is None else loss + reg).backward()
        engine.finish_step()
    return x, y


def test_inception_variable_names_match_reference_graph():
    model = build_model("inception_v3_head")
    names = {v.name: v for v in model_variables(model)}
    # head (recognition_flower/single/flower_train.py:72-84)
    assert names["fc_trainable/dense_prelogit1/kernel"].to_tf(
        names["fc_trainable/dense_prelogit1/kernel"].tensor).shape == (2048, 1024)
    assert names["fc_trainable/flower_logits/kernel"].to_tf(
        names["fc_trainable/flower_logits/kernel"].tensor).shape == (1024, 5)
    assert names["fc_trainable/flower_logits/bias"].to_tf(names["fc_trainable/flower_logits/bias"].tensor).shape == (5,)
    # slim trunk scopes, including the upstream naming quirks
    for key in ["InceptionV3/Conv2d_1a_3x3/weights", "InceptionV3/Conv2d_1a_3x3/BatchNorm/beta",
                "InceptionV3/Conv2d_1a_3x3/BatchNorm/moving_mean", "InceptionV3/Conv2d_1a_3x3/BatchNorm/moving_variance",
                "InceptionV3/Mixed_5c/Branch_1/Conv_1_0c_5x5/weights", "InceptionV3/Mixed_5c/Branch_1/Conv2d_0b_1x1/weights",
                "InceptionV3/Mixed_6a/Branch_0/Conv2d_1a_1x1/weights", "InceptionV3/Mixed_7b/Branch_1/Conv2d_0b_3x1/weights",
                "InceptionV3/Mixed_7c/Branch_1/Conv2d_0c_3x1/weights", "InceptionV3/Mixed_7c/Branch_2/Conv2d_0d_3x1/weights",
                "InceptionV3/Logits/Conv2d_1c_1x1/weights", "InceptionV3/Logits/Conv2d_1c_1x1/biases",
                "InceptionV3/AuxLogits/Conv2d_2b_1x1/weights", "InceptionV3/AuxLogits/Conv2d_2b_1x1/biases",
                "InceptionV3/AuxLogits/Conv2d_2a_5x5/BatchNorm/beta"]:
        assert key in names, key
    assert not any(k.endswith("BatchNorm/gamma") for k in names)  # slim arg scope: scale=False
    # HWIO layout with the 3 real input channels of the stem
    w = names["InceptionV3/Conv2d_1a_3x3/weights"]
    assert w.to_tf(w.tensor).shape == (3, 3, 3, 32)
    assert names["InceptionV3/Mixed_6b/Branch_1/Conv2d_0b_1x7/weights"].to_tf(
        names["InceptionV3/Mixed_6b/Branch_1/Conv2d_0b_1x7/weights"].tensor).shape == (1, 7, 128, 128)
    # 94 trunk convs + 2 aux convs with BN, (weights, beta, mean, var) each; 2 biased convs; 4 head tensors
    n_bn_convs = sum(1 for k in names if k.endswith("/BatchNorm/beta"))
    assert n_bn_convs == 96
    assert len(names) == 96 * 4 + 2 * 2 + 4
    total = sum(int(np.prod(v.to_tf(v.tensor).shape)) for v in names.values() if not v.name.startswith("fc_trainable"))
    assert 27_000_000 < total < 27_300_000  # ~27.2 M variables in slim's inception_v3(num_classes=1001)


def test_save_restore_roundtrip_with_adam_slots(tmp_path):
    torch.manual_seed(0)
    model = build_model("flower_cnn", widths=(8, 8, 16, 16, 16))
    eng = DataParallelEngine(model, mode="sync_dp", optimizer="adam", lr=1e-2)
    x, y = _train(model, eng, 3)
    saver = Saver(model, eng)
    prefix = saver.save(str(tmp_path), 3)
    keys = [e[0] for e in bundle.list_bundle(prefix)]
    assert "global_step" in keys and "train/beta1_power" in keys and "train/beta2_power" in keys
    assert "fc_trainable/dense_prelogit1/kernel/Adam" in keys and "fc_trainable/dense_prelogit1/kernel/Adam_1" in keys
    assert "FlowerCNN/conv1/weights/Adam" in keys and "FlowerCNN/conv1/BatchNorm/moving_mean" in keys
    data = bundle.read_bundle(prefix)
    assert data["global_step"].dtype == np.int64 and int(data["global_step"]) == 3
    np.testing.assert_allclose(data["train/beta1_power"], 0.9 ** 4, rtol=1e-6)
    assert data["FlowerCNN/conv1/weights"].shape == (3, 3, 3, 8)

    # continue training in the original, and in a restored copy: identical trajectories
    torch.manual_seed(1)
    model2 = build_model("flower_cnn", widths=(8, 8, 16, 16, 16))
    eng2 = DataParallelEngine(model2, mode="sync_dp", optimizer="adam", lr=1e-2)
    step = Saver(model2, eng2).restore(prefix, strict=True)
    assert step == 3 and eng2.optimizer.step_count == 3
    model.eval(); model2.eval()
    with torch.no_grad():
        torch.testing.assert_close(model(x)[0], model2(x)[0])
    model.train(); model2.train()
    for m in (model, model2):  # no dropout noise
        m.head.dropout_rate = 0.0
    _train(model, eng, 2, seed=5)
    _train(model2, eng2, 2, seed=5)
    for (n1, p1), (_, p2) in zip(model.named_parameters(), model2.named_parameters()):
        torch.testing.assert_close(p1, p2, msg=n1)


def test_restore_into_different_mode_and_momentum(tmp_path):
    torch.manual_seed(0)
    model = build_model("flower_cnn", widths=(8, 8, 16, 16, 16))
    eng = DataParallelEngine(model, mode="sharded_ps", optimizer="momentum", lr=0.05,
                             opt_kwargs={"momentum": 0.9, "weight_decay": 1e-4})
    _train(model, eng, 2)
    prefix = Saver(model, eng).save(str(tmp_path), 2)
    keys = [e[0] for e in bundle.list_bundle(prefix)]
    assert "FlowerCNN/conv2/weights/Momentum" in keys and "train/beta1_power" not in keys
    model2 = build_model("flower_cnn", widths=(8, 8, 16, 16, 16))
    eng2 = DataParallelEngine(model2, mode="sync_dp", optimizer="momentum", lr=0.05,
                              opt_kwargs={"momentum": 0.9, "weight_decay": 1e-4})
    Saver(model2, eng2).restore(prefix, strict=True)
    for (n1, p1), (_, p2) in zip(model.named_parameters(), model2.named_parameters()):
        torch.testing.assert_close(p1, p2, msg=n1)
    torch.testing.assert_close(eng.full_slots()["Momentum/fp32"], eng2.optimizer.slots()["Momentum/fp32"])


def test_restore_accepts_unnamed_step_variable(tmp_path):
    model = build_model("flower_cnn", widths=(8, 8, 16, 16, 16))
    tensors = Saver(model).variables()
    tensors["Variable"] = np.array(41, dtype=np.int32)  # recognition_flower/single/flower_train.py:114
    prefix = os.path.join(str(tmp_path), "my_flower_model")
    bundle.write_bundle(prefix, tensors)
    assert Saver(build_model("flower_cnn", widths=(8, 8, 16, 16, 16))).restore(prefix, strict=True) == 41


def test_model_only_checkpoint_is_subset_restore(tmp_path):
    """save_model.py restores only model variables from a training checkpoint (a superset)."""
    torch.manual_seed(0)
    model = build_model("flower_cnn", widths=(8, 8, 16, 16, 16))
    eng = DataParallelEngine(model, mode="sync_dp", optimizer="adam", lr=1e-2)
    x, _ = _train(model, eng, 2)
    prefix = Saver(model, eng).save(str(tmp_path), 2)
    fresh = build_model("flower_cnn", widths=(8, 8, 16, 16, 16))
    Saver(fresh).restore(prefix, strict=True, restore_slots=False)
    model.eval(); fresh.eval()
    with torch.no_grad():
        torch.testing.assert_close(model(x)[0], fresh(x)[0])


def test_meta_file_lists_the_saved_variables(tmp_path):
    """Every checkpoint comes with a ``.meta`` (MetaGraphDef wire format): SaverDef V2 and the variable
    collections, so the file set is the one ``tf.train.Saver`` leaves behind."""
    from distributed_tensorflow_amd.io.metagraph import decode_meta_graph

    torch.manual_seed(0)
    model = build_model("flower_cnn", widths=(8, 8, 16, 16, 32))
    eng = DataParallelEngine(model, mode="sync_dp", optimizer="adam", lr=1e-3)
    saver = Saver(model, eng)
    prefix = saver.save(str(tmp_path), 7)
    files = sorted(os.listdir(tmp_path))
    assert files == ["checkpoint", "model.ckpt-7.data-00000-of-00001", "model.ckpt-7.index", "model.ckpt-7.meta"]
    meta = decode_meta_graph(open(prefix + ".meta", "rb").read())
    assert meta["tags"] == ["train"] and meta["saver"]["version"] == 2 and meta["saver"]["max_to_keep"] == 5
    assert meta["saver"]["restore_op_name"] == "save/restore_all"
    names = {n for n, _ in meta["collections"]["variables"]}
    keys = {k + ":0" for k in bundle.read_bundle(prefix)}
    assert names == keys
    trainable = {n for n, _ in meta["collections"]["trainable_variables"]}
    assert "fc_trainable/flower_logits/kernel:0" in trainable and "FlowerCNN/conv1/BatchNorm/moving_mean:0" not in trainable
    assert "fc_trainable/flower_logits/kernel/Adam:0" in names and "fc_trainable/flower_logits/kernel/Adam:0" not in trainable
    assert meta["collections"]["global_step"] == [("global_step:0", False)]
