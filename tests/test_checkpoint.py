"""Saver: TensorFlow variable names, layouts, round trips, cross-mode restore."""
import os

import numpy as np
import pytest
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


def test_saved_model_pb_carries_tags_saver_collections_and_signatures(tmp_path):
    """``saved_model.pb`` of an export: schema version 1, one meta graph tagged ``serve``, the two signatures of
    the reference (recognition_flower/single/flower_train.py:207-249) with their TensorInfo, ``legacy_init_op``.
    The hand-written wire encoding is cross-checked by parsing the same bytes with protobuf-runtime message
    classes declared from the public saved_model / meta_graph schemas (field numbers written out below)."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

    from distributed_tensorflow_amd.io.export import export_servable, read_signatures, signature_manifest
    from distributed_tensorflow_amd.io.metagraph import decode_saved_model

    model = build_model("flower_cnn", num_classes=3, device=torch.device("cpu"))
    path = export_servable(model, ["a", "b", "c"], str(tmp_path), 4, model_name="flower_cnn", image_size=32)
    raw = open(os.path.join(path, "saved_model.pb"), "rb").read()
    sm = decode_saved_model(raw)
    assert sm["schema_version"] == 1 and sm["tags"] == ["serve"] and sm["saver"]["version"] == 2
    assert sm["node_lists"] == {"legacy_init_op": ["legacy_init_op"]}
    keys = {k for k, *_ in bundle.list_bundle(os.path.join(path, "variables", "variables"))}
    assert {n[:-2] for n, _ in sm["collections"]["variables"]} == keys
    want = signature_manifest()
    got = read_signatures(path)
    assert set(got) == {"serving_default", "predict_images"}
    for key in want:
        assert got[key]["method_name"] == want[key]["method_name"]
        for side in ("inputs", "outputs"):
            assert {n: (d["tensor"], d["dtype"], d["shape"]) for n, d in got[key][side].items()} == \
                   {n: (d["tensor"], d["dtype"], d["shape"]) for n, d in want[key][side].items()}

    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name, fd.package, fd.syntax = "dtfa_test/saved_model.proto", "dtfa_test", "proto3"

    def msg(name, fields, maps=()):
        m = fd.message_type.add()
        m.name = name
        for fname, num, typ, rep in fields:
            f = m.field.add()
            f.name, f.number, f.label = fname, num, F.LABEL_REPEATED if rep else F.LABEL_OPTIONAL
            if typ.startswith("."):
                f.type, f.type_name = F.TYPE_MESSAGE, typ
            else:
                f.type = getattr(F, "TYPE_" + typ.upper())
        for fname, num, vtyp in maps:
            e = m.nested_type.add()
            e.name = "".join(p.capitalize() for p in fname.split("_")) + "Entry"
            e.options.map_entry = True
            k = e.field.add()
            k.name, k.number, k.label, k.type = "key", 1, F.LABEL_OPTIONAL, F.TYPE_STRING
            v = e.field.add()
            v.name, v.number, v.label, v.type, v.type_name = "value", 2, F.LABEL_OPTIONAL, F.TYPE_MESSAGE, vtyp
            f = m.field.add()
            f.name, f.number, f.label, f.type = fname, num, F.LABEL_REPEATED, F.TYPE_MESSAGE
            f.type_name = ".dtfa_test.%s.%s" % (name, e.name)

    msg("Dim", [("size", 1, "int64", False)])
    msg("Shape", [("dim", 2, ".dtfa_test.Dim", True), ("unknown_rank", 3, "bool", False)])
    msg("TensorInfo", [("name", 1, "string", False), ("dtype", 2, "int32", False), ("tensor_shape", 3, ".dtfa_test.Shape", False)])
    msg("SignatureDef", [("method_name", 3, "string", False)],
        maps=[("inputs", 1, ".dtfa_test.TensorInfo"), ("outputs", 2, ".dtfa_test.TensorInfo")])
    msg("MetaInfoDef", [("tags", 4, "string", True)])
    msg("SaverDef", [("filename_tensor_name", 1, "string", False), ("sharded", 5, "bool", False), ("version", 7, "int32", False)])
    msg("MetaGraphDef", [("meta_info_def", 1, ".dtfa_test.MetaInfoDef", False), ("saver_def", 3, ".dtfa_test.SaverDef", False)],
        maps=[("signature_def", 5, ".dtfa_test.SignatureDef")])
    msg("SavedModel", [("saved_model_schema_version", 1, "int64", False), ("meta_graphs", 2, ".dtfa_test.MetaGraphDef", True)])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    SavedModel = message_factory.GetMessageClass(pool.FindMessageTypeByName("dtfa_test.SavedModel"))
    pm = SavedModel.FromString(raw)
    assert pm.saved_model_schema_version == 1 and len(pm.meta_graphs) == 1
    mg = pm.meta_graphs[0]
    assert list(mg.meta_info_def.tags) == ["serve"] and mg.saver_def.version == 2 and mg.saver_def.sharded
    sd = mg.signature_def["serving_default"]
    assert sd.method_name == "tensorflow/serving/classify"
    assert sd.inputs["inputs"].name == "tf_example:0" and sd.inputs["inputs"].dtype == 7
    assert sd.inputs["inputs"].tensor_shape.unknown_rank and len(sd.inputs["inputs"].tensor_shape.dim) == 0
    pi = mg.signature_def["predict_images"]
    assert pi.method_name == "tensorflow/serving/predict" and [d.size for d in pi.inputs["images"].tensor_shape.dim] == [-1]
    assert pi.outputs["scores"].dtype == 1 and [d.size for d in pi.outputs["scores"].tensor_shape.dim] == [-1, 1]
    assert pi.outputs["classes"].dtype == 7


def test_trunk_bn_moving_keeps_trunk_normalisation_frozen():
    """``trunk_bn="moving"``: train() leaves the trunk's conv+bn layers on their moving statistics (and does not
    update them), PreLogits dropout and the head still follow the mode; the variable set is unchanged."""
    from distributed_tensorflow_amd.models.layers import ConvBnAct

    m = build_model("inception_v3_head", trunk_bn="moving").train()
    layers = [x for x in m.trunk.modules() if isinstance(x, ConvBnAct)]
    assert len(layers) == 96 and not any(x.training for x in layers)  # 94 executed + 2 AuxLogits (variables only)
    assert m.trunk.training and m.head.training
    m.eval()
    assert not m.trunk.training and not any(x.training for x in layers)
    ref_names = {v.name for v in model_variables(build_model("inception_v3_head"))}
    assert {v.name for v in model_variables(m)} == ref_names
    with pytest.raises(ValueError):
        build_model("inception_v3_head", trunk_bn="frozen")


def test_background_save_snapshots_now_and_caches_frozen_variables(tmp_path):
    """save(background=True): the files hold the values at the time of the call although training goes on while the
    helper thread writes; frozen parameters are converted once and re-converted after a restore."""
    model = build_model("inception_v3_head")
    eng = DataParallelEngine(model, mode="sync_dp", optimizer="adam", lr=1e-2)
    saver = Saver(model, eng)
    before = {g.name: g.master.clone() for g in eng.flat.groups}
    saver.save(str(tmp_path), 1, background=True)
    for g in eng.flat.groups:           # "training" continues: the live master changes under the writer
        g.master.add_(1.0)
    saver.wait()
    data = bundle.read_bundle(bundle.latest_checkpoint(str(tmp_path)))
    k = data["fc_trainable/flower_logits/kernel"]
    assert k.shape == (1024, 5)
    want = model.head.flower_logits.weight.detach()[:5].t() - 1.0   # parameters view the (moved) master
    np.testing.assert_allclose(k, want.numpy(), rtol=0, atol=1e-6)   # (x + 1) - 1 rounds; the moved values are 1 off
    assert int(data["global_step"]) == 1
    # frozen trunk: same array object on the next save, a fresh one after the parameter is rewritten
    name = "InceptionV3/Conv2d_1a_3x3/weights"
    first = saver.variables()[name]
    assert saver.variables()[name] is first and not first.flags.writeable
    with torch.no_grad():
        model.trunk.Conv2d_1a_3x3.weight.mul_(2.0)
    second = saver.variables()[name]
    assert second is not first and np.allclose(second, first * 2)
    # saves queue up in step order and restore joins the writer
    saver.save(str(tmp_path), 2, background=True)
    saver.save(str(tmp_path), 3, background=True)
    assert saver.restore_latest(str(tmp_path)) == 3
    assert bundle.read_checkpoint_state(str(tmp_path))[0] == "model.ckpt-3"
    del before
