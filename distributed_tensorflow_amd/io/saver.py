"""``tf.train.Saver`` work-alike: TensorFlow variable names <-> model tensors.

Checkpoints written here are Saver-V2 bundles whose keys are the variable
names the reference's graphs create:

* trunk: ``InceptionV3/<scope>/weights`` (HWIO), ``.../BatchNorm/beta``,
  ``.../BatchNorm/moving_mean``, ``.../BatchNorm/moving_variance``
  (slim arg scope; recognition_flower/single/flower_train.py:67-68);
* head: ``fc_trainable/dense_prelogit1/{kernel,bias}``,
  ``fc_trainable/flower_logits/{kernel,bias}`` (``[in, out]`` kernels;
  recognition_flower/single/flower_train.py:72-84);
* Adam slots ``<var>/Adam``, ``<var>/Adam_1`` and the ``beta1_power`` /
  ``beta2_power`` accumulators, which TensorFlow creates under the ``train``
  name scope of ``optimizer.minimize`` (recognition_flower/single/flower_train.py:140-152);
  momentum slots ``<var>/Momentum``;
* ``global_step`` (int64; recognition_flower/distributed/train_dataset.py:150).  The
  single-process reference script uses an unnamed ``tf.Variable(0)`` whose key is
  ``Variable`` (recognition_flower/single/flower_train.py:114); that key is accepted
  on restore.

Tensor layouts are converted at this boundary: conv weights ``[Cout,KH,KW,Cin]``
-> ``[KH,KW,Cin,Cout]`` (dropping the zero input channels of padded stems),
dense weights ``[Out_padded, In]`` -> ``[In, Out]``.  Because keys and shapes are
logical (never sharded), a checkpoint restores into any world size or
parallel mode.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import bundle


@dataclass
class VarSpec:
    name: str                      # TensorFlow variable name
    tensor: torch.Tensor           # parameter or buffer (compute copy)
    to_tf: Callable[[torch.Tensor], torch.Tensor]
    from_tf: Callable[[torch.Tensor, torch.Tensor], torch.Tensor]  # (tf value, current tensor) -> new tensor value
    trainable: bool


def _identity_spec(name, t, trainable):
    return VarSpec(name, t, lambda x: x, lambda v, cur: v.reshape(cur.shape), trainable)


def model_variables(model: nn.Module) -> List[VarSpec]:
    """Every checkpointed variable of ``model`` with its TensorFlow name."""
    from ..models.layers import Conv2d, ConvBnAct, Dense

    out: List[VarSpec] = []
    for mod in model.modules():
        scope = getattr(mod, "tf_scope", None)
        if scope is None:
            continue
        if isinstance(mod, (ConvBnAct, Conv2d)):
            cin = getattr(mod, "cin_logical", mod.weight.shape[3])

            def w_to_tf(x, cin=cin):
                return x[..., :cin].permute(1, 2, 3, 0).contiguous()

            def w_from_tf(v, cur, cin=cin):
                new = torch.zeros_like(cur, dtype=torch.float32)
                new[..., :cin] = v.permute(3, 0, 1, 2)
                return new

            out.append(VarSpec(scope + "/weights", mod.weight, w_to_tf, w_from_tf, mod.weight.requires_grad))
            if isinstance(mod, ConvBnAct):
                bn = scope + "/BatchNorm"
                if mod.gamma is not None:
                    out.append(_identity_spec(bn + "/gamma", mod.gamma, mod.gamma.requires_grad))
                out.append(_identity_spec(bn + "/beta", mod.beta, mod.beta.requires_grad))
                out.append(_identity_spec(bn + "/moving_mean", mod.running_mean, False))
                out.append(_identity_spec(bn + "/moving_variance", mod.running_var, False))
            elif mod.bias is not None:
                out.append(_identity_spec(scope + "/biases", mod.bias, mod.bias.requires_grad))
        elif isinstance(mod, Dense):
            n_out = mod.out_features

            def k_to_tf(x, n_out=n_out):
                return x[:n_out].t().contiguous()

            def k_from_tf(v, cur, n_out=n_out):
                new = torch.zeros_like(cur, dtype=torch.float32)
                new[:n_out] = v.t()
                return new

            def b_to_tf(x, n_out=n_out):
                return x[:n_out].contiguous()

            def b_from_tf(v, cur, n_out=n_out):
                new = torch.zeros_like(cur, dtype=torch.float32)
                new[:n_out] = v
                return new

            out.append(VarSpec(scope + "/kernel", mod.weight, k_to_tf, k_from_tf, mod.weight.requires_grad))
            out.append(VarSpec(scope + "/bias", mod.bias, b_to_tf, b_from_tf, mod.bias.requires_grad))
    names = [v.name for v in out]
    if len(set(names)) != len(names):
        raise ValueError("duplicate TensorFlow variable names in model")
    return out


SLOT_SUFFIX = {"Adam": "Adam", "Adam_1": "Adam_1", "Momentum": "Momentum"}


class Saver:
    """Save / restore model variables, optimizer slots and the global step.

    ``engine`` (a :class:`parallel.engine.DataParallelEngine`) is optional;
    with it, fp32 master weights and optimizer slots are saved (gathered from
    their owners in sharded mode); without it only the model tensors are.
    """

    def __init__(self, model: nn.Module, engine=None, max_to_keep: int = 5, basename: str = "model.ckpt",
                 power_scope: str = "train/"):
        self.model, self.engine = model, engine
        self.max_to_keep, self.basename, self.power_scope = max_to_keep, basename, power_scope
        self.specs = model_variables(model)
        # TensorFlow-layout copies of frozen parameters (the Inception trunk: 22 M of the flower model's 24 M
        # values), reused from save to save until the parameter is written or replaced
        self._frozen: Dict[str, tuple] = {}
        self._writer = None       # background file writer (save(background=True))
        self._write_future = None

    # ------------------------------------------------------------------ collect
    def _segment_views(self):
        """param id -> (group name, offset, numel) for parameters living in flat buffers."""
        m = {}
        if self.engine is None:
            return m
        for g in self.engine.flat.groups:
            for seg in g.segments:
                m[id(seg.param)] = (g.name, seg.offset, seg.numel)
        return m

    def variables(self, include_slots: bool = True, detach_host: bool = False) -> Dict[str, np.ndarray]:
        """TensorFlow-named arrays of every variable.  ``detach_host``: arrays never alias live host tensors
        (what a background writer needs while training keeps updating them)."""

        def host(t):
            a = t.cpu().numpy()
            return a.copy() if (detach_host and t.device.type == "cpu") else a

        segs = self._segment_views()
        master = self.engine.full_master() if self.engine is not None else {}
        slots = self.engine.full_slots() if (self.engine is not None and include_slots) else {}
        out: Dict[str, np.ndarray] = {}
        for spec in self.specs:
            t = spec.tensor
            loc = segs.get(id(t))
            if loc is not None:
                gname, off, n = loc
                val = master[gname][off:off + n].view(t.shape)
            else:
                val = t.detach()
            # a parameter nothing trains changes only through copy_ / .data assignment (restore, casts), which
            # move its version counter or its storage; running statistics are written by kernels behind
            # torch's back and are never cached
            frozen = (loc is None and isinstance(t, nn.Parameter) and not t.requires_grad
                      and not getattr(self, "_frozen_disabled", False))  # (switch for tools/bench_ckpt.py)
            key = (t.data_ptr(), t._version, tuple(t.shape), t.dtype) if frozen else None
            hit = self._frozen.get(spec.name) if frozen else None
            if hit is not None and hit[0] == key:
                out[spec.name] = hit[1]
            else:
                out[spec.name] = host(spec.to_tf(val.float()))
                if frozen:
                    out[spec.name].setflags(write=False)
                    self._frozen[spec.name] = (key, out[spec.name])
            if loc is not None:
                for key, flat in slots.items():
                    slot, g2 = key.rsplit("/", 1)
                    if g2 != gname:
                        continue
                    sv = flat[off:off + n].view(t.shape)
                    out[spec.name + "/" + SLOT_SUFFIX.get(slot, slot)] = host(spec.to_tf(sv.float()))
        return out

    def gather_only(self) -> None:
        """Non-chief ranks of a sharded run: participate in the collectives ``save`` performs."""
        if self.engine is not None:
            self.engine.full_master()
            self.engine.full_slots()

    # ------------------------------------------------------------------ save
    def wait(self) -> None:
        """Block until a background save has written its files (raises what the writer raised)."""
        fut, self._write_future = self._write_future, None
        if fut is not None:
            fut.result()

    def save(self, checkpoint_dir: str, global_step: int, write_state: bool = True,
             extra: Optional[Dict[str, np.ndarray]] = None, write_meta: bool = True, background: bool = False) -> str:
        """Write ``<checkpoint_dir>/<basename>-<global_step>.{index,data-00000-of-00001,meta}`` and update the
        ``checkpoint`` state file.  With ``background`` the variables are snapshotted now (host copies in
        TensorFlow layout) and the files are written by a helper thread while training continues; the next
        save, :meth:`wait` or a restore joins it first, so files appear in step order."""
        self.wait()
        os.makedirs(checkpoint_dir, exist_ok=True)
        tensors = self.variables(detach_host=background)
        fetched = getattr(self.engine, "fetched_global_step", None)
        if fetched is not None:
            global_step = fetched  # asynchronous ps: label the files with the step the fetched state belongs to
        tensors["global_step"] = np.array(int(global_step), dtype=np.int64)
        opt = getattr(self.engine, "optimizer", None)
        if opt is not None and getattr(opt, "beta1_power", None) is not None:
            tensors[self.power_scope + "beta1_power"] = np.array(opt.beta1_power, dtype=np.float32)
            tensors[self.power_scope + "beta2_power"] = np.array(opt.beta2_power, dtype=np.float32)
        if extra:
            tensors.update(extra)
        base = "%s-%d" % (self.basename, int(global_step))
        prefix = os.path.join(checkpoint_dir, base)

        def write():
            bundle.write_bundle(prefix, tensors)
            if write_meta:
                from .metagraph import encode_meta_graph

                trainable = {v.name: v.trainable for v in self.specs}
                names = [(n, trainable.get(n, False)) for n in sorted(tensors, key=lambda k: k.encode())]
                tmp = prefix + ".meta.tmp%d" % os.getpid()
                with open(tmp, "wb") as f:
                    f.write(encode_meta_graph(names, self.max_to_keep))
                os.replace(tmp, prefix + ".meta")
            if write_state:
                bundle.update_checkpoint_state(checkpoint_dir, base, self.max_to_keep)

        if background:
            if self._writer is None:
                import concurrent.futures as cf

                self._writer = cf.ThreadPoolExecutor(1, thread_name_prefix="dtfa-ckpt")
            self._write_future = self._writer.submit(write)
        else:
            write()
        return prefix

    # ------------------------------------------------------------------ restore
    def restore(self, prefix: str, strict: bool = False, restore_slots: bool = True) -> int:
        """Load by variable name; returns the checkpoint's global step (0 when absent)."""
        self.wait()
        data = bundle.read_bundle(prefix)
        segs = self._segment_views()
        missing = []
        new_slots: Dict[str, torch.Tensor] = {}
        slots = self.engine.full_slots() if (self.engine is not None and restore_slots) else {}
        touched_master = False
        for spec in self.specs:
            if spec.name not in data:
                missing.append(spec.name)
                continue
            t = spec.tensor
            v = torch.from_numpy(np.asarray(data[spec.name], dtype=np.float32))
            new = spec.from_tf(v, t.detach().float().cpu())
            loc = segs.get(id(t))
            if loc is not None:
                gname, off, n = loc
                g = next(g for g in self.engine.flat.groups if g.name == gname)
                g.master[off:off + n].copy_(new.reshape(-1).to(g.master.device))
                touched_master = True
                for key, flat in slots.items():
                    slot, g2 = key.rsplit("/", 1)
                    name = spec.name + "/" + SLOT_SUFFIX.get(slot, slot)
                    if g2 == gname and name in data:
                        sv = spec.from_tf(torch.from_numpy(np.asarray(data[name], dtype=np.float32)),
                                          t.detach().float().cpu())
                        if key not in new_slots:
                            new_slots[key] = flat.clone()
                        new_slots[key][off:off + n].copy_(sv.reshape(-1).to(flat.device))
            else:
                with torch.no_grad():
                    t.copy_(new.to(device=t.device, dtype=t.dtype))
        if strict and missing:
            raise KeyError("variables missing from checkpoint: %s" % missing[:5])
        step = 0
        for key in ("global_step", "Variable"):
            if key in data:
                step = int(np.asarray(data[key]).reshape(-1)[0])
                break
        # the step count first: a parameter-server engine uploads it together with the variables (it is also
        # what Adam's beta powers are derived from)
        opt = getattr(self.engine, "optimizer", None)
        if opt is not None:
            opt.step_count = step
        if self.engine is not None:
            if touched_master:
                self.engine.load_master()
            if new_slots:
                self.engine.load_full_slots(new_slots)
        return step

    def restore_latest(self, checkpoint_dir: str, **kw) -> Optional[int]:
        self.wait()
        prefix = bundle.latest_checkpoint(checkpoint_dir)
        if prefix is None:
            return None
        return self.restore(prefix, **kw)
