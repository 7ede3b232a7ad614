"""Training session loop and hooks (``MonitoredTrainingSession`` work-alike).

The reference drives training with ``sess.run`` loops: the single-process
script logs loss / accuracy / learning-rate every step and writes summaries
(recognition_flower/single/flower_train.py:177-193); the distributed script
wraps the loop in ``MonitoredTrainingSession`` with a ``CheckpointSaverHook``
and a ``SummarySaverHook`` firing every 2 global steps
(recognition_flower/distributed/train_dataset.py:176-192).  ``TrainingSession``
reproduces that structure on top of the data-parallel engine; hooks receive
the step results and decide by *global step* (not by epoch, and only on the
chief for file-writing hooks).
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, List, Optional

import torch

from ..ops import functional as F


@dataclass
class StepResult:
    global_step: int
    loss: float
    accuracy: float
    learning_rate: float
    batch_size: int
    step_time: float
    logits: Optional[torch.Tensor] = None
    extra: Dict[str, Any] = field(default_factory=dict)


class Hook:
    chief_only = False

    def begin(self, session: "TrainingSession") -> None:
        pass

    def after_step(self, session: "TrainingSession", r: StepResult) -> None:
        pass

    def end(self, session: "TrainingSession") -> None:
        pass


class EveryNSteps:
    """Trigger for "every ``n`` global steps" that also works when the observed step sequence has gaps.

    With several asynchronous workers the global step a worker sees after its own update advances by more
    than one (other workers stepped meanwhile); a ``step % n == 0`` test then never fires for a worker that
    keeps seeing odd steps.  Like TensorFlow's step timer (the hooks of
    recognition_flower/distributed/train_dataset.py:176-180) the trigger fires when a multiple of ``n`` has
    been reached or passed since the last time it fired; on a gap-free sequence that is exactly every
    multiple of ``n``."""

    def __init__(self, n: int):
        self.n = max(1, int(n))
        self.last = None  # last step the trigger looked at

    def reset(self, step: int) -> None:
        self.last = int(step)

    def __call__(self, step: int) -> bool:
        prev = self.last if self.last is not None else step - 1
        self.last = int(step)
        return step // self.n > prev // self.n


class StopAtStepHook(Hook):
    def __init__(self, last_step: int):
        self.last_step = last_step

    def after_step(self, session, r):
        if r.global_step >= self.last_step:
            session.request_stop()


class LoggingHook(Hook):
    """Console lines in the spirit of the reference's prints
    (recognition_flower/single/flower_train.py:190-193,
    recognition_flower/distributed/train_dataset.py:190-191)."""

    def __init__(self, every_n_steps: int = 1, worker_index: Optional[int] = None, printer: Callable = print):
        self.every, self.worker_index, self.printer = EveryNSteps(every_n_steps), worker_index, printer

    def begin(self, session):
        self.every.reset(session.global_step)

    def after_step(self, session, r):
        if not self.every(r.global_step):
            return
        if self.worker_index is None:
            self.printer("step %d: learning rate %.6f accuracy %.4f loss %.6f (%.1f images/sec)" % (
                r.global_step, r.learning_rate, r.accuracy, r.loss, r.batch_size / max(r.step_time, 1e-9)))
        else:
            self.printer("Worker %d: training step %d done , loss is %f" % (self.worker_index, r.global_step, r.loss))


class ThroughputMeter(Hook):
    """Images/sec over a sliding window of steps (the reference measures no time at all)."""

    def __init__(self, window: int = 20, world_size: int = 1):
        self.window, self.world = window, world_size
        self.times: List[float] = []
        self.images: List[int] = []
        self.total_images = 0
        self.t_begin = None

    def begin(self, session):
        self.t_begin = time.perf_counter()

    def after_step(self, session, r):
        self.times.append(r.step_time)
        self.images.append(r.batch_size * self.world)
        self.total_images += r.batch_size * self.world
        if len(self.times) > self.window:
            self.times.pop(0)
            self.images.pop(0)

    @property
    def images_per_sec(self) -> float:
        t = sum(self.times)
        return sum(self.images) / t if t > 0 else 0.0


class CheckpointSaverHook(Hook):
    """Save every ``save_steps`` global steps (and at the end) through a :class:`io.saver.Saver`."""

    chief_only = True
    collective = True  # with sharded ownership every rank must take part in gathering the shards

    def __init__(self, checkpoint_dir: str, save_steps: int, saver, save_at_end: bool = True, background: bool = True):
        """``background``: the variables are snapshotted in the step, the files are written by a helper thread
        while the next steps run (the reference saves every 2 steps: written in line, the 100 MB of the
        Inception model cost many times the step itself).  ``end`` waits for the last write."""
        self.dir, self.save_steps, self.saver, self.save_at_end = checkpoint_dir, max(1, save_steps), saver, save_at_end
        self.background = background
        self.last_saved = -1
        self._due = EveryNSteps(self.save_steps)

    def begin(self, session):
        self._due.reset(session.global_step)

    def after_step(self, session, r):
        if self._due(r.global_step) and r.global_step != self.last_saved:
            session.save_checkpoint(self.saver, self.dir, r.global_step, background=self.background)
            self.last_saved = r.global_step

    def end(self, session):
        if self.save_at_end and session.global_step != self.last_saved and session.global_step > 0:
            session.save_checkpoint(self.saver, self.dir, session.global_step)
            self.last_saved = session.global_step
        self.saver.wait()


class SummarySaverHook(Hook):
    """Scalars ``loss`` / ``accuracy`` / ``learning_rate`` and the ``logits histogram``
    every ``save_steps`` global steps."""

    chief_only = True

    def __init__(self, save_steps: int, output_dir: str, histogram: bool = True):
        self.save_steps = max(1, save_steps)
        self._due = EveryNSteps(self.save_steps)
        self.output_dir = output_dir
        self.writer = None  # opened in begin(): hooks dropped on non-chief ranks never create a file
        self.histogram = histogram

    def begin(self, session):
        from ..io.events import EventWriter

        if self.writer is None:
            self.writer = EventWriter(self.output_dir)
        self._due.reset(session.global_step)

    def after_step(self, session, r):
        if not self._due(r.global_step):
            return
        scalars = {"loss": r.loss, "accuracy": r.accuracy, "learning_rate": r.learning_rate}
        scalars.update({k: float(v) for k, v in r.extra.items() if isinstance(v, (int, float))})
        hist = {}
        if self.histogram and r.logits is not None:
            hist["logits histogram"] = r.logits.detach().float().cpu().numpy()
        self.writer.add_summary(r.global_step, scalars, hist)
        self.writer.flush()

    def end(self, session):
        if self.writer is not None:
            self.writer.close()
            self.writer = None


class ProfilerHook(Hook):
    """Trace a window of steps with ``torch.profiler`` (CPU + device activities) and write a Chrome trace;
    every step is also bracketed by a marker range (``roctx`` on ROCm) so that ``rocprofv3 --marker-trace``
    shows step boundaries.  The reference has no tracing at all (its only structural dump is the graph it
    hands to ``FileWriter``, recognition_flower/single/flower_train.py:174)."""

    def __init__(self, output_dir: str, start_step: int = 5, num_steps: int = 3):
        self.output_dir, self.start, self.n = output_dir, max(1, start_step), max(1, num_steps)
        self.prof = None
        self.trace_path: Optional[str] = None

    def before_step(self, session, step: int) -> None:
        if step == self.start and self.prof is None:
            import torch.profiler as tp

            acts = [tp.ProfilerActivity.CPU] + ([tp.ProfilerActivity.CUDA] if torch.cuda.is_available() else [])
            self.prof = tp.profile(activities=acts)
            self.prof.__enter__()
        if torch.cuda.is_available():
            torch.cuda.nvtx.range_push("train_step_%d" % step)

    def after_step(self, session, r):
        import os

        if torch.cuda.is_available():
            torch.cuda.nvtx.range_pop()
        if self.prof is not None and r.global_step >= self.start + self.n - 1:
            self.prof.__exit__(None, None, None)
            os.makedirs(self.output_dir, exist_ok=True)
            self.trace_path = os.path.join(self.output_dir, "trace_steps_%d_%d.json" % (self.start, r.global_step))
            self.prof.export_chrome_trace(self.trace_path)
            self.prof = None
            self.start = 1 << 62  # one window per run

    def end(self, session):
        if self.prof is not None:
            self.prof.__exit__(None, None, None)
            self.prof = None


class TrainingSession:
    """Runs training steps of ``model`` through ``engine`` and dispatches hooks.

    ``model(images)`` returns logits or ``(logits, regulariser)``.  The loss is
    mean sparse softmax cross-entropy, plus ``reg_weight * regulariser`` and an
    optional L2 activity term on the logits -- the reference's loss
    (recognition_flower/single/flower_train.py:143-145; the distributed script
    scales the regulariser by 0.01, recognition_flower/distributed/train_dataset.py:159-160).
    """

    def __init__(self, model, engine, *, num_classes: int, hooks: Optional[List[Hook]] = None, is_chief: bool = True,
                 reg_weight: float = 1.0, logits_act_l2: float = 0.0, global_step: int = 0,
                 keep_logits: bool = False, hip_graph: bool = False, graph_warmup_steps: int = 2):
        self.model, self.engine = model, engine
        self.num_classes = num_classes
        self.is_chief = is_chief
        # sharded ownership: every rank joins the gathers of a save; delayed-apply async: every rank must drain
        # its in-flight gradients at the same step as the chief
        needs_all_ranks = getattr(engine, "world", 1) > 1 and (
            getattr(engine, "mode", "") == "sharded_ps"
            or (getattr(engine, "mode", "") == "async_ssp" and getattr(engine, "staleness", 0) > 0))
        self.hooks = [h for h in (hooks or [])
                      if is_chief or not h.chief_only or (needs_all_ranks and getattr(h, "collective", False))]
        self.reg_weight, self.logits_act_l2 = reg_weight, logits_act_l2
        self.global_step = global_step
        self.keep_logits = keep_logits or any(isinstance(h, SummarySaverHook) for h in self.hooks)
        self._stop = False
        self._begun = False
        # whole-step HIP graph: the first ``graph_warmup_steps`` steps run eagerly (they create
        # plans, workspaces and collective communicators), then one step is captured and replayed
        self.hip_graph = bool(hip_graph) and getattr(engine, "graph_capturable", False)
        self.graph_warmup_steps = max(1, int(graph_warmup_steps))
        self._graph = None
        self._eager_steps = 0

    # ------------------------------------------------------------------ control
    def request_stop(self) -> None:
        self._stop = True

    def should_stop(self) -> bool:
        return self._stop

    def __enter__(self):
        self._begin()
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            self.close()
        return False

    def _begin(self):
        if not self._begun:
            for h in self.hooks:
                h.begin(self)
            self._begun = True

    def close(self):
        for h in self.hooks:
            h.end(self)
        self.release_graph()

    def release_graph(self):
        """Drop the captured step with the device idle.  A graph that holds a communicator's kernels must not outlive
        the process group (the communicator's watchdog thread aborts the process when it does)."""
        if self._graph is not None:
            self._graph.graph = None
            self._graph = None
            import gc

            gc.collect()
            if torch.cuda.is_available():
                torch.cuda.synchronize()

    def save_checkpoint(self, saver, directory, step, background: bool = False):
        drain = getattr(self.engine, "drain", None)
        if drain is not None:
            drain()
        if not self.is_chief:
            saver.gather_only()  # join the shard all-gathers; only the chief writes files
            return None
        return saver.save(directory, step, background=background)

    # ------------------------------------------------------------------ steps
    def compute_loss(self, images, labels):
        out = self.model(images)
        logits, reg = out if isinstance(out, tuple) else (out, None)
        xent, top1, score, correct = F.softmax_xent(logits, labels, self.num_classes, act_l2=self.logits_act_l2)
        total = xent if reg is None or not self.reg_weight else xent + self.reg_weight * reg
        return total, xent, logits, top1, score, correct

    def _graph_step(self, images, labels):
        from .graph import StepGraph

        if self._graph is not None and not self._graph.compatible(images, labels):
            return None  # odd-sized batch (end of an epoch): run it eagerly
        if self._graph is None:
            if self._eager_steps < self.graph_warmup_steps:
                return None
            self._graph = StepGraph(self.engine, self.compute_loss, images, labels)
        return self._graph.run(images, labels)

    def run_step(self, images, labels) -> StepResult:
        self._begin()
        for h in self.hooks:
            before = getattr(h, "before_step", None)
            if before is not None:
                before(self, self.global_step + 1)
        t0 = time.perf_counter()
        self.model.train()
        lr = self.engine.optimizer.current_lr()
        out = self._graph_step(images, labels) if (self.hip_graph and images.is_cuda) else None
        if out is None:
            if images.is_cuda:
                from ..ops import kernels as K

                K.rng_step(images.device).add_(1)
            self.engine.begin_step()
            out = self.compute_loss(images, labels)
            out[0].backward()
            self.engine.finish_step()
            self._eager_steps += 1
        total, xent, logits, top1, score, correct = out
        # a parameter-server engine reports the shared global step its update produced (other workers advance it
        # too); the collective engines count one per step
        shared = getattr(self.engine, "global_step", None)
        self.global_step = int(shared) if shared is not None else self.global_step + 1
        # one device-to-host transfer for the three scalars; it synchronises with the device, so step_time covers
        # the whole step
        loss_v, acc, xent_v = torch.stack([total.detach().float().reshape(()), correct.float().mean(),
                                           xent.detach().float().reshape(())]).tolist()
        dt = time.perf_counter() - t0
        r = StepResult(self.global_step, loss_v, acc, lr, int(labels.shape[0]), dt,
                       logits.detach()[:, :self.num_classes] if self.keep_logits else None,
                       {"xent": xent_v})
        for h in self.hooks:
            h.after_step(self, r)
        return r

    @torch.no_grad()
    def evaluate(self, images, labels) -> Dict[str, float]:
        """Inference-mode forward (batch-norm running statistics, no dropout)."""
        was = self.model.training
        self.model.eval()
        try:
            total, xent, logits, top1, score, correct = self.compute_loss(images, labels)
        finally:
            self.model.train(was)
        return {"loss": float(total.float().item()), "accuracy": float(correct.float().mean().item())}
