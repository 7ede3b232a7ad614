"""Cluster specification: ``TF_CONFIG`` (Kubeflow TFJob style) and torchrun environments.

The reference derives its role from the ``TF_CONFIG`` environment variable::

    {"cluster": {"ps": ["host:port", ...], "worker": ["host:port", ...]},
     "task": {"type": "ps" | "worker", "index": N}}

(recognition_flower/distributed/train_dataset.py:123-139, distributed/run.sh:3).
A missing variable yields an empty spec (task type ``None``), which this
framework treats as single-process training.  For the collective GPU modes the
same spec is mapped onto a ``torch.distributed`` rendezvous: the workers form
the process group (rank = worker index), the first worker address is the
rendezvous endpoint; plain ``torchrun`` variables (RANK / WORLD_SIZE /
LOCAL_RANK / MASTER_ADDR / MASTER_PORT) are honoured when ``TF_CONFIG`` is
absent.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple


@dataclass
class ClusterSpec:
    jobs: Dict[str, List[str]] = field(default_factory=dict)
    task_type: Optional[str] = None
    task_index: int = 0

    @property
    def ps_hosts(self) -> List[str]:
        return self.jobs.get("ps", [])

    @property
    def worker_hosts(self) -> List[str]:
        return self.jobs.get("worker", [])

    @property
    def chief_hosts(self) -> List[str]:
        """A dedicated chief replica (job ``chief``, or ``master`` in older operators), as Kubeflow's TFJob
        writes it; the reference's own clusters have none and use worker 0."""
        return self.jobs.get("chief", []) or self.jobs.get("master", [])

    @property
    def trainer_hosts(self) -> List[str]:
        """Every task that trains: the chief replica first, then the workers."""
        return self.chief_hosts + self.worker_hosts

    @property
    def num_workers(self) -> int:
        return max(1, len(self.trainer_hosts))

    @property
    def trainer_index(self) -> int:
        """Rank of this task among the training tasks (chief replica = 0, worker i = i + #chiefs)."""
        if self.task_type == "worker":
            return self.task_index + len(self.chief_hosts)
        return self.task_index

    @property
    def is_trainer(self) -> bool:
        return self.task_type in (None, "worker", "chief", "master")

    @property
    def is_chief(self) -> bool:
        """The first training task is the chief.  (The reference computes ``index == 0`` for
        every job type, so ps 0 also believes it is chief,
        recognition_flower/distributed/train_dataset.py:138; only training tasks are
        chiefs here.)"""
        return self.is_trainer and self.trainer_index == 0

    def address(self, job: str, index: int) -> Tuple[str, int]:
        host, _, port = self.jobs[job][index].rpartition(":")
        return host or "127.0.0.1", int(port)

    def my_address(self) -> Tuple[str, int]:
        return self.address(self.task_type, self.task_index)


def parse_tf_config(value: Optional[str] = None) -> ClusterSpec:
    raw = os.environ.get("TF_CONFIG") if value is None else value
    cfg = json.loads(raw or "{}")
    cluster = cfg.get("cluster", {}) or {}
    task = cfg.get("task", {}) or {}
    jobs = {str(k): [str(h) for h in v] for k, v in cluster.items()}
    ttype = task.get("type")
    idx = int(task.get("index", 0) or 0)
    if ttype is not None:
        if ttype not in jobs:
            raise ValueError("TF_CONFIG task type %r is not a job of the cluster %s" % (ttype, sorted(jobs)))
        if not 0 <= idx < len(jobs[ttype]):
            raise ValueError("TF_CONFIG task index %d out of range for job %r" % (idx, ttype))
    return ClusterSpec(jobs, ttype, idx)


def torchrun_env() -> Optional[Tuple[int, int, int]]:
    """(rank, world_size, local_rank) when launched by torchrun, else None."""
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        return int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    return None


def rendezvous_from_cluster(spec: ClusterSpec) -> Dict[str, str]:
    """Environment for ``torch.distributed.init_process_group(init_method='env://')``
    built from the worker list of a TF_CONFIG cluster."""
    if not spec.trainer_hosts:
        raise ValueError("cluster has no workers")
    host, _, port = spec.trainer_hosts[0].rpartition(":")
    return {"MASTER_ADDR": "127.0.0.1" if host in ("localhost", "") else host,
            "MASTER_PORT": str(int(port) + 1000),
            "RANK": str(spec.trainer_index), "WORLD_SIZE": str(len(spec.trainer_hosts))}


def local_device_index(task_index: int, n_devices: int):
    """GPU of trainer ``task_index`` when several TF_CONFIG workers share a machine with ``n_devices`` visible GPUs
    (round-robin), or None when there is at most one device to choose from."""
    if n_devices <= 1:
        return None
    return int(task_index) % int(n_devices)
