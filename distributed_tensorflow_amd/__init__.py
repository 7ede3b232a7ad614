"""MI355X-native distributed image-classifier training framework.

Package layout::

    models/    Inception-v3 (slim variable names) + flower head, flower CNN, ResNet-50
    ops/       autograd wrappers over the gfx950 HIP kernels + fp32 reference ops
    optim/     fused Adam / momentum-SGD on flat shards, LR schedules
    parallel/  sync data-parallel, sharded parameter ownership (RS/AG),
               staleness-bounded async, TF_CONFIG ps/worker runtime
    data/      TFRecord / JPEG-directory / synthetic input pipelines, device decode+resize
    io/        Saver-V2 tensor-bundle checkpoints, TensorBoard events, servable export
    train/     session loop, hooks, throughput meter
    utils/     flags, cluster spec, logging
"""

import os as _os

# Peer-mapped device memory (RCCL over xGMI, the hipIpc-mapped mailboxes of parallel/p2p.py and parallel/xps.py) needs
# the dmabuf IPC path of the ROCm runtime; the legacy path fails with ``hipIpcGetMemHandle: invalid argument``.  The
# runtime reads the variable when it initialises, which is at the first HIP call, so importing the package before
# that is early enough.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

# The communicator's flight recorder (a ring of the last collectives and whether the watchdog thread has retired them)
# is what train/graph.py reads to know that no eager collective is still being polled when a step graph is captured.
# Read when the process group is created.
_os.environ.setdefault("TORCH_FR_BUFFER_SIZE", "2000")

__version__ = "0.1.0"
