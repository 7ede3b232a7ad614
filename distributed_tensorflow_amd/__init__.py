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

__version__ = "0.1.0"
