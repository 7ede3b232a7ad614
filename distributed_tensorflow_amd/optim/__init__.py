from .fused import FusedAdam, FusedMomentum, OptRegion, make_optimizer, regions_from_flat  # noqa: F401
from .schedules import constant, exponential_decay, warmup_cosine  # noqa: F401
