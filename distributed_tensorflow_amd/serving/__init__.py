"""Model serving: version manager, dynamic batcher and the REST model server for exported servables."""
from .batcher import DynamicBatcher, QueueFull  # noqa: F401
from .manager import ModelManager, list_versions, wait_for_version  # noqa: F401
from .server import ModelServer  # noqa: F401
