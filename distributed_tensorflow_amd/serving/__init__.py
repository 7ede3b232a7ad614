"""Model serving: version manager, dynamic batcher, REST and gRPC model servers for exported servables."""
from .batcher import DynamicBatcher, QueueFull  # noqa: F401
from .manager import ModelManager, list_versions, wait_for_version  # noqa: F401
from .server import ModelServer  # noqa: F401


def __getattr__(name):  # the gRPC front end needs grpcio + protobuf; import it on demand
    if name == "GrpcModelServer":
        from .grpc_server import GrpcModelServer
        return GrpcModelServer
    raise AttributeError(name)
