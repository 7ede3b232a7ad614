#!/usr/bin/env python
"""Model server for the servables written by ``save_model.py`` / ``flower_train.py``.

The reference ends with an exported, versioned model directory
(recognition_flower/distributed/save_model.py:116-171) and leaves serving to an
external model server.  This entry point is that server::

    python serve_model.py --model_base_path=./output/models --model_name=flowers --rest_api_port=8501

    curl localhost:8501/v1/models/flowers
    curl -d '{"instances": [{"b64": "'$(base64 -w0 rose.jpg)'"}]}' localhost:8501/v1/models/flowers:predict
    curl -d '{"examples": [{"image/encoded": {"b64": "'$(base64 -w0 rose.jpg)'"}}]}' localhost:8501/v1/models/flowers:classify

gRPC clients reach ``tensorflow.serving.PredictionService`` (Predict, Classify) and
``tensorflow.serving.ModelService`` (GetModelStatus) on ``--port``.

The newest integer-named version under ``--model_base_path`` is served and the
directory is polled for newer ones.  Concurrent requests are merged into GPU
batches of up to ``--max_batch_size`` images (``--batch_timeout_ms`` bounds the
wait of a lone request).
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for peer-mapped GPU memory (RCCL, xGMI mailboxes)

from distributed_tensorflow_amd.utils import flags  # noqa: E402

flags.DEFINE_string("model_base_path", "./output/models", "Directory holding <version>/ servable exports.")
flags.DEFINE_string("model_name", "flowers", "Name in the request path /v1/models/<name>.")
flags.DEFINE_integer("rest_api_port", 8501, "HTTP port (0 picks a free one).")
flags.DEFINE_integer("port", 8500, "gRPC port of PredictionService / ModelService (-1 disables gRPC, 0 picks a free one).")
flags.DEFINE_string("host", "127.0.0.1", "Interface to bind.")
flags.DEFINE_integer("max_batch_size", 64, "Largest number of images merged into one forward pass.")
flags.DEFINE_float("batch_timeout_ms", 2.0, "Longest wait for a batch to fill.")
flags.DEFINE_integer("max_enqueued_images", 4096, "Requests beyond this queue depth are answered 503.")
flags.DEFINE_float("file_system_poll_wait_seconds", 2.0, "Version directory polling period (0 disables polling).")
flags.DEFINE_string("device", "auto", "cuda | cpu | auto")
flags.DEFINE_float("wait_for_model_seconds", 0.0, "Wait this long for a first version to appear before giving up.")

FLAGS = flags.FLAGS


def main(argv):
    import torch

    from distributed_tensorflow_amd.serving import ModelManager, ModelServer, wait_for_version

    device = FLAGS.device
    if device == "auto":
        device = "cuda" if torch.cuda.is_available() else "cpu"
    manager = ModelManager(FLAGS.model_base_path, FLAGS.model_name, device=device,
                           poll_seconds=FLAGS.file_system_poll_wait_seconds)
    manager.start()
    if manager.version is None:
        if FLAGS.wait_for_model_seconds <= 0:
            raise SystemExit("no servable version under %s" % FLAGS.model_base_path)
        wait_for_version(manager, FLAGS.wait_for_model_seconds)
    server = ModelServer(manager, FLAGS.host, FLAGS.rest_api_port, max_batch=FLAGS.max_batch_size,
                         batch_timeout_ms=FLAGS.batch_timeout_ms, max_queue=FLAGS.max_enqueued_images)
    grpc_server = None
    if FLAGS.port >= 0:
        from distributed_tensorflow_amd.serving import GrpcModelServer

        grpc_server = GrpcModelServer(manager, server.batcher, FLAGS.host, FLAGS.port).start()
    print("serving model %r version %s on http://%s:%d%s (device %s)"
          % (FLAGS.model_name, manager.version, FLAGS.host, server.port,
             " and grpc://%s:%d" % (FLAGS.host, grpc_server.port) if grpc_server else "", device), flush=True)
    try:
        server.serve_forever()
    except KeyboardInterrupt:
        pass
    finally:
        if grpc_server is not None:
            grpc_server.stop()
        server.stop()
    return 0


if __name__ == "__main__":
    flags.run(main)
