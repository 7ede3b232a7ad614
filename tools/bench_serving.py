"""Serving throughput of an exported servable on one GPU.

    python tools/bench_serving.py [--model inception_v3_head] [--image_size 299] [--clients 32] [--seconds 5]

Part 1: the servable called in-process with batches of JPEGs (decode + resize + network), per batch size.
Part 2: the REST server under ``--clients`` concurrent single-image clients (dynamic batching), reporting
requests/s, mean latency and the mean batch the batcher formed.
"""
import argparse
import base64
import http.client
import io
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.io.export import Servable, export_servable  # noqa: E402
from distributed_tensorflow_amd.models import build_model  # noqa: E402
from distributed_tensorflow_amd.serving import ModelManager, ModelServer  # noqa: E402


def synthetic_jpegs(n, hw=(320, 400), seed=0):
    from PIL import Image

    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        yy, xx = np.mgrid[0:hw[0], 0:hw[1]]
        img = np.stack([(yy * (i + 3) + xx) % 256, (xx * 2 + i * 17) % 256, (yy + xx * (i + 1)) % 256], -1)
        img = (img + rng.integers(0, 24, img.shape)).clip(0, 255).astype(np.uint8)
        b = io.BytesIO()
        Image.fromarray(img).save(b, format="JPEG", quality=85)  # baseline, 4:2:0
        out.append(b.getvalue())
    return out


def run_grpc_clients(port, threads_n, seconds):
    """Client process: THREADS channels sending single-image Predict requests for SECONDS."""
    import grpc

    from distributed_tensorflow_amd.serving import tfs_proto as P

    jpegs = synthetic_jpegs(16)
    reqs = []
    for j in jpegs:
        r = P.PredictRequest()
        r.model_spec.name, r.model_spec.signature_name = "flowers", "predict_images"
        r.inputs["images"].CopyFrom(P.string_tensor([j]))
        reqs.append(r.SerializeToString())
    counts, lat = [0] * threads_n, [0.0] * threads_n
    t_start = time.perf_counter()
    stop = t_start + seconds

    def client(i):
        ch = grpc.insecure_channel("127.0.0.1:%d" % port, options=[("grpc.max_send_message_length", 64 << 20)])
        call = ch.unary_unary("/tensorflow.serving.PredictionService/Predict", request_serializer=lambda b: b,
                              response_deserializer=P.PredictResponse.FromString)
        k = i
        while time.perf_counter() < stop:
            t0 = time.perf_counter()
            call(reqs[k % len(reqs)], timeout=60)
            lat[i] += time.perf_counter() - t0
            counts[i] += 1
            k += 1
        ch.close()

    threads = [threading.Thread(target=client, args=(i,)) for i in range(threads_n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    print(json.dumps({"requests": sum(counts), "latency_sum": sum(lat), "seconds": time.perf_counter() - t_start}))
    return 0


def run_clients(spec):
    """Client process: THREADS keep-alive connections posting single-image predict requests for SECONDS."""
    if spec.startswith("grpc:"):
        port, threads_n, seconds = spec[5:].split(",")
        return run_grpc_clients(int(port), int(threads_n), float(seconds))
    port, threads_n, seconds = spec.split(",")
    port, threads_n, seconds = int(port), int(threads_n), float(seconds)
    jpegs = synthetic_jpegs(16)
    bodies = [json.dumps({"instances": [{"b64": base64.b64encode(j).decode()}]}) for j in jpegs]
    counts, lat = [0] * threads_n, [0.0] * threads_n
    t_start = time.perf_counter()
    stop = t_start + seconds

    def client(i):
        conn = http.client.HTTPConnection("127.0.0.1", port, timeout=60)
        k = i
        while time.perf_counter() < stop:
            t0 = time.perf_counter()
            conn.request("POST", "/v1/models/flowers:predict", bodies[k % len(bodies)])
            r = conn.getresponse()
            r.read()
            if r.status != 200:
                raise RuntimeError("status %d" % r.status)
            lat[i] += time.perf_counter() - t0
            counts[i] += 1
            k += 1
        conn.close()

    threads = [threading.Thread(target=client, args=(i,)) for i in range(threads_n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    print(json.dumps({"requests": sum(counts), "latency_sum": sum(lat), "seconds": time.perf_counter() - t_start}))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="inception_v3_head")
    ap.add_argument("--image_size", type=int, default=299)
    ap.add_argument("--clients", type=int, default=32)
    ap.add_argument("--seconds", type=float, default=5.0)
    ap.add_argument("--max_batch", type=int, default=64)
    ap.add_argument("--batch_timeout_ms", type=float, default=2.0)
    ap.add_argument("--hip_graphs", type=int, default=1)
    ap.add_argument("--client_procs", type=int, default=4, help="client processes (threads are spread over them)")
    ap.add_argument("--client_of", default="", help=argparse.SUPPRESS)  # internal: PORT,THREADS,SECONDS
    args = ap.parse_args()
    if args.client_of:
        return run_clients(args.client_of)
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    classes = ["daisy", "dandelion", "roses", "sunflowers", "tulips"]
    base = tempfile.mkdtemp(prefix="dtfa_serving_")
    model = build_model(args.model, num_classes=len(classes), device=device)
    export_servable(model, classes, base, 1, model_name=args.model, image_size=args.image_size)
    del model
    jpegs = synthetic_jpegs(64)
    s = Servable(os.path.join(base, "1"), device=device, hip_graphs=bool(args.hip_graphs))
    print("in-process servable (%s @%d, %s, hip_graphs=%d)" % (args.model, args.image_size, device, s.hip_graphs))
    for bs in (1, 8, 32, 64):
        batch = jpegs[:bs]
        for _ in range(3):
            s.predict_images(batch)
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < 2.0:
            s.predict_images(batch)
            n += 1
        dt = time.perf_counter() - t0
        print("  batch %3d: %8.1f images/s  %7.2f ms/call" % (bs, n * bs / dt, dt / n * 1e3), flush=True)
    del s

    manager = ModelManager(base, "flowers", device=device, poll_seconds=0,
                           loader=lambda path: Servable(path, device=device, hip_graphs=bool(args.hip_graphs))).start()
    server = ModelServer(manager, port=0, max_batch=args.max_batch, batch_timeout_ms=args.batch_timeout_ms).start()
    # warm-up request per bucket the batcher can form, then the timed run with clients in their own processes
    # (they share no interpreter lock with the server)
    warm = http.client.HTTPConnection("127.0.0.1", server.port, timeout=300)
    for k in (1, 2, 4, 8, 16, 32, 64):
        if k <= args.max_batch:
            warm.request("POST", "/v1/models/flowers:predict",
                         json.dumps({"instances": [{"b64": base64.b64encode(j).decode()} for j in jpegs[:k]]}))
            warm.getresponse().read()
    procs_n = max(1, min(args.client_procs, args.clients))
    per = [args.clients // procs_n + (1 if i < args.clients % procs_n else 0) for i in range(procs_n)]
    from distributed_tensorflow_amd.serving import GrpcModelServer

    grpc_server = GrpcModelServer(manager, server.batcher, port=0, max_workers=max(16, args.clients)).start()

    def timed(label, spec_of):
        b0 = server.batcher.stats()
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--client_of", spec_of(n)],
                                  stdout=subprocess.PIPE, text=True) for n in per]
        results = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in procs]
        dt = time.perf_counter() - t0
        total = sum(r["requests"] for r in results)
        lat_sum = sum(r["latency_sum"] for r in results)
        span = max(r["seconds"] for r in results)
        st = server.batcher.stats()
        batches, items = st["batches"] - b0["batches"], st["items"] - b0["items"]
        print("%s, %d clients x 1 image: %.1f requests/s, mean latency %.2f ms, mean batch %.1f, "
              "device busy %.0f%%" % (label, args.clients, total / span, lat_sum / max(total, 1) * 1e3,
                                      items / max(batches, 1), 100 * (st["busy_seconds"] - b0["busy_seconds"]) / dt),
              flush=True)

    timed("REST server", lambda n: "%d,%d,%f" % (server.port, n, args.seconds))
    timed("gRPC server (PredictionService/Predict)", lambda n: "grpc:%d,%d,%f" % (grpc_server.port, n, args.seconds))
    grpc_server.stop()
    server.stop()


if __name__ == "__main__":
    main()
