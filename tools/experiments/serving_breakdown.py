"""Where a single-image request spends its time inside the servable (batch 1, Inception-v3 + flower head)."""
import os, sys, tempfile, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_serving import synthetic_jpegs  # noqa: E402
from distributed_tensorflow_amd.io.export import Servable, export_servable  # noqa: E402
from distributed_tensorflow_amd.models import build_model  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
model = build_model("inception_v3_head", num_classes=5, device=dev)
d = tempfile.mkdtemp()
path = export_servable(model, ["a", "b", "c", "d", "e"], d, 1, model_name="inception_v3_head", image_size=299)
sv = Servable(path, device=dev)
jp = synthetic_jpegs(16)
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
k = [0]
def req():
    k[0] += 1
    return [jp[(k[0] + i) % 16] for i in range(B)]
full = timeit(lambda: sv.predict_images(req()))
pre = timeit(lambda: sv._preprocess(req()))
x = sv._preprocess(req())
fwd = timeit(lambda: sv._forward(x))
def post():
    logits = sv._forward(x)[:, :5].float()
    s, i = logits.max(dim=1, keepdim=True)
    p = torch.softmax(logits, dim=1)
    i.view(-1).tolist(); s.cpu().numpy(); p.cpu().numpy()
fp = timeit(post)
entry = sv._graphs[next(b for b in sv.GRAPH_BUCKETS if b >= B)]
rep = timeit(lambda: entry[0].replay())
from distributed_tensorflow_amd.data import jpeg as J
ent = timeit(lambda: J.decode_batch_to_device(req(), dev))
print("batch %d: request %.3f ms = preprocess %.3f (of it JPEG decode to device %.3f) + forward %.3f (graph replay alone %.3f) + top-1 / softmax / copies %.3f"
      % (B, full, pre, ent, fwd, rep, fp - fwd))
