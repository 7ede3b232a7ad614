"""Where do the gradient collectives of a captured step see themselves?  Prints, for every bucket callback during the
capture, the thread, the current stream and whether it is capturing."""
import os, sys, threading, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ["DTFA_FORCE_COLLECTIVES"] = "1"
from distributed_tensorflow_amd.models import build_model
from distributed_tensorflow_amd.ops import functional as F
from distributed_tensorflow_amd.parallel.engine import DataParallelEngine
from distributed_tensorflow_amd.train.graph import StepGraph
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
m = build_model("flower_cnn", num_classes=5, device=dev); m.train()
eng = DataParallelEngine(m, mode=sys.argv[1] if len(sys.argv) > 1 else "sync_dp", optimizer="momentum", lr=0.02, bucket_mb=1)
orig = eng._on_bucket_ready
def spy(b):
    print("bucket ready: thread %s main=%s stream %#x capturing=%s" % (
        threading.current_thread().name, threading.current_thread() is threading.main_thread(),
        torch.cuda.current_stream().cuda_stream, torch.cuda.is_current_stream_capturing()), flush=True)
    orig(b)
eng.flat.on_bucket_ready = spy
x = torch.zeros(32, 224, 224, 8, device=dev, dtype=torch.bfloat16); x[..., :3].normal_()
y = torch.randint(0, 5, (32,), device=dev).int()
def loss_fn(im, lb):
    out = m(im)
    logits, reg = out if isinstance(out, tuple) else (out, None)
    loss = F.softmax_xent(logits, lb, 5)[0]
    return (loss if reg is None else loss + reg), loss
print("== eager"); 
for _ in range(2):
    eng.begin_step(); loss_fn(x, y)[0].backward(); eng.finish_step()
print("== capture")
sg = StepGraph(eng, loss_fn, x, y)
print("== replay")
for _ in range(3): sg.run(x, y)
torch.cuda.synchronize()
print("loss", float(sg.outputs[1]))
import time; time.sleep(1.0)
dist.barrier(); torch.cuda.synchronize(); time.sleep(1.0)
sg.graph = None; sg = None
import gc; gc.collect(); torch.cuda.synchronize()
dist.barrier(); dist.destroy_process_group(); print("done")
