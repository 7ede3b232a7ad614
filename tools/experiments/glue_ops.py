"""Which torch operators (not this framework's kernels) run inside a ResNet-50 training step, with their input shapes:
torch.profiler over one forward + backward."""
import sys
import torch
from torch.profiler import ProfilerActivity, profile
from distributed_tensorflow_amd.models import build_model
from distributed_tensorflow_amd.ops import functional as F


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda")
    model = build_model("resnet50", num_classes=5, device=dev)
    model.train()
    cpad = getattr(model, "input_channels", 8)
    images = torch.randn(B, 224, 224, cpad, device=dev).bfloat16()
    labels = torch.randint(0, 5, (B,), device=dev).to(torch.int32)

    def step():
        out = model(images)
        logits, reg = out if isinstance(out, tuple) else (out, None)
        loss, _, _, _ = F.softmax_xent(logits, labels, 5)
        loss.backward()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6):
        if e.key in ("aten::copy_", "aten::contiguous", "aten::add", "aten::add_", "aten::cat", "aten::fill_", "aten::zero_", "aten::mul", "aten::to", "aten::_to_copy", "aten::clone"):
            dt = getattr(e, "device_time_total", None) or getattr(e, "cuda_time_total", 0)
            if dt < 20:
                continue
            stack = [s for s in e.stack if "distributed_tensorflow_amd" in s or "bench" in s][:3]
            print("%-18s x%-3d %8.0f us  %s\n      %s" % (e.key, e.count, dt, e.input_shapes, " <- ".join(stack)))


if __name__ == "__main__":
    main()
