"""Tensors beyond 2^31 elements (ResNet-50 at batch 4096: the stem output and the stage-1 activations are 3.3 G elements
each).  The batch is the same B / 2 images twice over, so within ONE run
  * the logits of sample i and of sample i + B / 2 must be the same bits (batch statistics are shared, everything else is
    per sample), and
  * the weight gradients of a loss over the first half only and of a loss over the second half only must agree up to the
    order of fp32 sums.
A kernel whose indices wrapped, or that treated the upper part of a tensor differently, shows as a difference."""
import sys
import torch
from distributed_tensorflow_amd.models import build_model


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    h = B // 2
    dev = torch.device("cuda")
    torch.manual_seed(7)
    model = build_model("resnet50", num_classes=5, device=dev)
    model.train()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("gamma"):   # (the last batch-norm of a block starts at zero: nothing would flow through the blocks)
                p.copy_(torch.rand_like(p) + 0.5)
    cpad = getattr(model, "input_channels", 8)
    g = torch.Generator(device=dev).manual_seed(11)
    half = torch.zeros(h, 224, 224, cpad, device=dev, dtype=torch.bfloat16)
    half[..., :3] = (torch.rand(h, 224, 224, 3, device=dev, generator=g) * 2 - 1).bfloat16()
    images = torch.cat([half, half])
    del half
    labels = torch.randint(0, 5, (h,), device=dev, generator=g)

    def run(lo):
        for p in model.parameters():
            p.grad = None
        out = model(images)
        logits = out[0] if isinstance(out, tuple) else out
        loss = torch.nn.functional.cross_entropy(logits[lo:lo + h].float(), labels)
        loss.backward()
        torch.cuda.synchronize()
        return logits.detach(), float(loss.item()), {n: p.grad.detach().float().clone() for n, p in model.named_parameters()
                                                      if p.grad is not None}

    logits, l1, g1 = run(0)
    same = torch.equal(logits[:h], logits[h:])
    print("batch %d: logits of the two halves %s (largest difference %.3g)" %
          (B, "are the same bits" if same else "DIFFER", (logits[:h].float() - logits[h:].float()).abs().max().item()))
    _, l2, g2 = run(h)
    print("loss over the first half %.6f, over the second half %.6f" % (l1, l2))
    worst, zero = 0.0, 0
    for n in g1:
        scale = g1[n].abs().max().item()
        if scale == 0.0:
            zero += 1
            continue
        d = (g1[n] - g2[n]).abs().max().item() / scale
        worst = max(worst, d)
        if d > 1e-3:
            print("DIFFERS %s: %.3g of its largest entry" % (n, d))
    print("largest gradient difference over %d tensors (%d all zero): %.3g of the tensor's largest entry" % (len(g1), zero, worst))
    print("peak memory %.1f GB" % (torch.cuda.max_memory_allocated() / 2**30))


if __name__ == "__main__":
    main()
