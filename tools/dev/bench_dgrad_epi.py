import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_tensorflow_amd.ops import conv as cv
BF16 = torch.bfloat16
dev = torch.device("cuda", 0)
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
B = 2048
for hw, cin, cout in [(56, 256, 64), (28, 512, 128), (14, 1024, 256)]:
    plan = cv.get_plan(hw, hw, cin, cout, 1, 1, 1, 1, (0, 0, 0, 0), dev)
    dy = torch.randn(B, hw, hw, cout, device=dev).to(BF16)
    w = (torch.randn(cout, 1, 1, cin, device=dev) * cin ** -0.5).to(BF16)
    addend = torch.randn(B, hw, hw, cin, device=dev).to(BF16)
    mask = torch.randint(0, 256, (B * hw * hw * cin // 8,), device=dev, dtype=torch.uint8)
    tap = cv.BnBackwardTap(); tap.mask, tap.relu_mode, tap.gate_store = mask, 3, True
    t_tap = timeit(lambda: cv.conv_dgrad(dy, w, plan, addend=addend, bn_tap=tap))
    t_add = timeit(lambda: cv.conv_dgrad(dy, w, plan, addend=addend))
    t_plain = timeit(lambda: cv.conv_dgrad(dy, w, plan))
    gb = (dy.numel() + 2 * addend.numel()) * 2 / 1e9 + mask.numel() / 1e9
    print("%dx%d %d<-%d: tap+addend %.0f us (%.2f TB/s) addend %.0f us plain %.0f us" % (hw, hw, cin, cout, t_tap, gb / t_tap * 1e3, t_add, t_plain))
