"""Do a weight gradient and the next layer's data gradient gain from running on two streams?  Sequential on one stream
against forked onto two (events), ResNet-50 14x14 stage shapes at batch 2048."""
import torch
from distributed_tensorflow_amd.ops import conv as cv


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps


def main():
    B = 2048
    dev = "cuda"
    p3 = cv.get_plan(14, 14, 256, 256, 3, 3, 1, 1, (1, 1, 1, 1), dev)
    p1 = cv.get_plan(14, 14, 1024, 256, 1, 1, 1, 1, (0, 0, 0, 0), dev)
    x3 = torch.randn(B, 14, 14, 256, device=dev).bfloat16()
    dy3 = torch.randn(B, 14, 14, 256, device=dev).bfloat16()
    w1 = (torch.randn(256, 1, 1, 1024, device=dev) * 0.03).bfloat16()
    dy1 = torch.randn(B, 14, 14, 256, device=dev).bfloat16()
    side = torch.cuda.Stream()

    def wgrad():
        return cv.conv_wgrad(x3, dy3, p3)

    def dgrad():
        return cv.conv_dgrad(dy1, w1, p1)

    def seq():
        wgrad()
        dgrad()

    def forked():
        main_s = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(main_s)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            wgrad()
        dgrad()
        ev2 = torch.cuda.Event()
        ev2.record(side)
        main_s.wait_event(ev2)

    print("wgrad 3x3 14x14x256: %.1f us" % timed(wgrad))
    print("dgrad 1x1 1024->256: %.1f us" % timed(dgrad))
    print("one stream:  %.1f us" % timed(seq))
    print("two streams: %.1f us" % timed(forked))
    print("one stream:  %.1f us" % timed(seq))
    print("two streams: %.1f us" % timed(forked))


if __name__ == "__main__":
    main()
