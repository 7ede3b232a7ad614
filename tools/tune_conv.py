"""Measure the launch configurations of the convolution kernels on the shapes the models run and write the table
``ops/tune.py`` looks configurations up in.

    python tools/tune_conv.py --out distributed_tensorflow_amd/ops/tuned_gfx950.json \
        --configs resnet50:2048,flower_cnn:1024,inception_v3_head:256,inception_v3_finetune:512

Every configuration is a model and a per-GPU batch; two eager training steps of it run with ``DTFA_AUTOTUNE=1``,
so each convolution launch (forward, data gradient per phase, weight gradient) whose shape is not in the table yet
times its candidates on its real operands, inside the step, and keeps the fastest.  Existing entries of ``--out`` are
kept (``--fresh`` starts over)."""
import argparse
import os
import sys
import time

os.environ["DTFA_AUTOTUNE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "distributed_tensorflow_amd", "ops", "tuned_gfx950.json"))
    ap.add_argument("--configs", default="resnet50:2048,flower_cnn:1024,inception_v3_head:256,inception_v3_finetune:512")
    ap.add_argument("--fresh", action="store_true")
    ap.add_argument("--drop", default="", help="kinds of entries (igemm, wgrad) to measure again: removed from --out first")
    a = ap.parse_args()
    os.environ["DTFA_TUNE_DB"] = a.out
    if a.fresh and os.path.exists(a.out):
        os.remove(a.out)
    if a.drop and os.path.exists(a.out):
        import json
        with open(a.out) as f:
            table = json.load(f)
        for kind in a.drop.split(","):
            table[kind] = {}
        with open(a.out, "w") as f:
            json.dump(table, f, indent=0, sort_keys=True)
    import bench  # noqa: E402  (after the environment is set: ops.tune reads it at import)
    from distributed_tensorflow_amd.ops import tune

    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    for cfg in a.configs.split(","):
        model, batch = cfg.split(":")
        # bench.py's own defaults (whatever options it has grown), eager steps
        args = bench.parse_args(["--hip_graph", "0", "--warmup", "2", "--steps", "1", "--also", "", "--also_modes", ""])
        t0 = time.time()
        before = tune.stats()
        r = bench.run_config(args, model, int(batch), device, 0, 1)
        after = tune.stats()
        print("%s batch %s: %d new entries in %.1f s (step %.1f ms after tuning)" % (
            model, batch, after["measured_here"] - before["measured_here"], time.time() - t0, r["ms_per_step"]), flush=True)
        tune.save(a.out)
        import gc
        gc.collect()
        torch.cuda.empty_cache()
    print("wrote", a.out, tune.stats())


if __name__ == "__main__":
    main()
