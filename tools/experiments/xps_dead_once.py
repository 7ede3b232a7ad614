"""One run of the dead-owner scenario of tests/test_gpu_xps.py, printing what each survivor saw (seconds per step, timeouts,
dropped pushes, applied steps per shard)."""
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import test_gpu_xps as T  # noqa: E402

with tempfile.TemporaryDirectory() as d:
    procs, outs = T._launch("dead", 3, d)
    for r, (p, o) in enumerate(zip(procs, outs)):
        line = [x for x in o.splitlines() if x.startswith("XPS_OK")]
        if not line:
            print("rank", r, "rc", p.returncode, o[-1500:])
            continue
        res = json.loads(line[0][7:])
        st = res["state"]
        print("rank", r, "steps", ["%.2f" % x for x in res["step_seconds"]], "timeouts", st["timeouts"], "dropped",
              st["dropped_pushes"], "applied", st["applied_per_shard"], "adopters", st["adopters"], "dead_mask", st["dead_mask"])
