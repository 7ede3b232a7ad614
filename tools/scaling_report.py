"""Run bench.py on 1, 2, 4, 8 GPUs of this node and print the weak-scaling table.

    python tools/scaling_report.py [--gpus 1,2,4,8] [-- extra bench.py flags]

One process per GPU over RCCL (``torch.distributed.run``); efficiency(N) = value(N) / (N * value(1))."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(n, extra, port):
    if n == 1:
        cmd = [sys.executable, "bench.py", "--gpus", "1"] + extra
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
               "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", str(n)] + extra
    r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise SystemExit("bench.py failed on %d GPU(s):\n%s" % (n, r.stderr[-3000:]))
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def main():
    argv = sys.argv[1:]
    extra = []
    if "--" in argv:
        i = argv.index("--")
        argv, extra = argv[:i], argv[i + 1:]
    gpus = [1, 2, 4, 8]
    if "--gpus" in argv:
        gpus = [int(v) for v in argv[argv.index("--gpus") + 1].split(",")]
    import torch

    avail = torch.cuda.device_count()
    rows = []
    for k, n in enumerate(g for g in gpus if g <= avail):
        rows.append(run(n, extra, 29500 + k))
        print(json.dumps(rows[-1]), flush=True)
    if not rows:
        raise SystemExit("no GPUs visible")
    base = rows[0]["value"] / rows[0]["n_gpus"]
    print("\n| GPUs | %s | ms/step | efficiency |\n|---:|---:|---:|---:|" % rows[0]["unit"])
    for r in rows:
        print("| %d | %.0f | %.2f | %.1f %% |" % (r["n_gpus"], r["value"], r["ms_per_step"],
                                                100.0 * r["value"] / (r["n_gpus"] * base)))
    # the second model bench.py measures in the same run (flower CNN by default)
    for name in (rows[0].get("also") or {}):
        vals = [(r["n_gpus"], (r.get("also") or {}).get(name) or {}) for r in rows]
        if any("value" not in v for _, v in vals):
            continue
        b = vals[0][1]["value"] / vals[0][0]
        print("\n%s\n\n| GPUs | images/sec | ms/step | efficiency |\n|---:|---:|---:|---:|" % name)
        for n, v in vals:
            print("| %d | %.0f | %.2f | %.1f %% |" % (n, v["value"], v["ms_per_step"], 100.0 * v["value"] / (n * b)))


if __name__ == "__main__":
    main()
