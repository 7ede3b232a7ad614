"""Per-kernel counter table of one training step from the three passes of tools/pmc_step.sh.

    python tools/pmc_step_summary.py gpurun_out/pmc_step/<tag> [title]

Columns: calls, mean time per launch, matrix-core busy share (SQ_VALU_MFMA_BUSY_CYCLES over SIMD cycles available:
1024 SIMDs x time x 2.4 GHz), HBM read / write bytes per launch (FETCH_SIZE / WRITE_SIZE, reported in KiB) and the
rate they imply, L2 hit rate, share of wave time spent waiting on any instruction counter, LDS bank-conflict share.
Times are those of the counter run (kernels serialised), so rates are per kernel, not per step.
"""
import csv
import glob
import sys
from collections import defaultdict

SIMDS, CLOCK = 1024, 2.4e9


def load(d):
    """{kernel: {counter: [values]}} plus {kernel: [durations ns]} of one pass."""
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)
    if not f:
        return {}, {}
    vals = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(dict)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return vals, {k: list(v.values()) for k, v in dur.items()}


def mean(x):
    return sum(x) / len(x) if x else 0.0


def main():
    base = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else base
    a, da = load(base + "_a")
    b, db = load(base + "_b")
    c, dc = load(base + "_c")
    rows = []
    for k in a:
        t = mean(da[k]) * 1e-9
        calls = len(da[k])
        mfma = mean(a[k].get("SQ_VALU_MFMA_BUSY_CYCLES", []))
        wave = mean(a[k].get("SQ_WAVE_CYCLES", []))
        wait = mean(a[k].get("SQ_WAIT_INST_ANY", []))
        lds_act = mean(a[k].get("SQ_LDS_IDX_ACTIVE", []))
        lds_conf = mean(a[k].get("SQ_LDS_BANK_CONFLICT", []))
        rd = mean(b.get(k, {}).get("FETCH_SIZE", [])) * 1024
        tb = mean(db.get(k, [])) * 1e-9
        wr = mean(c.get(k, {}).get("WRITE_SIZE", [])) * 1024
        tc = mean(dc.get(k, [])) * 1e-9
        hit, miss = mean(c.get(k, {}).get("TCC_HIT_sum", [])), mean(c.get(k, {}).get("TCC_MISS_sum", []))
        rows.append((calls * t, k, calls, t, mfma / (SIMDS * t * CLOCK) if t else 0, rd, wr,
                     (rd / tb if tb else 0) + (wr / tc if tc else 0), hit / (hit + miss) if hit + miss else 0,
                     wait / wave if wave else 0, lds_conf / lds_act if lds_act else 0))
    rows.sort(reverse=True)
    total = sum(r[0] for r in rows)
    print("# %s\n" % title)
    print("| kernel | calls | us / launch | share | MFMA busy | HBM read MB | HBM write MB | HBM TB/s | L2 hit | wave wait | LDS conflicts |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for tot, k, calls, t, mf, rd, wr, bw, hit, wait, conf in rows[:28]:
        print("| `%s` | %d | %.1f | %.1f%% | %.0f%% | %.1f | %.1f | %.2f | %.0f%% | %.0f%% | %.1f%% |" % (
            k[:64].replace("|", "/"), calls, t * 1e6, 100 * tot / total, 100 * mf, rd / 1e6, wr / 1e6, bw / 1e12,
            100 * hit, 100 * wait, 100 * conf))


if __name__ == "__main__":
    main()
