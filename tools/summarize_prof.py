"""Summarise a rocprofv3 --kernel-trace --stats run: per-kernel totals, grouped categories, markdown output."""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    title = sys.argv[3] if len(sys.argv) > 3 else d
    found = glob.glob(d + "/**/*_kernel_stats.csv", recursive=True)
    if found:
        rows = list(csv.DictReader(open(found[0])))
    else:
        # rocprofv3's default output is a rocpd SQLite database; build the same per-kernel rows from its dispatch table
        import sqlite3
        db = sqlite3.connect(glob.glob(d + "/**/*_results.db", recursive=True)[0])
        q = db.execute("select name, count(*), sum(end - start), avg(end - start) from kernels group by name order by 3 desc").fetchall()
        total = float(sum(r[2] for r in q))
        rows = [{"Name": n, "Calls": str(c), "TotalDurationNs": str(t), "AverageNs": str(a), "Percentage": str(100.0 * t / total)}
                for n, c, t, a in q]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    cats = {}

    def cat(n):
        if "conv_igemm" in n or "conv_p3" in n or "conv_p8" in n: return "conv fwd/dgrad (implicit GEMM, MFMA)"
        if "conv_wgrad" in n or "wgrad_reduce" in n or "conv_wp3" in n or "conv_wp8" in n: return "conv wgrad (+ split reduction)"
        if n.startswith("void bn_") or "bn_" in n or "col_partials" in n or "partial_reduce" in n: return "batch-norm (stats / apply / backward)"
        if "pool" in n or "gap_" in n: return "pooling"
        if "weight_permute" in n: return "dgrad weight permute"
        if "momentum" in n or "adam" in n or "cast_f32" in n: return "optimizer"
        if "xent" in n or "mean_f32" in n: return "loss"
        if "at::native" in n or "rocclr" in n or "elementwise" in n: return "torch glue (autograd add / fill / copy)"
        return "other"

    for r in rows:
        c = cat(r["Name"])
        cats[c] = cats.get(c, 0.0) + float(r["TotalDurationNs"])
    print("# %s\n" % title)
    print("Total GPU kernel time: %.2f ms over %g steps = **%.2f ms/step**\n" % (tot / 1e6, steps, tot / 1e6 / steps))
    print("| category | ms/step | share |\n|---|---:|---:|")
    for c, v in sorted(cats.items(), key=lambda kv: -kv[1]):
        print("| %s | %.2f | %.1f%% |" % (c, v / 1e6 / steps, 100 * v / tot))
    print("\n| kernel | calls | total ms | avg us | share |\n|---|---:|---:|---:|---:|")
    for r in rows[:int(sys.argv[4]) if len(sys.argv) > 4 else 30]:
        print("| `%s` | %s | %.2f | %.1f | %.1f%% |" % (r["Name"][:80].replace("|", "/"), r["Calls"],
                                                     float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                                                     float(r["Percentage"])))


if __name__ == "__main__":
    main()
