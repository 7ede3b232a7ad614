"""List the kernels of a rocprofv3 --kernel-trace run that are not this framework's (torch's own elementwise / fill / copy
launches), with their full names: what is left of autograd glue in a step."""
import glob
import sqlite3
import sys

db = sqlite3.connect(glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True)[0])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = db.execute("select name, count(*), sum(end - start), avg(end - start) from kernels group by name order by 3 desc").fetchall()
for n, c, t, a in rows:
    if "at::native" in n or "rocclr" in n or "Memcpy" in n:
        print("%6.1f calls/step %9.1f us/step %8.1f us avg  %s" % (c / steps, t / 1e3 / steps, a / 1e3, n[:400]))
