"""Average the counters of a rocprofv3 --pmc run per kernel name."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if "conv" not in k and "col_partials" not in k and "bn_" not in k:
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print("    %-32s %16.0f" % (c, sum(v) / len(v)))
