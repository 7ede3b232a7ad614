"""Turn the raw output of tools/tune_conv.py (every candidate's time) into the table that ships with the package:
one entry per measured shape, the rule's choice unless the fastest candidate beat it by more than 3 %."""
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
d = json.load(open(src))
out = {"_about": "launch configurations measured on MI355X (gfx950) by tools/tune_conv.py inside training steps of the "
                 "models at their benchmark batch sizes; us = [time of the rule's choice, time of the entry]; where the "
                 "fastest candidate was within 3 % of the rule's choice the entry is the rule's choice",
       "igemm": {}, "wgrad": {}}
changed = 0
for kind in ("igemm", "wgrad"):
    for k, v in d[kind].items():
        r = ",".join(map(str, v["rule"]))
        b = ",".join(map(str, v["best"]))
        ur, ub = v["us"].get(r), v["us"][b]
        if ur is not None and ub > 0.97 * ur:
            out[kind][k] = {"best": v["rule"], "us": [ur, ur]}
        else:
            out[kind][k] = {"best": v["best"], "us": [ur, ub]}
            changed += 1
json.dump(out, open(dst, "w"), indent=0, sort_keys=True)
print({k: len(out[k]) for k in ("igemm", "wgrad")}, "entries that differ from the rules:", changed)
