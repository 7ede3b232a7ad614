"""Merge a table measured by tools/tune_conv.py into the shipped one.

    python tools/merge_tuned.py gpurun_out/tune_w.json [--kinds wgrad] [--margin 0.03]

An entry of the shipped table (``ops/tuned_gfx950.json``) is replaced when the new measurement's fastest candidate beats
the shipped choice -- timed in the same run -- by more than ``margin``; new keys are added (the rule's choice unless
something beats it by the margin).  Entries keep ``us`` = [time of the rule's choice, time of the entry]."""
import argparse
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("measured")
    ap.add_argument("--table", default=os.path.join(ROOT, "distributed_tensorflow_amd", "ops", "tuned_gfx950.json"))
    ap.add_argument("--kinds", default="igemm,wgrad")
    ap.add_argument("--margin", type=float, default=0.03)
    a = ap.parse_args()
    with open(a.measured) as f:
        new = json.load(f)
    with open(a.table) as f:
        table = json.load(f)
    changed = added = 0
    for kind in a.kinds.split(","):
        for key, m in new.get(kind, {}).items():
            if "rule" not in m or not isinstance(m.get("us"), dict):
                continue   # carried over, not measured in that run
            us = m["us"]
            name = lambda c: ",".join(map(str, c))
            best, rule = m["best"], m["rule"]
            old = table[kind].get(key)
            keep = old["best"] if old is not None and name(old["best"]) in us else rule
            if name(keep) not in us:
                keep = best
            if us[name(best)] < (1.0 - a.margin) * us[name(keep)]:
                keep = best
            entry = {"best": list(keep), "us": [us.get(name(rule), us[name(keep)]), us[name(keep)]]}
            if old is None:
                added += 1
            elif list(old["best"]) != list(keep):
                changed += 1
                print("%s %s: %s -> %s (%.1f -> %.1f us)" % (kind, key, old["best"], keep, us[name(old["best"])], us[name(keep)]))
            table[kind][key] = entry
    with open(a.table, "w") as f:
        json.dump(table, f, indent=0, sort_keys=True)
    print("changed %d, added %d" % (changed, added))


if __name__ == "__main__":
    main()
