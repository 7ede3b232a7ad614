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

            def t_of(cfg):
                """Time of ``cfg`` in this run (entries written before the reduction split existed lack its element)."""
                for k in (",".join(map(str, cfg)), ",".join(map(str, cfg)) + ",1"):
                    if k in us:
                        return us[k]
                return None

            best, rule = m["best"], m["rule"]
            old = table[kind].get(key)
            keep = old["best"] if old is not None and t_of(old["best"]) is not None else rule
            if t_of(keep) is None:
                keep = best
            if t_of(best) < (1.0 - a.margin) * t_of(keep):
                keep = best
            keep = list(keep)
            if kind == "igemm" and len(keep) == 6 and keep[5] == 1 and tuple(keep[:4]) != (64, 64, 64, 2):
                keep = keep[:5]   # no reduction split, on a tile that has none: the table's five-element form
            entry = {"best": list(keep), "us": [t_of(rule) if t_of(rule) is not None else t_of(keep), t_of(keep)]}
            if old is None:
                added += 1
            elif list(old["best"]) != list(keep):
                changed += 1
                print("%s %s: %s -> %s (%.1f -> %.1f us)" % (kind, key, old["best"], keep, t_of(old["best"]), t_of(keep)))
            table[kind][key] = entry
    with open(a.table, "w") as f:
        json.dump(table, f, indent=0, sort_keys=True)
    print("changed %d, added %d" % (changed, added))


if __name__ == "__main__":
    main()
