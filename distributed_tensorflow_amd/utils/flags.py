"""``tf.app.flags`` / ``tf.app.run`` work-alike.

The reference scripts declare their configuration with ``tf.app.flags`` and
start through ``tf.app.run()`` (e.g. recognition_flower/single/flower_train.py:10-31,254-255).
This module provides the same surface: ``DEFINE_*`` functions, a ``FLAGS``
object whose first attribute access parses ``sys.argv`` lazily (the reference
relies on that: it reads ``FLAGS`` at import time,
recognition_flower/distributed/train_dataset.py:59-65), ``--name=value`` /
``--name value`` / ``--[no]bool`` syntax, and ``run(main)``.
"""
from __future__ import annotations

import sys
from typing import Any, Callable, Dict, List, Optional


class _Flag:
    def __init__(self, name, default, help_, parser, kind):
        self.name, self.default, self.help, self.parser, self.kind = name, default, help_, parser, kind
        self.value = default


def _parse_bool(s: str) -> bool:
    if s.lower() in ("1", "true", "t", "yes", "y"):
        return True
    if s.lower() in ("0", "false", "f", "no", "n"):
        return False
    raise ValueError("not a boolean: %r" % s)


class FlagValues:
    def __init__(self):
        self.__dict__["_flags"] = {}
        self.__dict__["_parsed"] = False
        self.__dict__["_remaining"] = []

    # -- definition -------------------------------------------------------
    def _define(self, name, default, help_, parser, kind):
        if name in self._flags:
            raise ValueError("flag %r defined twice" % name)
        self._flags[name] = _Flag(name, default, help_, parser, kind)
        if self._parsed:  # defined after parsing (e.g. by a late import): re-parse remaining args
            self.__dict__["_parsed"] = False

    # -- parsing ------------------------------------------------------------
    def parse(self, argv: Optional[List[str]] = None) -> List[str]:
        args = list(sys.argv[1:] if argv is None else argv)
        rest: List[str] = []
        i = 0
        while i < len(args):
            a = args[i]
            i += 1
            if a == "--":
                rest.extend(args[i:])
                break
            if not a.startswith("-") or a == "-":
                rest.append(a)
                continue
            body = a.lstrip("-")
            name, eq, val = body.partition("=")
            if name in ("help", "h") and not eq:
                print(self.usage())
                raise SystemExit(0)
            flag = self._flags.get(name)
            if flag is None and name.startswith("no") and name[2:] in self._flags and \
                    self._flags[name[2:]].kind == "bool" and not eq:
                self._flags[name[2:]].value = False
                continue
            if flag is None:
                rest.append(a)  # unknown flags are left for the caller, like absl's known-only mode
                continue
            if flag.kind == "bool" and not eq:
                flag.value = True
                continue
            if not eq:
                if i >= len(args):
                    raise ValueError("flag --%s needs a value" % name)
                val = args[i]
                i += 1
            try:
                flag.value = flag.parser(val)
            except Exception as e:  # noqa: BLE001
                raise ValueError("bad value %r for --%s: %s" % (val, name, e))
        self.__dict__["_parsed"] = True
        self.__dict__["_remaining"] = rest
        return rest

    def usage(self) -> str:
        lines = ["flags:"]
        for f in self._flags.values():
            lines.append("  --%s (%s, default %r)\n      %s" % (f.name, f.kind, f.default, f.help))
        return "\n".join(lines)

    def reset(self) -> None:
        """Forget parsed values (tests)."""
        for f in self._flags.values():
            f.value = f.default
        self.__dict__["_parsed"] = False

    # -- access ---------------------------------------------------------------
    def __getattr__(self, name: str) -> Any:
        flags = self.__dict__["_flags"]
        if name not in flags:
            raise AttributeError("unknown flag %r" % name)
        if not self.__dict__["_parsed"]:
            self.parse()
        return flags[name].value

    def __setattr__(self, name: str, value: Any) -> None:
        if name not in self._flags:
            raise AttributeError("unknown flag %r" % name)
        self._flags[name].value = value

    def __contains__(self, name: str) -> bool:
        return name in self._flags

    def flag_values_dict(self) -> Dict[str, Any]:
        if not self._parsed:
            self.parse()
        return {k: f.value for k, f in self._flags.items()}


FLAGS = FlagValues()


def DEFINE_string(name, default, help_="", flag_values: FlagValues = FLAGS):
    flag_values._define(name, default, help_, str, "string")


def DEFINE_integer(name, default, help_="", flag_values: FlagValues = FLAGS):
    flag_values._define(name, default, help_, int, "integer")


def DEFINE_float(name, default, help_="", flag_values: FlagValues = FLAGS):
    flag_values._define(name, default, help_, float, "float")


def DEFINE_boolean(name, default, help_="", flag_values: FlagValues = FLAGS):
    flag_values._define(name, default, help_, _parse_bool, "bool")


DEFINE_bool = DEFINE_boolean


def run(main: Callable[[List[str]], Any], argv: Optional[List[str]] = None, flag_values: FlagValues = FLAGS):
    """Parse flags, then call ``main([program] + unparsed_args)`` and exit with its result."""
    rest = flag_values.parse(argv)
    sys.exit(main([sys.argv[0]] + rest))
