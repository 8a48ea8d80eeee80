"""Minimal composition of the reference's Hydra config tree, for containers without hydra / omegaconf.

Covers what ``config/default.yaml`` + ``config/task/dsnt-*.yaml`` of the reference use (hydra-core ~1.2 semantics):

* defaults lists with config groups (``- task: ???``), same-group includes (``- task_default``), nested groups whose
  package follows the path (``task/model`` -> ``task.model``), ``override group: option`` entries, ``_self_``;
* command-line style overrides: ``group=option`` selects a group option, ``a.b.c=value`` assigns a (YAML-typed) value;
* interpolations ``${a.b}``, ``${choices.task/model}`` and the resolvers the reference registers or uses:
  ``oc.env``, ``oc.select``, ``hydra:runtime.choices`` / ``hydra:runtime.cwd``, ``sys.num_workers``, and ``if`` /
  ``labels`` / ``frac`` of the reference's ``runner.py:17-27``.

When hydra IS installed the reference's own ``runner.py`` composes the very same YAML files; this module is only the
stand-in.
"""
from __future__ import annotations

import os
import re
from pathlib import Path
from typing import Any, Dict, List, Optional, Sequence

import yaml

from contour_uncertainty._compat import AttrDict, to_attr

MISSING = "???"
_NOTHING = object()


def _load(path: Path) -> dict:
    if not path.exists():
        raise FileNotFoundError(f"config file {path} not found")
    return yaml.safe_load(path.read_text()) or {}


def _merge(dst: dict, src: dict) -> dict:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def _set_path(cfg: dict, dotted: str, value):
    keys = dotted.split(".")
    for k in keys[:-1]:
        cfg = cfg.setdefault(k, {})
    cfg[keys[-1]] = value


def _entries(defaults) -> List:
    return list(defaults or [])


class _Composer:
    def __init__(self, config_dir: Path, cli_groups: Dict[str, str]):
        self.dir = config_dir
        self.choices: Dict[str, Optional[str]] = dict(cli_groups)       # group path -> option (command line wins)
        self.fixed = set(cli_groups)

    # -- pass 1: `override group: option` entries anywhere in the chain of a group file
    def collect_overrides(self, group: str, option: str):
        cfg = _load(self.dir / group / f"{option}.yaml" if group else self.dir / f"{option}.yaml")
        for e in _entries(cfg.get("defaults")):
            if isinstance(e, str):
                if e != "_self_":
                    self.collect_overrides(group, e)
            else:
                (key, opt), = e.items()
                sub = key.split()[-1]
                path = f"{group}/{sub}" if group else sub
                if key.startswith("override "):
                    if path not in self.fixed and path not in self.choices:
                        self.choices[path] = opt
                elif opt not in (None, MISSING) or path in self.choices:
                    # follow the option that will actually be loaded to find deeper overrides
                    pick = self.choices.get(path, opt)
                    if pick not in (None, MISSING):
                        self.collect_overrides(path, pick)

    # -- pass 2: merge
    def load(self, group: str, option: str) -> dict:
        cfg = _load(self.dir / group / f"{option}.yaml" if group else self.dir / f"{option}.yaml")
        entries = _entries(cfg.pop("defaults", None))
        out: dict = {}
        self_done = False
        for e in entries:
            if e == "_self_":
                _merge(out, cfg)
                self_done = True
            elif isinstance(e, str):
                _merge(out, self.load(group, e))                     # same group, same package
            else:
                (key, opt), = e.items()
                if key.startswith("override "):
                    continue                                           # applied where the group is declared
                sub = key
                path = f"{group}/{sub}" if group else sub
                pick = self.choices.get(path, opt)
                if pick is None:
                    continue
                if pick == MISSING:
                    raise ValueError(f"config group '{path}' needs a choice ({path}=<option>)")
                self.choices[path] = pick
                _merge(out.setdefault(sub, {}), self.load(path, pick))
        if not self_done:
            _merge(out, cfg)
        return out


# ------------------------------------------------------------------------------------------------ interpolation
def _split_args(s: str) -> List[str]:
    args, depth, cur = [], 0, ""
    for ch in s:
        if ch == "," and depth == 0:
            args.append(cur)
            cur = ""
            continue
        depth += ch == "{"
        depth -= ch == "}"
        cur += ch
    args.append(cur)
    return [a.strip() for a in args]


def _literal(s):
    if isinstance(s, str):
        t = s.strip()
        if len(t) >= 2 and t[0] == t[-1] and t[0] in "\"'":
            return t[1:-1]
        try:
            return yaml.safe_load(t) if t != "" else ""
        except yaml.YAMLError:
            return t
    return s


class _Resolver:
    def __init__(self, root: dict, choices: dict):
        self.root, self.choices = root, choices
        self.active = set()

    def lookup(self, dotted: str):
        if dotted == "choices" or dotted.startswith("choices."):
            node: Any = {"choices": self.choices}
            keys = ["choices"] + ([dotted[len("choices."):]] if "." in dotted else [])
        else:
            node, keys = self.root, dotted.split(".")
        for k in keys:
            if not isinstance(node, dict) or k not in node:
                raise KeyError(f"interpolation key '{dotted}' not found")
            node = node[k]
        if dotted in self.active:
            raise ValueError(f"interpolation cycle at '{dotted}'")
        self.active.add(dotted)
        try:
            return self.resolve(node)
        finally:
            self.active.discard(dotted)

    def call(self, name: str, argstr: str):
        args = [self.resolve(a) if "${" in a else _literal(a) for a in _split_args(argstr)] if argstr != "" else []
        if name == "oc.env":
            if args[0] in os.environ:
                return os.environ[args[0]]
            if len(args) > 1:
                return args[1]
            raise KeyError(f"environment variable {args[0]} is not set")
        if name == "oc.select":
            try:
                return self.lookup(str(args[0]))
            except KeyError:
                return args[1] if len(args) > 1 else None
        if name == "hydra":
            return {"runtime.choices": self.choices, "runtime.cwd": os.getcwd()}[args[0]]
        if name == "sys.num_workers":
            return max((os.cpu_count() or 2) - 1, 0)
        if name == "if":                   # reference runner.py:25-27
            cond, flip, a, b = args
            return a if bool(cond) == bool(flip) else b
        if name == "labels":               # reference runner.py:17-20
            x = args[0]
            return "-" + "-".join(str(n).lower() for n in x if n != "bg") if x is not None and len(x) != 4 else ""
        if name == "frac":
            return int(args[0] * 100)
        raise KeyError(f"unknown resolver '{name}'")

    def _one(self, expr: str):
        expr = expr.strip()
        m = re.match(r"^([A-Za-z_][\w.]*):(.*)$", expr, re.S)      # resolver call "name:args" (args may be empty)
        if m:
            return self.call(m.group(1), m.group(2))
        return self.lookup(expr)

    def resolve(self, value):
        if isinstance(value, dict):
            return {k: self.resolve(v) for k, v in value.items()}
        if isinstance(value, list):
            return [self.resolve(v) for v in value]
        if not isinstance(value, str) or "${" not in value:
            return value
        # innermost-first substitution; a string that is ONE interpolation keeps the value's type
        out, i = "", 0
        whole = _NOTHING
        while i < len(value):
            if value.startswith("${", i):
                depth, j = 0, i
                while j < len(value):
                    if value.startswith("${", j):
                        depth += 1
                        j += 2
                        continue
                    if value[j] == "}":
                        depth -= 1
                        if depth == 0:
                            break
                    j += 1
                if depth != 0:
                    raise ValueError(f"unbalanced interpolation in '{value}'")
                inner = value[i + 2:j]
                res = self._one(inner)
                if i == 0 and j == len(value) - 1:
                    whole = res
                out += "" if res is None else str(res)
                i = j + 1
            else:
                out += value[i]
                i += 1
        return out if whole is _NOTHING else whole


def compose(config_dir, config_name: str = "default", overrides: Sequence[str] = ()) -> AttrDict:
    """-> the composed, fully resolved config (attribute-access dicts)."""
    config_dir = Path(config_dir)
    groups, assigns = {}, []
    for ov in overrides:
        key, _, val = ov.partition("=")
        key = key.lstrip("+")
        if (config_dir / key).is_dir():
            groups[key] = val
        else:
            assigns.append((key, yaml.safe_load(val) if val != "" else ""))
    comp = _Composer(config_dir, groups)
    comp.collect_overrides("", config_name)
    cfg = comp.load("", config_name)
    for key, val in assigns:
        _set_path(cfg, key, val)
    cfg["choices"] = dict(comp.choices)
    res = _Resolver(cfg, cfg["choices"])
    return to_attr(res.resolve(cfg))
