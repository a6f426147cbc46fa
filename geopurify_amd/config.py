"""Flat experiment config with `KEY VALUE` overrides -- host mirror of util/config.py.

Behaviour reproduced (util/config.py:58-146):
  * load: the yaml's top-level sections (DATA, Model, ...) are merged into ONE flat namespace whose
    keys are also attributes; nested mappings (e.g. category_split) stay attribute-accessible;
  * override: trailing command-line pairs `KEY VALUE`; VALUE is python-literal-evaluated when
    possible; the key (last dotted component) must already exist; the new value must have the old
    value's type, except that anything may replace None and list <-> tuple are converted.
"""
import ast
import copy
import os

import yaml


class CfgNode(dict):
    """dict with attribute access; nested dicts are wrapped recursively."""

    def __init__(self, mapping=None, key_list=None, new_allowed=False):
        super().__init__()
        for k, v in (mapping or {}).items():
            self[k] = CfgNode(v) if type(v) is dict else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def __str__(self):
        lines = []
        for k in sorted(self):
            v = self[k]
            if isinstance(v, CfgNode):
                body = "\n".join("  " + ln for ln in str(v).split("\n"))
                lines.append(f"{k}:\n{body}")
            else:
                lines.append(f"{k}: {v}")
        return "\n".join(lines)

    def __repr__(self):
        return f"CfgNode({dict.__repr__(self)})"


def load_cfg_from_cfg_file(file):
    if not (os.path.isfile(file) and file.endswith(".yaml")):
        raise AssertionError(f"{file} is not a yaml file")
    with open(file, "r") as fh:
        sections = yaml.safe_load(fh)
    flat = {}
    for section in sections.values():
        flat.update(section)
    return CfgNode(flat)


def _literal(text):
    if not isinstance(text, str):
        return text
    try:
        return ast.literal_eval(text)
    except (ValueError, SyntaxError):
        return text


def _coerce(new, old, full_key):
    if old is None or type(new) is type(old):
        return new
    if isinstance(new, tuple) and isinstance(old, list):
        return list(new)
    if isinstance(new, list) and isinstance(old, tuple):
        return tuple(new)
    raise ValueError(f"Type mismatch ({type(old)} vs. {type(new)}) with values ({old} vs. {new}) "
                     f"for config key: {full_key}")


def merge_cfg_from_list(cfg, cfg_list):
    if len(cfg_list) % 2:
        raise AssertionError("overrides must be KEY VALUE pairs")
    out = copy.deepcopy(cfg)
    for full_key, raw in zip(cfg_list[0::2], cfg_list[1::2]):
        key = full_key.split(".")[-1]
        if key not in cfg:
            raise AssertionError(f"Non-existent key: {full_key}")
        out[key] = _coerce(_literal(raw), cfg[key], full_key)
    return out
