"""YAML configuration loading with the call surface of reference src/config/configloading.py:4-47:
`load_config(path, default_path=None)` follows `inherit_from` links and `update_recursive(dict1, dict2)` merges nested
mappings in place.  Written for this package (iterative chain resolution, cycle detection, empty files tolerated)."""
from __future__ import annotations

import os

import yaml


def _read_yaml(path):
    with open(path, "r") as stream:
        doc = yaml.load(stream, Loader=yaml.Loader)
    return {} if doc is None else doc


def load_config(path, default_path=None):
    """Settings of `path` layered over everything it inherits from.

    The chain of `inherit_from` links is walked from the requested file up to its root; the result starts from
    `default_path` (only when given) and the chain is folded in root first, so the most specific file wins."""
    chain, visited = [], set()
    cursor = path
    while cursor is not None:
        key = os.path.abspath(cursor)
        if key in visited:
            raise ValueError(f"inherit_from cycle through {cursor}")
        visited.add(key)
        doc = _read_yaml(cursor)
        chain.append(doc)
        cursor = doc.get("inherit_from")
    merged = _read_yaml(default_path) if default_path is not None else {}
    for doc in reversed(chain):
        update_recursive(merged, doc)
    return merged


def update_recursive(dict1, dict2):
    """Merge `dict2` into `dict1` in place: mappings are merged key by key, everything else is overwritten."""
    for key, value in dict2.items():
        if isinstance(value, dict):
            child = dict1.get(key)
            if not isinstance(child, dict):
                child = dict1[key] = {}
            update_recursive(child, value)
        else:
            dict1[key] = value
