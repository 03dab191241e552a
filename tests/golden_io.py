"""Loaders for the committed golden fixtures (tests/golden/)."""
import gzip
import json
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _conv(o):
    if isinstance(o, str) and o in ("inf", "-inf", "nan"):
        return float(o)
    if isinstance(o, list):
        return [_conv(v) for v in o]
    if isinstance(o, dict):
        return {k: _conv(v) for k, v in o.items()}
    return o


def load(name):
    p = os.path.join(GOLDEN, name)
    if name.endswith(".gz"):
        with gzip.open(p, "rt") as f:
            return _conv(json.load(f))
    with open(p) as f:
        return _conv(json.load(f))
