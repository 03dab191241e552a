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


def splitmix_sym(state, n):
    """n values 2u-1 of the fixture generator's helper stream (splitmix64, oracle/ref_driver.cc) from `state`"""
    M = (1 << 64) - 1
    out = []
    s = int(state)
    for _ in range(n):
        s = (s + 0x9E3779B97F4A7C15) & M
        z = s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        z ^= z >> 31
        out.append(2 * ((z >> 11) * (1.0 / 9007199254740992.0)) - 1)
    return out


def trace_deltas(g):
    """the scripted proposal offsets [Nt][nsteps][D] of a trace: stored, or (compact fixtures) regenerated as the generator
    made them: sym() * step_scale / sqrt(max(beta_r, 0.02)), rung-major"""
    import numpy as np
    if "deltas" in g:
        return np.array(g["deltas"])
    Nt, ns, D = g["Nt"], g["nsteps"], g["D"]
    u = np.array(splitmix_sym(g["delta_state"], Nt * ns * D)).reshape(Nt, ns, D)
    sc = g["step_scale"] / np.sqrt(np.maximum(np.array(g["invtemps"]), 0.02))
    return u * sc[:, None, None]
