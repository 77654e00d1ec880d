"""Deterministic parameter fill rule shared by the golden-capture script and the tests.

Freshly constructed reference models output exactly 0 (``zero_module`` on each
ResBlock's second conv, every attention ``proj_out`` and the head conv:
reference unet.py:210-212, 294, 615), and the real checkpoints are not
available offline.  Golden vectors are therefore captured with EVERY parameter
overwritten by this rule, which depends only on the parameter's state-dict
name and shape, so both sides can regenerate identical weights without any
weight file travelling.

value[i] = scale(name, shape) * u(i, name),  u in [-1, 1) from a splitmix64
hash of (flat index, crc32(name)); float32.
"""
from __future__ import annotations

import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _uniform(name: str, n: int) -> np.ndarray:
    seed = np.uint64(zlib.crc32(name.encode("utf-8")))
    with np.errstate(over="ignore"):
        h = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        h = h + seed * np.uint64(0xBF58476D1CE4E5B9) + np.uint64(0x632BE59BD9B4E019)
        h ^= h >> np.uint64(30)
        h = h * np.uint64(0xBF58476D1CE4E5B9)
        h ^= h >> np.uint64(27)
        h = h * np.uint64(0x94D049BB133111EB)
        h ^= h >> np.uint64(31)
    u24 = (h >> np.uint64(40)).astype(np.float64)  # 24 random bits
    return (u24 / float(1 << 23) - 1.0).astype(np.float32)


def fill_array(name: str, shape) -> np.ndarray:
    """Return the float32 fill for one parameter."""
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if shape else 1
    u = _uniform(name, n).reshape(shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "positional_embedding":
        return (u * (1.0 / np.sqrt(shape[0]))).astype(np.float32)
    if name.endswith("label_emb.weight"):
        return (0.5 * u).astype(np.float32)
    if len(shape) >= 2:  # conv / linear weight: variance 1/fan_in
        fan_in = int(np.prod(shape[1:]))
        return (u * np.sqrt(3.0 / fan_in)).astype(np.float32)
    if leaf == "weight":  # 1-D weight == GroupNorm gamma
        return (1.0 + 0.2 * u).astype(np.float32)
    return (0.1 * u).astype(np.float32)  # biases / GroupNorm beta


def fill_state_dict(shapes) -> dict:
    """name -> float32 ndarray for a {name: shape} mapping."""
    return {k: fill_array(k, s) for k, s in shapes.items()}
