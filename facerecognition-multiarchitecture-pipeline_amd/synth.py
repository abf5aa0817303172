"""Seeded synthetic inputs / weights / galleries.

There is no network on the build or GPU boxes, so ImageNet / VGGFace2 weights
(`/root/reference/src/face_models.py:67,269,463,658`; `src/app.py:281`) cannot be fetched.  Every
benchmark and parity test therefore uses tensors generated here.  The generator is numpy-PCG64
keyed by ``(seed, canonical parameter name)``, so it does not depend on torch's RNG, on parameter
iteration order, or on which of the reference's aliased key sets (``backbone.*`` vs ``features.*``,
`face_models.py:463-464,658-660`) is used to ask for a tensor.
"""
from __future__ import annotations

import math
import os
import zlib
from typing import Dict, Iterable, Mapping, Tuple

import numpy as np
import torch


def _rng(seed: int, tag: str) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([int(seed), zlib.crc32(tag.encode())]))


def randn(seed: int, shape: Iterable[int], tag: str = "x") -> torch.Tensor:
    """i.i.d. N(0,1) float32 tensor — the stand-in for an ImageNet-normalised crop batch."""
    a = _rng(seed, tag).standard_normal(tuple(shape), dtype=np.float32)
    return torch.from_numpy(a)


def unit_rows(seed: int, n: int, d: int, tag: str = "gallery") -> torch.Tensor:
    """Random unit-norm float32 rows (a synthetic identity gallery)."""
    a = _rng(seed, tag).standard_normal((n, d)).astype(np.float64)
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    return torch.from_numpy(a.astype(np.float32))


# ``features.N`` is `nn.Sequential(*list(resnet.children())[:-1|-2])` in the reference, i.e. the same
# tensors as the named ResNet children.  Index -> child name:
_FEATURES_IDX = {"0": "conv1", "1": "bn1", "4": "layer1", "5": "layer2", "6": "layer3", "7": "layer4"}


def canonical_key(key: str, trunk_prefix: str | None) -> str:
    """Map an aliased ``features.<i>.*`` key onto ``<trunk_prefix>.<child>.*``."""
    if trunk_prefix and key.startswith("features."):
        parts = key.split(".")
        child = _FEATURES_IDX.get(parts[1])
        if child is not None:
            return ".".join([trunk_prefix, child] + parts[2:])
    return key


def trunk_prefix_of(keys: Iterable[str]) -> str | None:
    ks = list(keys)
    for p in ("backbone", "cnn"):
        if any(k.startswith(p + ".conv1.") for k in ks):
            return p
    return None


def synth_state_dict(shapes: Mapping[str, Tuple[Tuple[int, ...], torch.dtype]], seed: int) -> Dict[str, torch.Tensor]:
    """Build a state_dict for the given ``{key: (shape, dtype)}`` map.

    Rules (by key suffix): conv / linear ``weight`` -> N(0, 2/fan_in); norm-layer ``weight`` ->
    U(0.5, 1.5); ``bias`` -> N(0, 0.05²); ``running_mean`` -> N(0, 0.1²); ``running_var`` ->
    U(0.5, 1.5); ``num_batches_tracked`` and ``arcface.u`` -> 0; ``pos_encoding`` -> N(0, 0.02²)
    (`face_models.py:668`); ``gamma`` -> U(0.3, 0.8).
    """
    keys = list(shapes.keys())
    tp = trunk_prefix_of(keys)
    bn_prefixes = {k[: -len("running_mean")] for k in keys if k.endswith("running_mean")}
    out: Dict[str, torch.Tensor] = {}
    for key in keys:
        shape, dtype = shapes[key]
        ck = canonical_key(key, tp)
        g = _rng(seed, ck)
        leaf = key.rsplit(".", 1)[-1]
        pre = key[: len(key) - len(leaf)]
        if leaf == "num_batches_tracked":
            t = torch.zeros(shape, dtype=dtype)
        elif leaf == "u":
            t = torch.zeros(shape, dtype=dtype)
        elif leaf == "running_mean":
            t = torch.from_numpy((0.1 * g.standard_normal(shape)).astype(np.float32))
        elif leaf == "running_var":
            t = torch.from_numpy(g.uniform(0.5, 1.5, shape).astype(np.float32))
        elif leaf == "gamma":  # AttentionModule.gamma (`face_models.py:220`, zeros at init: the branch would be inert)
            t = torch.from_numpy(g.uniform(0.3, 0.8, shape).astype(np.float32))
        elif leaf == "pos_encoding":
            t = torch.from_numpy((0.02 * g.standard_normal(shape)).astype(np.float32))
        elif leaf in ("bias", "in_proj_bias"):
            t = torch.from_numpy((0.05 * g.standard_normal(shape)).astype(np.float32))
        elif leaf in ("weight", "in_proj_weight", "weights"):
            if len(shape) >= 2:
                fan_in = int(np.prod(shape[1:]))
                t = torch.from_numpy((math.sqrt(2.0 / fan_in) * g.standard_normal(shape)).astype(np.float32))
            elif pre in bn_prefixes or len(shape) == 1:
                t = torch.from_numpy(g.uniform(0.5, 1.5, shape).astype(np.float32))
            else:
                t = torch.from_numpy(g.standard_normal(shape).astype(np.float32))
        else:
            t = torch.from_numpy((0.05 * g.standard_normal(shape)).astype(np.float32))
        out[key] = t.to(dtype)
    return out


def shapes_of(module: torch.nn.Module) -> Dict[str, Tuple[Tuple[int, ...], torch.dtype]]:
    return {k: (tuple(v.shape), v.dtype) for k, v in module.state_dict().items()}


# ------------------------------------------------------------------------------------------------
# BatchNorm-calibrated synthetic weights without a forward pass
# ------------------------------------------------------------------------------------------------
# A seeded random-init network maps every input to almost the same embedding (SURVEY.md §7 hard part 2), so the
# parity tests calibrate each BatchNorm's running statistics on a seeded batch (oracle/weights.py, a CPU forward of
# the oracle).  The benchmark must run the SAME network the tests check, but the product package may not import the
# oracle and has no CPU forward of its own: the calibrated statistics of the benchmarked (model, seed) pairs are
# therefore committed as data (``data/bn_stats_<model>_<seed>.npz``, written by ``oracle/gen_bench_calib.py`` from
# ``oracle.weights.calibrated_state_dict``; ~40 KB each) and merged into the seeded init here.
_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def bn_stats_path(model_type: str, seed: int) -> str:
    return os.path.join(_DATA_DIR, f"bn_stats_{model_type}_{int(seed)}.npz")


def calibrated_state_dict(model_type: str, shapes: Mapping[str, Tuple[Tuple[int, ...], torch.dtype]], seed: int) -> Dict[str, torch.Tensor]:
    """``synth_state_dict(shapes, seed)`` with every BatchNorm ``running_mean`` / ``running_var`` replaced by the
    committed calibrated statistics of (model_type, seed).  Raises if that pair has no committed statistics."""
    path = bn_stats_path(model_type, seed)
    if not os.path.isfile(path):
        raise FileNotFoundError(f"no committed BatchNorm statistics for ({model_type!r}, seed {seed}): {path} "
                                "(generate with `python oracle/gen_bench_calib.py` in the build container)")
    sd = synth_state_dict(shapes, seed)
    stats = np.load(path)
    tp = trunk_prefix_of(sd.keys())
    for key in sd:
        if key.endswith("running_mean") or key.endswith("running_var"):
            ck = canonical_key(key, tp)
            if ck not in stats.files:
                raise KeyError(f"{path} has no entry {ck!r}")
            t = torch.from_numpy(np.asarray(stats[ck], dtype=np.float32))
            if tuple(t.shape) != tuple(sd[key].shape):
                raise ValueError(f"{ck}: stored shape {tuple(t.shape)} != {tuple(sd[key].shape)}")
            sd[key] = t.clone()
    return sd
