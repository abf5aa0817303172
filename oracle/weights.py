"""Seeded, BN-calibrated synthetic weights for parity tests — test infrastructure only.

``calibrated_state_dict`` = ``frmap_amd.synth.synth_state_dict`` (seeded init keyed by parameter
name) followed by ``face_oracle.calibrate_bn`` on a seeded N(0,1) batch.  Both the golden
generator (container, reference classes) and the tests (container and GPU box) build their weights
through this one function, so they agree up to fp32 round-off of the calibration pass.
"""
from __future__ import annotations

import os
import sys
from typing import Dict, Mapping, Tuple

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

import frmap_amd.synth as synth  # noqa: E402  (pure numpy/torch helper, no HIP)
from oracle import face_oracle  # noqa: E402

# SURVEY.md §8d suggested seeds
SEEDS = {"baseline": (1001, 2001, 3001), "cnn": (1002, 2002, 3002), "arcface": (1004, 2004, 3004),
         "siamese": (1006, 2006, 3006), "hybrid": (1005, 2005, 3005), "attention": (1007, 2007, 3007)}
N_CALIB = 32
N_GOLDEN = 16


def calibrated_state_dict(model_type: str,
                          shapes: Mapping[str, Tuple[Tuple[int, ...], torch.dtype]],
                          seed: int, n_calib: int = N_CALIB) -> Dict[str, torch.Tensor]:
    sd = synth.synth_state_dict(shapes, seed)
    xcal = synth.randn(seed + 50000, (n_calib, 3, 224, 224), tag="calib")
    nt = torch.get_num_threads()
    face_oracle.calibrate_bn(model_type, sd, xcal)
    torch.set_num_threads(nt)
    return sd


def golden_inputs(model_type: str, n: int = N_GOLDEN) -> torch.Tensor:
    return synth.randn(SEEDS[model_type][1], (n, 3, 224, 224), tag="x")
