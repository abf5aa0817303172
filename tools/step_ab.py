#!/usr/bin/env python3
"""In-process A/B of whole embed+match steps (eager, one stream) under different kernel-selection settings: the settings are
toggled between interleaved rounds in ONE process on ONE device (box-to-box and run-to-run spread is larger than most deltas)."""
import argparse, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import frmap_amd
from frmap_amd import _lib, synth

SETTINGS = {
    "default": [],
    "pp_off": [("frmap_conv_pp_tuning", (0, -1, -1))],
    "ds_gen1": [("frmap_conv_pp_ds", (0,))],
    "ds_pp": [("frmap_conv_pp_ds", (1,))],
}
RESET = [("frmap_conv_pp_tuning", (-1, -1, -1)), ("frmap_conv_pp_ds", (-1,))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256); ap.add_argument("--model", default="cnn")
    ap.add_argument("--settings", default="ds_gen1,ds_pp"); ap.add_argument("--rounds", type=int, default=6); ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    dev = "cuda"
    m = frmap_amd.get_model(a.model, 36)
    m.load_state_dict(synth.calibrated_state_dict(a.model, synth.shapes_of(m), 1002 if a.model == "cnn" else 1004))
    m = m.to(dev).eval().set_compute_dtype(torch.bfloat16)
    gal = frmap_amd.Gallery([f"id{i}" for i in range(36)], synth.unit_rows(3002, 36, 512), dev)
    x = torch.randn((a.batch, 3, 224, 224), device=dev)
    names = a.settings.split(",")
    res = {n: [] for n in names}
    with torch.no_grad():
        for _ in range(60): frmap_amd.embed_and_match(m, x, gal, 1.0, normalize=True)
        for rnd in range(a.rounds):
            for n in names:
                for fn, args in RESET + SETTINGS[n]: getattr(raw, fn)(*args)
                for _ in range(3): frmap_amd.embed_and_match(m, x, gal, 1.0, normalize=True)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(a.reps): frmap_amd.embed_and_match(m, x, gal, 1.0, normalize=True)
                torch.cuda.synchronize(); res[n].append((time.perf_counter() - t0) / a.reps * 1e6)
    for fn, args in RESET: getattr(raw, fn)(*args)
    for n, t in res.items():
        t = sorted(t)
        print(f"{n:10s} min {t[0]:8.1f} us  median {t[len(t) // 2]:8.1f} us  -> {a.batch / t[len(t) // 2] * 1e6:9.0f} faces/s")


if __name__ == "__main__":
    main()
