#!/usr/bin/env python3
"""A host that drives the model-level C ABI with NOTHING from the `frmap_amd` Python package: ctypes for the calls, torch
only as the device-memory allocator (any allocator that yields device pointers would do), numpy for files.

This is the binding a maintainer of the reference would write to swap `model(images)` (`/root/reference/src/testing.py:255-273`)
and `compare_faces(get_embedding(...), refs)` (`src/app.py:44,50-64`) for the HIP path without adopting this repo's Python
surface: create a handle, feed it the checkpoint's `state_dict` entry by entry under the reference's own key names, finalize,
then call `frmap_model_forward` / `frmap_model_embed_and_match` with caller-owned buffers.

    python examples/cabi_model_client.py --model cnn --dtype f16 --weights sd.npz --inputs x.npz --out result.npz [--gallery g.npz]

  sd.npz      the state_dict: one array per key (fp32)            x.npz   {"x": fp32 [B,3,H,W]}
  g.npz       {"gallery": fp32 [G,512], "thresh": float, "normalize": 0/1}
  result.npz  logits, embedding (and with a gallery: idx, dist, ids)
`tests/test_model_cabi_gpu.py` runs it in a subprocess and checks the outputs against the reference-generated goldens.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "facerecognition-multiarchitecture-pipeline_amd", "libfrmap_hip.so")
BF16, F16 = 0, 1
IN_F32 = 0
OUT_MAP, OUT_POOLED, OUT_EMB, OUT_LOGITS = 0, 1, 2, 3
vp, i32, f32, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t


def bind():
    lib = C.CDLL(LIB)
    lib.frmap_last_error.restype = C.c_char_p
    lib.frmap_model_create.argtypes = [C.POINTER(vp), C.c_char_p, i32, i32]
    lib.frmap_model_load_tensor.argtypes = [vp, C.c_char_p, vp, sz, i32]
    lib.frmap_model_finalize.argtypes = [vp, vp]
    lib.frmap_model_workspace_bytes.restype = sz
    lib.frmap_model_workspace_bytes.argtypes = [vp, i32, i32, i32]
    lib.frmap_model_match_workspace_bytes.restype = sz
    lib.frmap_model_match_workspace_bytes.argtypes = [vp, i32, i32, i32, i32]
    lib.frmap_model_forward.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]
    lib.frmap_model_embed_and_match.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp, i32, f32, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.frmap_match_gallery_pack_bytes.restype = sz
    lib.frmap_match_gallery_pack_bytes.argtypes = [i32, i32]
    lib.frmap_match_pack_gallery.argtypes = [vp, vp, vp, i32, i32, vp]
    lib.frmap_model_embedding_dim.argtypes = [vp]
    lib.frmap_model_destroy.argtypes = [vp]
    lib.frmap_model_destroy.restype = None
    return lib


def check(lib, rc, what):
    if rc < 0:
        raise RuntimeError(f"{what}: rc {rc}: {lib.frmap_last_error().decode()}")
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", required=True, choices=["baseline", "cnn", "siamese", "arcface", "hybrid"])
    ap.add_argument("--dtype", default="f16", choices=["bf16", "f16"])
    ap.add_argument("--classes", type=int, default=36)
    ap.add_argument("--weights", required=True)
    ap.add_argument("--inputs", required=True)
    ap.add_argument("--gallery")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()

    lib = bind()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    st = torch.cuda.current_stream().cuda_stream

    h = vp()
    check(lib, lib.frmap_model_create(C.byref(h), a.model.encode(), a.classes, BF16 if a.dtype == "bf16" else F16), "model_create")
    sd = np.load(a.weights)
    used = 0
    for key in sd.files:                                   # the checkpoint, entry by entry, under the reference's key names
        t = np.ascontiguousarray(sd[key], dtype=np.float32)
        rc = check(lib, lib.frmap_model_load_tensor(h, key.encode(), t.ctypes.data_as(vp), t.size, 0), f"load_tensor({key})")
        used += rc == 0
    check(lib, lib.frmap_model_finalize(h, st), "model_finalize")

    x = torch.from_numpy(np.load(a.inputs)["x"]).to(dev)
    B, _, H, W = x.shape
    ws = torch.empty(lib.frmap_model_workspace_bytes(h, B, H, W), dtype=torch.uint8, device=dev)
    D = lib.frmap_model_embedding_dim(h)                     # 512; SiameseNet: 256
    logits = torch.empty((B, a.classes), dtype=torch.float32, device=dev)
    emb = torch.empty((B, D), dtype=torch.float32, device=dev)
    check(lib, lib.frmap_model_forward(h, x.data_ptr(), IN_F32, B, H, W, OUT_EMB, emb.data_ptr(), ws.data_ptr(), st), "forward(emb)")
    result = {"embedding": None, "tensors_used": np.int64(used)}
    if a.model in ("baseline", "cnn", "hybrid"):           # forward() = classifier logits ('arcface': the labels path, see the header; 'siamese': none)
        check(lib, lib.frmap_model_forward(h, x.data_ptr(), IN_F32, B, H, W, OUT_LOGITS, logits.data_ptr(), ws.data_ptr(), st), "forward(logits)")
        result["logits"] = None
    if a.gallery:
        g = np.load(a.gallery)
        gal = torch.from_numpy(np.ascontiguousarray(g["gallery"], dtype=np.float32)).to(dev)
        G = gal.shape[0]
        packed = stat = None
        if G >= 512:
            packed = torch.empty(lib.frmap_match_gallery_pack_bytes(G, D), dtype=torch.uint8, device=dev)
            stat = torch.empty((G, 4), dtype=torch.float32, device=dev)
            check(lib, lib.frmap_match_pack_gallery(gal.data_ptr(), packed.data_ptr(), stat.data_ptr(), G, D, st), "pack_gallery")
        ws2 = torch.empty(lib.frmap_model_match_workspace_bytes(h, B, H, W, G), dtype=torch.uint8, device=dev)
        idx = torch.empty(B, dtype=torch.int32, device=dev)
        ids = torch.empty(B, dtype=torch.int32, device=dev)
        dist = torch.empty(B, dtype=torch.float32, device=dev)
        check(lib, lib.frmap_model_embed_and_match(h, x.data_ptr(), IN_F32, B, H, W, gal.data_ptr(),
                                                   packed.data_ptr() if packed is not None else None,
                                                   stat.data_ptr() if stat is not None else None, G, float(g["thresh"]),
                                                   int(g["normalize"]), idx.data_ptr(), dist.data_ptr(), ids.data_ptr(), None, None,
                                                   ws2.data_ptr(), st), "embed_and_match")
        torch.cuda.synchronize()
        result.update(idx=idx.cpu().numpy(), dist=dist.cpu().numpy(), ids=ids.cpu().numpy())
    torch.cuda.synchronize()
    result["embedding"] = emb.cpu().numpy()
    if "logits" in result:
        result["logits"] = logits.cpu().numpy()
    lib.frmap_model_destroy(h)
    assert not any(m == "frmap_amd" or m.startswith("frmap_amd.") for m in sys.modules), "this client must not import the package"
    np.savez(a.out, **result)
    print("ok", a.model, a.dtype, "B", B, "tensors used", used)


if __name__ == "__main__":
    main()
