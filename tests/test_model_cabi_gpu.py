"""GPU: the model-level C ABI (`frmap_model_*`, include/frmap_hip.h) driven by a host that imports NOTHING from the Python
package (`examples/cabi_model_client.py`, run in a subprocess: ctypes + a device allocator), checked against

  * the reference-generated goldens `tests/golden/cnn.npz` / `arcface.npz` (outputs of the reference's own classes),
  * config 2's oracle top-1 (cnn, 256 faces, 36-ID enrolment gallery) and a 10 000-ID arcface match,
  * the Python planner (`face_models._PyTrunkPlan`, FRMAP_PY_PLAN=1: one C-ABI op per call): same kernels in the same order on
    the same folded weights => bit-identical outputs; and the default Python surface, which is itself a client of the handle.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import frmap_amd  # noqa: E402
from frmap_amd import synth  # noqa: E402
from oracle import face_oracle as fo  # noqa: E402
from oracle import weights  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLIENT = os.path.join(ROOT, "examples", "cabi_model_client.py")
DEV = "cuda"


def _run_client(tmp_path, model, dtype, sd, x, gallery=None, thresh=1.0, normalize=0):
    np.savez(tmp_path / "sd.npz", **{k: v.numpy() for k, v in sd.items() if v.dtype.is_floating_point})
    np.savez(tmp_path / "x.npz", x=x.numpy())
    cmd = [sys.executable, CLIENT, "--model", model, "--dtype", dtype, "--weights", str(tmp_path / "sd.npz"),
           "--inputs", str(tmp_path / "x.npz"), "--out", str(tmp_path / "out.npz")]
    if gallery is not None:
        np.savez(tmp_path / "g.npz", gallery=gallery.numpy(), thresh=np.float32(thresh), normalize=np.int32(normalize))
        cmd += ["--gallery", str(tmp_path / "g.npz")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(tmp_path / "out.npz")


def _py_model(mt, sd, dtype, py_plan=False):
    """The package's module; `py_plan`: planned in Python (per-op calls) instead of on a model handle."""
    import frmap_amd.face_models as fm
    m = frmap_amd.get_model(mt, 36)
    m.load_state_dict(sd)
    m = m.to(DEV).eval().set_compute_dtype(dtype)
    old = fm._PY_PLAN
    fm._PY_PLAN = bool(py_plan)
    try:
        m._get_plan()
    finally:
        fm._PY_PLAN = old
    assert (m.model_handle() is None) == bool(py_plan)
    return m


@pytest.mark.parametrize("mt", ["cnn", "arcface", "baseline", "siamese", "hybrid"])
def test_ctypes_only_host_reproduces_the_reference_goldens(mt, tmp_path, gold_dir, calibrated_sd):
    z = np.load(os.path.join(gold_dir, f"{mt}.npz"))
    sd = calibrated_sd(mt)
    x = weights.golden_inputs(mt)
    out = _run_client(tmp_path, mt, "f16", sd, x)
    emb, gold = torch.from_numpy(out["embedding"]), torch.from_numpy(z["embedding"])
    rel = float((emb - gold).norm() / gold.norm())
    print(f"C-ABI {mt} fp16 embedding vs golden(reference): rel-L2 {rel:.2e}")
    if mt in ("arcface", "siamese"):
        assert float((1 - F.cosine_similarity(emb, gold, dim=1)).max()) < 1e-3            # north_star: cosine <= 1e-3 in fp16
        if mt == "arcface":
            assert int(out["tensors_used"]) >= 100 + 5 + 2     # trunk (100 tensors; its features.* aliases load again) + embedding/bn + val_classifier
    elif mt in ("baseline", "hybrid"):
        assert rel < 8e-3
        ref = torch.from_numpy(z["forward"])
        assert float((torch.from_numpy(out["logits"]) - ref).norm() / ref.norm()) < 8e-3  # forward(): fc2 / fc
    else:
        assert rel < 5e-3
        ref = torch.from_numpy(z["forward"])
        assert float((torch.from_numpy(out["logits"]) - ref).norm() / ref.norm()) < 5e-3  # forward(): resnet.fc
        assert int(out["tensors_used"]) == 100 + 2
    # the Python planner issues the same launches on the same folded weights: bit-identical; so is the default surface
    for py_plan in (True, False):
        m = _py_model(mt, sd, torch.float16, py_plan)
        with torch.no_grad():
            e_py = m.get_embedding(x.to(DEV)).float().cpu().reshape(emb.shape)
        assert torch.equal(e_py, emb), py_plan


def test_ctypes_only_host_config2_top1_identical_to_oracle(tmp_path, calibrated_sd):
    """BASELINE.json configs[1] through `frmap_model_embed_and_match`: 256 faces, 36-ID enrolment gallery (oracle embeddings),
    bf16; the 32 enrolled probes must return the reference function's top-1."""
    sd = calibrated_sd("cnn")
    enrol = synth.randn(7101, (36, 3, 224, 224), "enrol2")
    probes = enrol[:32] + 0.02 * synth.randn(7102, (32, 3, 224, 224), "pert2")
    fill = synth.randn(7103, (224, 3, 224, 224), "fill2")
    with torch.no_grad():
        gal = F.normalize(fo.cnn_embedding(sd, enrol), dim=1)
        ref_emb = F.normalize(fo.cnn_embedding(sd, probes), dim=1)
    refs = [{"name": f"id{i}", "embedding": gal[i:i + 1]} for i in range(36)]
    ref_ans = [fo.compare_faces(ref_emb[i:i + 1], refs, 1.0) for i in range(32)]
    x = torch.cat([probes, fill])
    out = _run_client(tmp_path, "cnn", "bf16", sd, x, gallery=gal, thresh=1.0, normalize=1)
    assert out["ids"][:32].tolist() == [a[2] for a in ref_ans] == list(range(32))
    assert float(np.abs(out["dist"][:32] - np.array([a[1] for a in ref_ans])).max()) < 5e-3      # config 2's bf16 bound
    # and equal to the Python surface's answer on the same batch, bit for bit
    for py_plan in (True, False):
        m = _py_model("cnn", sd, torch.bfloat16, py_plan)
        with torch.no_grad():
            ids, dists = frmap_amd.embed_and_match(m, x.to(DEV), frmap_amd.Gallery([f"id{i}" for i in range(36)], gal, DEV), 1.0, normalize=True)
        assert ids.cpu().numpy().tolist() == out["ids"].tolist() and np.array_equal(dists.cpu().numpy(), out["dist"]), py_plan


def test_ctypes_only_host_arcface_large_gallery(tmp_path, calibrated_sd):
    """'arcface' through `frmap_model_embed_and_match` against a 10 000-ID gallery (packed: MFMA match path) holding the
    oracle embeddings of 8 enrolled faces."""
    sd = calibrated_sd("arcface")
    enrol = synth.randn(7401, (8, 3, 224, 224), "enrol4")
    probes = enrol + 0.02 * synth.randn(7402, (8, 3, 224, 224), "pert4")
    rows = [1201 * i + 3 for i in range(8)]
    with torch.no_grad():
        gal = synth.unit_rows(3004, 10000, 512)
        gal[rows] = fo.arcface_embedding(sd, enrol)
        ref_emb = fo.arcface_embedding(sd, probes)
    want_i, want_d = fo.match_top1(ref_emb, gal)
    assert want_i.tolist() == rows
    out = _run_client(tmp_path, "arcface", "f16", sd, probes, gallery=gal, thresh=1.0, normalize=0)
    assert out["idx"].tolist() == rows and out["ids"].tolist() == rows
    assert float(np.abs(out["dist"] - want_d.numpy()).max()) < 6e-3


def _write_weights_bin(path, sd):
    import struct
    items = [(k, v.numpy().astype(np.float32).ravel()) for k, v in sd.items() if v.dtype.is_floating_point]
    with open(path, "wb") as f:
        f.write(struct.pack("<i", len(items)))
        for k, a in items:
            kb = k.encode()
            f.write(struct.pack("<i", len(kb))); f.write(kb); f.write(struct.pack("<q", a.size)); f.write(a.tobytes())


@pytest.mark.parametrize("mt,G", [("cnn", 36), ("arcface", 10000)])
def test_plain_c_host_runs_model_and_match(mt, G, tmp_path, gold_dir, calibrated_sd):
    """`examples/cabi_host.c` - a C program (gcc, HIP runtime for device memory, libfrmap_hip.so; no Python anywhere in the
    process) - loads the checkpoint under the reference's keys, embeds the golden inputs and matches them against a gallery
    that holds their oracle embeddings: embeddings vs the reference-generated golden, top-1 = own row."""
    import struct
    host = os.path.join(ROOT, "examples", "cabi_host")
    assert os.path.isfile(host), "examples/cabi_host is built by csrc/build.sh (__graft_entry__.build())"
    z = np.load(os.path.join(gold_dir, f"{mt}.npz"))
    sd = calibrated_sd(mt)
    x = weights.golden_inputs(mt)
    _write_weights_bin(tmp_path / "w.bin", sd)
    with open(tmp_path / "x.bin", "wb") as f:
        f.write(struct.pack("<iii", x.shape[0], x.shape[2], x.shape[3])); f.write(x.numpy().astype(np.float32).tobytes())
    gold = torch.from_numpy(z["embedding"])
    gal = synth.unit_rows(3004, G, 512)
    rows = [(G // 16) * i + 1 for i in range(16)]
    gal[rows] = F.normalize(gold, dim=1)                                       # enrol the reference's own embeddings
    with open(tmp_path / "g.bin", "wb") as f:
        f.write(struct.pack("<iifi", G, 512, 1.0, 1 if mt == "cnn" else 0)); f.write(gal.numpy().astype(np.float32).tobytes())
    r = subprocess.run([host, mt, "36", "f16", str(tmp_path / "w.bin"), str(tmp_path / "x.bin"), str(tmp_path / "g.bin"),
                        str(tmp_path / "o.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    raw = open(tmp_path / "o.bin", "rb").read()
    B, D = struct.unpack("<ii", raw[:8])
    emb = torch.from_numpy(np.frombuffer(raw, np.float32, B * D, 8).reshape(B, D).copy())
    ids = np.frombuffer(raw, np.int32, B, 8 + 4 * B * D)
    dist = np.frombuffer(raw, np.float32, B, 8 + 4 * B * D + 4 * B)
    assert (B, D) == (16, 512)
    assert float((emb - gold).norm() / gold.norm()) < 8e-3
    assert ids.tolist() == rows and float(dist.max()) < 5e-2                   # every face finds its own enrolment
    print(r.stdout.strip())


def test_model_handle_rejections():
    import ctypes as C
    from frmap_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.frmap_model_create(C.byref(h), b"resnet50", 36, 0) == -1
    assert "Invalid model type: resnet50" in lib.frmap_last_error().decode()              # the reference's message, face_models.py:813
    assert lib.frmap_model_create(C.byref(h), b"cnn", 36, 1) == 0
    w = np.zeros(10, dtype=np.float32)
    assert lib.frmap_model_load_tensor(h, b"resnet.conv1.weight", w.ctypes.data, 10, 0) == -1      # wrong size
    assert lib.frmap_model_load_tensor(h, b"resnet.bn1.num_batches_tracked", w.ctypes.data, 1, 0) == 1   # ignored key
    assert lib.frmap_model_finalize(h, None) == -1 and "never loaded" in lib.frmap_last_error().decode()
    x = torch.zeros((1, 3, 224, 224), device=DEV)
    o = torch.zeros((1, 512), device=DEV)
    assert lib.frmap_model_forward(h, x.data_ptr(), 0, 1, 224, 224, 2, o.data_ptr(), o.data_ptr(), None) == -1   # not finalized
    lib.frmap_model_destroy(h)
