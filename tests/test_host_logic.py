"""CPU: host-side mirror of the reference interface, gallery file format, C-ABI loading."""
import ctypes
import io
import json
import os
import pickle
import re

import numpy as np
import pytest
import torch

import frmap_amd
from frmap_amd import _lib, gallery_io, matching, synth
from frmap_amd import dist as fdist


def test_get_model_types_and_errors():
    assert frmap_amd.MODEL_TYPES == ['baseline', 'cnn', 'siamese', 'attention', 'arcface', 'hybrid', 'ensemble']
    with pytest.raises(ValueError, match="Invalid model type"):       # face_models.py:813
        frmap_amd.get_model("nope")
    att = frmap_amd.get_model("attention", 36)                       # face_models.py:797-799
    assert isinstance(att, frmap_amd.AttentionNet) and att.fc.out_features == 36
    ens = frmap_amd.get_model("ensemble", 36)                        # face_models.py:805-808: cnn + attention + arcface, 'average'
    assert [type(m).__name__ for m in ens.models] == ["ResNetTransfer", "AttentionNet", "ArcFaceNet"]
    assert ens.ensemble_method == "average" and not ens.weights.requires_grad
    ens2 = frmap_amd.get_model(["baseline", "cnn"], 36)               # face_models.py:809-811: a list builds an ensemble
    assert len(ens2.models) == 2
    m = frmap_amd.get_model("arcface", 36)
    assert m.training and next(m.parameters()).device.type == "cpu" and next(m.parameters()).dtype == torch.float32
    with pytest.raises(ValueError, match="Labels must be provided during training"):   # face_models.py:528-529
        m(torch.zeros(1, 3, 224, 224))


def test_state_dict_keys_match_reference(gold_dir):
    table = json.load(open(os.path.join(gold_dir, "state_dict_keys.json")))
    for mt, ref in table.items():
        sd = frmap_amd.get_model(mt, 36).state_dict()
        assert set(sd) == set(ref), (mt, set(sd) ^ set(ref))
        for k, v in sd.items():
            assert list(v.shape) == ref[k][0] and str(v.dtype).replace("torch.", "") == ref[k][1], (mt, k)


def test_no_cpu_fallback():
    m = frmap_amd.get_model("baseline", 36).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(2, 3, 224, 224))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.get_embedding(torch.zeros(2, 3, 224, 224))
    from frmap_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.l2_normalize(torch.zeros(2, 8))


def test_compare_faces_sentinels_without_gpu():
    assert matching.compare_faces(None, [{"name": "a", "embedding": torch.zeros(1, 512)}], 1.0) == ("Unknown", float("inf"), None)
    assert matching.compare_faces(torch.zeros(1, 512), [], 1.0) == ("Unknown", float("inf"), None)


def test_gallery_file_roundtrip_and_reference_format(tmp_path, gold_dir):
    doc = json.load(open(os.path.join(gold_dir, "face_references.json")))
    recs = [{"name": n, "embedding_numpy": np.asarray(doc["embeddings"][i], np.float32)[None, :], "image_path": p}
            for i, (n, p) in enumerate(zip(doc["names"], doc["image_paths"]))]
    f = tmp_path / "face_references.pkl"
    gallery_io.write_gallery_file(str(f), recs)
    back = gallery_io.read_gallery_file(str(f))
    assert [r["name"] for r in back] == doc["names"]
    for a, b in zip(back, recs):
        assert a["embedding_numpy"].shape == (1, 512) and a["embedding_numpy"].dtype == np.float32
        assert np.array_equal(a["embedding_numpy"], b["embedding_numpy"])
    # what the reference's own pickle.load would see is the same structure (our own file: safe to unpickle)
    plain = pickle.load(open(f, "rb"))
    assert isinstance(plain, list) and set(plain[0]) == {"name", "embedding_numpy", "image_path"}


def test_gallery_reader_refuses_code():
    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned",))
    with pytest.raises(gallery_io.UnsafeGalleryError):
        gallery_io.safe_load_pickle(pickle.dumps([{"name": "x", "embedding_numpy": Evil()}], protocol=4))
    with pytest.raises(gallery_io.UnsafeGalleryError):
        gallery_io.safe_load_pickle(pickle.dumps(Evil(), protocol=2))


def test_load_refs_drops_missing_images(tmp_path):
    from PIL import Image
    img = np.zeros((8, 8, 3), np.uint8)
    refs = [{"name": "a b", "embedding": torch.ones(1, 512), "image": img},
            {"name": "c", "embedding": torch.zeros(1, 512), "image": img}]
    f = str(tmp_path / "face_references.pkl")
    assert matching.save_refs(refs, f)
    loaded = matching.load_refs(f)
    assert [r["name"] for r in loaded] == ["a b", "c"] and loaded[0]["embedding"].shape == (1, 512)
    saved = gallery_io.read_gallery_file(f)
    assert os.path.basename(saved[0]["image_path"]).startswith("a_b_")            # app.py:76
    os.remove(saved[1]["image_path"])
    assert [r["name"] for r in matching.load_refs(f)] == ["a b"]                   # app.py:110,119
    assert matching.load_refs(str(tmp_path / "missing.pkl")) == []                 # app.py:105


def test_cabi_exports_every_declared_symbol():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "frmap_hip.h")).read()
    declared = set(re.findall(r"\b(frmap_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert _lib.lib_available(), "libfrmap_hip.so not built (run __graft_entry__.build())"
    lib = _lib.load()
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.frmap_abi_version() == _lib.ABI_VERSION
    # ... and nothing undeclared: every `frmap_*` symbol the library exports with C linkage is in the header
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if ln.split() and ln.split()[-1].startswith("frmap_")}
    assert exported == declared, exported ^ declared
    assert lib.frmap_small_cin_kpad(7, 7) == 240 and lib.frmap_small_cin_kpad(3, 3) == 80   # K + 16: the bank-conflict-free pitch
    assert lib.frmap_head_workspace_bytes(4, 4) >= 8 * 16


def test_shard_bounds_and_packing():
    for total in (0, 1, 7, 8, 8192, 1000):
        for world in (1, 2, 3, 8):
            spans = [fdist.shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    ids = torch.tensor([0, -1, 9999, 2 ** 31 - 1], dtype=torch.int32)
    d = torch.tensor([0.0, float("inf"), 1.25, -3.5e-7])
    i2, d2 = fdist.unpack_results(fdist.pack_results(ids, d))
    assert torch.equal(i2, ids) and torch.equal(d2, d)


def test_synth_is_alias_consistent():
    m = frmap_amd.get_model("arcface", 36)
    sd = synth.synth_state_dict(synth.shapes_of(m), 7)
    assert torch.equal(sd["backbone.layer2.0.conv1.weight"], sd["features.5.0.conv1.weight"])
    assert torch.equal(sd["backbone.bn1.running_var"], sd["features.1.running_var"])
    m.load_state_dict(sd)


def test_missing_extension_fails_loudly(monkeypatch):
    """No fallback: without the built library every compute entry raises, it does not route elsewhere."""
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libfrmap_hip.so")
    assert not _lib.lib_available()
    with pytest.raises(RuntimeError, match="HIP extension not built"):
        _lib.load()


def test_product_package_never_imports_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "facerecognition-multiarchitecture-pipeline_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn


def test_predict_image_discovery_errors_match_the_reference(tmp_path):
    """`src/testing.py:532-558`: the checks that run before any model is built raise the reference's messages (no GPU needed)."""
    from frmap_amd import evaluate
    ck, proc = tmp_path / "checkpoints", tmp_path / "processed"
    ck.mkdir()
    with pytest.raises(ValueError, match="No trained models found for type: cnn"):
        evaluate.predict_image("cnn", "x.png", checkpoints_dir=str(ck), proc_data_dir=str(proc))
    with pytest.raises(ValueError, match="Model not found: cnn_v3"):
        evaluate.predict_image("cnn", "x.png", model_name="cnn_v3", checkpoints_dir=str(ck), proc_data_dir=str(proc))
    (ck / "cnn_v1").mkdir(); (ck / "cnn_v2").mkdir(); (ck / "siamese_v1").mkdir()
    with pytest.raises(ValueError, match="No processed datasets found."):
        evaluate.predict_image("cnn", "x.png", checkpoints_dir=str(ck), proc_data_dir=str(proc))
    (proc / "lfw" / "train" / "alice").mkdir(parents=True)
    with pytest.raises(ValueError, match="Siamese model can't be used for direct prediction"):
        evaluate.predict_image("siamese", "x.png", checkpoints_dir=str(ck), proc_data_dir=str(proc))
    with pytest.raises(FileNotFoundError):                       # the image itself (the latest directory, cnn_v2, was picked)
        evaluate.predict_image("cnn", str(tmp_path / "missing.png"), checkpoints_dir=str(ck), proc_data_dir=str(proc))


def test_evaluate_trained_discovery_matches_the_reference(tmp_path):
    """`src/testing.py:31-70,129`: dataset discovery order / names and the errors raised before any model runs (no GPU needed)."""
    from frmap_amd import evaluate
    proc = tmp_path / "processed"
    for d in ("cfgA/test/x", "cfgB/lfw/test/x", "cfgB/celeb/test/x", "cfgB/notes", "test/x", "train/x"):
        (proc / d).mkdir(parents=True)
    found = evaluate.find_processed_datasets(str(proc))
    assert sorted(n for _, n in found) == sorted(["cfgA", "cfgB/lfw", "cfgB/celeb", "processed (root)"])
    assert found[-1][1] == "processed (root)"                                   # the root's own split comes last (`:62-65`)
    ck = tmp_path / "checkpoints"
    ck.mkdir()
    with pytest.raises(ValueError, match="No trained models found for type: baseline"):
        evaluate.evaluate_trained("baseline", checkpoints_dir=str(ck), proc_data_dir=str(proc))
    (ck / "baseline_v1").mkdir()
    with pytest.raises(ValueError, match="Model not found: baseline_v7"):
        evaluate.evaluate_trained("baseline", "baseline_v7", checkpoints_dir=str(ck), proc_data_dir=str(proc))
    with pytest.raises(ValueError, match="No processed datasets found with test data."):
        evaluate.evaluate_trained("baseline", checkpoints_dir=str(ck), proc_data_dir=str(tmp_path / "empty"))
    with pytest.raises(ValueError, match="dataset_index"):
        evaluate.evaluate_trained("baseline", auto_dataset=False, dataset_index=9, checkpoints_dir=str(ck), proc_data_dir=str(proc))
