"""GPU: BASELINE.json's configs at their FULL sizes against the CPU oracle, through the code paths the benchmark times.

  config 2  cnn, bf16, B = 256, 36-ID gallery: `embed_and_match` (the fused pool + normalise + match tail) and
            `GraphedEmbedMatch` (HIP-graph replay, 2 micro-batch streams) on the full batch; the first 32 faces
            are checked against oracle `cnn_embedding -> normalize -> compare_faces` (enrolment-style gallery:
            probes are perturbed copies of enrolled faces, so top-1 has a real answer), top-1 identical and
            the distance within a stated bound; the other 224 faces through per-face independence.
  config 3  ArcMarginProduct eval head, B = 1024 x 1000 IDs, directly against the oracle restatement of
            `face_models.py:351-429` (whose known answers are the reference class's, tests/golden/arcmargin.npz).
  config 5  hybrid, bf16, B = 256 per GPU: 16 oracle rows + per-face independence at the full batch.
  boundary  the reference's threading pattern (`src/app.py:331-335,639`): a daemon thread runs model(x) in a
            loop while the main thread runs compare_faces; results equal the single-threaded ones.
"""
import math
import queue
import threading

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import frmap_amd  # noqa: E402
from frmap_amd import ops, synth  # noqa: E402
from oracle import face_oracle as fo  # noqa: E402
from oracle import weights  # noqa: E402

DEV = "cuda"


def _model(mt, sd, dtype):
    m = frmap_amd.get_model(mt, 36)
    m.load_state_dict(sd)
    return m.to(DEV).eval().set_compute_dtype(dtype)


# measured on MI355X (printed by the test): max |d - d_oracle| = 2.1e-3 (bf16), 1.3e-4 (fp16); bound = ~2.5x
DIST_BOUND = {torch.bfloat16: 5e-3, torch.float16: 3.5e-4}


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_config2_cnn_b256_top1_and_distance_vs_oracle(dtype, calibrated_sd):
    sd = calibrated_sd("cnn")
    enrol = synth.randn(7101, (36, 3, 224, 224), "enrol2")
    probes = enrol[:32] + 0.02 * synth.randn(7102, (32, 3, 224, 224), "pert2")
    fill = synth.randn(7103, (224, 3, 224, 224), "fill2")
    with torch.no_grad():
        gal = F.normalize(fo.cnn_embedding(sd, enrol), dim=1)                 # oracle enrolment
        ref_emb = F.normalize(fo.cnn_embedding(sd, probes), dim=1)
    refs = [{"name": f"id{i}", "embedding": gal[i:i + 1]} for i in range(36)]
    ref_ans = [fo.compare_faces(ref_emb[i:i + 1], refs, 1.0) for i in range(32)]
    assert [a[2] for a in ref_ans] == list(range(32))                          # every probe finds its own enrolment
    d_all = torch.cdist(ref_emb, gal)
    margin = float((d_all.sort(dim=1).values[:, 1] - d_all.sort(dim=1).values[:, 0]).min())
    assert margin > 4 * DIST_BOUND[dtype], margin                              # the top-1 answer is not a coin flip

    m = _model("cnn", sd, dtype)
    g = frmap_amd.Gallery([f"id{i}" for i in range(36)], gal, DEV)
    x = torch.cat([probes, fill]).to(DEV)                                       # the benchmark's batch: 256 faces
    assert x.shape[0] == 256
    with torch.no_grad():
        ids, dists = frmap_amd.embed_and_match(m, x, g, 1.0, normalize=True)    # fused gap_norm_match tail
        pipe = frmap_amd.GraphedEmbedMatch(m, g, x.clone(), 1.0, normalize=True, streams=2)
        pipe()
        torch.cuda.synchronize()
        ids_g, dists_g = pipe.ids().clone(), pipe.dists().clone()
        ids_half, dists_half = frmap_amd.embed_and_match(m, x[128:], g, 1.0, normalize=True)
    want_ids = torch.tensor([a[2] for a in ref_ans], dtype=torch.int32)
    want_d = torch.tensor([a[1] for a in ref_ans])
    for name, i_, d_ in (("embed_and_match", ids, dists), ("GraphedEmbedMatch", ids_g, dists_g)):
        err = float((d_[:32].cpu() - want_d).abs().max())
        print(f"config 2 {dtype} {name}: top-1 {'identical' if torch.equal(i_[:32].cpu(), want_ids) else 'DIFFERS'}, "
              f"max |dist - oracle| = {err:.2e} (oracle margin {margin:.3f})")
        assert torch.equal(i_[:32].cpu(), want_ids), name
        assert err < DIST_BOUND[dtype], (name, err)
    # graph replay (2 micro-batches of 128 on 2 streams) vs the eager full batch vs a 128-face sub-batch: the same
    # top-1 for all 256 faces; distances equal up to the fp32 summation order (the 3x3 kernel picks its tile layout -
    # pixel split or in-workgroup split-K - from the number of tiles, i.e. from the batch size)
    dtol = 2e-3 if dtype == torch.bfloat16 else 2e-4
    # (the 224 filler faces are unrelated to the gallery: their best and second-best distances can tie to within that
    #  order effect, so their top-1 is required to agree in distance everywhere and in index almost everywhere)
    assert float((dists_g - dists).abs().max()) < dtol and float((ids_g == ids).float().mean()) > 0.97
    assert float((dists_half - dists[128:]).abs().max()) < dtol and float((ids_half == ids[128:]).float().mean()) > 0.97
    assert torch.equal(ids_g[:32], ids[:32])
    assert torch.equal(ids_half, ids_g[128:]) and torch.equal(dists_half, dists_g[128:])   # same batch size: bit for bit


# config 4, measured on MI355X (printed by the test): max |d - d_oracle| = 1.49e-2 (bf16), 2.34e-3 (fp16) on distances of
# 0.1-0.2 (1 - cos to the oracle embedding 1.85e-3 / 3.84e-5; the oracle's top-1 margin is 1.04); bound = 2.5x
DIST_BOUND4 = {torch.bfloat16: 3.7e-2, torch.float16: 6e-3}


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_config4_arcface_b1024_g10000_top1_and_distance_vs_oracle(dtype, calibrated_sd):
    """BASELINE.json configs[3], per-GPU shape: ArcFaceNet eval forward (`face_models.py:573-590`) of 1024 faces matched
    against a 10 000-ID gallery (`app.py:50-64`), on CALIBRATED weights.  32 faces are enrolled with their ORACLE embeddings
    at scattered gallery rows; their perturbed copies sit at scattered positions inside the 1024-face batch (so they pass
    through the kernels and tile layouts only a 1024-face batch selects); `embed_and_match` and `GraphedEmbedMatch`
    must return the reference function's top-1 for each of them, with the distance inside a stated bound."""
    sd = calibrated_sd("arcface")
    enrol = synth.randn(7401, (32, 3, 224, 224), "enrol4")
    probes = enrol + 0.02 * synth.randn(7402, (32, 3, 224, 224), "pert4")
    rows = [311 * i + 17 for i in range(32)]                                    # gallery rows of the enrolled faces (17 .. 9658)
    pos = [31 * i + 5 for i in range(32)]                                       # batch positions of their probes (5 .. 966)
    with torch.no_grad():
        gal = synth.unit_rows(3004, 10000, 512)
        gal[rows] = fo.arcface_embedding(sd, enrol)                             # oracle enrolment (unit-norm)
        ref_emb = fo.arcface_embedding(sd, probes)
    refs = [{"name": f"id{i}", "embedding": gal[i:i + 1]} for i in range(10000)]
    ref_ans = [fo.compare_faces(ref_emb[i:i + 1], refs, 1.0) for i in range(32)]
    assert [a[2] for a in ref_ans] == rows                                      # every probe finds its own enrolment
    d_all = torch.cdist(ref_emb, gal).sort(dim=1).values
    margin = float((d_all[:, 1] - d_all[:, 0]).min())
    assert margin > 4 * DIST_BOUND4[dtype], margin

    m = _model("arcface", sd, dtype)
    g = frmap_amd.Gallery([f"id{i}" for i in range(10000)], gal, DEV)
    gen = torch.Generator(device=DEV); gen.manual_seed(7403)
    x = torch.randn((1024, 3, 224, 224), device=DEV, generator=gen)
    x[pos] = probes.to(DEV)
    want_ids = torch.tensor(rows, dtype=torch.int32)
    want_d = torch.tensor([a[1] for a in ref_ans])
    with torch.no_grad():
        ids, dists = frmap_amd.embed_and_match(m, x, g, 1.0)
        pipe = frmap_amd.GraphedEmbedMatch(m, g, x.clone(), 1.0, streams=2)
        pipe()
        torch.cuda.synchronize()
        ids_g, dists_g = pipe.ids().clone(), pipe.dists().clone()
        emb = m(x)
    cosdev = float((1 - F.cosine_similarity(emb[pos].cpu(), ref_emb, dim=1)).max())
    for name, i_, d_ in (("embed_and_match", ids, dists), ("GraphedEmbedMatch", ids_g, dists_g)):
        err = float((d_[pos].cpu() - want_d).abs().max())
        print(f"config 4 {dtype} {name}: top-1 {'identical' if torch.equal(i_[pos].cpu(), want_ids) else 'DIFFERS'}, "
              f"max |dist - oracle| = {err:.2e} (oracle margin {margin:.3f}), max 1-cos(emb, oracle) = {cosdev:.2e}")
        assert torch.equal(i_[pos].cpu(), want_ids), name
        assert err < DIST_BOUND4[dtype], (name, err)
    assert cosdev < (1e-3 if dtype == torch.float16 else 2e-2)                  # north_star: cosine <= 1e-3 in fp16
    # the filler faces are unrelated to the gallery (best distance ~1.3): above the threshold in both paths
    others = torch.ones(1024, dtype=torch.bool); others[pos] = False
    assert (ids[others.to(DEV)] == -1).all() and (ids_g[others.to(DEV)] == -1).all()


def test_config1_baseline_b64_vs_oracle(calibrated_sd):
    """BASELINE.json configs[0]: BaselineNet forward + get_embedding on 64 x 3 x 224 x 224 (the reference's CPU-runnable case,
    `face_models.py:36-60`), embeddings L2-normalised and matched against a 36-ID gallery - against the fp32 CPU oracle at the
    config's own batch size (the model-parity tests run 16 faces)."""
    sd = calibrated_sd("baseline")
    x = synth.randn(2001, (64, 3, 224, 224), "cfg1.x")
    enrol = x[:36] + 0.05 * synth.randn(3001, (36, 3, 224, 224), "cfg1.enrol")
    with torch.no_grad():
        want_logits, want_emb = fo.baseline_forward(sd, x), fo.baseline_embedding(sd, x)
        gal = F.normalize(fo.baseline_embedding(sd, enrol), dim=1)
    want_idx, want_d = fo.match_top1(F.normalize(want_emb, dim=1), gal)
    d_all = torch.cdist(F.normalize(want_emb, dim=1), gal).sort(dim=1).values
    m = _model("baseline", sd, torch.float16)
    g = frmap_amd.Gallery([f"id{i}" for i in range(36)], gal, DEV)
    with torch.no_grad():
        logits, emb = m(x.to(DEV)), m.get_embedding(x.to(DEV))
        ids, dists = frmap_amd.embed_and_match(m, x.to(DEV), g, 10.0, normalize=True)
    rel_l = float((logits.float().cpu() - want_logits).norm() / want_logits.norm())
    rel_e = float((emb.float().cpu() - want_emb).norm() / want_emb.norm())
    derr = float((dists.cpu() - want_d).abs().max())
    print(f"config 1 baseline fp16 B=64: rel-L2 logits {rel_l:.2e}, embedding {rel_e:.2e}, max |dist - oracle| {derr:.2e}")
    assert rel_l < 5e-3 and rel_e < 5e-3 and derr < 2e-3
    safe = (d_all[:, 1] - d_all[:, 0]) > 4 * derr
    assert torch.equal(ids.cpu()[safe], want_idx[safe]) and int(safe.sum()) >= 36
    assert torch.equal(ids.cpu()[:36], torch.arange(36, dtype=torch.int32))    # every enrolled face finds its own row


def test_config3_arcmargin_head_full_size():
    """B = 1024 embeddings x 1000 class centres, s = 30, m = 0.5 (BASELINE.json configs[2]) in fp32, and the
    label-free cosine top-1 of the same shapes."""
    B, D, C = 1024, 512, 1000
    w = synth.randn(1003, (C, D), tag="arcmargin.weight")
    x = synth.randn(2003, (B, D), tag="cfg3.x")
    lab = torch.from_numpy(np.random.Generator(np.random.PCG64(4003)).integers(0, C, B)).long()
    want = fo.arcmargin_eval(w, x, lab, s=30.0, m=0.5)
    out, mm = ops.arcmargin_eval(x.to(DEV), w.to(DEV), lab.to(DEV), 30.0, 0.5, want_minmax=True)
    err = float((out.cpu() - want).abs().max())
    print(f"config 3: max |logit - oracle| = {err:.2e}")
    assert err < 2e-4                                                          # fp32 MFMA vs fp32 CPU, scale 24
    cos = F.linear(F.normalize(x), F.normalize(w))
    assert abs(float(mm[0]) - float(cos.max())) < 1e-5 and abs(float(mm[1]) - float(cos.min())) < 1e-5
    logits, arg = ops.cosine_logits(x.to(DEV), w.to(DEV), s=1.0)
    rl, ra = fo.class_centre_match(x, w, 1.0)
    assert float((logits.cpu() - rl).abs().max()) < 1e-5
    top2 = rl.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 1e-5
    assert torch.equal(arg.cpu()[safe].long(), ra[safe]) and int(safe.sum()) > 1000
    head = frmap_amd.ArcMarginProduct(D, C, s=30.0, m=0.5).to(DEV).eval()
    with torch.no_grad():
        head.weight.copy_(w.to(DEV))
        assert float((head(x.to(DEV), lab.to(DEV)).cpu() - want).abs().max()) < 2e-4
    st = head.get_margin_stats()
    assert abs(st["max_cos_theta"] - float(cos.max())) < 1e-5


def test_config5_hybrid_b256(calibrated_sd):
    sd = calibrated_sd("hybrid")
    x16 = weights.golden_inputs("hybrid")                                       # 16 faces with oracle rows
    fill = synth.randn(7201, (240, 3, 224, 224), "fill5")
    with torch.no_grad():
        want = fo.hybrid_embedding(sd, x16)
    m = _model("hybrid", sd, torch.bfloat16)
    x = torch.cat([x16, fill]).to(DEV)
    with torch.no_grad():
        e = m.get_embedding(x)
        assert e.shape == (256, 512) and torch.isfinite(e).all()
        cosdev = float((1 - F.cosine_similarity(e[:16].float().cpu(), want, dim=1)).max())
        rel = float((e[:16].float().cpu() - want).norm() / want.norm())
        print(f"config 5 hybrid bf16 B=256: max 1-cos {cosdev:.2e}, rel-L2 {rel:.2e}")
        assert cosdev < 2e-3 and rel < 7e-2                                    # measured 7.9e-4 / 3.3e-2 (bf16)
        # per-face independence at the full batch.  Not bit-exact for this model: the token GEMMs pick their split-K
        # factor from the row count (frmap_linear_mfma), so the fp32 summation order differs between batch sizes.
        for lo, hi in ((0, 16), (100, 101), (250, 256)):
            sub = m.get_embedding(x[lo:hi]).reshape(hi - lo, 512)
            assert float((1 - F.cosine_similarity(sub.float(), e[lo:hi].float(), dim=1)).max()) < 1e-4, (lo, hi)
            assert float((sub.float() - e[lo:hi].float()).abs().max()) < 5e-2, (lo, hi)
        assert m(x).shape == (256, 36)


def test_large_gallery_duplicate_rows_return_first():
    """`app.py:58-62` keeps the FIRST strict minimum.  The large-gallery path (G > 64) takes the arg-min from the
    expanded form on the MFMA; identical rows give bit-identical scores and the (score, index) key keeps the lower
    index, so duplicates resolve as the reference's loop does."""
    G, D = 1000, 512
    gal = synth.unit_rows(3301, G, D)
    gal[700] = gal[3]
    gal[901] = gal[900]
    gal[64] = gal[63]
    probes = torch.cat([gal[[3, 900, 63, 700, 901, 64]], F.normalize(gal[[3, 900]] + 0.01 * synth.randn(3302, (2, D), "n"), dim=1)])
    idx, dist = ops.match_top1(probes.to(DEV), gal.to(DEV))
    assert idx.cpu().tolist() == [3, 900, 63, 3, 900, 63, 3, 900]
    want_i, want_d = fo.match_top1(probes, gal)
    assert want_i.tolist() == idx.cpu().tolist() and torch.allclose(dist.cpu(), want_d, atol=1e-5)


def test_daemon_thread_forward_beside_main_thread_matching(calibrated_sd):
    """`src/app.py:331-335` -> `:236` -> `:44`: model(x) runs on a daemon thread, results cross a queue.Queue;
    the main thread calls compare_faces (`:639`) meanwhile.  Same answers as the single-threaded calls."""
    sd = calibrated_sd("arcface")
    m = _model("arcface", sd, torch.float16)
    x = weights.golden_inputs("arcface", 8).to(DEV)
    gal = synth.unit_rows(3001, 36, 512)
    refs = [{"name": f"id{i}", "embedding": gal[i:i + 1]} for i in range(36)]
    with torch.no_grad():
        base = [m(x[i:i + 1]).clone() for i in range(8)]
        base_ans = [frmap_amd.compare_faces(e, refs, 2.0) for e in base]
    torch.cuda.synchronize()
    q, errors, stop = queue.Queue(), [], threading.Event()

    def producer():
        try:
            with torch.no_grad():
                for rep in range(6):
                    for i in range(8):
                        if stop.is_set():
                            return
                        q.put((i, m(x[i:i + 1])))
        except Exception as e:  # surfaced on the main thread
            errors.append(e)
        finally:
            q.put(None)

    t = threading.Thread(target=producer, daemon=True)
    t.start()
    n = 0
    try:
        while True:
            item = q.get(timeout=120)
            if item is None:
                break
            i, emb = item
            assert torch.equal(emb, base[i]), i                               # the forward is unaffected by the other thread
            assert frmap_amd.compare_faces(emb, refs, 2.0) == base_ans[i], i
            n += 1
    finally:
        stop.set()
        t.join(60)
    assert not errors, errors
    assert n == 48


def test_operands_on_one_device_and_launch_on_their_device():
    a = torch.zeros((2, 8), device=DEV)
    assert torch.equal(ops.l2_normalize(a), a)
    if torch.cuda.device_count() > 1:                                         # multi-GPU hosts only
        b = synth.unit_rows(5, 4, 512).to("cuda:1")
        with torch.cuda.device(0):
            out = ops.l2_normalize(b * 3)                                      # current device 0, operands on 1
        assert out.device == b.device and torch.allclose(out.cpu(), b.cpu(), atol=1e-6)
        with pytest.raises(ValueError, match="different devices"):
            ops.match_top1(b, synth.unit_rows(6, 3, 512).to("cuda:0"))


def test_evaluate_model_harness_on_synthetic_folders(tmp_path, calibrated_sd):
    """§8(f)-2: `evaluate.evaluate_model` (the loop + metrics + JSON of `src/testing.py:26-394`) on a synthetic
    ImageFolder (5 classes x 9 PNG files of assorted sizes; 45 images = one full batch of 32 + a ragged one) against
    the oracle's restatement of the loop on the same resized pixels."""
    import json
    from PIL import Image
    from frmap_amd import evaluate
    g = np.random.Generator(np.random.PCG64(99))
    root = tmp_path / "test"
    for c in range(5):
        d = root / f"person_{c:02d}"
        d.mkdir(parents=True)
        for i in range(9):
            h, w = int(g.integers(120, 300)), int(g.integers(120, 300))
            Image.fromarray(g.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(d / f"img_{i}.png")
    samples, classes = evaluate.image_folder(str(root))
    assert classes == [f"person_{c:02d}" for c in range(5)] and len(samples) == 45
    assert [s[1] for s in samples] == sorted(s[1] for s in samples)

    def cpu_batches():
        for lo in range(0, 45, 32):
            chunk = samples[lo: lo + 32]
            u8 = evaluate.resize_to_u8([Image.open(p) for p, _ in chunk])
            x = (u8.permute(0, 3, 1, 2).float() / 255 - torch.tensor(evaluate.IMAGENET_MEAN).view(1, 3, 1, 1)) / torch.tensor(evaluate.IMAGENET_STD).view(1, 3, 1, 1)
            yield x, torch.tensor([c for _, c in chunk])

    for mt in ("baseline", "arcface"):
        sd = calibrated_sd(mt)
        m = _model(mt, sd, torch.float16)
        out = tmp_path / f"out_{mt}"
        res = evaluate.evaluate_model(m, mt, str(root), out_dir=str(out), model_name=f"{mt}_v1", dataset_name="synthetic")
        ref = fo.evaluate_loop(mt, sd, cpu_batches())
        probs, rprobs = np.array(res["probabilities"]), ref["probabilities"]
        assert probs.shape == rprobs.shape == (45, 36) and res["targets"] == ref["targets"].tolist()
        perr = float(np.abs(probs - rprobs).max())
        assert perr < 3e-3
        top2 = np.sort(rprobs, axis=1)[:, -2:]
        safe = (top2[:, 1] - top2[:, 0]) > 4 * perr                             # decisions the fp16 error cannot flip
        print(f"evaluate_model {mt}: max |p - p_oracle| {perr:.2e}, {int(safe.sum())}/45 decisions outside the error band")
        assert (np.array(res["predictions"])[safe] == ref["predictions"][safe]).all() and safe.sum() >= 10
        assert (np.array(res["predictions"]) == ref["predictions"]).mean() > 0.8
        assert abs(res["test_loss"] - ref["test_loss"]) < 5e-3
        want = evaluate.classification_metrics(ref["targets"], np.array(res["predictions"]), rprobs)
        for k in ("accuracy", "precision", "recall", "f1"):
            assert res["metrics"][k] == pytest.approx(want[k], abs=1e-9)
        assert set(res["metrics"]) == {"accuracy", "precision", "recall", "f1", "roc_auc", "pr_auc", "inference_time"}
        assert res["metrics"]["inference_time"] > 0 and res["class_names"] == classes
        doc = json.load(open(out / f"{mt}_model_results.json" if mt not in ("arcface",) else out / "arcface_model_results.json"))
        assert set(doc) >= {"predictions", "targets", "probabilities", "class_names", "metrics"}       # `testing.py:346-362`
        summ = json.load(open(out / "experiment_summary.json"))
        assert summ["model_type"] == mt and summ["model_name"] == f"{mt}_v1" and summ["dataset"] == "synthetic"
    # ArcFace with an explicit classifier (`testing.py:134-136,262-263`)
    sd = calibrated_sd("arcface")
    m = _model("arcface", sd, torch.float16)
    w, b = synth.randn(31, (36, 512), "clsw") * 0.05, synth.randn(32, (36,), "clsb") * 0.01
    res = evaluate.evaluate_model(m, "arcface", str(root), arcface_classifier=(w, b))
    ref = fo.evaluate_loop("arcface", sd, cpu_batches(), arcface_classifier=(w, b))
    assert np.abs(np.array(res["probabilities"]) - ref["probabilities"]).max() < 3e-3
    # Siamese pairs (`testing.py:170-182`)
    sd = calibrated_sd("siamese")
    m = _model("siamese", sd, torch.float16)
    x = weights.golden_inputs("siamese", 12)
    lab = torch.tensor([1, 0, 1, 0, 1, 0])
    pairs = [(x[:4], x[4:8], lab[:4]), (x[8:10], x[10:12], lab[4:])]
    res = evaluate.evaluate_model(m, "siamese", [(a.to(DEV), b_.to(DEV), l) for a, b_, l in pairs], out_dir=str(tmp_path / "s"))
    ref = fo.evaluate_loop("siamese", sd, pairs)
    assert np.abs(np.array(res["probabilities"]) - ref["probabilities"]).max() < 2e-2
    assert res["predictions"] == ref["predictions"].tolist() and res["class_names"] == ["Same", "Different"]
    assert (tmp_path / "s" / "siamese_network_results.json").exists()
    with pytest.raises(FileNotFoundError):
        evaluate.evaluate_model(m, "baseline", str(tmp_path / "nope_empty_dir_missing"))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_uint8_input_stem_is_bit_identical_to_normalize_then_fp32_path(dtype, calibrated_sd):
    """§8(f)-1: the fused stem fed by the uint8 HWC image (`src/testing.py:99-104` applied while the rows are staged)
    against ToTensor + Normalize as its own kernel followed by the fp32-NCHW stem: same rounding points, so the
    pooled NHWC maps — and everything downstream — are bit-identical; and against the oracle on the normalised floats."""
    from frmap_amd import evaluate
    g = np.random.Generator(np.random.PCG64(4242))
    sd = calibrated_sd("cnn")
    m = _model("cnn", sd, dtype)
    import frmap_amd.face_models as fm
    plan = fm._PyTrunkPlan(m.resnet, dtype)          # the packed stem weights, for the kernel-level comparison below
    for (B, H, W), pool3 in (((3, 224, 224), True), ((2, 160, 160), True), ((2, 112, 96), True), ((2, 224, 224), False), ((1, 64, 72), False)):
        x8 = torch.from_numpy(g.integers(0, 256, (B, H, W, 3), dtype=np.uint8)).to(DEV)
        xf = ops.normalize_u8(x8, evaluate.IMAGENET_MEAN, evaluate.IMAGENET_STD)[0]
        a = ops.stem7x7_maxpool_u8(x8, plan.stem.wpk, plan.stem.shift, evaluate.IMAGENET_MEAN, evaluate.IMAGENET_STD, dtype, pool3=pool3)
        b = ops.stem7x7_maxpool(xf, plan.stem.wpk, plan.stem.shift, dtype, pool3=pool3)
        assert a.shape == b.shape and torch.equal(a, b), (B, H, W, pool3)
    x8 = torch.from_numpy(g.integers(0, 256, (6, 224, 224, 3), dtype=np.uint8)).to(DEV)
    xf = evaluate.preprocess(x8)
    with torch.no_grad():
        assert torch.equal(m.get_embedding(x8), m.get_embedding(xf))
        assert torch.equal(m(x8), m(xf))
        gal = frmap_amd.Gallery([f"id{i}" for i in range(36)], synth.unit_rows(3002, 36, 512), DEV)
        i8, d8 = frmap_amd.embed_and_match(m, x8, gal, 1.5, normalize=True)
        i_f, d_f = frmap_amd.embed_and_match(m, xf, gal, 1.5, normalize=True)
        assert torch.equal(i8, i_f) and torch.equal(d8, d_f)
        want = fo.cnn_embedding(sd, xf.cpu())
        got = m.get_embedding(x8).float().cpu()
        assert float((got - want).norm() / want.norm()) < (2.5e-2 if dtype == torch.bfloat16 else 4e-3)
        # the other first layers: BaselineNet (NHWC4 straight from the bytes), SiameseNet (2x2-pool stem), odd widths (unfused)
        for mt in ("baseline", "siamese"):
            mm = _model(mt, calibrated_sd(mt), dtype)
            assert torch.equal(mm.get_embedding(x8[:3]), mm.get_embedding(xf[:3])), mt
        x8o = torch.from_numpy(g.integers(0, 256, (2, 225, 230, 3), dtype=np.uint8)).to(DEV)
        assert torch.equal(m.get_embedding(x8o), m.get_embedding(evaluate.preprocess(x8o)))
        m.set_input_normalization((0.5, 0.5, 0.5), (0.5, 0.5, 0.5))      # `src/app.py:41`
        assert torch.equal(m.get_embedding(x8[:2]), m.get_embedding(evaluate.preprocess(x8[:2], evaluate.FACENET_MEAN, evaluate.FACENET_STD)))
    with pytest.raises(ValueError):
        m(torch.zeros((2, 3, 224, 224), dtype=torch.uint8, device=DEV))    # uint8 must be HWC


def test_device_resize_is_bit_exact_with_pillow(tmp_path):
    """`frmap_resize_bilinear_u8` (`transforms.Resize((224, 224))` of `src/testing.py:99-100` on the device)
    against Pillow, bit for bit: a ragged batch of images of different sizes in ONE launch (up- and down-scaling, one axis
    unchanged, same size, 1x1, a 20x reduction), the 160x160 target of `src/app.py:39`, and the evaluation harness's
    batches (decoded files -> device resize -> normalise) against the host path."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    from frmap_amd import evaluate, resize
    rng = np.random.default_rng(11)
    shapes = [(224, 224), (300, 400), (112, 112), (1, 1), (2, 3), (500, 37), (37, 500), (225, 223), (900, 1000), (224, 100), (223, 224),
              (4480, 30), (640, 480), (480, 640), (131, 977)]
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]
    for (oh, ow) in [(224, 224), (160, 160), (96, 128)]:
        got = resize.resize_bilinear_u8(imgs, (oh, ow), "cuda").cpu().numpy()
        for i, a in enumerate(imgs):
            ref = np.asarray(Image.fromarray(a).resize((ow, oh), Image.BILINEAR))
            assert np.array_equal(got[i], ref), (shapes[i], oh, ow, int(np.abs(got[i].astype(int) - ref.astype(int)).max()))
    pil = [Image.fromarray(a) for a in imgs[:6]]
    host = evaluate.resize_to_u8(pil, (224, 224))
    dev = evaluate.resize_to_u8(pil, (224, 224), device="cuda")
    assert dev.is_cuda and torch.equal(dev.cpu(), host)
    assert resize.resize_bilinear_u8([], (224, 224), "cuda").shape == (0, 224, 224, 3)
    with pytest.raises(ValueError):
        resize.resize_bilinear_u8([np.zeros((4, 4), np.uint8)], (224, 224), "cuda")


def test_gallery_prepared_pack_lifecycle():
    """`Gallery.prepared` (the fp16-split pack of the MFMA match path): absent below `ops.MATCH_MFMA_MIN_G` rows, built once,
    rebuilt after an in-place edit of the gallery matrix, and `compare_faces` / `match_batch` through it agree with the
    fp32 path on an enrolment-style gallery of 1 500 identities (first index wins on duplicated rows)."""
    from frmap_amd import matching
    small = frmap_amd.Gallery([f"s{i}" for i in range(100)], synth.unit_rows(71, 100, 512, "g"), "cuda")
    assert small.prepared is None
    rows = synth.unit_rows(72, 1500, 512, "g")
    rows[900] = rows[7]                                   # a later duplicate of row 7
    big = frmap_amd.Gallery([f"id{i}" for i in range(1500)], rows, "cuda")
    p1 = big.prepared
    assert p1 is not None and big.prepared is p1          # cached
    probes = (rows[[7, 33, 1499]] + 2e-3 * synth.randn(73, (3, 512), "n")).cuda()
    idx, dist = matching.match_batch(probes, big)
    idx32, dist32 = ops.match_top1(probes, big.matrix)
    assert idx.tolist() == [7, 33, 1499] and torch.equal(idx, idx32) and torch.equal(dist, dist32)
    name, d, i = frmap_amd.compare_faces(probes[1:2].cpu(), big, 1.0)
    assert (name, i) == ("id33", 33) and abs(d - float(dist[1])) < 1e-6
    big.matrix[33] = big.matrix[34]                       # in-place edit: the pack must be rebuilt, row 33 no longer matches itself
    p2 = big.prepared
    assert p2 is not p1
    idx2, _ = matching.match_batch(probes[1:2], big)
    idx2_32, _ = ops.match_top1(probes[1:2], big.matrix)
    assert torch.equal(idx2, idx2_32)


def test_compare_faces_sees_appended_edited_replaced_and_reallocated_refs():
    """`matching._as_gallery` keeps a device copy of the demo's `refs` list between frames (`app.py:639` passes the same list
    object every frame).  Whatever the host does to that list - enrol an identity (`app.py:428-436`: append), edit an enrolled
    embedding in place, replace an entry, free the list and build another - the next `compare_faces` must answer from the
    CURRENT contents, as the reference's loop over the live list does.  Small (exact-scan) and large (MFMA pack) galleries."""
    from frmap_amd import matching
    for G in (7, 600):
        emb = synth.unit_rows(8100 + G, G + 4, 512, "g")
        refs = [{"name": f"id{i}", "embedding": emb[i:i + 1].clone()} for i in range(G)]
        probe = emb[G:G + 1]                                                   # not enrolled yet
        want = fo.compare_faces(probe, refs, 10.0)
        got = frmap_amd.compare_faces(probe.to(DEV), refs, 10.0)
        assert (got[0], got[2]) == (want[0], want[2]) and abs(got[1] - want[1]) < 2e-6
        g0 = matching._gallery_cache[id(refs)][1]
        refs.append({"name": "new", "embedding": probe.clone()})               # enrolment: append
        got = frmap_amd.compare_faces(probe.to(DEV), refs, 10.0)
        assert (got[0], got[2]) == ("new", G) and abs(got[1] - math.sqrt(512) * 1e-6) < 2e-6
        assert matching._gallery_cache[id(refs)][1] is g0 and len(g0) == G + 1  # one row appended on the device, no rebuild
        refs[2]["embedding"].copy_(probe)                                      # in-place edit of an enrolled embedding
        got = frmap_amd.compare_faces(probe.to(DEV), refs, 10.0)
        assert (got[0], got[2]) == ("id2", 2)                                  # first strict minimum: row 2 now ties with the appended row
        refs[1] = {"name": "swapped", "embedding": probe.clone()}              # replaced entry
        got = frmap_amd.compare_faces(probe.to(DEV), refs, 10.0)
        assert (got[0], got[2]) == ("swapped", 1)
        assert fo.compare_faces(probe, refs, 10.0)[2] == 1
        del refs[1:3]                                                          # removal
        want = fo.compare_faces(probe, refs, 10.0)
        got = frmap_amd.compare_faces(probe.to(DEV), refs, 10.0)
        assert (got[0], got[2]) == (want[0], want[2]) == ("new", G - 2)
    for trial in range(6):                                                     # freed-and-reallocated lists (ids may repeat)
        e = synth.unit_rows(8200 + trial, 5, 512, "g")
        lst = [{"name": f"t{trial}_{i}", "embedding": e[i:i + 1]} for i in range(4)]
        got = frmap_amd.compare_faces(e[3:4].to(DEV), lst, 10.0)
        assert (got[0], got[2]) == (f"t{trial}_3", 3)
        del lst


def test_gallery_append_equals_rebuild():
    """`Gallery.append` (one row written, one 64-row tile of the MFMA pack re-packed, capacity doubling) leaves exactly the
    gallery a fresh build holds: same pack bytes, same statistics, same match results - across a capacity regrow and across
    256-row padding boundaries of the pack."""
    from frmap_amd import _lib
    rows = synth.unit_rows(8300, 1400, 512, "g")
    g = frmap_amd.Gallery([f"id{i}" for i in range(1000)], rows[:1000], DEV)
    assert g.prepared is not None
    for i in range(1000, 1400):
        assert g.append(f"id{i}", rows[i]) == i
    fresh = frmap_amd.Gallery([f"id{i}" for i in range(1400)], rows, DEV)
    assert len(g) == 1400 and torch.equal(g.matrix, fresh.matrix)
    nb = _lib.load().frmap_match_gallery_pack_bytes(1400, 512)
    torch.cuda.synchronize()
    assert torch.equal(g.prepared.packed[:nb], fresh.prepared.packed[:nb])
    assert torch.equal(g.prepared.stat_w[:1400], fresh.prepared.stat_w[:1400])
    probes = (rows[[5, 1000, 1023, 1024, 1279, 1280, 1399]] + 1e-3 * synth.randn(8301, (7, 512), "n")).to(DEV)
    idx, dist = frmap_amd.match_batch(probes, g)
    idx_f, dist_f = frmap_amd.match_batch(probes, fresh)
    assert idx.tolist() == [5, 1000, 1023, 1024, 1279, 1280, 1399] and torch.equal(idx, idx_f) and torch.equal(dist, dist_f)
    small = frmap_amd.Gallery([], torch.zeros((0, 1)), DEV)                     # enrolment from an empty gallery
    for i in range(40):
        small.append(f"s{i}", rows[i])
    idx, _ = frmap_amd.match_batch(rows[[0, 17, 39]].to(DEV), small)
    assert idx.tolist() == [0, 17, 39] and small.prepared is None
    with pytest.raises(ValueError):
        small.append("bad", rows[0][:256])


def test_predict_image_mirror(tmp_path, calibrated_sd):
    """`src/testing.py:532-595`: latest `<type>_*` checkpoint directory, class names from `<processed>/<dataset>/train`,
    Resize((224, 224)) -> ToTensor -> Normalize -> forward -> softmax -> max -> (class name, probability); against the oracle on
    the Pillow-resized pixels; the reference's error paths."""
    from PIL import Image
    from frmap_amd import evaluate
    ck, proc = tmp_path / "checkpoints", tmp_path / "processed"
    classes = [f"person_{i:02d}" for i in range(36)]
    for c in classes:
        (proc / "lfw" / "train" / c).mkdir(parents=True)
    g = np.random.Generator(np.random.PCG64(77))
    img = tmp_path / "probe.png"
    Image.fromarray(g.integers(0, 256, (301, 263, 3), dtype=np.uint8)).save(img)
    with pytest.raises(ValueError, match="No trained models found for type: baseline"):
        evaluate.predict_image("baseline", str(img), checkpoints_dir=str(ck), proc_data_dir=str(proc))
    for mt in ("baseline", "cnn"):
        sd = calibrated_sd(mt)
        for ver in ("v1", "v2"):
            (ck / f"{mt}_{ver}").mkdir(parents=True)
        torch.save(sd, ck / f"{mt}_v2" / "best_model.pth")                      # the LATEST directory is the one used (`:538-541`)
        name, prob = evaluate.predict_image(mt, str(img), checkpoints_dir=str(ck), proc_data_dir=str(proc))
        u8 = evaluate.resize_to_u8([Image.open(img)])                           # Pillow on the host
        x = (u8.permute(0, 3, 1, 2).float() / 255 - torch.tensor(evaluate.IMAGENET_MEAN).view(1, 3, 1, 1)) / torch.tensor(evaluate.IMAGENET_STD).view(1, 3, 1, 1)
        with torch.no_grad():
            p = torch.softmax(fo.FORWARD[mt](sd, x), dim=1)
        want_p, want_i = p.max(dim=1)
        top2 = p.topk(2, dim=1).values[0]
        print(f"predict_image {mt}: {name} p={prob:.4f} oracle {classes[int(want_i)]} p={float(want_p):.4f}")
        assert abs(prob - float(want_p)) < 5e-3
        if float(top2[0] - top2[1]) > 1e-2:
            assert name == classes[int(want_i)]
        with pytest.raises(FileNotFoundError):
            evaluate.predict_image(mt, str(img), model_name=f"{mt}_v1", checkpoints_dir=str(ck), proc_data_dir=str(proc))
    with pytest.raises(ValueError, match="Model not found: cnn_v9"):
        evaluate.predict_image("cnn", str(img), model_name="cnn_v9", checkpoints_dir=str(ck), proc_data_dir=str(proc))
    (ck / "siamese_v1").mkdir()
    with pytest.raises(ValueError, match="Siamese model can't be used for direct prediction"):
        evaluate.predict_image("siamese", str(img), checkpoints_dir=str(ck), proc_data_dir=str(proc))
    with pytest.raises(ValueError, match="No processed datasets found"):
        evaluate.predict_image("cnn", str(img), checkpoints_dir=str(ck), proc_data_dir=str(tmp_path / "nowhere"))


def test_evaluate_trained_end_to_end(tmp_path, calibrated_sd):
    """`src/testing.py:26-394` as one call: latest checkpoint directory + first processed dataset with a test split -> the same
    results as handing the loaded model and the folder to `evaluate_model`, and the reference's JSON files under <out>/<model_name>."""
    import json
    from PIL import Image
    from frmap_amd import evaluate
    g = np.random.Generator(np.random.PCG64(123))
    proc, ck, out = tmp_path / "processed", tmp_path / "checkpoints", tmp_path / "outputs"
    test_dir = proc / "cfg" / "lfw" / "test"
    for c in range(36):
        d = test_dir / f"person_{c:02d}"
        d.mkdir(parents=True)
        if c < 6:
            for i in range(2):
                Image.fromarray(g.integers(0, 256, (150 + 9 * c, 130 + 7 * i, 3), dtype=np.uint8)).save(d / f"img_{i}.png")
    (ck / "baseline_v1").mkdir(parents=True); (ck / "baseline_v2").mkdir()
    torch.save(calibrated_sd("baseline"), ck / "baseline_v2" / "best_checkpoint.pth")      # (`:121-127`: the fallback file name)
    res = evaluate.evaluate_trained("baseline", checkpoints_dir=str(ck), proc_data_dir=str(proc), out_root=str(out))
    m = _model("baseline", calibrated_sd("baseline"), torch.bfloat16)
    ref = evaluate.evaluate_model(m, "baseline", str(test_dir))
    assert res["predictions"] == ref["predictions"] and res["targets"] == ref["targets"] and len(res["targets"]) == 12
    assert np.allclose(np.array(res["probabilities"]), np.array(ref["probabilities"]), atol=1e-6)
    summary = json.load(open(out / "baseline_v2" / "experiment_summary.json"))
    assert summary["model_name"] == "baseline_v2" and summary["dataset"] == "cfg/lfw"
    assert (out / "baseline_v2" / "baseline_model_results.json").exists()
