"""GPU: whole-model parity of the HIP path (through get_model / forward / get_embedding /
compare_faces) against the committed reference goldens and the CPU oracle, same seeded inputs.

Tolerances (north star): fp16 — embedding 1-cos <= 1e-3 and top-1 identical (measured 2e-5 .. 4e-4);
bf16 — the measured deviation (1.3e-3 .. 2e-2 mean-centred: bf16 cannot meet 1e-3 on decorrelated embeddings,
SURVEY.md §7 hard part 3) is gated at 2.5x its measured value per model (table MEASURED).  Non-normalised / non-negative embeddings (baseline, cnn) are compared by relative L2
and by cosine after removing the batch mean (raw cosine is ~1 for any output there)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import frmap_amd  # noqa: E402
from frmap_amd import matching, synth  # noqa: E402
from oracle import face_oracle as fo  # noqa: E402
from oracle import weights  # noqa: E402

DEV = "cuda"
TOL = {torch.float16: dict(cos=1e-3, rel=1.5e-2), torch.bfloat16: dict(cos=1e-2, rel=6e-2)}   # forward()/unit-embedding checks
# Embedding deviation from the reference goldens as measured on MI355X (round 2, printed by the test):
# (relative L2, max mean-centred 1-cos).  The gate is 2.5x the measured value, so a regression of the arithmetic
# (an accumulation in 16 bits, a dropped rounding step) fails here long before it reaches the north-star bounds.
MEASURED = {
    ("baseline", torch.float16): (5.1e-4, 3.9e-4), ("baseline", torch.bfloat16): (3.7e-3, 2.0e-2),
    ("cnn", torch.float16): (1.1e-3, 3.0e-5), ("cnn", torch.bfloat16): (8.8e-3, 1.8e-3),
    ("arcface", torch.float16): (6.3e-3, 3.2e-5), ("arcface", torch.bfloat16): (5.0e-2, 1.8e-3),
    ("siamese", torch.float16): (4.3e-3, 2.2e-5), ("siamese", torch.bfloat16): (3.2e-2, 1.3e-3),
    ("hybrid", torch.float16): (4.1e-3, 6.0e-5), ("hybrid", torch.bfloat16): (3.3e-2, 4.2e-3),
    ("attention", torch.float16): (2.8e-3, 1.6e-4), ("attention", torch.bfloat16): (2.4e-2, 9.1e-3),
}
GATE = 2.5


def _model(mt, sd, dtype):
    m = frmap_amd.get_model(mt, 36)
    m.load_state_dict(sd)
    return m.to(DEV).eval().set_compute_dtype(dtype)


def _centered_cos(a, b):
    mu = b.mean(0, keepdim=True)
    return F.cosine_similarity(a - mu, b - mu, dim=1)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("mt", ["baseline", "cnn", "arcface", "siamese", "hybrid", "attention"])
def test_model_parity(mt, dtype, gold_dir, calibrated_sd):
    z = np.load(os.path.join(gold_dir, f"{mt}.npz"))
    sd = calibrated_sd(mt)
    x = weights.golden_inputs(mt)
    m = _model(mt, sd, dtype)
    tol = TOL[dtype]
    gold = torch.from_numpy(z["embedding"])
    with torch.no_grad():
        emb = m.get_embedding(x.to(DEV)).float().cpu()
        oracle = fo.EMBEDDING[mt](sd, x)
    assert emb.shape == gold.shape
    for name, ref in (("golden(reference)", gold), ("oracle", oracle)):
        rel = float((emb - ref).norm() / ref.norm())
        cosdev = float((1 - _centered_cos(emb, ref)).max())
        print(f"{mt} {dtype} vs {name}: rel-L2 {rel:.2e}  max(1-cos centred) {cosdev:.2e}")
        m_rel, m_cos = MEASURED[(mt, dtype)]
        assert rel < GATE * m_rel, (name, rel, m_rel)
        assert cosdev < GATE * m_cos, (name, cosdev, m_cos)
    if mt in ("arcface", "siamese"):
        assert float((1 - F.cosine_similarity(emb, gold, dim=1)).max()) < tol["cos"]
        assert torch.allclose(emb.norm(dim=1), torch.ones(emb.shape[0]), atol=1e-4)
    # forward()
    with torch.no_grad():
        if mt == "siamese":
            o1, o2 = m(x[:8].to(DEV), x[8:].to(DEV))
            assert float((1 - F.cosine_similarity(o1.cpu(), torch.from_numpy(z["forward_out1"]), dim=1)).max()) < tol["cos"]
            d = F.pairwise_distance(o1.cpu(), o2.cpu())
            assert torch.allclose(d, torch.from_numpy(z["pair_dist"]), atol=5e-2)
            assert m.get_debug_info()["flattened"] == torch.Size((8, 18432))
        elif mt == "arcface":
            out = m(x.to(DEV)).cpu()
            assert float((1 - F.cosine_similarity(out, torch.from_numpy(z["forward"]), dim=1)).max()) < tol["cos"]
            lab = torch.from_numpy(z["labels"]).long()
            logits = m(x.to(DEV), lab.to(DEV)).cpu()
            assert torch.allclose(logits, torch.from_numpy(z["forward_labels"]), atol=0.05 if dtype == torch.float16 else 0.15)
            e1 = m.get_embedding(x[:1].to(DEV)).cpu()
            assert e1.shape == (1, 512)
        else:
            out = m(x.to(DEV)).cpu()
            ref = torch.from_numpy(z["forward"])
            assert float((out - ref).norm() / ref.norm()) < tol["rel"]
            if mt == "cnn":
                assert m.get_embedding(x[:1].to(DEV)).shape == (512,)       # the reference's .squeeze()


def test_attention_map_and_kernel(calibrated_sd):
    """frmap_cnn_attention against the oracle's AttentionModule on the same (storage-rounded) trunk map: the attended map
    itself, not only its pooled embedding."""
    from frmap_amd import ops
    sd = calibrated_sd("attention")
    x = weights.golden_inputs("attention")[:4]
    m = _model("attention", sd, torch.float16)
    with torch.no_grad():
        amap = m.attention_map(x.to(DEV)).float().cpu().permute(0, 3, 1, 2)       # B×512×7×7
        f = fo.resnet18_trunk(sd, "backbone.", x, pool=False)
        ref = fo.attention_module(sd, "attention.", f)
    assert amap.shape == ref.shape
    rel = float((amap - ref).norm() / ref.norm())
    print(f"attention map rel-L2 {rel:.2e}")
    assert rel < 1.5e-2
    assert m.get_attention_params()["gamma"] == pytest.approx(float(sd["attention.gamma"]))
    # direct kernel call on a 5x5 map with 256 channels and a 3x3 gate (the non-default template paths)
    B, H, W, C, Cq = 3, 5, 5, 256, 32
    xm = synth.randn(71, (B, H, W, C), "xm").half()
    qkv = (synth.randn(72, (B, H, W, 2 * Cq + C), "qkv") * 0.3).half()
    gamma, sw, sb = torch.tensor([0.7]), synth.randn(73, (1, 2, 3, 3), "sw") * 0.3, torch.tensor([0.1])
    om, op = ops.cnn_attention(qkv.to(DEV), xm.to(DEV), gamma.to(DEV), sw.to(DEV), sb.to(DEV), Cq, want_map=True)
    q = qkv[..., :Cq].float().reshape(B, H * W, Cq)
    k = qkv[..., Cq:2 * Cq].float().reshape(B, H * W, Cq)
    v = qkv[..., 2 * Cq:].float().reshape(B, H * W, C)
    att = torch.softmax(torch.bmm(q, k.transpose(1, 2)), dim=-1)
    y = 0.7 * torch.bmm(att, v) + xm.float().reshape(B, H * W, C)
    ymap = y.reshape(B, H, W, C).permute(0, 3, 1, 2)
    pooled = torch.cat([ymap.mean(1, keepdim=True), ymap.max(1, keepdim=True)[0]], dim=1)
    gate = torch.sigmoid(F.conv2d(pooled, sw, sb, padding=1))
    want = (ymap * gate).permute(0, 2, 3, 1)
    assert torch.allclose(om.float().cpu(), want, atol=2e-2, rtol=2e-3)
    assert torch.allclose(op.cpu(), want.reshape(B, H * W, C).mean(1), atol=2e-3, rtol=2e-3)


def test_ensemble_on_gpu(gold_dir, calibrated_sd):
    """EnsembleModel (`face_models.py:843-941`): members run on the HIP path, the merge follows the reference's rules."""
    members = []
    for mt in ("cnn", "attention", "arcface"):
        members.append(_model(mt, calibrated_sd(mt), torch.float16))
    x = weights.golden_inputs("cnn")[:6]
    with torch.no_grad():
        logits = [fo.cnn_forward(calibrated_sd("cnn"), x), fo.attention_forward(calibrated_sd("attention"), x),
                  F.linear(fo.arcface_embedding(calibrated_sd("arcface"), x), F.normalize(calibrated_sd("arcface")["arcface.weight"]))]
    w = torch.tensor([0.3, -0.2, 0.8])
    for method in ("average", "weighted", "max"):
        ens = frmap_amd.EnsembleModel(members, ensemble_method=method).to(DEV).eval()
        with torch.no_grad():
            ens.weights.copy_(w.to(DEV))
            out = ens(x.to(DEV)).float().cpu()
        want = fo.ensemble_combine(logits, method, w)
        assert out.shape == want.shape == (6, 36)
        assert float((out - want).norm() / want.norm()) < 2e-2, method
    ens = frmap_amd.EnsembleModel(members, ensemble_method="attention").to(DEV).eval()
    with pytest.raises(ValueError, match="Unknown ensemble method"):
        ens(x.to(DEV))
    emb = frmap_amd.EnsembleModel(members, "average").to(DEV).eval().get_embedding(x.to(DEV))
    assert emb.shape == (6, 512 * 3)                                   # concatenated member embeddings (:932-934)
    single = frmap_amd.EnsembleModel(members[:1], "max").to(DEV).eval()
    with torch.no_grad():
        assert torch.equal(single(x.to(DEV)), members[0](x.to(DEV)))   # one valid member: returned as is (:902-904)


def test_fused_pool_norm_match_equals_unfused(calibrated_sd):
    """embed_and_match's one-launch tail for ResNetTransfer (pool + normalise + match) against the three-kernel path."""
    from frmap_amd import ops
    sd = calibrated_sd("cnn")
    m = _model("cnn", sd, torch.float16)
    x = weights.golden_inputs("cnn").to(DEV)
    gal = synth.unit_rows(3002, 36, 512).to(DEV)
    with torch.no_grad():
        emb = ops.l2_normalize(m.get_embedding(x), 1e-12)
        idx_u, dist_u, ids_u = ops.match_top1(emb, gal, 1.3)
        ids_f, dist_f = matching.embed_and_match(m, x, matching.Gallery([f"id{i}" for i in range(36)], gal, DEV), 1.3, normalize=True)
        idx2, dist2, ids2, pk, emb2 = ops.gap_norm_match(m.trunk_map(x), gal, 1.3, normalize=True, want_emb=True, packed=True)
    assert torch.equal(ids_f.cpu(), ids_u.cpu()) and torch.equal(idx2.cpu(), idx_u.cpu())
    assert torch.allclose(dist_f.cpu(), dist_u.cpu(), atol=1e-5) and torch.allclose(emb2.cpu(), emb.cpu(), atol=1e-6)
    assert torch.equal(pk[:, 0].cpu(), ids2.cpu()) and torch.equal(pk.view(torch.float32)[:, 1].cpu(), dist2.cpu())
    # raw (unnormalised) pooled embedding, empty gallery -> the "Unknown"/inf sentinel of compare_faces
    i0, d0, s0, _, e0 = ops.gap_norm_match(m.trunk_map(x), None, 1.0, normalize=False, want_emb=True)
    assert torch.allclose(e0.cpu(), m.get_embedding(x).cpu(), atol=1e-6)
    assert bool((i0 == -1).all()) and bool(torch.isinf(d0).all()) and bool((s0 == -1).all())


def test_arcface_top1_identical_fp16(calibrated_sd):
    """Enrolment-style gallery (SURVEY.md §7 hard part 3): gallery = oracle embeddings of 36 faces,
    probes = the same faces perturbed; HIP top-1 and distance must equal the CPU reference's."""
    sd = calibrated_sd("arcface")
    x = synth.randn(7001, (36, 3, 224, 224), "enrol")
    probes = (x[:16] + 0.05 * synth.randn(7002, (16, 3, 224, 224), "pert"))
    with torch.no_grad():
        gal = fo.arcface_embedding(sd, x)
        ref_emb = fo.arcface_embedding(sd, probes)
    refs = [{"name": f"id{i}", "embedding": gal[i:i + 1]} for i in range(36)]
    ref_ans = [fo.compare_faces(ref_emb[i:i + 1], refs, 1.0) for i in range(16)]
    m = _model("arcface", sd, torch.float16)
    with torch.no_grad():
        emb = m(probes.to(DEV))
        ids, dists = frmap_amd.embed_and_match(m, probes.to(DEV), refs, 1.0)
    for i in range(16):
        name, d, idx = frmap_amd.compare_faces(emb[i:i + 1], refs, 1.0)
        assert (name, idx) == (ref_ans[i][0], ref_ans[i][2]), i
        assert abs(d - ref_ans[i][1]) < 2e-2
        assert int(ids[i]) == (ref_ans[i][2] if ref_ans[i][2] is not None else -1)
    assert [a[2] for a in ref_ans] == list(range(16))          # the test is not vacuous: every probe finds its source


def test_compare_faces_on_reference_gallery(gold_dir):
    doc = json.load(open(os.path.join(gold_dir, "face_references.json")))
    emb = torch.tensor(doc["embeddings"], dtype=torch.float32)
    refs = [{"name": n, "embedding": emb[i:i + 1]} for i, n in enumerate(doc["names"])]
    for i in range(7):
        name, d, idx = frmap_amd.compare_faces(emb[i:i + 1].to(DEV), refs, 1.0)
        exp = doc["compare_full_thresh1"][i]
        assert (name, idx) == (exp[0], exp[2]) and abs(d - exp[1]) < 2e-6
        name, d, idx = frmap_amd.compare_faces(emb[i:i + 1], refs[:i] + refs[i + 1:], 1.0)     # CPU probe tensor is accepted
        exp = doc["compare_leave_one_out_thresh1"][i]
        assert name == "Unknown" and idx is None and abs(d - exp[1]) < 1e-5
        name, d, idx = frmap_amd.compare_faces(emb[i:i + 1].to(DEV), refs[:i] + refs[i + 1:], 2.0)
        exp = doc["compare_leave_one_out_thresh2"][i]
        assert (name, idx) == (exp[0], exp[2]) and abs(d - exp[1]) < 1e-5
    assert frmap_amd.compare_faces(None, refs, 1.0) == ("Unknown", float("inf"), None)
    assert frmap_amd.compare_faces(emb[:1].to(DEV), [], 1.0) == ("Unknown", float("inf"), None)


def test_match_goldens_gpu(gold_dir):
    z = np.load(os.path.join(gold_dir, "match.npz"))
    for G in (36, 1000):
        gal = synth.unit_rows(3000 + G, G, 512)
        g = frmap_amd.Gallery([f"id{i}" for i in range(G)], gal, DEV)
        probes = synth.unit_rows(3500 + G, 16, 512, tag="probes")
        idx, dist = frmap_amd.match_batch(probes.to(DEV), g)
        margin = z[f"g{G}_rand_margin"]
        for b in range(16):
            assert int(idx[b]) == int(z[f"g{G}_rand_id"][b]) or margin[b] < 1e-6
        assert np.allclose(dist.cpu().numpy(), z[f"g{G}_rand_dist"], atol=1e-5)
        src = torch.arange(16) * (G // 16)
        pe = F.normalize(gal[src] + 0.02 * synth.randn(3600 + G, (16, 512), tag="noise"), dim=1)
        idx, dist = frmap_amd.match_batch(pe.to(DEV), g)
        assert idx.cpu().tolist() == z[f"g{G}_enrol_id"].tolist()
        assert np.allclose(dist.cpu().numpy(), z[f"g{G}_enrol_dist"], atol=1e-5)


def test_state_dict_reload_invalidates_plan(calibrated_sd):
    sd = calibrated_sd("baseline")
    m = _model("baseline", sd, torch.float16)
    x = weights.golden_inputs("baseline", 2).to(DEV)
    a = m(x).clone()
    sd2 = {k: (v * 1.5 if k == "fc2.weight" else v) for k, v in sd.items()}
    m.load_state_dict(sd2)
    b = m(x)
    assert not torch.allclose(a, b)
    with torch.no_grad():
        m.conv1.weight.mul_(0.5)                # in-place edit: version counter bump -> re-pack
    c = m(x)
    assert not torch.allclose(b, c)
    m.train()
    with pytest.raises(NotImplementedError):
        m(x)


def test_eval_step_and_siamese_verify(gold_dir, calibrated_sd):
    """The callers around the path: testing.py's forward→softmax→argmax step, the
    ArcFace class-centre scoring, and the Siamese dist<0.5 decision, against the CPU oracle."""
    from frmap_amd import evaluate
    sd = calibrated_sd("baseline")
    x = weights.golden_inputs("baseline", 8)
    m = _model("baseline", sd, torch.float16)
    out, probs, pred = evaluate.predict_batch(m, x.to(DEV), "baseline")
    ref = fo.baseline_forward(sd, x)
    assert torch.allclose(probs.cpu(), F.softmax(ref, 1), atol=2e-3)
    assert pred.cpu().tolist() == ref.argmax(1).tolist()
    sd = calibrated_sd("arcface")
    x = weights.golden_inputs("arcface", 8)
    m = _model("arcface", sd, torch.float16)
    logits, arg = evaluate.arcface_validate(m, x.to(DEV))
    rl, ra = fo.class_centre_match(fo.arcface_embedding(sd, x), sd["arcface.weight"], 32.0)
    assert torch.allclose(logits.cpu(), rl, atol=0.05)
    top2 = rl.topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]          # random class centres: some rows are near-ties
    out, probs, pred = evaluate.predict_batch(m, x.to(DEV), "arcface")
    for b in range(8):
        if margin[b] > 0.1:                   # decisions must agree wherever fp16 error cannot flip them
            assert int(arg[b]) == int(ra[b]) == int(pred[b]), b
    assert (margin > 0.1).sum() >= 4
    assert int(arg.cpu()[0]) == int(logits.cpu()[0].argmax())   # arg-max is consistent with its own logits
    sd = calibrated_sd("siamese")
    x = weights.golden_inputs("siamese", 8)
    m = _model("siamese", sd, torch.float16)
    z = np.load(os.path.join(gold_dir, "siamese.npz"))
    d, p = evaluate.siamese_verify(m, x[:4].to(DEV), x[4:8].to(DEV))
    rd, rp = fo.siamese_decision(*fo.siamese_forward(sd, x[:4], x[4:8]))
    assert torch.allclose(d.cpu(), rd, atol=2e-2) and p.cpu().tolist() == rp.tolist()


def test_graphed_multistream_pipeline_matches_eager(calibrated_sd):
    """HIP-graph replay with the batch split over 2 concurrent streams returns exactly what the eager
    single-stream call returns (same kernels, same per-face arithmetic), and tracks new inputs."""
    sd = calibrated_sd("cnn")
    m = _model("cnn", sd, torch.bfloat16)
    gal = frmap_amd.Gallery([f"id{i}" for i in range(36)], synth.unit_rows(3002, 36, 512), DEV)
    x = synth.randn(8101, (12, 3, 224, 224), "gx").to(DEV)
    ids0, d0 = frmap_amd.embed_and_match(m, x, gal, 1.2, normalize=True)
    pipe = frmap_amd.GraphedEmbedMatch(m, gal, x.clone(), 1.2, normalize=True, streams=2)
    pipe()
    assert torch.equal(pipe.ids(), ids0) and torch.equal(pipe.dists(), d0)
    x2 = synth.randn(8102, (12, 3, 224, 224), "gx2").to(DEV)
    ids1, d1 = frmap_amd.embed_and_match(m, x2, gal, 1.2, normalize=True)
    pipe(x2)
    assert torch.equal(pipe.ids(), ids1) and torch.equal(pipe.dists(), d1)
    assert not torch.equal(d0, d1)


def test_app_get_embedding_mirror(calibrated_sd):
    """`app.py:32-48`: BGR crop -> 160x160 -> [-1,1] -> model; None on empty input or any failure."""
    from PIL import Image
    sd = calibrated_sd("arcface")
    m = _model("arcface", sd, torch.float16)
    g = np.random.Generator(np.random.PCG64(77))
    bgr = g.integers(0, 256, (97, 83, 3), dtype=np.uint8)
    emb = frmap_amd.get_embedding(bgr, m)
    assert emb.shape == (1, 512) and emb.is_cuda
    pil = Image.fromarray(np.ascontiguousarray(bgr[:, :, ::-1])).resize((160, 160), Image.BILINEAR)
    x = (torch.from_numpy(np.array(pil)).permute(2, 0, 1).float().div(255) - 0.5) / 0.5
    ref = fo.arcface_embedding(sd, x.unsqueeze(0))
    assert float(1 - F.cosine_similarity(emb.cpu(), ref, dim=1)) < 1e-3
    assert frmap_amd.get_embedding(None, m) is None
    assert frmap_amd.get_embedding(np.zeros((0, 0, 3), np.uint8), m) is None
    assert frmap_amd.get_embedding(np.zeros((5, 5), np.uint8), m) is None          # wrong rank -> exception -> None
