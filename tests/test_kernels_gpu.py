"""GPU: every HIP kernel, called through the C ABI, against a plain PyTorch fp32 reference of the
same op computed on the CPU from the SAME rounded operands (so the only differences are fp32
accumulation order and the final rounding to the storage dtype)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from frmap_amd import ops, synth  # noqa: E402

DEV = "cuda"
DTYPES = [torch.float16, torch.bfloat16]


def _nhwc(x_nchw):
    return x_nchw.permute(0, 2, 3, 1).contiguous()


def _tol(dtype):
    # one rounding of the output to the storage dtype + fp32 accumulation noise
    return (2e-3, 2e-3) if dtype == torch.float16 else (1.6e-2, 1.6e-2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,Cin,Cout,dsC,s", [
    (3, 28, 28, 128, 128, 64, 2),      # layer2.0: conv2 + downsample(56x56x64, stride 2)
    (5, 14, 14, 256, 256, 128, 2),     # layer3.0
    (9, 7, 7, 512, 512, 256, 2),       # layer4.0: many images per tile, 8 shortcut stages
    (2, 14, 14, 128, 64, 128, 1),      # stride-1 projection (ds_Cin == Cin: a shortcut stage after every main chunk)
])
def test_conv_igemm_fused_shortcut(dtype, B, H, W, Cin, Cout, dsC, s):
    """frmap_conv_igemm_ds == conv3x3(h) + conv1x1_stride(x) + shift, ReLU (torchvision BasicBlock with downsample)."""
    h = synth.randn(21, (B, Cin, H, W), "h").to(dtype)
    xd = synth.randn(22, (B, dsC, (H - 1) * s + 1 + (s - 1), (W - 1) * s + 1 + (s - 1)), "xd").to(dtype)
    w = (synth.randn(23, (Cout, Cin, 3, 3), "w") * math.sqrt(2.0 / (Cin * 9))).to(dtype)
    wd = (synth.randn(24, (Cout, dsC, 1, 1), "wd") * math.sqrt(1.0 / dsC)).to(dtype)
    shift = synth.randn(25, (Cout,), "b") * 0.1
    ref = F.relu(F.conv2d(h.float(), w.float(), None, padding=1) + F.conv2d(xd.float(), wd.float(), None, stride=s)
                 + shift.view(1, -1, 1, 1))
    assert ops.conv_ds_supported(B, H, W, Cin, Cout, xd.shape[2], xd.shape[3], dsC, s)
    y = ops.conv_igemm_ds(_nhwc(h).to(DEV), ops.pack_conv_weight(w.float().to(DEV), dtype), shift.to(DEV), Cout,
                          _nhwc(xd).to(DEV), ops.pack_conv_weight(wd.float().to(DEV), dtype), s, True)
    y = y.float().cpu().permute(0, 3, 1, 2)
    atol, rtol = _tol(dtype)
    assert y.shape == ref.shape
    assert torch.allclose(y, ref, atol=atol, rtol=rtol), (y - ref).abs().max()
    # shapes the fused kernel does not take are reported, not run
    assert not ops.conv_ds_supported(B, H, W, Cin, Cout, xd.shape[2], xd.shape[3], 2 * Cin, s)   # more shortcut stages than main chunks
    assert not ops.conv_ds_supported(B, H, W, Cin, Cout, xd.shape[2] + 4, xd.shape[3], dsC, s)    # geometry mismatch


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,res,relu", [
    (2, 56, 56, 64, 64, 3, 1, True, True),       # resnet layer1 (wave-autonomous weights-resident kernel)
    (3, 16, 24, 64, 128, 3, 1, True, True),      # same kernel: 2 channel tiles, 18 patches (fewer than waves)
    (1, 8, 8, 64, 64, 3, 1, False, False),       # same kernel: a single 8x8 patch
    (5, 24, 8, 64, 64, 3, 1, False, True),       # same kernel: one patch column
    (2, 20, 56, 64, 64, 3, 1, True, True),       # H not a multiple of 8 -> register-prefetch kernel
    (3, 56, 56, 64, 128, 3, 2, False, True),     # layer2.0.conv1 (stride 2)
    (3, 56, 56, 64, 128, 1, 2, False, False),    # layer2.0.downsample
    (5, 28, 28, 128, 128, 3, 1, True, True),
    (7, 14, 14, 256, 256, 3, 1, False, False),
    (9, 7, 7, 512, 512, 3, 1, True, True),       # many images per tile, ragged tail (441 px)
    (2, 14, 14, 256, 512, 3, 2, False, True),
    (1, 112, 112, 32, 64, 3, 1, False, True),    # BaselineNet conv2
    (2, 10, 6, 32, 64, 3, 1, False, False),      # odd small geometry
    (37, 1, 1, 1024, 512, 1, 1, False, True),    # Linear as 1x1
])
def test_conv_igemm(dtype, B, H, W, Cin, Cout, k, s, res, relu):
    pad = 1 if k == 3 else 0
    x = synth.randn(11, (B, Cin, H, W), "x").to(dtype)
    w = (synth.randn(12, (Cout, Cin, k, k), "w") * math.sqrt(2.0 / (Cin * k * k))).to(dtype)
    shift = synth.randn(13, (Cout,), "b") * 0.1
    y_ref = F.conv2d(x.float(), w.float(), None, stride=s, padding=pad) + shift.view(1, -1, 1, 1)
    r = None
    if res:
        r = synth.randn(14, tuple(y_ref.shape), "r").to(dtype)
        y_ref = y_ref + r.float()
    if relu:
        y_ref = F.relu(y_ref)
    wpk = ops.pack_conv_weight(w.float().to(DEV), dtype)
    y = ops.conv_igemm(_nhwc(x).to(DEV), wpk, shift.to(DEV), Cout, k, s, pad, relu,
                       _nhwc(r).to(DEV) if r is not None else None)
    y = y.float().cpu().permute(0, 3, 1, 2)
    atol, rtol = _tol(dtype)
    assert y.shape == y_ref.shape
    assert torch.allclose(y, y_ref, atol=atol, rtol=rtol), (y - y_ref).abs().max()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,Cout,k,s,pad", [(2, 224, 224, 64, 7, 2, 3), (3, 64, 48, 64, 7, 2, 3),
                                                 (2, 224, 224, 32, 3, 1, 1), (3, 20, 12, 32, 3, 1, 1)])
def test_conv_small_cin(dtype, B, H, W, Cout, k, s, pad):
    x = synth.randn(21, (B, 3, H, W), "x")
    w = (synth.randn(22, (Cout, 3, k, k), "w") * math.sqrt(2.0 / (3 * k * k))).to(dtype)
    shift = synth.randn(23, (Cout,), "b") * 0.1
    x4 = ops.pack_input(x.to(DEV), dtype)
    # pack_input itself: NHWC4 with a zero 4th channel, rounded to dtype
    x4c = x4.float().cpu()
    assert torch.equal(x4c[..., :3], _nhwc(x).to(dtype).float()) and float(x4c[..., 3].abs().max()) == 0.0
    y_ref = F.relu(F.conv2d(x.to(dtype).float(), w.float(), None, stride=s, padding=pad) + shift.view(1, -1, 1, 1))
    y = ops.conv_small_cin(x4, ops.pack_conv_weight_c3(w.float().to(DEV), dtype), shift.to(DEV), Cout, k, s, pad, True)
    y = y.float().cpu().permute(0, 3, 1, 2)
    atol, rtol = _tol(dtype)
    assert torch.allclose(y, y_ref, atol=atol, rtol=rtol), (y - y_ref).abs().max()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W", [(3, 224, 224), (2, 112, 96), (2, 61, 37), (1, 200, 224)])
def test_fused_stem_maxpool(dtype, B, H, W):
    x = synth.randn(25, (B, 3, H, W), "x")
    w = (synth.randn(26, (64, 3, 7, 7), "w") * math.sqrt(2.0 / 147)).to(dtype)
    shift = synth.randn(27, (64,), "b") * 0.3
    conv = F.relu(F.conv2d(x.to(dtype).float(), w.float(), None, stride=2, padding=3) + shift.view(1, -1, 1, 1))
    ref = F.max_pool2d(conv.to(dtype).float(), 3, 2, 1)       # unfused path rounds the conv map first; max commutes
    y = ops.stem7x7_maxpool(x.to(DEV), ops.pack_conv_weight_c3(w.float().to(DEV), dtype), shift.to(DEV), dtype)
    y = y.float().cpu().permute(0, 3, 1, 2)
    atol, rtol = _tol(dtype)
    assert y.shape == ref.shape
    assert torch.allclose(y, ref, atol=atol, rtol=rtol), (y - ref).abs().max()


@pytest.mark.parametrize("B,H,W", [(1, 224, 224), (2, 61, 36), (1, 226, 220), (2, 34, 200), (300, 64, 64), (7, 160, 160), (3, 35, 8)])
def test_stem_s2d_segments_borders_and_many_units(B, H, W):
    """`stem_s2d_kernel` (W % 4 == 0) on the shapes its structure branches on: one image cut into 19 row segments (every
    segment but the first computes its carry row alone), odd conv heights / widths (the row / column past the last conv
    position acts as -inf: masking path), a single column half, several units per persistent workgroup, tiny maps; fp32 and the
    uint8 twin; against fp32 torch on the same rounded operands."""
    dtype = torch.float16
    x = synth.randn(125, (B, 3, H, W), "x")
    w = (synth.randn(126, (64, 3, 7, 7), "w") * math.sqrt(2.0 / 147)).to(dtype)
    shift = synth.randn(127, (64,), "b") * 0.3
    conv = F.relu(F.conv2d(x.to(dtype).float(), w.float(), None, stride=2, padding=3) + shift.view(1, -1, 1, 1))
    ref = F.max_pool2d(conv.to(dtype).float(), 3, 2, 1)
    wpk = ops.pack_conv_weight_c3(w.float().to(DEV), dtype)
    y = ops.stem7x7_maxpool(x.to(DEV), wpk, shift.to(DEV), dtype).float().cpu().permute(0, 3, 1, 2)
    atol, rtol = _tol(dtype)
    assert y.shape == ref.shape
    assert torch.allclose(y, ref, atol=atol, rtol=rtol), (y - ref).abs().max()
    if B <= 8:   # uint8 twin: bit-identical to normalise -> fp32 entry
        g = np.random.Generator(np.random.PCG64(128))
        x8 = torch.from_numpy(g.integers(0, 256, (B, H, W, 3), dtype=np.uint8)).to(DEV)
        mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
        xf = ops.normalize_u8(x8, mean, std)[0]
        a = ops.stem7x7_maxpool_u8(x8, wpk, shift.to(DEV), mean, std, dtype)
        b = ops.stem7x7_maxpool(xf.to(dtype).float(), wpk, shift.to(DEV), dtype)
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W", [(3, 224, 224), (2, 100, 84), (2, 61, 37)])
def test_fused_stem_maxpool2x2(dtype, B, H, W):
    x = synth.randn(28, (B, 3, H, W), "x")
    w = (synth.randn(29, (64, 3, 7, 7), "w") * math.sqrt(2.0 / 147)).to(dtype)
    shift = synth.randn(30, (64,), "b") * 0.3
    conv = F.relu(F.conv2d(x.to(dtype).float(), w.float(), None, stride=2, padding=3) + shift.view(1, -1, 1, 1))
    ref = F.max_pool2d(conv.to(dtype).float(), 2, 2)
    y = ops.stem7x7_maxpool(x.to(DEV), ops.pack_conv_weight_c3(w.float().to(DEV), dtype), shift.to(DEV), dtype, pool3=False)
    y = y.float().cpu().permute(0, 3, 1, 2)
    atol, rtol = _tol(dtype)
    assert y.shape == ref.shape, (y.shape, ref.shape)
    assert torch.allclose(y, ref, atol=atol, rtol=rtol), (y - ref).abs().max()


@pytest.mark.parametrize("dtype", DTYPES)
def test_pools(dtype):
    x = synth.randn(31, (3, 64, 30, 22), "x").to(dtype)
    xd = _nhwc(x).to(DEV)
    for k, s, p in ((3, 2, 1), (2, 2, 0)):
        y = ops.maxpool(xd, k, s, p).float().cpu().permute(0, 3, 1, 2)
        assert torch.equal(y, F.max_pool2d(x.float(), k, s, p))          # exact: max of representable values
    g = ops.avgpool_global(xd).cpu()
    assert torch.allclose(g, x.float().mean(dim=(2, 3)), atol=1e-5)
    x2 = synth.randn(32, (2, 512, 14, 14), "x").to(dtype)
    a = ops.avgpool_adaptive(_nhwc(x2).to(DEV), 6, 6).float().cpu().permute(0, 3, 1, 2)
    atol, rtol = _tol(dtype)
    assert torch.allclose(a, F.adaptive_avg_pool2d(x2.float(), (6, 6)), atol=atol, rtol=rtol)
    for dt in (dtype,):
        t = synth.randn(33, (1000,), "c")
        assert torch.equal(ops.cast_from_f32(t.to(DEV), dt).cpu(), t.to(dt))
        assert torch.equal(ops.cast_to_f32(t.to(dt).to(DEV)).cpu(), t.to(dt).float())


@pytest.mark.parametrize("B,K,N,relu", [(1, 512, 36, False), (70, 128, 512, True), (256, 512, 512, False), (3, 512, 1000, False)])
def test_linear_f32_and_normalize(B, K, N, relu):
    x = synth.randn(41, (B, K), "x")
    w = synth.randn(42, (N, K), "w") / math.sqrt(K)
    sc = synth.randn(43, (N,), "s").abs() + 0.5
    sh = synth.randn(44, (N,), "h")
    ref = (x.double() @ w.double().t()) * sc.double() + sh.double()
    if relu:
        ref = ref.clamp_min(0)
    y = ops.linear_f32(x.to(DEV), w.to(DEV), sc.to(DEV), sh.to(DEV), relu).cpu()
    assert torch.allclose(y.double(), ref, atol=2e-5, rtol=2e-5), (y.double() - ref).abs().max()
    y2 = ops.linear_f32(x.to(DEV), w.to(DEV)).cpu()
    assert torch.allclose(y2.double(), x.double() @ w.double().t(), atol=2e-5, rtol=2e-5)
    n = ops.l2_normalize(x.to(DEV), 1e-12).cpu()
    assert torch.allclose(n, F.normalize(x, p=2, dim=1, eps=1e-12), atol=1e-6)
    z = torch.zeros(2, K)
    assert torch.equal(ops.l2_normalize(z.to(DEV), 1e-12).cpu(), z)       # x / max(0, eps) = 0, no NaN


def _ref_match(e, g):
    d = torch.sqrt(((e[:, None, :].double() - g[None].double() + 1e-6) ** 2).sum(-1))
    dist, idx = d.min(1)
    return idx.int(), dist.float(), d


@pytest.mark.parametrize("B,G,D", [(1, 7, 512), (16, 36, 512), (100, 1000, 512), (257, 1300, 512), (5, 129, 256)])
def test_match_top1(B, G, D):
    gal = synth.unit_rows(51, G, D)
    pr = synth.unit_rows(52, B, D, tag="p")
    pr[0] = gal[G // 2]                     # exact hit: distance sqrt(D)*1e-6
    idx, dist = ops.match_top1(pr.to(DEV), gal.to(DEV))
    ridx, rdist, dm = _ref_match(pr, gal)
    idx, dist = idx.cpu(), dist.cpu()
    top2 = dm.topk(2, dim=1, largest=False).values if G > 1 else None
    for b in range(B):
        if idx[b] != ridx[b]:               # only a sub-ulp near-tie may differ
            assert top2 is not None and float(top2[b, 1] - top2[b, 0]) < 1e-6, (b, idx[b], ridx[b])
    assert int(idx[0]) == G // 2 and abs(float(dist[0]) - math.sqrt(D) * 1e-6) < 2e-6
    assert torch.allclose(dist, rdist, atol=2e-6, rtol=1e-5)
    thr = float(rdist.median())
    i2, d2, ids = ops.match_top1(pr.to(DEV), gal.to(DEV), thr)
    assert torch.equal(i2.cpu(), idx) and torch.equal(d2.cpu(), dist)
    assert torch.equal(ids.cpu(), torch.where(dist <= thr, idx, torch.full_like(idx, -1)))
    pk = ops.match_top1(pr.to(DEV), gal.to(DEV), thr, packed=True)[3].cpu()
    assert torch.equal(pk[:, 0], ids.cpu()) and torch.equal(pk.view(torch.float32)[:, 1], dist)


def test_match_top1_edge_cases():
    D = 512
    pr = synth.unit_rows(61, 3, D)
    idx, dist = ops.match_top1(pr.to(DEV), torch.zeros((0, D), device=DEV))
    assert idx.cpu().tolist() == [-1, -1, -1] and torch.isinf(dist).all()      # empty gallery
    gal = synth.unit_rows(62, 9, D)
    gal[6] = gal[2]                                                                # duplicate rows: FIRST minimum wins
    idx, _ = ops.match_top1(gal[6:7].to(DEV), gal.to(DEV))
    assert int(idx[0]) == 2


def test_cosine_logits_and_arcmargin(gold_dir):
    import os
    z = np.load(os.path.join(gold_dir, "arcmargin.npz"))
    w = synth.randn(1003, (1000, 512), tag="arcmargin.weight")
    x = synth.randn(2003, (32, 512), tag="arcmargin.x")
    lab = torch.from_numpy(z["labels"]).long()
    for name, kw in (("s30_m05", dict(s=30.0, m=0.5)), ("s32_m05", dict(s=32.0, m=0.5)),
                     ("s16_m03_easy", dict(s=16.0, m=0.3, easy_margin=True))):
        out, mm = ops.arcmargin_eval(x.to(DEV), w.to(DEV), lab.to(DEV), kw["s"], kw["m"], kw.get("easy_margin", False),
                                     want_minmax=True)
        ref = torch.from_numpy(z[name])
        assert torch.allclose(out.cpu(), ref, atol=2e-4), (name, (out.cpu() - ref).abs().max())
    cos = F.linear(F.normalize(x), F.normalize(w))
    assert abs(float(mm[0]) - float(cos.max())) < 1e-5 and abs(float(mm[1]) - float(cos.min())) < 1e-5
    logits, arg = ops.cosine_logits(x.to(DEV), w.to(DEV), s=32.0)
    assert torch.allclose(logits.cpu(), cos * 32.0, atol=1e-4)
    assert arg.cpu().tolist() == (cos * 32.0).argmax(1).tolist()
    _, arg2 = ops.cosine_logits(x.to(DEV), w.to(DEV), s=1.0, want_logits=False)
    assert arg2.cpu().tolist() == arg.cpu().tolist()


@pytest.mark.parametrize("dtype", DTYPES)
def test_transformer_token_kernels(dtype):
    B, L, D, H = 5, 49, 512, 4
    atol, rtol = _tol(dtype)
    x = synth.randn(71, (B, L, D), "x").to(dtype)
    pos = synth.randn(72, (L, D), "p") * 0.1
    g1, b1 = synth.randn(73, (D,), "g").abs() + 0.5, synth.randn(74, (D,), "b") * 0.1
    t, y = ops.add_pos_layernorm(x.to(DEV), pos.to(DEV), g1.to(DEV), b1.to(DEV), want_sum=True)
    t_ref = (x.float() + pos).to(dtype)
    assert torch.equal(t.cpu(), t_ref)
    y_ref = F.layer_norm(t_ref.float(), (D,), g1, b1, 1e-5)
    assert torch.allclose(y.float().cpu(), y_ref, atol=atol * 2, rtol=rtol)
    _, y2 = ops.add_pos_layernorm(x.to(DEV), None, g1.to(DEV), b1.to(DEV))
    assert torch.allclose(y2.float().cpu(), F.layer_norm(x.float(), (D,), g1, b1, 1e-5), atol=atol * 2, rtol=rtol)
    # attention
    qkv = (synth.randn(75, (B, L, 3 * D), "qkv") * 0.5).to(dtype)
    o = ops.mha_tokens(qkv.to(DEV), H).float().cpu()
    q, k, v = qkv.float().split(D, dim=-1)
    hd = lambda z: z.view(B, L, H, D // H).transpose(1, 2)
    att = torch.softmax(hd(q) @ hd(k).transpose(-1, -2) / math.sqrt(D // H), dim=-1) @ hd(v)
    o_ref = att.transpose(1, 2).reshape(B, L, D)
    assert torch.allclose(o, o_ref, atol=atol, rtol=rtol), (o - o_ref).abs().max()
    # token mean + LayerNorm
    m = ops.mean_layernorm(x.to(DEV), g1.to(DEV), b1.to(DEV)).cpu()
    assert torch.allclose(m, F.layer_norm(x.float().mean(1), (D,), g1, b1, 1e-5), atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_gelu_residual_on_conv_kernel(dtype):
    M, K, N = 3 * 49, 512, 2048
    x = synth.randn(81, (M, K), "x").to(dtype)
    w = (synth.randn(82, (N, K), "w") / math.sqrt(K)).to(dtype)
    bias = synth.randn(83, (N,), "b") * 0.1
    wpk = ops.pack_conv_weight(w.float().view(N, K, 1, 1).to(DEV), dtype)
    y = ops.conv_igemm(x.view(M, 1, 1, K).to(DEV), wpk, bias.to(DEV), N, 1, 1, 0, 2).view(M, N).float().cpu()
    ref = F.gelu(x.float() @ w.float().t() + bias)
    atol, rtol = _tol(dtype)
    assert torch.allclose(y, ref, atol=atol, rtol=rtol), (y - ref).abs().max()
    w2 = (synth.randn(84, (K, N), "w2") / math.sqrt(N)).to(dtype)
    r = synth.randn(85, (M, K), "r").to(dtype)
    wpk2 = ops.pack_conv_weight(w2.float().view(K, N, 1, 1).to(DEV), dtype)
    y2 = ops.conv_igemm(y.to(dtype).view(M, 1, 1, N).to(DEV), wpk2, torch.zeros(K, device=DEV), K, 1, 1, 0, 0,
                        r.view(M, 1, 1, K).to(DEV)).view(M, K).float().cpu()
    ref2 = y.to(dtype).float() @ w2.float().t() + r.float()
    assert torch.allclose(y2, ref2, atol=atol * 2, rtol=rtol), (y2 - ref2).abs().max()


def test_preprocess_and_eval_glue():
    from frmap_amd import evaluate
    g = np.random.Generator(np.random.PCG64(91))
    img = torch.from_numpy(g.integers(0, 256, (3, 40, 36, 3), dtype=np.uint8))
    for mean, std in ((evaluate.IMAGENET_MEAN, evaluate.IMAGENET_STD), (evaluate.FACENET_MEAN, evaluate.FACENET_STD)):
        ref = (img.permute(0, 3, 1, 2).float().div(255) - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
        out = evaluate.preprocess(img, mean, std)
        assert torch.allclose(out.cpu(), ref, atol=1e-6, rtol=1e-6), (out.cpu() - ref).abs().max()
        _, x4 = ops.normalize_u8(img.to(DEV), mean, std, want_nchw=False, nhwc4_dtype=torch.float16)
        assert torch.equal(x4[..., :3].float().cpu(), ref.permute(0, 2, 3, 1).to(torch.float16).float())
        assert float(x4[..., 3].abs().max()) == 0.0
    logits = synth.randn(92, (37, 36), "l") * 3
    logits[5, 7] = logits[5, 11] = logits[5].max() + 1.0           # tie: first index wins (torch.max semantics)
    probs, pred = ops.softmax_argmax(logits.to(DEV))
    assert torch.allclose(probs.cpu(), F.softmax(logits, dim=1), atol=1e-6)
    assert pred.cpu().tolist() == logits.argmax(1).tolist() and int(pred[5]) == 7
    a, b = synth.randn(93, (19, 256), "a"), synth.randn(94, (19, 256), "b")
    d, same = ops.pairwise_distance(a.to(DEV), b.to(DEV), 22.6)
    dref = F.pairwise_distance(a, b)
    assert torch.allclose(d.cpu(), dref, rtol=1e-6) and same.cpu().tolist() == (dref < 22.6).int().tolist()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,K,N,act,res", [(256, 18432, 1024, 1, False), (37, 1024, 512, 1, False), (5, 512, 256, 0, True),
                                           (300, 2048, 512, 2, True)])
def test_linear_mfma_splitk(dtype, M, K, N, act, res):
    x = synth.randn(95, (M, K), "x").to(dtype)
    w = (synth.randn(96, (N, K), "w") / math.sqrt(K)).to(dtype)
    shift = synth.randn(97, (N,), "b") * 0.1
    r = synth.randn(98, (M, N), "r").to(dtype) if res else None
    ref = x.float() @ w.float().t() + shift
    if res:
        ref = ref + r.float()
    ref = F.relu(ref) if act == 1 else (F.gelu(ref) if act == 2 else ref)
    wpk = ops.pack_conv_weight(w.float().view(N, K, 1, 1).to(DEV), dtype)
    y = ops.linear_mfma(x.to(DEV), wpk, shift.to(DEV), N, act, r.to(DEV) if res else None).float().cpu()
    atol, rtol = _tol(dtype)
    assert torch.allclose(y, ref, atol=atol, rtol=rtol), (y - ref).abs().max()


def test_conv_igemm_random_geometries():
    """40 seeded random geometries (odd sizes, ragged tiles, every kernel variant: register-prefetch,
    generic, stride-2 row-parity split and its odd-height fallback, 1x1 gather) against fp32 torch."""
    rng = np.random.Generator(np.random.PCG64(2024))
    for case in range(40):
        dtype = DTYPES[case % 2]
        k = int(rng.choice([1, 3, 3, 3]))
        s_ = int(rng.choice([1, 1, 2]))
        B = int(rng.integers(1, 6))
        H, W = int(rng.integers(3, 40)), int(rng.integers(3, 40))
        Cin, Cout = int(rng.choice([32, 64, 96])), int(rng.choice([64, 128, 192]))
        pad = 1 if k == 3 else 0
        relu, res = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        x = synth.randn(3000 + case, (B, Cin, H, W), "x").to(dtype)
        w = (synth.randn(4000 + case, (Cout, Cin, k, k), "w") * math.sqrt(2.0 / (Cin * k * k))).to(dtype)
        shift = synth.randn(5000 + case, (Cout,), "b") * 0.1
        ref = F.conv2d(x.float(), w.float(), None, stride=s_, padding=pad) + shift.view(1, -1, 1, 1)
        r = synth.randn(6000 + case, tuple(ref.shape), "r").to(dtype) if res else None
        if res:
            ref = ref + r.float()
        if relu:
            ref = F.relu(ref)
        y = ops.conv_igemm(_nhwc(x).to(DEV), ops.pack_conv_weight(w.float().to(DEV), dtype), shift.to(DEV), Cout, k, s_, pad,
                           relu, _nhwc(r).to(DEV) if res else None).float().cpu().permute(0, 3, 1, 2)
        atol, rtol = _tol(dtype)
        assert y.shape == ref.shape, (case, y.shape, ref.shape)
        assert torch.allclose(y, ref, atol=atol, rtol=rtol), (case, B, H, W, Cin, Cout, k, s_, float((y - ref).abs().max()))


def test_conv_wave_and_shortcut_random_geometries():
    """Seeded random shapes for the two newest kernels: the wave-autonomous Cin=64 kernel (H, W multiples of 8, 1-3
    channel tiles, with/without residual and activation incl. GELU) and the fused projection shortcut."""
    rng = np.random.Generator(np.random.PCG64(77))
    for case in range(16):
        dtype = DTYPES[case % 2]
        B = int(rng.integers(1, 7))
        H, W = 8 * int(rng.integers(1, 6)), 8 * int(rng.integers(1, 6))
        Cout = int(rng.choice([64, 128, 192]))
        act, res = int(rng.integers(0, 3)), bool(rng.integers(0, 2))
        x = synth.randn(7000 + case, (B, 64, H, W), "x").to(dtype)
        w = (synth.randn(7100 + case, (Cout, 64, 3, 3), "w") * math.sqrt(2.0 / 576)).to(dtype)
        shift = synth.randn(7200 + case, (Cout,), "b") * 0.1
        ref = F.conv2d(x.float(), w.float(), None, padding=1) + shift.view(1, -1, 1, 1)
        r = synth.randn(7300 + case, tuple(ref.shape), "r").to(dtype) if res else None
        if res:
            ref = ref + r.float()
        ref = F.relu(ref) if act == 1 else (F.gelu(ref) if act == 2 else ref)
        y = ops.conv_igemm(_nhwc(x).to(DEV), ops.pack_conv_weight(w.float().to(DEV), dtype), shift.to(DEV), Cout, 3, 1, 1, act,
                           _nhwc(r).to(DEV) if res else None).float().cpu().permute(0, 3, 1, 2)
        atol, rtol = _tol(dtype)
        assert torch.allclose(y, ref, atol=atol, rtol=rtol), ("wave", case, B, H, W, Cout, act, res, float((y - ref).abs().max()))
    for case in range(12):
        dtype = DTYPES[case % 2]
        B = int(rng.integers(1, 6))
        H, W = int(rng.integers(3, 30)), int(rng.integers(3, 30))
        Cin = int(rng.choice([64, 96, 128]))
        dsC = int(rng.choice([c for c in (32, 64, 96, 128) if c <= Cin]))
        Cout, s_ = int(rng.choice([64, 128])), int(rng.choice([1, 2]))
        Hd, Wd = (H - 1) * s_ + 1 + int(rng.integers(0, s_)), (W - 1) * s_ + 1 + int(rng.integers(0, s_))
        if not ops.conv_ds_supported(B, H, W, Cin, Cout, Hd, Wd, dsC, s_):
            continue
        h = synth.randn(8000 + case, (B, Cin, H, W), "h").to(dtype)
        xd = synth.randn(8100 + case, (B, dsC, Hd, Wd), "xd").to(dtype)
        w = (synth.randn(8200 + case, (Cout, Cin, 3, 3), "w") * math.sqrt(2.0 / (Cin * 9))).to(dtype)
        wd = (synth.randn(8300 + case, (Cout, dsC, 1, 1), "wd") * math.sqrt(1.0 / dsC)).to(dtype)
        shift = synth.randn(8400 + case, (Cout,), "b") * 0.1
        ref = F.relu(F.conv2d(h.float(), w.float(), None, padding=1) + F.conv2d(xd.float(), wd.float(), None, stride=s_)
                     + shift.view(1, -1, 1, 1))
        y = ops.conv_igemm_ds(_nhwc(h).to(DEV), ops.pack_conv_weight(w.float().to(DEV), dtype), shift.to(DEV), Cout,
                              _nhwc(xd).to(DEV), ops.pack_conv_weight(wd.float().to(DEV), dtype), s_, True)
        y = y.float().cpu().permute(0, 3, 1, 2)
        atol, rtol = _tol(dtype)
        assert torch.allclose(y, ref, atol=atol, rtol=rtol), ("ds", case, B, H, W, Cin, dsC, Cout, s_, float((y - ref).abs().max()))


def test_rejections_do_not_launch():
    with pytest.raises(ValueError):
        ops.conv_igemm(torch.zeros(1, 8, 8, 48, device=DEV, dtype=torch.float16),
                       torch.zeros(48 * 64 * 9, device=DEV, dtype=torch.float16), torch.zeros(64, device=DEV),
                       64, 3, 1, 1, True)            # Cin not a multiple of 32
    with pytest.raises(TypeError):
        ops.pack_input(torch.zeros(1, 3, 8, 8, device=DEV), torch.float32)


def test_conv_pp_kernel_shapes_and_tilings():
    """The LDS-DMA ping-pong 3x3 stride-1 kernel (conv_pp.hip) against fp32 torch: the ResNet layer-2/3/4 shapes, ragged
    batches (tiles that cross image boundaries, a partial last tile), odd geometries, both channel tiles (128 / 256),
    forced tile sizes that are not whole rows, residual / activation variants, one vs many 32-channel chunks; and the
    first-generation kernels on the same inputs (the two generations must agree to rounding)."""
    from frmap_amd import _lib
    lib = _lib.load()
    cases = [  # B, H, W, Cin, Cout, residual, act, tile_px, bn
        (3, 14, 14, 256, 256, False, 1, -1, -1), (5, 14, 14, 256, 256, True, 1, -1, 256), (5, 14, 14, 256, 256, True, 1, -1, 128),
        (9, 7, 7, 512, 512, True, 1, -1, -1), (9, 7, 7, 512, 512, False, 0, -1, 256), (2, 28, 28, 128, 128, True, 1, -1, -1),
        (3, 28, 28, 128, 128, False, 1, 196, -1), (2, 13, 17, 128, 128, True, 2, -1, -1), (1, 56, 56, 128, 128, False, 1, -1, -1),
        (2, 9, 40, 160, 384, True, 1, -1, -1), (4, 14, 14, 256, 256, False, 1, 100, -1), (7, 5, 3, 32, 128, True, 0, -1, -1),
        (2, 3, 224, 64, 128, False, 1, -1, -1), (33, 7, 7, 128, 256, True, 1, 37, 256), (1, 1, 1, 1024, 128, False, 1, -1, -1),
        (2, 56, 56, 64, 128, True, 1, -1, -1),
        # bn = 1282: 128-channel tiles with the two wave groups splitting K (accumulators merged through LDS)
        (9, 7, 7, 512, 512, True, 1, -1, 1282), (5, 14, 14, 256, 256, False, 1, -1, 1282), (2, 28, 28, 128, 128, True, 2, -1, 1282),
        (3, 13, 17, 128, 256, True, 0, -1, 1282), (33, 7, 7, 128, 256, True, 1, 37, 1282), (2, 20, 12, 64, 128, False, 1, -1, 1282),
        (2, 9, 40, 160, 384, True, 1, -1, 1282),   # odd chunk count: split-K not applicable, falls back to the pixel split
    ]
    try:
        for ci, (B, H, W, Cin, Cout, res, act, px, bn) in enumerate(cases):
            for dtype in DTYPES:
                x = synth.randn(9100 + ci, (B, Cin, H, W), "x").to(dtype)
                w = (synth.randn(9200 + ci, (Cout, Cin, 3, 3), "w") * math.sqrt(2.0 / (Cin * 9))).to(dtype)
                shift = synth.randn(9300 + ci, (Cout,), "b") * 0.1
                ref = F.conv2d(x.float(), w.float(), None, stride=1, padding=1) + shift.view(1, -1, 1, 1)
                r = synth.randn(9400 + ci, tuple(ref.shape), "r").to(dtype) if res else None
                if res:
                    ref = ref + r.float()
                ref = F.relu(ref) if act == 1 else (F.gelu(ref) if act == 2 else ref)
                wpk = ops.pack_conv_weight(w.float().to(DEV), dtype)
                xin, rin = _nhwc(x).to(DEV), (_nhwc(r).to(DEV) if res else None)
                lib.frmap_conv_pp_tuning(1, px, bn)
                lib.frmap_conv_pp_ri(0)
                y_pp = ops.conv_igemm(xin, wpk, shift.to(DEV), Cout, 3, 1, 1, act, rin)
                lib.frmap_conv_pp_ri(1)      # fragment reads interleaved with the MFMAs: same arithmetic, same order
                y_ri = ops.conv_igemm(xin, wpk, shift.to(DEV), Cout, 3, 1, 1, act, rin)
                lib.frmap_conv_pp_tuning(0, -1, -1)
                y_g1 = ops.conv_igemm(xin, wpk, shift.to(DEV), Cout, 3, 1, 1, act, rin)
                atol, rtol = _tol(dtype)
                y = y_pp.float().cpu().permute(0, 3, 1, 2)
                assert y.shape == ref.shape
                assert torch.allclose(y, ref, atol=atol, rtol=rtol), (ci, dtype, float((y - ref).abs().max()))
                assert torch.allclose(y_pp.float(), y_g1.float(), atol=atol, rtol=rtol), (ci, dtype, "generations differ")
                assert torch.equal(y_ri, y_pp), (ci, dtype, "interleaved-read form differs", float((y_ri.float() - y_pp.float()).abs().max()))
    finally:
        lib.frmap_conv_pp_tuning(-1, -1, -1)
        lib.frmap_conv_pp_ri(-1)


def test_conv_pp_stride2_kernel():
    """The stride-2 form of the LDS-DMA ping-pong kernel (space-to-depth addressing: four half-resolution halo images
    per 32-channel chunk, one to four taps each) against fp32 torch and against the first-generation row-parity kernel:
    the three ResNet stride-2 shapes, ragged batches, tiles crossing image boundaries, odd aspect ratios, forced tile
    sizes, both channel tiles, residual / activation variants."""
    from frmap_amd import _lib
    lib = _lib.load()
    cases = [  # B, H, W, Cin, Cout, residual, act, tile_px, bn
        (3, 28, 28, 128, 256, False, 1, -1, -1), (2, 56, 56, 64, 128, False, 1, -1, -1), (5, 14, 14, 256, 512, False, 1, -1, -1),
        (5, 14, 14, 256, 512, True, 0, -1, 128), (2, 20, 36, 96, 128, True, 1, -1, -1), (4, 8, 6, 32, 128, False, 2, -1, -1),
        (33, 14, 14, 128, 256, True, 1, 37, 256), (3, 28, 28, 128, 256, False, 1, 100, 128), (1, 2, 2, 64, 128, False, 1, -1, -1),
        (2, 4, 112, 32, 128, True, 1, -1, -1), (7, 28, 12, 64, 256, False, 1, -1, -1),
    ]
    try:
        for ci, (B, H, W, Cin, Cout, res, act, px, bn) in enumerate(cases):
            for dtype in DTYPES:
                x = synth.randn(9500 + ci, (B, Cin, H, W), "x").to(dtype)
                w = (synth.randn(9600 + ci, (Cout, Cin, 3, 3), "w") * math.sqrt(2.0 / (Cin * 9))).to(dtype)
                shift = synth.randn(9700 + ci, (Cout,), "b") * 0.1
                ref = F.conv2d(x.float(), w.float(), None, stride=2, padding=1) + shift.view(1, -1, 1, 1)
                r = synth.randn(9800 + ci, tuple(ref.shape), "r").to(dtype) if res else None
                if res:
                    ref = ref + r.float()
                ref = F.relu(ref) if act == 1 else (F.gelu(ref) if act == 2 else ref)
                wpk = ops.pack_conv_weight(w.float().to(DEV), dtype)
                xin, rin = _nhwc(x).to(DEV), (_nhwc(r).to(DEV) if res else None)
                assert lib.frmap_conv_pp_tuning(1, px, bn) == 0
                assert lib.frmap_conv3x3s2_pp_layout(B, H, W, Cin, Cout) in (1, 2), (ci, "not taken")
                y_pp = ops.conv_igemm(xin, wpk, shift.to(DEV), Cout, 3, 2, 1, act, rin)
                atol, rtol = _tol(dtype)
                y = y_pp.float().cpu().permute(0, 3, 1, 2)
                assert y.shape == ref.shape
                assert torch.allclose(y, ref, atol=atol, rtol=rtol), (ci, dtype, float((y - ref).abs().max()))
                if min(H, W) >= 4:   # (the first-generation stride-2 kernels do not take 2-pixel-wide inputs)
                    lib.frmap_conv_pp_tuning(0, -1, -1)
                    y_g1 = ops.conv_igemm(xin, wpk, shift.to(DEV), Cout, 3, 2, 1, act, rin)
                    assert torch.allclose(y_pp.float(), y_g1.float(), atol=atol, rtol=rtol), (ci, dtype, "generations differ")
    finally:
        lib.frmap_conv_pp_tuning(-1, -1, -1)


def test_conv_pp_fused_shortcut():
    """conv3x3_pp_kernel<..., DS = true>: BasicBlock.conv2 + projection shortcut (extra one-tap phases fed by a gathered
    image of the strided input) against fp32 torch and the first-generation fused kernel: the three ResNet shapes, both
    pixel-split layouts, ragged batches, a stride-1 projection, shortcut depths of 1..8 chunks, forced tile sizes."""
    from frmap_amd import _lib
    lib = _lib.load()
    cases = [  # B, H, W, Cin, Cout, dsC, s, act, tile_px, bn
        (3, 28, 28, 128, 128, 64, 2, 1, -1, -1), (5, 14, 14, 256, 256, 128, 2, 1, -1, -1), (9, 7, 7, 512, 512, 256, 2, 1, -1, 256),
        (2, 14, 14, 128, 256, 128, 1, 1, -1, -1), (5, 14, 14, 256, 256, 128, 2, 0, -1, 128), (4, 10, 6, 64, 128, 32, 2, 2, -1, -1),
        (33, 7, 7, 128, 256, 96, 2, 1, 37, 256), (2, 28, 28, 128, 128, 128, 1, 1, 300, -1), (1, 3, 5, 256, 128, 256, 2, 1, -1, -1),
    ]
    try:
        for ci, (B, H, W, Cin, Cout, dsC, sd, act, px, bn) in enumerate(cases):
            for dtype in DTYPES:
                h = synth.randn(9900 + ci, (B, Cin, H, W), "h").to(dtype)
                xd = synth.randn(9910 + ci, (B, dsC, (H - 1) * sd + 1 + (sd - 1), (W - 1) * sd + 1 + (sd - 1)), "xd").to(dtype)
                w = (synth.randn(9920 + ci, (Cout, Cin, 3, 3), "w") * math.sqrt(2.0 / (Cin * 9))).to(dtype)
                wd = (synth.randn(9930 + ci, (Cout, dsC, 1, 1), "wd") * math.sqrt(1.0 / dsC)).to(dtype)
                shift = synth.randn(9940 + ci, (Cout,), "b") * 0.1
                ref = F.conv2d(h.float(), w.float(), None, padding=1) + F.conv2d(xd.float(), wd.float(), None, stride=sd) + shift.view(1, -1, 1, 1)
                ref = F.relu(ref) if act == 1 else (F.gelu(ref) if act == 2 else ref)
                assert ops.conv_ds_supported(B, H, W, Cin, Cout, xd.shape[2], xd.shape[3], dsC, sd)
                args = (_nhwc(h).to(DEV), ops.pack_conv_weight(w.float().to(DEV), dtype), shift.to(DEV), Cout,
                        _nhwc(xd).to(DEV), ops.pack_conv_weight(wd.float().to(DEV), dtype), sd, act)
                lib.frmap_conv_pp_tuning(1, px, bn)
                assert lib.frmap_conv3x3_pp_ds_layout(B, H, W, Cin, Cout, xd.shape[2], xd.shape[3], dsC, sd) in (1, 2), (ci, "not taken")
                y_pp = ops.conv_igemm_ds(*args)
                lib.frmap_conv_pp_tuning(0, -1, -1)
                y_g1 = ops.conv_igemm_ds(*args)
                atol, rtol = _tol(dtype)
                y = y_pp.float().cpu().permute(0, 3, 1, 2)
                assert y.shape == ref.shape
                assert torch.allclose(y, ref, atol=atol, rtol=rtol), (ci, dtype, float((y - ref).abs().max()))
                assert torch.allclose(y_pp.float(), y_g1.float(), atol=atol, rtol=rtol), (ci, dtype, "generations differ")
    finally:
        lib.frmap_conv_pp_tuning(-1, -1, -1)


def test_conv_fused_maxpool2x2():
    """conv 3x3 s1 p1 + shift + ReLU + MaxPool2d(2, 2) in one launch (`face_models.py:38-40,121-141`; pool-major pixel
    order + conv_epilogue_pool2) against fp32 torch and against the two-launch path: the BaselineNet / SiameseNet shapes,
    windows that straddle tile and image boundaries, ragged batches, no activation, the Cin = 3 first layer."""
    cases = [  # B, H, W, Cin, Cout, relu
        (2, 112, 112, 32, 64, 1), (3, 56, 56, 64, 128, 1), (5, 6, 6, 64, 64, 1), (3, 12, 20, 32, 64, 0), (7, 2, 2, 32, 64, 1),
        (2, 56, 56, 128, 128, 1), (3, 28, 28, 256, 256, 1), (1, 4, 130, 32, 128, 1), (9, 14, 10, 96, 192, 1),
        (5, 8, 24, 32, 64, 1), (3, 16, 8, 64, 192, 0), (37, 8, 8, 64, 64, 1),   # 8-aligned maps with Cin 32 / 64: the wave-autonomous kernel
        # row pairs that tile a 112-pixel slice, Cin >= 128, Cout % 128 == 0: the ping-pong kernel (both layouts, ragged tails,
        # tiles straddling images, every admissible width)
        (5, 14, 14, 128, 256, 1), (3, 8, 8, 128, 128, 1), (9, 4, 4, 256, 128, 0), (33, 2, 2, 128, 128, 1), (3, 28, 28, 160, 384, 1),
        (1, 56, 56, 128, 256, 1), (7, 6, 14, 192, 128, 1),
    ]
    forms = {0: 2, 1: 2, 5: 3, 6: 3, 2: 1, 12: 3, 13: 3, 15: 3}
    for ci, (B, H, W, Cin, Cout, relu) in enumerate(cases):
        for dtype in DTYPES:
            h = synth.randn(9700 + ci, (B, Cin, H, W), "h").to(dtype)
            w = (synth.randn(9710 + ci, (Cout, Cin, 3, 3), "w") * math.sqrt(2.0 / (Cin * 9))).to(dtype)
            shift = synth.randn(9720 + ci, (Cout,), "b") * 0.1
            ref = F.conv2d(h.float(), w.float(), None, padding=1) + shift.view(1, -1, 1, 1)
            ref = F.max_pool2d(F.relu(ref) if relu else ref, 2, 2)
            assert ops.conv_pool2_form(B, H, W, Cin, Cout) == forms.get(ci, ops.conv_pool2_form(B, H, W, Cin, Cout)) != 0, (ci, "form")
            x, wpk, sh = _nhwc(h).to(DEV), ops.pack_conv_weight(w.float().to(DEV), dtype), shift.to(DEV)
            y = ops.conv_igemm_pool2(x, wpk, sh, Cout, relu)
            y2 = ops.maxpool(ops.conv_igemm(x, wpk, sh, Cout, 3, 1, 1, relu), 2, 2, 0)
            atol, rtol = _tol(dtype)
            yc = y.float().cpu().permute(0, 3, 1, 2)
            assert yc.shape == ref.shape
            assert torch.allclose(yc, ref, atol=atol, rtol=rtol), (ci, dtype, float((yc - ref).abs().max()))
            assert torch.allclose(y.float(), y2.float(), atol=atol, rtol=rtol), (ci, dtype, "fused and two-launch paths differ")
    for ci, (B, H, W) in enumerate([(2, 224, 224), (3, 10, 14), (1, 2, 2), (5, 36, 18)]):   # BaselineNet conv1 (Cin = 3 as NHWC4)
        for dtype in DTYPES:
            h = synth.randn(9750 + ci, (B, 3, H, W), "h").to(dtype)
            w = (synth.randn(9760 + ci, (32, 3, 3, 3), "w") * math.sqrt(2.0 / 27)).to(dtype)
            shift = synth.randn(9770 + ci, (32,), "b") * 0.1
            ref = F.max_pool2d(F.relu(F.conv2d(h.float(), w.float(), None, padding=1) + shift.view(1, -1, 1, 1)), 2, 2)
            x4 = torch.zeros(B, H, W, 4, dtype=dtype)
            x4[..., :3] = h.permute(0, 2, 3, 1)
            x4, wpk, sh = x4.to(DEV), ops.pack_conv_weight_c3(w.float().to(DEV), dtype), shift.to(DEV)
            y = ops.conv_small_cin_pool2(x4, wpk, sh, 32, True)
            y2 = ops.maxpool(ops.conv_small_cin(x4, wpk, sh, 32, 3, 1, 1, True), 2, 2, 0)
            atol, rtol = _tol(dtype)
            yc = y.float().cpu().permute(0, 3, 1, 2)
            assert torch.allclose(yc, ref, atol=atol, rtol=rtol), ("c3", ci, dtype, float((yc - ref).abs().max()))
            assert torch.equal(y, y2), ("c3", ci, dtype, "fused and two-launch paths differ")
    with pytest.raises(ValueError):
        ops.conv_igemm_pool2(torch.zeros(1, 5, 6, 32, dtype=torch.bfloat16, device=DEV),
                             torch.zeros(64 * 32 * 9, dtype=torch.bfloat16, device=DEV), torch.zeros(64, device=DEV), 64, True)


@pytest.mark.parametrize("dtype", DTYPES)
def test_fused_arcface_head(dtype):
    """`frmap_gap_linear_norm` (avgpool -> Linear(no bias) -> BatchNorm1d eval -> F.normalize, `face_models.py:573-590`) against
    fp32 torch and against the three-launch path, ragged batch sizes (the kernel walks 8 faces per workgroup)."""
    for ci, (B, H, W, K, N) in enumerate([(1, 7, 7, 512, 512), (19, 7, 7, 512, 512), (8, 3, 5, 256, 256), (33, 1, 1, 64, 512), (5, 14, 14, 2048, 256)]):
        fmap = torch.relu(synth.randn(9800 + ci, (B, H, W, K), "m")).to(dtype)
        w = synth.randn(9810 + ci, (N, K), "w") * (1.0 / math.sqrt(K))
        scale = 1.0 + 0.1 * synth.randn(9820 + ci, (N,), "s")
        shift = 0.1 * synth.randn(9830 + ci, (N,), "b")
        pooled = fmap.float().mean(dim=(1, 2))
        pre_ref = (pooled.double() @ w.double().t()).float() * scale + shift
        emb_ref = F.normalize(pre_ref, p=2, dim=1, eps=1e-12)
        emb, pre = ops.gap_linear_norm(fmap.to(DEV), w.t().contiguous().to(DEV), scale.to(DEV), shift.to(DEV), 1e-12, want_pre=True)
        assert torch.allclose(pre.cpu(), pre_ref, atol=2e-5, rtol=2e-5), (ci, float((pre.cpu() - pre_ref).abs().max()))
        assert torch.allclose(emb.cpu(), emb_ref, atol=2e-6, rtol=2e-5), (ci, float((emb.cpu() - emb_ref).abs().max()))
        assert torch.allclose(emb.norm(dim=1).cpu(), torch.ones(B), atol=1e-5)
        three = ops.l2_normalize(ops.linear_f32(ops.avgpool_global(fmap.to(DEV)), w.to(DEV), scale.to(DEV), shift.to(DEV)), 1e-12)
        assert torch.allclose(emb, three, atol=2e-6, rtol=2e-5), (ci, "fused and three-launch heads differ")
        # BaselineNet's head (`face_models.py:41-46`): bias instead of the folded BatchNorm, ReLU before the normalisation
        emb_r, pre_r = ops.gap_linear_norm(fmap.to(DEV), w.t().contiguous().to(DEV), None, shift.to(DEV), 1e-12, want_pre=True, relu=True)
        pre_rr = F.relu((pooled.double() @ w.double().t()).float() + shift)
        assert torch.allclose(pre_r.cpu(), pre_rr, atol=2e-5, rtol=2e-5), (ci, "relu head")
        assert torch.allclose(emb_r.cpu(), F.normalize(pre_rr, p=2, dim=1, eps=1e-12), atol=2e-6, rtol=2e-5), (ci, "relu head, normalised")
    with pytest.raises(ValueError):
        ops.gap_linear_norm(torch.zeros(2, 7, 7, 512, dtype=dtype, device=DEV), torch.zeros(512, 384, device=DEV), None, None)


def test_match_top1_mfma_path_equals_fp32_path():
    """Large galleries on the fp16 MFMA pipe (`frmap_match_top1_packed`: every fp32 operand split into a scaled fp16 (hi, lo)
    pair, one K = 3 D GEMM) against the fp32-GEMM path and against the oracle-style exact scan: same winner, bit-identical
    distance; ragged sizes, duplicate rows (first index wins), tiny and huge magnitudes, every stage width of the GEMM."""
    def exact(e, g):
        d = ((e[:, None, :].double() - g[None, :, :].double()) + 1e-6).pow(2).sum(-1).sqrt()
        return d.min(dim=1)
    for ci, (B, G, D, kind) in enumerate([(64, 1000, 512, "unit"), (1024, 10000, 512, "unit"), (37, 1001, 256, "unit"), (5, 777, 96, "unit"),
                                          (130, 2049, 64, "big"), (16, 640, 512, "tiny"), (33, 1500, 128, "dup")]):
        g = synth.unit_rows(9900 + ci, G, D, "gal")
        e = synth.unit_rows(9910 + ci, B, D, "emb")
        if kind == "big":
            g, e = g * 3000.0, e * 2500.0
        elif kind == "tiny":
            g, e = g * 1e-3, e * 1e-3
        elif kind == "dup":
            g[700:740] = g[3]          # later copies of row 3: the first index must win
            e[:8] = g[3] + 1e-4 * synth.randn(9920, (8, D), "n")
        elif ci == 0:
            e[:16] = g[100:116] + 5e-3 * synth.randn(9921, (16, D), "n")   # enrolment-style probes (near a gallery row)
        gd, ed = g.to(DEV), e.to(DEV)
        prep = ops.match_prepare(gd)
        idx_m, dist_m, ids_m = ops.match_top1(ed, gd, 1.0, prepared=prep)
        idx_f, dist_f, ids_f = ops.match_top1(ed, gd, 1.0)
        assert torch.equal(idx_m, idx_f), (ci, int((idx_m != idx_f).sum()))
        assert torch.equal(dist_m, dist_f) and torch.equal(ids_m, ids_f), ci
        dmin, imin = exact(e, g)
        # every in-band candidate is re-scored exactly (tests/test_match_exact_gpu.py): the winner IS the float64 scan's
        assert torch.equal(idx_m.cpu().long(), imin), (ci, int((idx_m.cpu().long() != imin).sum()))
        assert torch.allclose(dist_m.cpu().double(), dmin, rtol=2e-6, atol=1e-7), ci
        if kind == "dup":
            assert (idx_m[:8] == 3).all()
    g = synth.unit_rows(1, 600, 512, "gal").to(DEV)
    prep = ops.match_prepare(g)
    g[0, 0] += 1.0     # in-place edit: the pack is stale and must be refused
    with pytest.raises(ValueError):
        ops.match_top1(synth.unit_rows(2, 4, 512, "e").to(DEV), g, 1.0, prepared=prep)


def test_conv1x1_pp_kernel():
    """`conv1x1_pp_kernel` (1x1 convs / wide Linear layers on the LDS-DMA ping-pong pipeline) against fp32 torch and the
    first-generation 1x1 kernel: the three layouts (224 px x 256 ch, 448 px x 128 ch, split-K), stride 1 and 2, Linear shapes
    (H = W = 1: HybridNet's QKV / out-projection / MLP over 49 tokens x faces), residual + ReLU / GELU, ragged tails, a
    single k-step, odd k-step counts."""
    from frmap_amd import _lib
    lib = _lib.load()
    cases = [  # B, H, W, Cin, Cout, stride, residual, act, bn (-1 heuristic, 128, 256, 1282 split-K)
        (3, 14, 14, 256, 256, 1, False, 1, 256), (5, 14, 14, 256, 512, 1, True, 1, 128), (2, 28, 28, 128, 256, 2, False, 0, -1),
        (2, 56, 56, 64, 128, 2, False, 1, 128), (637, 1, 1, 512, 1536, 1, False, 0, 256), (637, 1, 1, 2048, 512, 1, True, 0, 1282),
        (1001, 1, 1, 512, 2048, 1, False, 2, 128), (9, 7, 7, 1024, 384, 1, True, 2, -1), (300, 1, 1, 32, 128, 1, False, 1, 256),
        (450, 1, 1, 96, 256, 1, True, 1, 1282), (4, 13, 9, 160, 640, 1, False, 1, 1282), (1, 1, 1, 512, 512, 1, False, 0, 256),
    ]
    try:
        for ci, (B, H, W, Cin, Cout, sd, res, act, bn) in enumerate(cases):
            for dtype in DTYPES:
                x = synth.randn(9500 + ci, (B, Cin, H, W), "x").to(dtype)
                w = (synth.randn(9510 + ci, (Cout, Cin, 1, 1), "w") * math.sqrt(1.0 / Cin)).to(dtype)
                shift = synth.randn(9520 + ci, (Cout,), "b") * 0.1
                ref = F.conv2d(x.float(), w.float(), None, stride=sd) + shift.view(1, -1, 1, 1)
                r = synth.randn(9530 + ci, tuple(ref.shape), "r").to(dtype) if res else None
                if res:
                    ref = ref + r.float()
                ref = F.relu(ref) if act == 1 else (F.gelu(ref) if act == 2 else ref)
                wpk = ops.pack_conv_weight(w.float().to(DEV), dtype)
                xin, rin = _nhwc(x).to(DEV), (_nhwc(r).to(DEV) if res else None)
                lib.frmap_conv_pp_tuning(1, -1, bn)
                want = {256: 1 if Cout % 256 == 0 else 2, 128: 2, 1282: 3 if (Cin // 32) % 2 == 0 else None}.get(bn)
                lay = lib.frmap_conv1x1_pp_layout(B, H, W, Cin, Cout, sd)
                assert lay in (1, 2, 3) and (want is None or lay == want), (ci, lay, want)
                y_pp = ops.conv_igemm(xin, wpk, shift.to(DEV), Cout, 1, sd, 0, act, rin)
                lib.frmap_conv_pp_tuning(0, -1, -1)
                y_g1 = ops.conv_igemm(xin, wpk, shift.to(DEV), Cout, 1, sd, 0, act, rin)
                atol, rtol = _tol(dtype)
                y = y_pp.float().cpu().permute(0, 3, 1, 2)
                assert y.shape == ref.shape
                assert torch.allclose(y, ref, atol=atol, rtol=rtol), (ci, dtype, float((y - ref).abs().max()))
                assert torch.allclose(y_pp.float(), y_g1.float(), atol=atol, rtol=rtol), (ci, dtype, "generations differ")
    finally:
        lib.frmap_conv_pp_tuning(-1, -1, -1)
