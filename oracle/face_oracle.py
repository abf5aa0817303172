"""CPU ORACLE — test infrastructure, NOT product code.

A plain fp32 PyTorch-CPU restatement of the reference's embedding-extraction + gallery-matching
path (SURVEY.md §8a rows a2–a12).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; the product package
(``facerecognition-multiarchitecture-pipeline_amd``) never does and has no CPU fallback.

Every function is *functional*: it takes a ``state_dict`` with the reference's key names
(`/root/reference/src/face_models.py`, key sets listed in SURVEY.md §8b) and an input tensor, and
returns what the reference module would return in ``eval()`` mode under ``torch.no_grad()``.

Pinning: the reference has no tests or golden vectors of its own, so this oracle is
pinned against outputs of the reference classes themselves, executed in the build container by
``oracle/gen_golden.py`` (reference file loaded by path with a stub for the un-vendored
``torchvision.models.resnet18``) and committed as ``tests/golden/*.npz``.  The ResNet-18 topology
itself comes from torchvision (`requirements.txt:2`, ``torchvision>=0.14.0``, unpinned, absent from
``/root/reference``): it is restated here from its published definition (He et al. 2015 v1
BasicBlock; stride on the first 3×3 of each stage; 1×1-s2 conv + BN downsample; ``bias=False``;
BN eps 1e-5; 3×3-s2-p1 max-pool) and is **parity-unpinned** at that boundary.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------------
def _bn(sd: SD, p: str, x: torch.Tensor, calib: bool = False, eps: float = 1e-5) -> torch.Tensor:
    """Eval-mode BatchNorm{1,2}d: gamma*(x-mu)/sqrt(var+eps)+beta with running stats.

    ``calib=True`` first overwrites the running stats in ``sd`` with this batch's statistics
    (what one train-mode forward with momentum=1 would store) — used only to build non-degenerate
    synthetic weights (SURVEY.md §7 hard part 2), never on the inference path.
    """
    if calib:
        dims = [0] + list(range(2, x.dim()))
        sd[p + "running_mean"] = x.mean(dim=dims).detach().clone()
        sd[p + "running_var"] = x.var(dim=dims, unbiased=True).detach().clone()
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"],
                        sd[p + "weight"], sd[p + "bias"], training=False, eps=eps)


def _alias_bn_stats(sd: SD) -> None:
    """Keep the reference's duplicated ``features.*`` keys equal to ``backbone.*`` / ``cnn.*``."""
    idx = {"conv1": "0", "bn1": "1", "layer1": "4", "layer2": "5", "layer3": "6", "layer4": "7"}
    for tp in ("backbone", "cnn"):
        for k in list(sd.keys()):
            if k.startswith(tp + "."):
                parts = k.split(".")
                if parts[1] in idx:
                    alias = ".".join(["features", idx[parts[1]]] + parts[2:])
                    if alias in sd:
                        sd[alias] = sd[k]


# --------------------------------------------------------------------------------------------
# ResNet-18 trunk (torchvision definition; used through face_models.py:67,100,269,463-464,658-660)
# --------------------------------------------------------------------------------------------
def _basic_block(sd: SD, p: str, x: torch.Tensor, stride: int, calib: bool) -> torch.Tensor:
    out = F.conv2d(x, sd[p + "conv1.weight"], None, stride=stride, padding=1)
    out = F.relu(_bn(sd, p + "bn1.", out, calib))
    out = F.conv2d(out, sd[p + "conv2.weight"], None, stride=1, padding=1)
    out = _bn(sd, p + "bn2.", out, calib)
    if (p + "downsample.0.weight") in sd:
        idn = F.conv2d(x, sd[p + "downsample.0.weight"], None, stride=stride, padding=0)
        idn = _bn(sd, p + "downsample.1.", idn, calib)
    else:
        idn = x
    return F.relu(out + idn)


def resnet18_trunk(sd: SD, p: str, x: torch.Tensor, pool: bool = True, calib: bool = False) -> torch.Tensor:
    """``children()[:-1]`` (pool=True → B×512) or ``children()[:-2]`` (pool=False → B×512×7×7)."""
    x = F.conv2d(x, sd[p + "conv1.weight"], None, stride=2, padding=3)
    x = F.relu(_bn(sd, p + "bn1.", x, calib))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for li, stride in ((1, 1), (2, 2), (3, 2), (4, 2)):
        x = _basic_block(sd, f"{p}layer{li}.0.", x, stride, calib)
        x = _basic_block(sd, f"{p}layer{li}.1.", x, 1, calib)
    if pool:
        x = F.adaptive_avg_pool2d(x, 1).flatten(1)
    return x


# --------------------------------------------------------------------------------------------
# a2  BaselineNet  (face_models.py:16-60)
# --------------------------------------------------------------------------------------------
def baseline_embedding(sd: SD, x: torch.Tensor, calib: bool = False) -> torch.Tensor:
    """`face_models.py:51-60`: 3×[conv3×3 p1 + bias → BN → ReLU → maxpool 2×2] → GAP → ReLU(fc1)."""
    for i in (1, 2, 3):
        x = F.conv2d(x, sd[f"conv{i}.weight"], sd[f"conv{i}.bias"], padding=1)
        x = F.max_pool2d(F.relu(_bn(sd, f"bn{i}.", x, calib)), 2, 2)
    x = F.adaptive_avg_pool2d(x, 1).flatten(1)
    return F.relu(F.linear(x, sd["fc1.weight"], sd["fc1.bias"]))


def baseline_forward(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """`face_models.py:36-49` (dropout is identity in eval)."""
    return F.linear(baseline_embedding(sd, x), sd["fc2.weight"], sd["fc2.bias"])


# --------------------------------------------------------------------------------------------
# a3  ResNetTransfer  (face_models.py:62-102)
# --------------------------------------------------------------------------------------------
def cnn_embedding(sd: SD, x: torch.Tensor, calib: bool = False) -> torch.Tensor:
    """`face_models.py:98-102`: pooled trunk output, then ``.squeeze()`` (drops B when B == 1)."""
    return resnet18_trunk(sd, "resnet.", x, pool=True, calib=calib).squeeze()


def cnn_forward(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """`face_models.py:93-96`: ``resnet(x)`` with ``fc = Sequential(Dropout, Linear)`` (`:73-76`)."""
    f = resnet18_trunk(sd, "resnet.", x, pool=True)
    return F.linear(f, sd["resnet.fc.1.weight"], sd["resnet.fc.1.bias"])


# --------------------------------------------------------------------------------------------
# a4  ArcFaceNet eval  (face_models.py:511-590)     a5  ArcMarginProduct eval (face_models.py:334-429)
# --------------------------------------------------------------------------------------------
def arcface_pre_norm(sd: SD, x: torch.Tensor, calib: bool = False) -> torch.Tensor:
    """`face_models.py:585-588`: trunk → Linear(512,512,bias=False) → BatchNorm1d (before normalise)."""
    f = resnet18_trunk(sd, "backbone.", x, pool=True, calib=calib)
    e = F.linear(f, sd["embedding.weight"], None)
    e = _bn(sd, "bn.", e, calib)
    if calib:
        _alias_bn_stats(sd)
    return e


def arcface_embedding(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """`face_models.py:584-590` — also what eval ``forward(x)`` returns (`:573-582`)."""
    return F.normalize(arcface_pre_norm(sd, x), p=2, dim=1, eps=1e-12)


def arcface_forward(sd: SD, x: torch.Tensor, labels: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Eval branch `face_models.py:573-582`: with labels → ``val_classifier`` logits on row-normalised
    classifier weights (`:576`); without → unit embedding."""
    emb = arcface_embedding(sd, x)
    if labels is None:
        return emb
    w = F.normalize(sd["val_classifier.weight"], p=2, dim=1, eps=1e-12)
    return F.linear(emb, w, sd["val_classifier.bias"])


def arcmargin_eval(weight: torch.Tensor, x: torch.Tensor, label: torch.Tensor,
                   s: float = 32.0, m: float = 0.5, easy_margin: bool = False) -> torch.Tensor:
    """`face_models.py:351-429` with ``self.training == False``: full margin ``m``, scale
    ``min(s, 24.0)`` (`:401-404`), NaN/Inf → 0 (`:423-426`)."""
    xn = F.normalize(x, p=2, dim=1, eps=1e-12)
    wn = F.normalize(weight, p=2, dim=1, eps=1e-12)
    cos = F.linear(xn, wn)
    cos = torch.clamp(cos, min=-1.0 + 1e-7, max=1.0 - 1e-7)
    theta = torch.acos(cos)
    if easy_margin:
        phi = torch.where(cos > 0, torch.cos(theta + m), cos)
    else:
        phi = torch.cos(torch.minimum(torch.tensor(math.pi - 1e-4), theta + m))
    one_hot = torch.zeros_like(cos)
    one_hot.scatter_(1, label.view(-1, 1), 1)
    out = torch.where(one_hot.bool(), phi, cos) * min(s, 24.0)
    return torch.where(torch.isnan(out) | torch.isinf(out), torch.zeros_like(out), out)


# --------------------------------------------------------------------------------------------
# a6  SiameseNet  (face_models.py:104-192)
# --------------------------------------------------------------------------------------------
def siamese_pre_norm(sd: SD, x: torch.Tensor, calib: bool = False) -> torch.Tensor:
    """`face_models.py:113-156,161-176`."""
    x = F.conv2d(x, sd["conv.0.weight"], sd["conv.0.bias"], stride=2, padding=3)
    x = F.max_pool2d(F.relu(_bn(sd, "conv.1.", x, calib)), 2, 2)
    for ci, bi, pool in ((4, 5, False), (7, 8, True), (11, 12, False), (14, 15, True), (18, 19, False)):
        x = F.conv2d(x, sd[f"conv.{ci}.weight"], sd[f"conv.{ci}.bias"], padding=1)
        x = F.relu(_bn(sd, f"conv.{bi}.", x, calib))
        if pool:
            x = F.max_pool2d(x, 2, 2)
    x = F.adaptive_avg_pool2d(x, (6, 6))
    x = x.reshape(x.size(0), -1)
    x = F.linear(x, sd["fc.1.weight"], sd["fc.1.bias"])
    x = F.relu(_bn(sd, "fc.2.", x, calib))
    x = F.linear(x, sd["fc.5.weight"], sd["fc.5.bias"])
    x = F.relu(_bn(sd, "fc.6.", x, calib))
    return F.linear(x, sd["fc.8.weight"], sd["fc.8.bias"])


def siamese_forward_one(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """`face_models.py:161-180` / ``get_embedding`` `:187-188`."""
    return F.normalize(siamese_pre_norm(sd, x), p=2, dim=1)


def siamese_forward(sd: SD, x1: torch.Tensor, x2: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """`face_models.py:182-185`."""
    return siamese_forward_one(sd, x1), siamese_forward_one(sd, x2)


def siamese_decision(out1: torch.Tensor, out2: torch.Tensor, thresh: float = 0.5):
    """a12 — `src/testing.py:175-177`: ``dist = pairwise_distance``; ``pred = dist < 0.5``."""
    dist = F.pairwise_distance(out1, out2)
    return dist, (dist < thresh).float()


# --------------------------------------------------------------------------------------------
# a7  HybridNet + TransformerBlock  (face_models.py:618-721)
# --------------------------------------------------------------------------------------------
def _mha(sd: SD, p: str, x: torch.Tensor, num_heads: int = 4) -> torch.Tensor:
    """``nn.MultiheadAttention(512, 4)`` self-attention on L×B×D (`face_models.py:623,640`), eval."""
    L, B, D = x.shape
    dh = D // num_heads
    qkv = F.linear(x, sd[p + "in_proj_weight"], sd[p + "in_proj_bias"])
    q, k, v = qkv.split(D, dim=-1)

    def heads(t):  # L×B×D -> (B·H)×L×dh
        return t.reshape(L, B * num_heads, dh).transpose(0, 1)

    q, k, v = heads(q), heads(k), heads(v)
    att = torch.softmax(torch.bmm(q, k.transpose(1, 2)) / math.sqrt(dh), dim=-1)
    o = torch.bmm(att, v).transpose(0, 1).reshape(L, B, D)
    return F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def hybrid_embedding(sd: SD, x: torch.Tensor, calib: bool = False) -> torch.Tensor:
    """`face_models.py:705-721`."""
    f = resnet18_trunk(sd, "cnn.", x, pool=False, calib=calib)
    if calib:
        _alias_bn_stats(sd)
    B = f.shape[0]
    t = f.reshape(B, 512, -1).permute(2, 0, 1) + sd["pos_encoding"]
    n1 = F.layer_norm(t, (512,), sd["transformer.norm1.weight"], sd["transformer.norm1.bias"])
    t = t + _mha(sd, "transformer.attention.", n1)
    n2 = F.layer_norm(t, (512,), sd["transformer.norm2.weight"], sd["transformer.norm2.bias"])
    ff = F.linear(F.gelu(F.linear(n2, sd["transformer.ff.0.weight"], sd["transformer.ff.0.bias"])),
                  sd["transformer.ff.3.weight"], sd["transformer.ff.3.bias"])
    t = t + ff
    return F.layer_norm(t.mean(dim=0), (512,), sd["norm.weight"], sd["norm.bias"])


def hybrid_forward(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """`face_models.py:680-703`."""
    return F.linear(hybrid_embedding(sd, x), sd["fc.weight"], sd["fc.bias"])


# --------------------------------------------------------------------------------------------
# §8(f) AttentionNet  (face_models.py:194-295)   EnsembleModel combination (face_models.py:880-926)
# --------------------------------------------------------------------------------------------
def attention_module(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """`face_models.py:213-262` (AttentionModule) + `:194-211` (SpatialAttention): 1×1 q/k/v projections,
    softmax(qᵀk) over the H·W positions, ``gamma * (v · attnᵀ) + x``, then the 7×7 spatial gate on the
    channel-mean / channel-max maps."""
    B, C, H, W = x.shape
    q = F.conv2d(x, sd[p + "query.weight"], sd[p + "query.bias"]).view(B, -1, H * W).permute(0, 2, 1)
    k = F.conv2d(x, sd[p + "key.weight"], sd[p + "key.bias"]).view(B, -1, H * W)
    v = F.conv2d(x, sd[p + "value.weight"], sd[p + "value.bias"]).view(B, -1, H * W)
    attention = F.softmax(torch.bmm(q, k), dim=-1)
    out = torch.bmm(v, attention.permute(0, 2, 1)).view(B, C, H, W)
    y = sd[p + "gamma"] * out + x
    pooled = torch.cat([y.mean(dim=1, keepdim=True), y.max(dim=1, keepdim=True)[0]], dim=1)
    gate = torch.sigmoid(F.conv2d(pooled, sd[p + "spatial_attention.conv.weight"], sd[p + "spatial_attention.conv.bias"],
                                  padding=sd[p + "spatial_attention.conv.weight"].shape[-1] // 2))
    return y * gate


def attention_embedding(sd: SD, x: torch.Tensor, calib: bool = False) -> torch.Tensor:
    """`face_models.py:284-288`: features → attention → global average pool → B×512."""
    f = resnet18_trunk(sd, "backbone.", x, pool=False, calib=calib)
    if calib:
        _alias_bn_stats(sd)
    return attention_module(sd, "attention.", f).mean(dim=(2, 3))


def attention_forward(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """`face_models.py:276-282`."""
    return F.linear(attention_embedding(sd, x), sd["fc.weight"], sd["fc.bias"])


def ensemble_combine(outputs: Sequence[torch.Tensor], method: str, weights: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`face_models.py:902-920`: how EnsembleModel merges its members' logits (a single member is returned as is;
    any method but 'average' / 'weighted' / 'max' — including the constructor's 'attention' — raises ValueError)."""
    outputs = list(outputs)
    if len(outputs) == 1:
        return outputs[0]
    if method == "average":
        return torch.mean(torch.stack(outputs), dim=0)
    if method == "max":
        probs = [F.softmax(o, dim=1) for o in outputs]
        return torch.log(torch.max(torch.stack(probs), dim=0)[0])
    if method == "weighted":
        w = F.softmax(weights, dim=0)
        return torch.sum(torch.stack([w[i] * outputs[i] for i in range(len(outputs))]), dim=0)
    raise ValueError(f"Unknown ensemble method: {method}")


# --------------------------------------------------------------------------------------------
# a8  compare_faces   (src/app.py:50-64)      a11 class-centre match (hyperparameter_tuning.py:1036-1046)
# --------------------------------------------------------------------------------------------
def compare_faces(emb, refs: Sequence[dict], thresh: float):
    """`src/app.py:50-64`: Euclidean ``pairwise_distance`` (eps=1e-6 added to the *difference*),
    first strict minimum, ``("Unknown", d, None)`` above threshold, sentinel for None/empty."""
    if emb is None or not refs:
        return "Unknown", float("inf"), None
    min_dist, best, best_idx = float("inf"), "Unknown", None
    e = emb.cpu()
    for i, ref in enumerate(refs):
        d = F.pairwise_distance(e, ref["embedding"].cpu()).item()
        if d < min_dist:
            min_dist, best, best_idx = d, ref["name"], i
    return (best, min_dist, best_idx) if min_dist <= thresh else ("Unknown", min_dist, None)


def match_top1(emb: torch.Tensor, gallery: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Batched form of ``compare_faces`` without the threshold: for each row of ``emb`` (B×D) the
    index of the first minimum of ``‖e − g_i + 1e-6‖₂`` over gallery rows (G×D) and that distance."""
    d = torch.sqrt(((emb[:, None, :].double() - gallery[None, :, :].double() + 1e-6) ** 2).sum(-1))
    dist, idx = d.min(dim=1)
    return idx.to(torch.int32), dist.float()


def class_centre_match(emb: torch.Tensor, centres: torch.Tensor, s: float = 1.0):
    """`hyperparameter_tuning.py:1038-1046`, `face_models.py:889-893`: cosine logits vs class
    centres (× s) and their arg-max."""
    logits = torch.matmul(F.normalize(emb, p=2, dim=1), F.normalize(centres, p=2, dim=1).t()) * s
    return logits, logits.argmax(dim=1)


def evaluate_loop(model_type: str, sd: SD, batches, arcface_classifier=None) -> dict:
    """The evaluation loop of `src/testing.py:167-283` restated on the CPU: per batch forward -> (ArcFace: classifier
    or cosine vs class centres, `:258-269`) -> CrossEntropyLoss (`:275-276`) -> softmax -> arg-max (`:278-279`);
    Siamese: ``pairwise_distance`` -> ``dist < 0.5`` (`:175-177`), distances collected as ``[d]`` rows (`:182`).
    Returns predictions / targets / probabilities (numpy) and the mean loss, as the reference accumulates them."""
    import numpy as np
    preds, targets, probs, total_loss, n = [], [], [], 0.0, 0
    with torch.no_grad():
        for batch in batches:
            if model_type == "siamese":
                x1, x2, labels = batch
                dist, pred = siamese_decision(*siamese_forward(sd, x1, x2))
                preds.extend(pred.numpy()); targets.extend(labels.numpy()); probs.extend(dist.numpy()[:, None])
                n += 1
                continue
            x, labels = batch
            if model_type == "arcface":
                emb = arcface_embedding(sd, x)
                if arcface_classifier is not None:
                    outputs = F.linear(emb, arcface_classifier[0], arcface_classifier[1])
                else:
                    outputs = F.linear(F.normalize(emb), F.normalize(sd["arcface.weight"]))
            else:
                outputs = FORWARD[model_type](sd, x)
            total_loss += float(F.cross_entropy(outputs, labels))
            p = F.softmax(outputs, dim=1)
            _, predicted = torch.max(outputs, 1)
            preds.extend(predicted.numpy()); targets.extend(labels.numpy()); probs.extend(p.numpy())
            n += 1
    return {"predictions": np.array(preds), "targets": np.array(targets), "probabilities": np.array(probs),
            "test_loss": total_loss / max(n, 1), "logit_batches": n}


# --------------------------------------------------------------------------------------------
# dispatch by reference model_type (face_models.py:785-813)
# --------------------------------------------------------------------------------------------
FORWARD = {"attention": attention_forward, "baseline": baseline_forward, "cnn": cnn_forward, "arcface": arcface_forward,
           "hybrid": hybrid_forward}
EMBEDDING = {"attention": attention_embedding, "baseline": baseline_embedding, "cnn": cnn_embedding, "arcface": arcface_embedding,
             "siamese": siamese_forward_one, "hybrid": hybrid_embedding}
_CALIB = {"attention": attention_embedding, "baseline": baseline_embedding, "cnn": cnn_embedding, "arcface": arcface_pre_norm,
          "siamese": siamese_pre_norm, "hybrid": hybrid_embedding}


def calibrate_bn(model_type: str, sd: SD, x: torch.Tensor) -> SD:
    """Overwrite every BatchNorm's running stats in ``sd`` (in place) with the statistics of its own
    input on batch ``x`` (layer by layer, upstream layers already calibrated).  Makes random-init
    networks non-degenerate so that embedding-cosine / top-1 parity checks mean something."""
    with torch.no_grad():
        _CALIB[model_type](sd, x, calib=True)
    return sd
