"""The callers either side of the hot path (SURVEY.md §8f "next" rows), on the device.

* input pipeline — ``transforms.Resize((224,224)) → ToTensor → Normalize(ImageNet)`` of
  `/root/reference/src/testing.py:99-104` (and the 160×160 / 0.5-0.5 variant of `src/app.py:39-42`):
  the resize stays on the host (PIL bilinear, exactly what torchvision's PIL backend calls), the
  uint8 → normalised-float step is ``ops.normalize_u8`` (3 bytes per pixel cross PCIe instead of 12);
* evaluation step — `src/testing.py:255-283`: forward → softmax → arg-max, with the ArcFace branch
  scoring embeddings against the class centres (`:264-269`; `hyperparameter_tuning.py:1038-1046`);
* Siamese verification — `src/testing.py:170-177`: ``dist = pairwise_distance(out1, out2)``,
  ``pred = dist < 0.5``.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops

IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)     # testing.py:102-103
FACENET_MEAN, FACENET_STD = (0.5, 0.5, 0.5), (0.5, 0.5, 0.5)                   # app.py:41


def resize_to_u8(images: Sequence, size: Tuple[int, int] = (224, 224)) -> torch.Tensor:
    """Host side of the transform: PIL images / HWC uint8 arrays → one uint8 [B, H, W, 3] tensor,
    resized the way ``transforms.Resize(size)`` does for PIL input (bilinear)."""
    from PIL import Image
    out = np.empty((len(images), size[0], size[1], 3), np.uint8)
    for i, im in enumerate(images):
        if not isinstance(im, Image.Image):
            im = Image.fromarray(np.asarray(im, np.uint8))
        out[i] = np.asarray(im.convert("RGB").resize((size[1], size[0]), Image.BILINEAR))
    return torch.from_numpy(out)


def preprocess(images_u8: torch.Tensor, mean=IMAGENET_MEAN, std=IMAGENET_STD, device="cuda") -> torch.Tensor:
    """uint8 [B, H, W, 3] → fp32 NCHW on the device == ``Normalize(mean, std)(ToTensor()(img))``."""
    return ops.normalize_u8(images_u8.to(device, non_blocking=True), mean, std)[0]


def predict_batch(model, images: torch.Tensor, model_type: str):
    """One evaluation-loop step (`testing.py:255-283`): returns ``(outputs, probs, predicted)`` on the
    device.  ``arcface``: logits = cosine(embedding, class centres) (`testing.py:264-269`)."""
    if model_type == 'arcface':
        emb = model(images)
        outputs, _ = ops.cosine_logits(emb, model.arcface.weight.detach(), s=1.0, want_argmax=False)
    else:
        outputs = model(images)
    probs, pred = ops.softmax_argmax(outputs)
    return outputs, probs, pred


def arcface_validate(model, images: torch.Tensor):
    """`hyperparameter_tuning.py:1038-1046`: logits = s · cos(embedding, class centres); arg-max."""
    emb = model.get_embedding(images)
    return ops.cosine_logits(emb, model.arcface.weight.detach(), s=float(model.arcface.s))


def siamese_verify(model, img1: torch.Tensor, img2: torch.Tensor, thresh: float = 0.5):
    """`testing.py:175-177`: returns ``(dist, pred)`` with ``pred = (dist < thresh)`` as float."""
    out1, out2 = model(img1, img2)
    dist, same = ops.pairwise_distance(out1, out2, thresh)
    return dist, same.to(torch.float32)
