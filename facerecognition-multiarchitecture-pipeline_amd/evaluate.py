"""The callers either side of the hot path (SURVEY.md §8f "next" rows), on the device.

* input pipeline — ``transforms.Resize((224,224)) → ToTensor → Normalize(ImageNet)`` of
  `/root/reference/src/testing.py:99-104` (and the 160×160 / 0.5-0.5 variant of `src/app.py:39-42`):
  the resize runs on the device (``resize.resize_bilinear_u8``, bit-exact with Pillow's bilinear resampler, which is what
  torchvision's PIL backend calls; without a device argument Pillow does it on the host), the uint8 → normalised-float step is
  ``ops.normalize_u8`` or the stem itself (3 bytes per pixel cross PCIe instead of 12);
* single-image prediction — `src/testing.py:532-595` ``predict_image``: Resize → ToTensor → Normalize → forward → softmax → max;
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


def resize_to_u8(images: Sequence, size: Tuple[int, int] = (224, 224), device=None) -> torch.Tensor:
    """``transforms.Resize(size)`` for PIL input (bilinear): PIL images / HWC uint8 arrays → one uint8 [B, H, W, 3] tensor.
    With ``device`` the decoded images are uploaded at their native size and resized on the GPU
    (`resize.resize_bilinear_u8`, bit-exact with Pillow); without it Pillow does it on the host."""
    from PIL import Image
    if device is not None:
        from . import resize as _resize
        return _resize.resize_bilinear_u8([np.asarray(im.convert("RGB") if isinstance(im, Image.Image) else im, np.uint8)
                                           for im in images], size, device)
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


# ------------------------------------------------------------------------------------------------
# §8(f)-2: the batched evaluation harness around the path (`src/testing.py:26-394`)
# ------------------------------------------------------------------------------------------------
IMG_EXTENSIONS = ('.jpg', '.jpeg', '.png', '.ppm', '.bmp', '.pgm', '.tif', '.tiff', '.webp')


def image_folder(root: str):
    """``datasets.ImageFolder(root)``'s sample list (`testing.py:110`): classes = sorted sub-directories, samples =
    every image file under each class directory in sorted walk order.  Returns ``(samples, classes)`` with
    ``samples = [(path, class_index), ...]``."""
    import os
    classes = sorted(d.name for d in os.scandir(root) if d.is_dir())
    if not classes:
        raise FileNotFoundError(f"Couldn't find any class folder in {root}.")
    samples = []
    for ci, c in enumerate(classes):
        for r, _, files in sorted(os.walk(os.path.join(root, c), followlinks=True)):
            for f in sorted(files):
                if f.lower().endswith(IMG_EXTENSIONS):
                    samples.append((os.path.join(r, f), ci))
    return samples, classes


def _folder_batches(samples, batch_size, device):
    from PIL import Image
    for lo in range(0, len(samples), batch_size):
        chunk = samples[lo: lo + batch_size]
        imgs = []
        for path, _ in chunk:
            with Image.open(path) as im:
                imgs.append(im.convert("RGB"))
        u8 = resize_to_u8(imgs, (224, 224), device=device)                    # transforms.Resize((224, 224)), `testing.py:100`: on the device
        yield preprocess(u8, IMAGENET_MEAN, IMAGENET_STD, device), torch.tensor([c for _, c in chunk], dtype=torch.int64)


def _results_file_name(model_type: str) -> str:
    """`testing.py:369-378`."""
    return {'siamese': 'siamese_network_results.json', 'arcface': 'arcface_model_results.json',
            'baseline': 'baseline_model_results.json', 'cnn': 'cnn_model_results.json'}.get(model_type, f'{model_type}_model_results.json')


def classification_metrics(all_targets: np.ndarray, all_predictions: np.ndarray, all_probs: np.ndarray, siamese: bool = False) -> dict:
    """The metric block of `testing.py:290-315` (sklearn, weighted averages, one-vs-rest ROC AUC, average precision;
    Siamese: curves on ``-distance``).  A metric sklearn cannot define on the given labels (a class with no sample)
    comes back as NaN instead of aborting the evaluation."""
    from sklearn.metrics import (accuracy_score, auc, average_precision_score, f1_score, precision_recall_curve,
                                 precision_score, recall_score, roc_auc_score, roc_curve)

    def safe(fn):
        try:
            return float(fn())
        except Exception:
            return float("nan")

    m = {"accuracy": safe(lambda: accuracy_score(all_targets, all_predictions)),
         "precision": safe(lambda: precision_score(all_targets, all_predictions, average='weighted', zero_division=0)),
         "recall": safe(lambda: recall_score(all_targets, all_predictions, average='weighted', zero_division=0)),
         "f1": safe(lambda: f1_score(all_targets, all_predictions, average='weighted', zero_division=0))}
    if siamese:
        def roc():
            fpr, tpr, _ = roc_curve(all_targets, -all_probs.ravel())
            return auc(fpr, tpr)

        def pr():
            p, r, _ = precision_recall_curve(all_targets, -all_probs.ravel())
            return auc(r, p)
        m["roc_auc"], m["pr_auc"] = safe(roc), safe(pr)
    else:
        m["roc_auc"] = safe(lambda: roc_auc_score(all_targets, all_probs, multi_class='ovr'))
        m["pr_auc"] = safe(lambda: average_precision_score(all_targets, all_probs))
    return m


def evaluate_model(model, model_type: str, test_data, class_names: Optional[Sequence[str]] = None,
                   out_dir: Optional[str] = None, model_name: Optional[str] = None, dataset_name: str = "test",
                   batch_size: int = 32, arcface_classifier: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                   device="cuda") -> dict:
    """`src/testing.py:26-394` around the HIP path: batches of 32 → forward → softmax → arg-max → accumulate →
    sklearn metrics → the same JSON files.  What differs, by necessity: the model (already loaded, on the GPU, in
    ``eval()``) and the test set are passed in instead of being discovered under ``PROC_DATA_DIR`` / ``CHECKPOINTS_DIR``
    (`:29-129`, the reference's project layout and interactive prompt — out of scope), and the per-batch "inference
    time" is bracketed by device synchronisation (the reference's `time.time()` pair at `:164,255-273` times only the
    asynchronous launches on a GPU).

    ``test_data``: a directory in ``ImageFolder`` layout (class sub-folders; images are resized to 224×224 with PIL
    and normalised on the device), or an iterable of ready batches — ``(images fp32 NCHW, labels)``, for
    ``model_type == 'siamese'`` ``(img1, img2, labels)`` (`:170-182`; label 1 = same, prediction = distance < 0.5).
    ``arcface``: logits = cosine(embedding, class centres) (`:264-269`) unless ``arcface_classifier=(weight, bias)`` is
    given (the reference scores with a freshly initialised ``nn.Linear(512, C)``, `:134-136,262-263`).
    Returns the ``model_results`` dict (`:346-362`); with ``out_dir`` also writes ``<type>_model_results.json`` and
    ``experiment_summary.json`` (`:364-394`)."""
    import json
    import os
    import time
    if isinstance(test_data, (str, os.PathLike)):
        if model_type == 'siamese':
            raise ValueError("siamese evaluation takes an iterable of (img1, img2, labels) batches")
        samples, classes = image_folder(str(test_data))
        if not samples:
            raise FileNotFoundError(f"Found 0 files in subfolders of: {test_data}")
        class_names = list(classes) if class_names is None else list(class_names)
        batches = _folder_batches(samples, batch_size, device)
    else:
        batches = iter(test_data)
    if model.training:
        model.eval()                                                           # `:131`
    siamese = model_type == 'siamese'
    all_predictions, all_targets, all_probs, inference_times = [], [], [], []
    total_loss, nbatches = 0.0, 0
    with torch.no_grad():
        for batch in batches:
            if siamese:
                img1, img2, labels = batch
                img1, img2 = img1.to(device), img2.to(device)
                torch.cuda.synchronize()
                t0 = time.time()
                dist, same = siamese_verify(model, img1, img2, 0.5)             # `:175-177`
                torch.cuda.synchronize()
                inference_times.append(time.time() - t0)
                all_predictions.extend(same.cpu().numpy())
                all_targets.extend(np.asarray(labels.cpu() if isinstance(labels, torch.Tensor) else labels))
                all_probs.extend(dist.cpu().numpy()[:, None])                    # distances stand in for probabilities (`:182`)
                nbatches += 1
                continue
            images, labels = batch
            images = images.to(device)
            labels = torch.as_tensor(labels).to(torch.int64)
            torch.cuda.synchronize()
            t0 = time.time()
            if model_type == 'arcface':
                emb = model(images)
                if arcface_classifier is not None:
                    w, b = arcface_classifier
                    outputs = ops.linear_f32(emb, w.to(device).float(), None, None if b is None else b.to(device).float())
                else:
                    outputs, _ = ops.cosine_logits(emb, model.arcface.weight.detach(), s=1.0, want_argmax=False)
            else:
                outputs = model(images)
            torch.cuda.synchronize()
            inference_times.append(time.time() - t0)
            probs, pred = ops.softmax_argmax(outputs)                           # `:278-279`
            logits = outputs.float().cpu()
            lse = torch.logsumexp(logits, dim=1)
            total_loss += float((lse - logits[torch.arange(logits.shape[0]), labels]).mean())   # nn.CrossEntropyLoss, `:275-276`
            nbatches += 1
            all_predictions.extend(pred.cpu().numpy())
            all_targets.extend(labels.numpy())
            all_probs.extend(probs.cpu().numpy())
    if nbatches == 0:
        raise ValueError("evaluate_model: the test set is empty")
    all_predictions, all_targets, all_probs = np.array(all_predictions), np.array(all_targets), np.array(all_probs)
    metrics = classification_metrics(all_targets, all_predictions, all_probs, siamese)
    metrics["inference_time"] = float(np.mean(inference_times))                # `:315`
    if siamese:
        class_names = ['Same', 'Different']                                     # `:330`
    elif class_names is None:
        class_names = [str(i) for i in range(all_probs.shape[1])]
    model_results = {"predictions": all_predictions.tolist(), "targets": all_targets.tolist(),
                     "probabilities": all_probs.tolist(), "class_names": list(class_names), "metrics": metrics}
    if not siamese:
        model_results["test_loss"] = total_loss / nbatches                      # printed at `:327`
    if out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, _results_file_name(model_type)), 'w') as f:
            json.dump(model_results, f, indent=2)
        with open(os.path.join(out_dir, 'experiment_summary.json'), 'w') as f:   # `:384-394`
            json.dump({"model_type": model_type, "model_name": model_name or model_type, "dataset": dataset_name,
                       "metrics": metrics, "class_names": list(class_names)}, f, indent=2)
    return model_results


# ------------------------------------------------------------------------------------------------
# `src/testing.py:532-595`: predict_image
# ------------------------------------------------------------------------------------------------
def predict_image(model_type: str, image_path: str, model_name: Optional[str] = None, checkpoints_dir: str = "outputs/checkpoints",
                  proc_data_dir: str = "data/processed", device="cuda") -> Tuple[str, float]:
    """Make a prediction for a single image: ``(class_name, probability)`` - `src/testing.py:532-595` on the HIP path.

    Same discovery rules and errors as the reference: the latest ``<model_type>_*`` directory under ``checkpoints_dir``
    (``CHECKPOINTS_DIR``, `base_config.py:18`) unless ``model_name`` is given; class names from ``ImageFolder(<first processed
    dataset>/train)`` under ``proc_data_dir`` (``PROC_DATA_DIR``, `base_config.py:15`); ``best_model.pth`` before
    ``best_checkpoint.pth``; ``'siamese'`` is refused.  The image is resized to 224x224 on the device (bit-exact with the PIL
    bilinear ``transforms.Resize``), ToTensor + Normalize run inside the stem, then forward -> softmax -> max.
    Checkpoints are read with ``weights_only=True`` (a ``state_dict`` of tensors; nothing in the file is executed)."""
    import os
    from pathlib import Path
    from PIL import Image
    from .face_models import get_model
    ckpt_root, proc_root = Path(checkpoints_dir), Path(proc_data_dir)
    if model_name is None:
        model_dirs = list(ckpt_root.glob(f'{model_type}_*'))
        if not model_dirs:
            raise ValueError(f"No trained models found for type: {model_type}")
        model_name = sorted(model_dirs)[-1].name
    model_checkpoint_dir = ckpt_root / model_name
    if not model_checkpoint_dir.exists():
        raise ValueError(f"Model not found: {model_name}")
    processed_dirs = [d for d in proc_root.iterdir() if d.is_dir() and (d / "train").exists()] if proc_root.exists() else []
    if not processed_dirs:
        raise ValueError("No processed datasets found.")
    if model_type == 'siamese':
        raise ValueError("Siamese model can't be used for direct prediction. Use it for verification.")
    classes = sorted(d.name for d in os.scandir(processed_dirs[0] / "train") if d.is_dir())   # ImageFolder(...).classes
    with Image.open(image_path) as im:
        u8 = resize_to_u8([im.convert('RGB')], (224, 224), device=device)                    # uint8 [1, 224, 224, 3] on the device
    model = get_model(model_type, num_classes=len(classes)).to(device)
    best_model_path, best_checkpoint_path = model_checkpoint_dir / 'best_model.pth', model_checkpoint_dir / 'best_checkpoint.pth'
    if best_model_path.exists():
        model.load_state_dict(torch.load(best_model_path, map_location=device, weights_only=True))
    elif best_checkpoint_path.exists():
        model.load_state_dict(torch.load(best_checkpoint_path, map_location=device, weights_only=True))
    else:
        raise FileNotFoundError(f"Neither best_model.pth nor best_checkpoint.pth found in {model_checkpoint_dir}")
    model.eval()
    with torch.no_grad():
        x = u8 if getattr(model, "supports_u8_input", False) else preprocess(u8, IMAGENET_MEAN, IMAGENET_STD, device)
        outputs = model(x)
        probs, pred = ops.softmax_argmax(outputs)
        pred_idx = int(pred[0])
        return classes[pred_idx], float(probs[0, pred_idx])


# ------------------------------------------------------------------------------------------------
# `src/testing.py:26-131`: the discovery half of the reference's evaluate_model (which checkpoint, which processed dataset)
# ------------------------------------------------------------------------------------------------
def find_processed_datasets(proc_data_dir: str):
    """`testing.py:43-70`: ``[(path, display name)]`` of the processed datasets that hold a ``test`` split - ``<config>/test``,
    ``<config>/<dataset>/test`` and the root's own ``test``, in the reference's order, de-duplicated by display name."""
    from pathlib import Path
    root = Path(proc_data_dir)
    found, names = [], set()
    if not root.exists():
        return found
    for config_dir in [d for d in root.iterdir() if d.is_dir() and d.name not in ["train", "val", "test"]]:
        if (config_dir / "test").exists():
            if config_dir.name not in names:
                found.append((config_dir, config_dir.name)); names.add(config_dir.name)
        else:
            for dataset_dir in config_dir.iterdir():
                if dataset_dir.is_dir() and (dataset_dir / "test").exists():
                    disp = f"{config_dir.name}/{dataset_dir.name}"
                    if disp not in names:
                        found.append((dataset_dir, disp)); names.add(disp)
    if (root / "test").exists() and "root" not in names:
        found.append((root, "processed (root)"))
    return found


def evaluate_trained(model_type: str, model_name: Optional[str] = None, auto_dataset: bool = True, dataset_index: int = 0,
                     checkpoints_dir: str = "outputs/checkpoints", proc_data_dir: str = "data/processed",
                     out_root: Optional[str] = "outputs", device="cuda") -> dict:
    """The reference's ``evaluate_model(model_type, model_name=None, auto_dataset=False)`` (`src/testing.py:26-394`) end to end on
    the HIP path: the latest ``<model_type>_*`` directory under ``checkpoints_dir`` unless ``model_name`` is given
    (``ValueError("No trained models found for type: …")`` / ``ValueError("Model not found: …")``, `:31-40`), the processed datasets
    that hold a ``test`` split (``ValueError("No processed datasets found with test data.")``, `:67`), ``best_model.pth`` before
    ``best_checkpoint.pth`` (``FileNotFoundError``, `:129`), then the loop + metrics + JSON of `evaluate_model` with results under
    ``<out_root>/<model_name>`` (`:94-96`).  The interactive dataset prompt (`:74-90`) becomes ``dataset_index`` (``auto_dataset``
    picks the first, as the reference's flag does).  Siamese pair datasets (`data_utils.SiameseDataset`) are outside the scope:
    pass pair batches to `evaluate_model` directly."""
    import os
    from pathlib import Path
    from .face_models import get_model
    ckpt_root = Path(checkpoints_dir)
    if model_name is None:
        model_dirs = list(ckpt_root.glob(f'{model_type}_*'))
        if not model_dirs:
            raise ValueError(f"No trained models found for type: {model_type}")
        model_name = sorted(model_dirs)[-1].name
    model_checkpoint_dir = ckpt_root / model_name
    if not model_checkpoint_dir.exists():
        raise ValueError(f"Model not found: {model_name}")
    processed_dirs = find_processed_datasets(proc_data_dir)
    if not processed_dirs:
        raise ValueError("No processed datasets found with test data.")
    if model_type == 'siamese':
        raise ValueError("siamese evaluation takes an iterable of (img1, img2, labels) batches: use evaluate_model")
    idx = 0 if auto_dataset else int(dataset_index)
    if not 0 <= idx < len(processed_dirs):
        raise ValueError(f"dataset_index {dataset_index} out of range (1..{len(processed_dirs)} datasets)")
    selected_data_dir, display = processed_dirs[idx]
    test_dir = selected_data_dir / "test"
    classes = sorted(d.name for d in os.scandir(test_dir) if d.is_dir())
    model = get_model(model_type, len(classes)).to(device)
    best_model_path, best_checkpoint_path = model_checkpoint_dir / 'best_model.pth', model_checkpoint_dir / 'best_checkpoint.pth'
    if best_model_path.exists():
        model.load_state_dict(torch.load(best_model_path, map_location=device, weights_only=True))
    elif best_checkpoint_path.exists():
        model.load_state_dict(torch.load(best_checkpoint_path, map_location=device, weights_only=True))
    else:
        raise FileNotFoundError(f"Neither best_model.pth nor best_checkpoint.pth found in {model_checkpoint_dir}")
    model.eval()
    out_dir = str(Path(out_root) / model_name) if out_root is not None else None
    return evaluate_model(model, model_type, str(test_dir), out_dir=out_dir, model_name=model_name, dataset_name=display, device=device)
