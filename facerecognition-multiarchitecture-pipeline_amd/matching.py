"""Gallery matching with the reference's call surface, computed on the GPU.

Mirrors ``/root/reference/src/app.py``:

* ``compare_faces(emb, refs, thresh) -> (name, dist, idx|None)``  (`app.py:50-64`): Euclidean
  ``F.pairwise_distance`` (eps = 1e-6 added to the *difference*), first strict minimum, the
  ``("Unknown", dist, None)`` result above the threshold and the ``("Unknown", inf, None)``
  sentinel for ``None`` / empty input — never raises on those.
* ``load_refs()`` / ``save_refs(refs)``  (`app.py:67-123`): same pickle file layout, entries whose
  image file is missing are dropped on load; the file is read with a non-executing parser
  (``gallery_io``), not ``pickle.load``.
* ``embed_and_match(model, x, gallery, thresh)``: the batched form: for every b,
  ``ids[b], dists[b] == compare_faces(model(x[b:b+1]), refs, thresh)[2], [1]``.

The per-entry Python loop of the reference becomes one fp32 MFMA kernel
(``frmap_match_top1``) over a device-resident G×D gallery matrix.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import gallery_io, ops

_MFMA_MATCH = os.environ.get("FRMAP_MATCH_MFMA", "1") != "0"   # A/B switch: 0 = large galleries on the fp32 GEMM
REC_THRESH = 1.0                       # `app.py:20`
REF_DIR = "face_references"            # `app.py:23`
REF_FILE = os.path.join(REF_DIR, "face_references.pkl")   # `app.py:24`
_save_counter = 0


class Gallery:
    """Device-resident gallery: names + one fp32 G×D matrix (row i = reference i's embedding), held in a buffer with spare
    capacity so that enrolling an identity (`app.py:428-436` appends one entry) writes ONE row - and, for galleries on the MFMA
    match path, re-packs one 64-row tile - instead of rebuilding and re-uploading everything."""

    def __init__(self, names: Sequence[str], embeddings: torch.Tensor, device: Union[str, torch.device] = "cuda"):
        emb = embeddings.detach().to(torch.float32)
        if emb.dim() == 3 and emb.shape[1] == 1:
            emb = emb[:, 0, :]
        if emb.dim() != 2 or emb.shape[0] != len(names):
            raise ValueError("Gallery: need one D-vector per name")
        self.names = list(names)
        self._buf = emb.to(device).contiguous()          # [capacity][D]; rows >= len(names) are spare
        self._pack = None
        self._refresh_pack()

    @property
    def matrix(self) -> torch.Tensor:
        return self._buf[: len(self.names)]

    def _wants_pack(self) -> bool:
        return len(self.names) >= ops.MATCH_MFMA_MIN_G and self._buf.shape[1] % 32 == 0 and _MFMA_MATCH

    def _refresh_pack(self) -> None:
        # built HERE (construction / enrolment time, on the caller's stream) and guarded by an event - not lazily on whichever
        # side stream first matches against it
        if self._wants_pack() and (self._pack is None or not self._pack.matches(self.matrix)):
            with torch.cuda.device(self._buf.device):
                self._pack = ops.MatchPack(self.matrix, capacity=self._buf.shape[0])

    @property
    def prepared(self):
        """The gallery split for the MFMA match path (galleries of >= `ops.MATCH_MFMA_MIN_G` rows), else None."""
        if not self._wants_pack():
            return None
        self._refresh_pack()         # (someone wrote into `matrix` behind our back: rebuild rather than match stale rows)
        return self._pack

    def append(self, name: str, embedding: torch.Tensor) -> int:
        """Enrol one identity (`app.py:428-436`): returns its row.  One row is written on the device; the MFMA pack (if any)
        re-packs only that row's tile.  Capacity doubles when exhausted."""
        e = embedding.detach().reshape(-1).to(torch.float32)
        G, D = len(self.names), self._buf.shape[1]
        if G == 0 and D != e.numel():
            self._buf = torch.empty((16, e.numel()), dtype=torch.float32, device=self._buf.device)
            D = e.numel()
        if e.numel() != D:
            raise ValueError(f"Gallery.append: embedding has {e.numel()} values, the gallery rows have {D}")
        with torch.cuda.device(self._buf.device):
            if G == self._buf.shape[0]:
                cap = max(2 * G, 16)
                cap = (cap + 255) // 256 * 256 if cap >= ops.MATCH_MFMA_MIN_G // 2 else cap
                grown = torch.empty((cap, D), dtype=torch.float32, device=self._buf.device)
                grown[:G] = self._buf[:G]
                self._buf, self._pack = grown, None
            self._buf[G].copy_(e.to(self._buf.device), non_blocking=True)
            self.names.append(name)
            if self._wants_pack():
                if self._pack is not None and self._pack.src_ptr == self._buf.data_ptr() and self._pack.G == G:
                    self._pack.update_rows(self.matrix, G, G + 1)
                else:
                    self._pack = ops.MatchPack(self.matrix, capacity=self._buf.shape[0])
        return G

    @classmethod
    def from_refs(cls, refs: Sequence[dict], device: Union[str, torch.device] = "cuda") -> "Gallery":
        if not refs:
            return cls([], torch.zeros((0, 1)), device)
        rows = [r["embedding"].detach().reshape(-1) for r in refs]
        if len({(t.device, t.dtype) for t in rows}) == 1:
            mat = torch.stack(rows)                      # one gather on the tensors' own device, one transfer
        else:
            mat = torch.stack([t.to(torch.float32).cpu() for t in rows])
        return cls([r["name"] for r in refs], mat, device)

    def __len__(self):
        return len(self.names)


_gallery_cache: dict = {}


def _ref_tag(r) -> tuple:
    e = r.get("embedding") if isinstance(r, dict) else None
    if isinstance(e, torch.Tensor):
        return (id(e), e.data_ptr(), e._version, tuple(e.shape), r.get("name"))
    return (id(e), r.get("name") if isinstance(r, dict) else None)


def _refs_tag(refs) -> tuple:
    """Content tag of a ``refs`` list: per entry the embedding object's identity, storage address and in-place
    version counter — an in-place edit of an enrolled embedding, a replaced entry or a list that was freed and
    reallocated at the same ``id`` all change it."""
    return (len(refs),) + tuple(_ref_tag(r) for r in refs)


def _as_gallery(refs, device) -> Gallery:
    if isinstance(refs, Gallery):
        return refs
    # the demo passes the same list object every frame (`app.py:639`): keep its device matrix while the list is unchanged,
    # and when entries were only APPENDED (enrolment, `app.py:428-436`) append their rows instead of rebuilding
    key = id(refs)
    tag = _refs_tag(refs)
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    hit = _gallery_cache.get(key)
    if hit is not None and hit[1].matrix.device == dev:
        old_tag, g = hit
        if old_tag == tag:
            return g
        n_old = old_tag[0]
        if 0 < n_old < tag[0] and tag[1: 1 + n_old] == old_tag[1:] and len(g) == n_old:
            try:
                for r in refs[n_old:]:
                    g.append(r["name"], r["embedding"])
                _gallery_cache[key] = (tag, g)
                return g
            except Exception:
                pass                                     # (ragged entry: fall through to the full rebuild and its error)
    g = Gallery.from_refs(refs, dev)
    if len(_gallery_cache) > 8:
        _gallery_cache.clear()
    _gallery_cache[key] = (tag, g)
    return g


def match_batch(emb: torch.Tensor, gallery: Gallery) -> Tuple[torch.Tensor, torch.Tensor]:
    """B×D device embeddings → (int32[B] first-arg-min index, fp32[B] distance), on the device."""
    return ops.match_top1(emb.to(torch.float32), gallery.matrix, prepared=gallery.prepared)


def get_embedding(face_img, model):
    """`app.py:32-48`: BGR uint8 crop → RGB → Resize((160,160)) → ToTensor → Normalize(0.5, 0.5) →
    ``model(x)`` on the model's device under ``no_grad``; ``None`` for an empty crop or on ANY
    exception (the reference swallows them, `:46-48`).  The crop is uploaded as it is; the resize (bit-exact with
    PIL's, `resize.resize_bilinear_u8`), the uint8 → normalised-float step and the model run on the GPU."""
    if face_img is None or getattr(face_img, "size", 0) == 0:
        return None
    try:
        from . import resize as _resize
        rgb = np.ascontiguousarray(np.asarray(face_img)[:, :, ::-1])
        dev = next(model.parameters()).device
        u8 = _resize.resize_bilinear_u8([rgb], (160, 160), dev)
        x = ops.normalize_u8(u8, (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))[0]
        with torch.no_grad():
            return model(x)
    except Exception:
        return None


def compare_faces(emb, refs, thresh):
    """`app.py:50-64` on the GPU.  ``refs``: the reference's list of dicts (its device copy is cached and re-validated against the
    live list on every call - an O(len(refs)) walk over version counters, ~1 us per entry: fine for the demo's tens of entries) or
    a `Gallery` (no walk: what a host with thousands of identities should hold; `Gallery.append` enrols in O(1))."""
    if emb is None or refs is None or len(refs) == 0:
        return "Unknown", float('inf'), None
    dev = emb.device if emb.is_cuda else torch.device("cuda")
    g = _as_gallery(refs, dev)
    e = emb.detach().reshape(1, -1).to(device=dev, dtype=torch.float32)
    # one launch sequence, ONE device -> host copy: the int32 [1, 2] record (index, bits of the distance)
    rec = ops.match_top1(e, g.matrix, float("inf"), packed=True, prepared=g.prepared)[3].cpu()
    best_ref_idx = int(rec[0, 0])
    min_dist = float(rec.view(torch.float32)[0, 1])
    if best_ref_idx < 0:                                  # every distance NaN: the reference's loop never updates its minimum
        return "Unknown", float('inf'), None
    if min_dist <= thresh:
        return g.names[best_ref_idx], min_dist, best_ref_idx
    return "Unknown", min_dist, None


def embed_and_match(model, x: torch.Tensor, gallery, thresh: float = REC_THRESH,
                    normalize: bool = False, packed: bool = False):
    """Embed a batch and match every face.  Returns ``(ids int32[B], dists fp32[B])`` on the device,
    ``ids[b] = -1`` where the best distance exceeds ``thresh`` (compare_faces' "Unknown").
    ``normalize=True`` L2-normalises the embeddings first (for models whose embedding is not
    unit-norm: 'baseline', 'cnn', 'hybrid').  ``packed=True`` returns instead the int32 ``[B, 2]``
    record tensor ``(id, bits(dist))`` the multi-GPU all-gather ships (``dist.gather_packed``); ``packed=<tensor>``
    writes those records into the given int32 ``[B, 2]`` buffer (a slice of a larger step buffer)."""
    g = _as_gallery(gallery, x.device if isinstance(x, torch.Tensor) and x.is_cuda else "cuda")
    h = model.model_handle() if hasattr(model, "model_handle") else None
    if h is not None:
        # 'cnn' / 'arcface': forward + match as ONE call on the model handle (`frmap_model_embed_and_match`)
        _idx, dist, ids, pk, _ = h.embed_and_match(model._check_input(x), g.matrix if len(g) else None, g.prepared if len(g) else None,
                                                   thresh, normalize, packed=packed)
        return pk if (packed is not None and packed is not False) else (ids, dist)
    fmap = model.trunk_map(x) if hasattr(model, "trunk_map") and len(g) <= 64 else None
    if fmap is not None:
        # embedding == global average pool of the trunk map (ResNetTransfer): pool + normalise + match in one launch
        _idx, dist, ids, pk, _ = ops.gap_norm_match(fmap, g.matrix if len(g) else None, thresh, normalize=normalize, packed=packed)
        return pk if (packed is not None and packed is not False) else (ids, dist)
    if normalize and hasattr(model, "unit_embedding"):
        emb = model.unit_embedding(x)          # the head kernel's unit-norm output (BaselineNet)
    else:
        emb = model.get_embedding(x)
        if emb.dim() == 1:
            emb = emb.unsqueeze(0)
        if normalize:
            emb = ops.l2_normalize(emb, 1e-12)
    g = _as_gallery(gallery, emb.device)
    if packed is not None and packed is not False:
        return ops.match_top1(emb.to(torch.float32), g.matrix, thresh, packed=packed, prepared=g.prepared)[3]
    _idx, dist, ids = ops.match_top1(emb.to(torch.float32), g.matrix, thresh, prepared=g.prepared)
    return ids, dist


class GraphedEmbedMatch:
    """The batched embed → (normalise) → match step captured once into a HIP graph and replayed.

    One step is ~30 kernel launches of 5–150 µs; driven from Python each costs ~10 µs of host time,
    which caps how many concurrent streams can be kept fed.  Capturing the launches (hipGraph via
    ``torch.cuda.CUDAGraph`` — our kernels are plain launches on the capturing stream) removes the
    per-launch host cost, and splitting the batch over ``streams`` concurrent branches lets one
    branch's tail waves run beside another's full waves (at 256 faces the late ResNet layers have
    only 392–784 tiles for 512 workgroup slots).

    ``x`` is the static input buffer (fp32 NCHW on the device): write the next batch into
    ``pipeline.x`` (or pass it to ``__call__``, which copies it) and call; the result is the int32
    ``[B, 2]`` record tensor ``(id-or-unknown, bits(dist))`` — ``ids()`` / ``dists()`` give views.
    """

    def __init__(self, model, gallery, x: torch.Tensor, thresh: float = REC_THRESH, normalize: bool = False,
                 streams: int = 1):
        if not x.is_cuda:
            raise RuntimeError("GraphedEmbedMatch needs a device-resident input buffer (no CPU fallback)")
        self.model, self.x, self.thresh, self.normalize = model, x, float(thresh), bool(normalize)
        self.gallery = _as_gallery(gallery, x.device)
        self.streams = max(1, min(int(streams), x.shape[0]))
        self._side = [torch.cuda.Stream(device=x.device) for _ in range(self.streams - 1)]
        self._xs = list(self.x.chunk(self.streams))
        warm = torch.cuda.Stream(device=x.device)
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm), torch.no_grad():   # plans, kernel attributes, allocator pools
            for _ in range(2):
                self._run()
        torch.cuda.current_stream().wait_stream(warm)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: calls other threads make meanwhile (e.g. the NCCL watchdog polling events) must not invalidate the capture
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"), torch.no_grad():
            self.records = self._run()

    def _one(self, xs, out):
        return embed_and_match(self.model, xs, self.gallery, self.thresh, normalize=self.normalize, packed=out)

    def _run(self):
        # every micro-batch writes its records straight into its slice of one buffer (no concatenation kernel at the join)
        rec = torch.empty((self.x.shape[0], 2), dtype=torch.int32, device=self.x.device)
        if self.streams == 1:
            return self._one(self.x, rec)
        main = torch.cuda.current_stream()
        lo, slices = 0, []
        for xs in self._xs:
            slices.append(rec[lo: lo + xs.shape[0]])
            lo += xs.shape[0]
        for i, st in enumerate(self._side):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                self._one(self._xs[i + 1], slices[i + 1])
        self._one(self._xs[0], slices[0])
        for st in self._side:
            main.wait_stream(st)
        return rec

    def __call__(self, x: Optional[torch.Tensor] = None) -> torch.Tensor:
        if x is not None and x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.records

    def ids(self) -> torch.Tensor:
        return self.records[:, 0]

    def dists(self) -> torch.Tensor:
        return self.records.view(torch.float32)[:, 1]


# ------------------------------------------------------------------------------------------------
# persistence (`app.py:67-123`)
# ------------------------------------------------------------------------------------------------
def _imread_bgr(path: str):
    try:
        from PIL import Image
        with Image.open(path) as im:
            return np.asarray(im.convert("RGB"))[:, :, ::-1].copy()
    except Exception:
        return None


def _imwrite_bgr(path: str, img) -> bool:
    try:
        from PIL import Image
        Image.fromarray(np.asarray(img)[:, :, ::-1]).save(path)
        return True
    except Exception:
        return False


def load_refs(ref_file: Optional[str] = None) -> List[dict]:
    """`app.py:104-123`: ``[]`` if the file is missing or unreadable; entries whose image is missing
    are skipped; embeddings come back as CPU fp32 tensors."""
    ref_file = ref_file or REF_FILE
    if not os.path.exists(ref_file):
        return []
    refs = []
    try:
        for rec in gallery_io.read_gallery_file(ref_file):
            p = rec["image_path"]
            if p and os.path.exists(p):                   # as stored, relative to the working directory (`app.py:110`)
                img = _imread_bgr(p)
                if img is not None:
                    refs.append({'name': rec['name'], 'embedding': torch.tensor(rec['embedding_numpy']).cpu(),
                                 'image': img})
        return refs
    except Exception:
        return []


def save_refs(refs: Sequence[dict], ref_file: Optional[str] = None) -> bool:
    """`app.py:67-91`: one JPEG per entry (``<name>_<counter:08x>.jpg``) + the pickle list."""
    global _save_counter
    ref_file = ref_file or REF_FILE
    ref_dir = os.path.dirname(os.path.abspath(ref_file))
    try:
        os.makedirs(ref_dir, exist_ok=True)
        records = []
        for ref in refs:
            _save_counter += 1
            img_file = f"{ref['name'].replace(' ', '_')}_{_save_counter:08x}.jpg"
            img_path = os.path.join(ref_dir, img_file)
            if _imwrite_bgr(img_path, ref['image']):
                records.append({'name': ref['name'], 'embedding_numpy': ref['embedding'].detach().cpu().numpy(),
                                'image_path': img_path})
        gallery_io.write_gallery_file(ref_file, records)
        return True
    except Exception:
        return False
