"""MI355X-native embedding-extraction + gallery-matching path (import name: ``frmap_amd``).

Mirrors the Python call surface of the reference's hot path:
``get_model`` / module ``forward`` / ``get_embedding`` (`/root/reference/src/face_models.py:785-813`)
and ``compare_faces`` / ``load_refs`` / ``save_refs`` (`/root/reference/src/app.py:50-123`), with the
compute in hand-written HIP kernels for gfx950 behind a C-ABI shared library
(``include/frmap_hip.h``).  There is no CPU fallback: calling a compute entry point without the
built library, or with host tensors, raises.
"""
from .face_models import (MODEL_TYPES, ArcFaceNet, ArcMarginProduct, AttentionNet, BaselineNet, EnsembleModel, HybridNet,
                          ResNetTransfer, SiameseNet, create_ensemble, get_model, set_default_compute_dtype)
from . import evaluate
from .matching import (REC_THRESH, Gallery, GraphedEmbedMatch, get_embedding, compare_faces, embed_and_match, load_refs, match_batch, save_refs)

__all__ = ["MODEL_TYPES", "get_model", "BaselineNet", "ResNetTransfer", "SiameseNet", "ArcFaceNet",
           "ArcMarginProduct", "HybridNet", "AttentionNet", "EnsembleModel", "create_ensemble", "set_default_compute_dtype", "compare_faces", "load_refs",
           "save_refs", "evaluate", "embed_and_match", "match_batch", "Gallery", "GraphedEmbedMatch", "get_embedding", "REC_THRESH"]
