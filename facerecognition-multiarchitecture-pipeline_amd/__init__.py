"""MI355X-native embedding-extraction + gallery-matching path (import name: ``frmap_amd``).

Mirrors the Python call surface of the reference's hot path:
``get_model`` / module ``forward`` / ``get_embedding`` (`/root/reference/src/face_models.py:785-813`)
and ``compare_faces`` / ``load_refs`` / ``save_refs`` (`/root/reference/src/app.py:50-123`), with the
compute in hand-written HIP kernels for gfx950 behind a C-ABI shared library
(``include/frmap_hip.h``).  There is no CPU fallback: calling a compute entry point without the
built library, or with host tensors, raises.
"""
__all__ = ["synth", "gallery_io"]
