"""Shared inputs of the near-duplicate gallery-match tests (`test_match_exact_gpu.py`, `test_oracle_golden.py`).

A case is a gallery that holds GROUPS of near-duplicate rows - the same identity enrolled several times
(`/root/reference/src/app.py:428-436` appends without de-duplicating) - at separations 1e-6 ... 1e-3 placed both BEFORE and
AFTER their base row, bit-identical copies, and probes that are equal to / within 1e-5 of one member of a group.  These are
the inputs on which the arg-min of an EXPANDED squared distance (|a|^2 + |g|^2 - 2 a.g, fp32) is decided by rounding noise
while the reference loop (`app.py:58-63`: exact ||(a - g) + 1e-6||_2 per row, first strict minimum) is not.
"""
import numpy as np
import torch

from frmap_amd import synth

SEPARATIONS = (1e-6, 1e-5, 1e-4, 1e-3)


def build_case(G: int, D: int, kind: str, seed: int):
    """Returns (probes [P, D], gallery [G, D], notes) as float32 CPU tensors.  kind: 'unit' (unit-norm rows, what
    ArcFace / Siamese / InceptionResnetV1 embeddings are) or 'raw' (un-normalised rows of norm ~ 17, what
    `ResNetTransfer.get_embedding` returns and `compare_faces` is equally happy to take)."""
    scale = 1.0 if kind == "unit" else 17.0
    gal = synth.unit_rows(seed, G, D, "gal") * scale
    rng = np.random.default_rng(seed + 7)
    probes, notes = [], []
    n_groups = len(SEPARATIONS)
    # base rows spread over the gallery (first slot, slot boundaries, last rows)
    bases = sorted(set(int(v) for v in np.linspace(5, G - 6, n_groups)))
    used = set(bases)

    def free(pos):
        pos %= G
        while pos in used:
            pos = (pos + 1) % G
        used.add(pos)
        return pos

    for gi, (base, sep) in enumerate(zip(bases, SEPARATIONS)):
        # copies a few rows BEFORE the base row, a few rows after it, and half a gallery away (another slot / tile)
        members = [base, free(base - 3), free(base + 9), free(base + G // 2)]
        for pos in members[1:]:
            n = rng.standard_normal(D)
            n /= np.linalg.norm(n)
            gal[pos] = (gal[base].double() + sep * scale * torch.from_numpy(n)).float()
        for member in members:
            probes.append(gal[member].clone())                                  # equal to a member
            notes.append(("equal", gi, member))
            n = rng.standard_normal(D)
            n /= np.linalg.norm(n)
            probes.append((gal[member].double() + 1e-5 * scale * torch.from_numpy(n)).float())   # within 1e-5 of it
            notes.append(("near", gi, member))
    # bit-identical copies: the FIRST must win
    src = free(bases[0] + 1)
    for pos in (free(src + 11), free(G - 2)):
        gal[pos] = gal[src]
    src = min(src, *[i for i in range(G) if torch.equal(gal[i], gal[src])])
    probes.append(gal[src].clone()); notes.append(("dup", -1, src))
    # ordinary probes
    rnd = synth.unit_rows(seed + 1, 6, D, "probe") * scale
    for r in rnd:
        probes.append(r); notes.append(("random", -1, -1))
    return torch.stack(probes).contiguous(), gal.contiguous(), notes


def reference_top1(fo, probes: torch.Tensor, gal: torch.Tensor):
    """The reference loop itself (`oracle.face_oracle.compare_faces`, pinned to `app.py:50-64`), one probe at a time."""
    refs = [{"name": str(i), "embedding": gal[i:i + 1]} for i in range(gal.shape[0])]
    idx, dist = [], []
    for p in probes:
        _, d, i = fo.compare_faces(p[None], refs, float("inf"))
        idx.append(-1 if i is None else i); dist.append(d)
    return torch.tensor(idx, dtype=torch.int64), torch.tensor(dist, dtype=torch.float64)


def float64_first_min(probes: torch.Tensor, gal: torch.Tensor):
    """What the HIP finalize step computes: fp32 elements ((a - g) + 1e-6f), squares summed in float64, first minimum."""
    diff = (probes[:, None, :] - gal[None, :, :]) + torch.tensor(1e-6, dtype=torch.float32)
    d2 = diff.double().pow(2).sum(-1)
    dmin, imin = d2.min(dim=1)
    first = (d2 == dmin[:, None]).float().argmax(dim=1)     # lowest index attaining the minimum
    return first, dmin.sqrt(), d2


CASES = [(G, D, kind) for G in (65, 129, 1000, 10000) for D in (256, 512) for kind in ("unit", "raw")]
