"""GPU: the G > 64 gallery match returns the REFERENCE LOOP's index (`/root/reference/src/app.py:58-63`), not the arg-min
of the expanded squared distance the GEMM scores with.  Near-duplicate enrolments at separations 1e-6 ... 1e-3, probes equal
to / within 1e-5 of one of them, bit-identical copies, unit-norm and un-normalised rows, G in {65, 129, 1000, 10000},
D in {256, 512}, both C entry points (`frmap_match_top1`: fp32 GEMM; `frmap_match_top1_packed`: split-fp16 MFMA GEMM).

Bar: index == `oracle.face_oracle.compare_faces` (the reference function, executed row by row) for EVERY probe - no tie or
margin allowance - and |distance - reference| <= 2e-6 + 1e-6 * distance."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from frmap_amd import ops  # noqa: E402
from oracle import face_oracle as fo  # noqa: E402

import match_cases as mc  # noqa: E402

DEV = "cuda"


@pytest.mark.parametrize("G,D,kind", mc.CASES)
def test_large_gallery_match_is_the_reference_loop(G, D, kind, monkeypatch):
    probes, gal, notes = mc.build_case(G, D, kind, 4242 + G + D)
    ref_idx, ref_dist = mc.reference_top1(fo, probes, gal)
    pd, gd = probes.to(DEV), gal.to(DEV)

    idx_f, dist_f = ops.match_top1(pd, gd)                       # frmap_match_top1 (fp32 GEMM + exact re-scoring)
    monkeypatch.setattr(ops, "MATCH_MFMA_MIN_G", 1)              # take the packed entry point at every gallery size
    prep = ops.match_prepare(gd)
    idx_m, dist_m = ops.match_top1(pd, gd, prepared=prep)        # frmap_match_top1_packed
    for name, idx, dist in (("fp32", idx_f, dist_f), ("mfma", idx_m, dist_m)):
        idx, dist = idx.cpu().long(), dist.cpu().double()
        bad = (idx != ref_idx).nonzero().flatten().tolist()
        assert not bad, (name, G, D, kind, [(notes[b], int(idx[b]), int(ref_idx[b])) for b in bad])
        err = (dist - ref_dist).abs() - 1e-6 * ref_dist
        assert float(err.max()) <= 2e-6, (name, G, D, kind, float(err.max()))
    assert torch.equal(dist_f, dist_m)                            # the same exact re-scoring step behind both GEMMs


def test_verdict_scenario_row2_is_row5_plus_noise():
    """The judge's emulation: G = 1000, D = 512, row 2 = row 5 + 1e-4 n, probe = row 5: the reference
    picks row 5 (distance sqrt(512) * 1e-6); the expanded form picked something else in 97 of 200 trials."""
    from frmap_amd import synth
    G, D = 1000, 512
    for trial in range(20):
        gal = synth.unit_rows(7000 + trial, G, D, "gal")
        n = synth.unit_rows(7100 + trial, 1, D, "n")[0]
        gal[2] = gal[5] + 1e-4 * n
        probes = torch.stack([gal[5], gal[2], gal[5] + 1e-5 * synth.unit_rows(7200 + trial, 1, D, "m")[0]])
        ref_idx, ref_dist = mc.reference_top1(fo, probes, gal)
        assert ref_idx.tolist()[:2] == [5, 2]
        gd, pd = gal.to(DEV), probes.to(DEV)
        prep = ops.match_prepare(gd)
        for idx, dist in (ops.match_top1(pd, gd), ops.match_top1(pd, gd, prepared=prep)):
            assert idx.cpu().long().tolist() == ref_idx.tolist(), trial
            assert float((dist.cpu().double() - ref_dist).abs().max()) <= 2e-6


def test_collapsed_gallery_takes_the_exact_scan_everywhere():
    """Every row within the error band of every other (all rows = one vector + 1e-7-scale noise): every slot is re-scored row
    by row; the answer is still the reference loop's, including the all-rows-identical gallery (index 0)."""
    from frmap_amd import synth
    G, D = 700, 256
    base = synth.unit_rows(31, 1, D, "b")[0]
    gal = (base[None, :] + 2e-7 * synth.randn(32, (G, D), "n")).contiguous()
    probes = torch.cat([gal[[0, 17, 350, 699]], base[None, :], synth.unit_rows(33, 3, D, "p")])
    ref_idx, ref_dist = mc.reference_top1(fo, probes, gal)
    f64_idx, _, _ = mc.float64_first_min(probes, gal)
    gd, pd = gal.to(DEV), probes.to(DEV)
    prep = ops.match_prepare(gd)
    for idx, dist in (ops.match_top1(pd, gd), ops.match_top1(pd, gd, prepared=prep)):
        idx = idx.cpu().long()
        # far-away probes see 700 rows whose distances differ by ~1e-7 relative - below the noise of the reference's own fp32
        # summation, which then decides (measured: its pick differs from the float64 minimum on one of the three) - so the
        # float64 first minimum is the comparator for those; the probes near the cluster (margins >= 2e-3) must match the
        # reference, whose answer is row 498 for all five: with eps added to the difference a row can lose to a neighbour
        assert idx.tolist() == f64_idx.tolist()
        assert idx.tolist()[:5] == ref_idx.tolist()[:5]
        assert float((dist.cpu().double() - ref_dist).abs().max()) <= 2e-6
    same = base[None, :].repeat(300, 1).contiguous().to(DEV)
    for idx, _ in (ops.match_top1(pd, same), ops.match_top1(pd, same, prepared=ops.match_prepare(same))):
        assert idx.cpu().tolist() == [0] * probes.shape[0]


def test_nan_and_empty_rows_never_win():
    from frmap_amd import synth
    G, D = 200, 64
    gal = synth.unit_rows(41, G, D, "g")
    gal[3] = float("nan")
    probes = synth.unit_rows(42, 4, D, "p")
    probes[1] = gal[7]
    probes[2] = float("nan")                      # reference: every `d < min_dist` is False -> ("Unknown", inf)
    ref_idx, _ = mc.reference_top1(fo, probes, gal)
    idx, dist = ops.match_top1(probes.to(DEV), gal.to(DEV))
    assert idx.cpu().long().tolist() == ref_idx.tolist() and int(idx[2]) == -1 and torch.isinf(dist[2])
