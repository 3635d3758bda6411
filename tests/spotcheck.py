"""Spot verification of a large batch against the oracle's fp64 anchor: whole recordings at benchmark size are far too long
for the CPU oracle, but any stretch of evaluations depends only on the samples under it.  Stretches are aimed at the places
a tiling bug would show: the first and last evaluations of a channel (ragged tail, 32-bit offsets of the last rows), and both
sides of the seams between workgroup segments.  Test infrastructure (uses the oracle): imported by tests/ and by bench.py's
verification leg only."""
import numpy as np

import pyoracle as po
import util


def stretches(E: int, seg: int, width: int = 160, max_seams: int = 3):
    """[(e0, e1)]: the head, the tail, and `width` evaluations around up to `max_seams` seams (first, middle, last)."""
    out = [(0, min(width, E)), (max(0, E - width), E)]
    if seg > 0:
        seams = list(range(seg, E, seg))
        pick = sorted({seams[0], seams[len(seams) // 2], seams[-1]}) if seams else []
        for s in pick[:max_seams]:
            out.append((max(0, s - width // 2), min(E, s + width // 2)))
    return sorted(set(out))


def plant(det, cfg, x, channels, width: int = 160, seed: int = 77) -> int:
    """Writes template syllables (synth.syllable_channel: the audio the sample network was built to fire on, at jittered
    positions and levels) over the stretches check() reads of the chosen channels of the device tensor x [C, S] -- the benchmark's
    ordinary audio never fires the sample network, so flags compared on it are all zero.  Returns the number of stretches
    written; 0 when the configuration is not the template's (bins x timeRange)."""
    import torch
    from syllable_detector_swift_amd import synth
    tpl = util.template()
    g = det.geometry
    if tpl.shape != (cfg.timeRange, g.f1 - g.f0):
        return 0
    S = int(x.shape[1])
    E = det.countEvaluations(S)
    T, W = cfg.timeRange, cfg.windowLength
    n = 0
    for c in channels:
        for e0, e1 in stretches(E, det.segmentEvaluations(S), width):
            s0 = e0 * g.hop
            s1 = (e1 - 1 + T - 1) * g.hop + g.gap + W
            a = synth.syllable_channel(s1 - s0, tpl, seed=seed + 131 * c + n, hop=g.hop, window=W, f0=g.f0,
                                       fourier_length=cfg.fourierLength, every=4000)
            x[c, s0:s1] = torch.from_numpy(a).to(x.device)
            n += 1
    return n


def check(det, cfg, x, outputs, flags, channels, width: int = 160, tol: float = util.TOL):
    """x [C, S], outputs [C, E, n_out], flags [C, E]: device tensors of one batch call of `det`.  Compares the chosen stretches
    of the chosen channels with the anchor: values to `tol`, flags exactly outside the guard band.  Returns a summary dict;
    raises AssertionError on a mismatch."""
    S = int(x.shape[1])
    E = det.countEvaluations(S)
    g = det.geometry
    T, W = cfg.timeRange, cfg.windowLength
    seg = det.segmentEvaluations(S)
    o = util.oracle_for(cfg)
    n_evals, worst, fired = 0, 0.0, 0
    for c in channels:
        for e0, e1 in stretches(E, seg, width):
            s0 = e0 * g.hop
            s1 = (e1 - 1 + T - 1) * g.hop + g.gap + W
            xs = x[c, s0:s1].cpu().numpy()
            _, _, w64 = o.run(xs, po.F64, cfg.rule)
            assert w64.shape[0] == e1 - e0, (w64.shape, e0, e1)
            got = outputs[c, e0:e1].cpu().numpy()
            gfl = flags[c, e0:e1].cpu().numpy()
            ok = np.isfinite(w64).all(axis=1)
            assert (np.isfinite(got).all(axis=1) == ok).all(), "channel %d evaluations %d..%d: NaN pattern differs" % (c, e0, e1)
            if ok.any():
                err = np.abs(got[ok] - w64[ok]) / np.maximum(1.0, np.abs(w64[ok]))
                worst = max(worst, float(err.max()))
                assert err.max() <= tol, "channel %d evaluations %d..%d: error %.3g > %.1g" % (c, e0, e1, err.max(), tol)
                util.assert_flags_exact(gfl[ok], w64[ok], cfg.thresholds, cfg.rule, tol)
            n_evals += e1 - e0
            fired += int(gfl[ok].sum())
    return {"channels": list(channels), "evaluations_checked": n_evals, "detections": fired, "segment_evaluations": seg,
            "stretches_per_channel": len(stretches(E, seg, width)), "max_error": worst, "tolerance": tol,
            "against": "oracle fp64 anchor (parity unpinned: no reference vectors exist)"}
