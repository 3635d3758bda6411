"""Both fused kernels against the oracle on inputs at the edge of what fp32 audio can hold: silence, NaN and infinite
samples, recordings scaled by 1e+-30, level steps of 240 dB inside a recording.  No allowance: values to 1e-5, flags exact,
NaN exactly where the reference has it."""
import numpy as np
import pytest

import pyoracle as po
import util
import syllable_detector_swift_amd as sd
from syllable_detector_swift_amd import synth

pytestmark = pytest.mark.gpu

S = 64 * 132 * 6 + 500


def _cases():
    base = synth.syllable_channel(S, util.template(), seed=5).astype(np.float32)
    out = {"plain": base.copy()}
    z = base.copy(); z[20000:30000] = 0.0; out["zero stretch"] = z
    out["all zero"] = np.zeros_like(base)
    z = base.copy(); z[25000] = np.nan; out["one NaN"] = z
    z = base.copy(); z[25000] = np.inf; out["one inf"] = z
    out["x 1e30"] = (base * np.float32(1e30)).astype(np.float32)
    out["x 1e-30"] = (base * np.float32(1e-30)).astype(np.float32)
    z = base.copy(); z[S // 2:] *= np.float32(1e-12); out["step 1e-12"] = z
    z = base.copy(); z[S // 2:] *= np.float32(1e12); out["step 1e12"] = z
    # what a cage recording does: a click 90 dB above a quiet stretch, a quiet stretch right behind a loud one
    z = (base * np.float32(3e-5)).astype(np.float32); z[26000] = 1.0; z[41000] = -0.7; out["clicks over quiet audio"] = z
    z = base.copy(); z[S // 2 + 777:] *= np.float32(1e-4); out["80 dB down"] = z
    z = base.copy(); z[:S // 2 + 777] *= np.float32(1e-5); out["100 dB up"] = z
    return out


@pytest.mark.parametrize("kernel", util.FUSED_KERNELS)
@pytest.mark.parametrize("name", list(_cases()))
def test_extreme_inputs(oracle_lib, monkeypatch, kernel, name):
    import torch
    util.select_fused(monkeypatch, kernel)
    cfg = util.sample_net()
    x = _cases()[name]
    with sd.SyllableDetector(cfg, channels=1) as det:
        det.profile(True)
        out, fl = det.run(torch.from_numpy(x[None]).cuda())
        torch.cuda.synchronize()
        assert det.lastTimings()[0][0] == kernel
        out, fl = out.cpu().numpy()[0], fl.cpu().numpy()[0]
        items, over = det.fixupStats()
    _, wfl, w64 = util.oracle_for(cfg).run(x, po.F64)
    ok, okg = np.isfinite(w64).all(axis=1), np.isfinite(out).all(axis=1)
    # NaN exactly where the reference has it (the windows that contain the offending sample, NeuralNet.swift:47-59): what the
    # block-floating-point grid cannot hold is recomputed from the samples (precision guard, include/syldet.h)
    assert (ok == okg).all(), "NaN evaluations must coincide with the reference's: %d extra, %d missing" % ((ok & ~okg).sum(), (okg & ~ok).sum())
    if ok.any():
        tol = 1e-5
        assert np.abs(out[ok] - w64[ok]).max() <= tol * max(1.0, np.abs(w64[ok]).max())
        util.assert_flags_exact(fl[ok], w64[ok], cfg.thresholds, cfg.rule, tol)
    assert not fl[~okg].any()
    # the guard is at work exactly where it has to be: ordinary audio (and whole-recording scalings) never reach the slow path
    if name in ("plain", "x 1e30"):
        assert items == 0, "%d work items for ordinary audio" % items
    # (the pass-scaled kernels send a 240 dB step to the exact path; the symmetric-fold kernel scales every frame by itself and
    # holds it natively -- only what no grid can hold, an infinite sample, goes to the slow path there)
    if name == "one inf" or (kernel != "fused_s_kernel" and name in ("step 1e-12", "step 1e12")):
        assert items > 0
    assert over == 0


# (|X|^2 of a recording at 1e30 is not a case: it does not fit fp32 -- the reference's vDSP_zvmags overflows too)
@pytest.mark.parametrize("name,spectrum", [(name, spectrum) for spectrum in (0, 1)
                                           for name in ("plain", "one NaN", "one inf", "x 1e30", "step 1e12", "step 1e-12", "clicks over quiet audio")
                                           if not (spectrum and name == "x 1e30")])
def test_extreme_inputs_spectrogram(oracle_lib, name, spectrum):
    """The spectrogram API (the fused engine's DFT half) on the same inputs: every column to 1e-5 of its largest value, NaN
    exactly in the frames that contain the offending sample."""
    import torch
    cfg = util.sample_net()
    cfg.spectrum = spectrum
    x = _cases()[name]
    with sd.SyllableDetector(cfg, channels=1, engine=(1 if spectrum else 0)) as det:      # (|X|^2 columns: generic network engine)
        det.profile(True)
        cols = det.spectrogram(torch.from_numpy(x[None]).cuda())
        torch.cuda.synchronize()
        names = util.launched(det)
        cols = cols.cpu().numpy()[0]
        items, over = det.fixupStats()
    o = util.oracle_for(cfg)
    want = o.spectrogram(x, po.F64)
    ok = np.isfinite(want).all(axis=1)
    assert (np.isfinite(cols).all(axis=1) == ok).all()
    util.assert_columns_close(cols[ok], want[ok])
    if names == ["fused_kernel (spectrogram)"]:
        assert over == 0 and (items > 0) == (name in ("one inf", "step 1e12"))
    if names == ["fused_s_kernel (spectrogram)"]:      # (every frame has its own scale: only what no grid holds is recomputed)
        assert over == 0 and (items > 0) == (name == "one inf")


@pytest.mark.parametrize("chain", [(), ("l2normalize",), ("l2normalize", "mapminmax"), ("normalize",), ("normalizestd", "mapstd"), ("mapstd", "mapminmax")])
@pytest.mark.parametrize("level", [1.0, 1e-3])
def test_guard_stays_quiet_on_ordinary_audio(oracle_lib, chain, level):
    """The slow, exact path is for what the grid cannot hold.  Ordinary audio at an ordinary level -- any input chain, both
    fused kernels (4 and 8 hidden units) -- must never reach it: a guard that cries wolf costs a factor of three in speed
    without anyone noticing (results are right either way)."""
    import torch
    from syllable_detector_swift_amd import nets
    base = util.sample_net()
    rng = np.random.default_rng(12)
    x = (synth.channels(2, 132 * 700 + 256, first=3) * level).astype(np.float32)
    for H in (4, 8):
        cfg = nets.variant(base, net=nets.random_net(rng, 290, (H,), 1, in_fns=chain))
        with sd.SyllableDetector(cfg, channels=2) as det:
            out, fl = det.run(torch.from_numpy(x).cuda())
            torch.cuda.synchronize()
            assert det.geometry.engine == 2
            items, over = det.fixupStats()
            # Without a normaliser the network sees the columns at the recording's level: at level 1 the bursts' columns
            # (|X| ~ 20, under gains of 1 .. 2: network inputs of 50) put the matrix-core arithmetic's 2^-21.4 of that level past the
            # 1e-5 bar where an fp32 FFT still holds it -- those windows ARE what the grid cannot hold to the contract, and the fold
            # kernel hands exactly them to the exact path (guard_loud).  Everything else must never reach it.
            loud = level == 1.0 and (not chain or chain[0] not in ("l2normalize", "normalize", "normalizestd"))
            assert over == 0 and ((items == 0) or loud), (chain, H, (items, over))
            if loud:
                assert items < (132 * 700 // 132) * 2 // 16 // 2, items            # at most the bursts, never the noise between them
            out = out.cpu().numpy()
        o = util.oracle_for(cfg)
        for c in range(2):
            _, _, w64 = o.run(x[c], po.F64)
            w32 = o.run(x[c], po.F32)[0]
            own = float(np.abs(w32 - w64).max())
            # (one bar for every chain: 1e-5, or 4x the fp32 port's own distance -- no allowance for the recording's level)
            util.assert_outputs_close(out[c], w64, max(util.TOL, 4 * own))


def test_profile_lists_the_exact_path_only_for_calls_that_gave_it_work():
    """syldet_timings (Time.swift:36-100's place): the fix-up launch behind a fused kernel is timed like every other kernel and listed
    for the calls whose work list was not empty -- a recording ten times full scale through a network without a normaliser spends most
    of its time there (MEASUREMENTS R5.7), ordinary audio none."""
    import torch
    from syllable_detector_swift_amd import nets
    base = util.sample_net()
    cfg = nets.variant(base, net=nets.random_net(np.random.default_rng(3), 290, (4,), 1, in_fns=()))
    x = torch.from_numpy(synth.channels(4, 132 * 2000 + 256, first=1).astype(np.float32)).cuda()
    with sd.SyllableDetector(cfg, channels=4) as det:
        det.profile(True, history=3)
        for level in (0.01, 10.0, 0.01):
            det.run(x * level)
        torch.cuda.synchronize()
        quiet2, loud, quiet1 = (det.timingsOf(k) for k in range(3))
        items, over = det.fixupStats()
    assert [n for n, _ in quiet1] == ["fused_s_kernel"] and [n for n, _ in quiet2] == ["fused_s_kernel"] and (items, over) == (0, 0)
    assert [n for n, _ in loud] == ["fused_s_kernel", "fixup_kernel"] and loud[1][1] > loud[0][1]
