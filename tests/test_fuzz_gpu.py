"""Seeded random sweep over the configuration space the fused engine accepts (windows, gaps, zero padding, bin
ranges, timeRange, layer shapes, input chains, scalings, spectrum modes, output rules, lengths, channel counts):
every draw is checked against the oracle's fp64 anchor to the 1e-5 bar, flags exactly, on whichever engine
`AUTO` selects -- and most draws must land on the fused one."""
import os

import numpy as np
import pytest

import pyoracle as po
import util
import syllable_detector_swift_amd as sd
from syllable_detector_swift_amd import _abi, nets, synth
from syllable_detector_swift_amd.config import SyllableDetectorConfig, frequencyIndexRange

pytestmark = pytest.mark.gpu

FS = 44100.0


def draw(rng):
    W = int(rng.choice([32, 64, 96, 128, 192, 256]))
    N = 1 << int(np.ceil(np.log2(W)))
    if rng.random() < 0.3 and N < 512:
        N *= 2                                                   # zero padding
    if rng.random() < 0.15:
        ov = -4 * int(rng.integers(1, 9))                        # a gap between windows
    else:
        ov = 4 * int(rng.integers(0, W // 8 + 1))                # hop = W - ov stays a multiple of 4
        ov = min(ov, W - 4)
    hop = max(0, -ov) + W - max(0, ov)
    if hop % 4 or hop > 140:
        ov = W - 4 * int(rng.integers(max(1, (W - 140 + 3) // 4), W // 4))
        ov = min(max(ov, -32), W - 4)
    # a band of at most 30 bins somewhere below Nyquist
    f0 = int(rng.integers(0, N // 2 - 2))
    F = int(rng.integers(1, min(30, N // 2 - f0) + 1))
    lo, hi = (f0 - 0.4) * FS / N, (f0 + F - 1 + 0.4) * FS / N
    lo = max(lo, 0.0)
    r = frequencyIndexRange(N, FS, lo, hi)
    F = r[1] - r[0]
    T = int(rng.integers(1, 13))
    chain = [(), ("l2normalize",), ("l2normalize", "mapminmax"), ("normalize",), ("normalizestd", "mapstd"), ("mapminmax",),
             ("mapstd", "mapminmax")][int(rng.integers(0, 7))]
    scaling = ["linear", "linear", "log", "db"][int(rng.integers(0, 4))]
    two_layers = rng.random() < 0.8
    H = int(rng.integers(1, 15 if chain[:1] == ("l2normalize",) else 16))
    n_out = int(rng.integers(1, 5)) if two_layers else H
    if not two_layers:
        n_out = min(H, 4)
        H = n_out
    tfs = ["TanSig", "LogSig", "PureLin", "SatLin"]
    hidden = (H,) if two_layers else ()
    if two_layers and rng.random() < 0.15:                      # three or four layers (any layerCount: NeuralNet.swift:310-313)
        hidden = (H,) + tuple(int(rng.integers(1, 9)) for _ in range(int(rng.integers(1, 3))))
    net = nets.random_net(rng, F * T, hidden, n_out,
                          transfer=tuple(tfs[int(rng.integers(0, 4))] for _ in range(len(hidden) + 1)) if len(hidden) > 1 else
                          (tfs[int(rng.integers(0, 4))], tfs[int(rng.integers(0, 4))]), in_fns=chain,
                          out_fns=[(), ("mapminmax",), ("mapstd",), ("mapminmax", "mapstd")][int(rng.integers(0, 4))])
    cfg = SyllableDetectorConfig(FS, N, W, ov, (lo, hi), T, scaling, [float(x) for x in rng.uniform(-0.5, 0.8, n_out)], net,
                                 window=int(rng.integers(0, 4)), spectrum=int(rng.integers(0, 2)), rule=int(rng.integers(0, 2)))
    return cfg


@pytest.mark.parametrize("seed", range(int(os.environ.get("SYLDET_FUZZ_DRAWS", "48"))))   # more draws: SYLDET_FUZZ_DRAWS=1000
def test_random_configuration(oracle_lib, seed):
    import torch
    rng = np.random.default_rng(9000 + seed)
    cfg = draw(rng)
    C = int(rng.integers(1, 4))
    hop = max(0, -cfg.windowOverlap) + cfg.windowLength - max(0, cfg.windowOverlap)
    frames = int(rng.integers(cfg.timeRange, 700))
    S = max(0, -cfg.windowOverlap) + cfg.windowLength + (frames - 1) * hop + int(rng.integers(0, hop))
    level = float(10.0 ** rng.uniform(-3, 1))
    x = synth.channels(C, S, first=seed * 7, fs=FS) * level
    stepped = rng.random() < 0.4
    if stepped:
        # level steps: block scales change between passes, and inside a pass the quiet part rides on the loud part's
        # scale.  Up to 50 dB per step here; the fused engine is block floating point per 128-frame pass and degrades
        # gradually once a frame sits ~70 dB under the loudest sample of its pass (numerics notes).
        env = np.ones(S)
        for _ in range(int(rng.integers(1, 4))):
            at = int(rng.integers(0, S))
            env[at:] *= float(10.0 ** rng.uniform(-2.5, 2.5))
        x = x * np.clip(env, 1e-3, 1e3)[None, :]
    x = x.astype(np.float32)
    o = util.oracle_for(cfg)
    runs = []
    with sd.SyllableDetector(cfg, channels=C) as det:
        out, fl = det.run(torch.from_numpy(x).cuda())
        torch.cuda.synchronize()
        runs.append((out.cpu().numpy(), fl.cpu().numpy(), det.geometry.engine, 1.0))
    if runs[0][2] != _abi.ENGINE_FUSED and cfg.spectrogramScaling != "linear":
        # AUTO keeps log / dB scalings on the generic engine (the fused one hands columns to the first layer as f16 hi + lo
        # pairs: 2^-22 relative on values up to ~100); there the fused engine is checked on request -- to the same bar and the
        # same evidence rule as AUTO's choice since round 6
        try:
            with sd.SyllableDetector(cfg, channels=C, engine=_abi.ENGINE_FUSED) as det:
                out, fl = det.run(torch.from_numpy(x).cuda())
                torch.cuda.synchronize()
                runs.append((out.cpu().numpy(), fl.cpu().numpy(), det.geometry.engine, 30.0))
        except sd.SyllableDetectorError:
            pass                                                # shape outside the fused engine's range
    for c in range(C):
        w32, _, w64 = o.run(x[c], po.F64, cfg.rule)
        ok = np.isfinite(w64).all(axis=1)
        o32 = o.run(x[c], po.F32, cfg.rule)[0]
        own = float((np.abs(o32[ok] - w64[ok]) / np.maximum(1.0, np.abs(w64[ok]))).max()) if ok.any() else 0.0
        names = [f.function for f in cfg.net.inputProcessing]
        few = cfg.net.layers[0].inputs <= 8 and any(f in ("normalize", "normalizestd") for f in names)
        if few and cfg.net.layers[0].inputs <= 3:
            continue            # (a - b) / |a - b| of two or three nearly equal values: a sign, not a number to compare
        # ONE rule for every draw (round 6: log / dB columns, |X|^2 columns, small normalisers and the fused engine on request
        # no longer start from 1e-4 or 30x the port's distance): the flat bar is 1e-5 (or 4x the fp32 port's own distance from the
        # anchor); beyond 1e-5 an evaluation needs the evidence of util.widened_evaluations -- fp32 itself beyond half the bar on
        # the evaluations that share a frame with it, or a conditioning floor beyond the bar: kappa 2^-23 behind l2normalize on
        # linear columns (util.band_condition), else how far the ANCHOR's output moves when every bin moves by 2^-23 of its frame's
        # norm, the error any fp32 transform leaves there (util.log_condition: the logarithm of a near-empty bin, a normaliser over a
        # handful of nearly equal values, squared columns) -- and `unexplained == 0` is asserted for every engine that ran.
        detector_mode = cfg.spectrogramScaling == "linear" and cfg.spectrum == _abi.SPECTRUM_POWER
        own_e = np.zeros(w64.shape[0])
        own_e[ok] = (np.abs(o32[ok] - w64[ok]) / np.maximum(1.0, np.abs(w64[ok]))).max(axis=1)
        T_ = cfg.timeRange
        # the fp32 port's own distance over the evaluations that share a frame with this one (their windows overlap: the
        # conditioning of a stretch of audio is not a property of one evaluation's rounding luck)
        own_nb = np.array([own_e[max(0, e - T_ + 1): e + T_].max() for e in range(len(own_e))])
        idx_ok = np.nonzero(ok)[0]
        flat = max(util.TOL, 4.0 * own)
        kappa_floor, cols64 = None, None
        if detector_mode and names[:1] == ["l2normalize"]:     # a band that holds little of its frames' energy: see util.band_condition
            kappa_floor = 2.0 ** -23 * util.band_condition(o, cfg, x[c])
        for out, fl, engine, widen in runs:
            assert out[c].shape == w64.shape
            assert (np.isfinite(out[c]).all(axis=1) == ok).all(), "NaN/inf evaluations must coincide"
            if ok.any():
                errv = (np.abs(out[c][ok] - w64[ok]) / np.maximum(1.0, np.abs(w64[ok]))).max(axis=1)
                tol = np.full(errv.shape, flat)
                floor_e = np.zeros(errv.shape)
                why = ""
                over = np.nonzero(errv > util.TOL)[0]
                if kappa_floor is not None:
                    floor_e = kappa_floor[ok].copy()
                    tol = np.maximum(tol, 4.0 * floor_e)
                    why = "kappa" if (errv > flat).any() else ""
                elif len(over):
                    # the conditioning floor of exactly the evaluations beyond 1e-5; the bar there: twice what the anchor itself moves
                    if cols64 is None:
                        cols64 = o.spectrogram(x[c], po.F64)
                    moves = util.log_condition(o, cfg, x[c], cols64, idx_ok[over])
                    floor_e[over] = moves
                    tol[over] = np.maximum(tol[over], 2.0 * moves)
                    if (errv > flat).any():
                        why = "log condition" if cfg.spectrogramScaling != "linear" else "perturbation floor"
                wide = util.widened_evaluations(errv, own_nb[ok], util.TOL, tol, floor_e)
                util.sweep_record("any configuration", seed, {1: "generic engine", 2: "fused engine", 3: "wide"}.get(engine, str(engine)) + (" (on request)" if widen != 1.0 else ""),
                                  errv.max(), own, flat, (errv / tol).max(), why, wide)
                # (every engine, AUTO's and the fused one on request: an error beyond 1e-5 must be one fp32 itself cannot avoid there)
                assert not wide or wide["unexplained"] == 0, "beyond the flat bar where fp32 holds it: %s" % wide
                util.assert_outputs_close(out[c][ok], w64[ok], tol)
                util.assert_flags_exact(fl[c][ok], w64[ok], cfg.thresholds, cfg.rule, tol)
            assert not fl[c][~ok].any()


def test_most_draws_of_the_detectors_mode_run_on_the_fused_engine():
    """AUTO really selects the fused engine: 48 draws of its own, put into the detector's mode (|X|, linear columns, a
    normaliser in front of a two-layer network -- without one AUTO keeps fp32 transforms outside the fold kernel's class);
    the fused engine takes all but the odd shape it cannot hold.  Only the geometry is asked for: nothing runs."""
    from syllable_detector_swift_amd.config import ProcessingFunction
    engines = []
    rng = np.random.default_rng(77)
    while len(engines) < 48:
        cfg = draw(rng)
        if len(cfg.net.layers) != 2:
            continue
        cfg.spectrogramScaling, cfg.spectrum = "linear", _abi.SPECTRUM_POWER
        names = [f.function for f in cfg.net.inputProcessing]
        if not names or names[0] not in ("l2normalize", "normalize", "normalizestd"):
            cfg.net.inputProcessing = [ProcessingFunction("l2normalize")] + list(cfg.net.inputProcessing)
        with sd.SyllableDetector(cfg, channels=2) as det:
            engines.append(det.geometry.engine)
    assert sum(e == _abi.ENGINE_FUSED for e in engines) >= len(engines) * 0.8, engines


@pytest.mark.parametrize("seed", range(12))
def test_random_configuration_streaming(oracle_lib, seed):
    """The reference's per-detector calls (appendAudioData / processNewValue / lastOutputs) with ragged appends on
    random configurations -- gaps, zero padding, short windows -- against the oracle's batch run."""
    rng = np.random.default_rng(500 + seed)
    cfg = draw(rng)
    if cfg.spectrogramScaling != "linear":
        cfg.spectrogramScaling = "linear"
    hop = max(0, -cfg.windowOverlap) + cfg.windowLength - max(0, cfg.windowOverlap)
    frames = int(rng.integers(cfg.timeRange + 1, 260))
    S = max(0, -cfg.windowOverlap) + cfg.windowLength + (frames - 1) * hop + int(rng.integers(0, hop))
    x = synth.channel(S, 900 + seed, fs=FS).astype(np.float32)
    o = util.oracle_for(cfg)
    w32, _, w64 = o.run(x, po.F64, cfg.rule)
    with sd.SyllableDetector(cfg, channels=1) as det:
        got, pos = [], 0
        while pos < S:
            n = int(rng.integers(1, 4 * hop + 50))
            det.appendAudioData(x[pos:pos + n])
            pos += n
            while det.processNewValue():
                got.append(det.lastOutputs)
        got = np.array(got, np.float64).reshape(-1, w64.shape[1])
    assert got.shape == w64.shape
    ok = np.isfinite(w64).all(axis=1)
    o32 = o.run(x, po.F32, cfg.rule)[0]
    own = float((np.abs(o32[ok] - w64[ok]) / np.maximum(1.0, np.abs(w64[ok]))).max()) if ok.any() else 0.0
    if ok.any():
        util.check_with_evidence(o, cfg, x, got, None, w64=w64, w32=o32, check_flags=False)


def draw_example_class(rng):
    """The reference's example detector class, which runs on kernels_fused_r.hip: N = 256, windows of 192 or 256 samples (or N = 128, windows of 96 or 128),
    any timeRange up to 12, hop a multiple of 4 up to 140, l2normalize
    first, <= 4 TanSig hidden units, one to four linear outputs, at most one output map; any band, window type, affine maps behind
    the normaliser, threshold, rule."""
    hop = int(rng.choice([16, 32, 48, 64, 64, 68, 80, 84, 96, 100, 112, 116, 120, 124, 128, 128, 132, 132, 132, 136, 140]))
    N = 256
    W = 256 if rng.random() < 0.75 else 192
    if rng.random() < 0.25:              # 128-point frames, windows of up to 128 samples: the kernel's four-k-step instantiations
        N = 128
        W = 128 if rng.random() < 0.7 else 96
        hop = int(rng.choice([16, 32, 32, 48, 64, 64, 36, 44, 60, 68, 80, 96, 100, 112, 128]))
    T = 10 if rng.random() < 0.4 else int(rng.integers(1, 13))
    f0 = int(rng.integers(0, N // 2 - 28))
    F = int(rng.integers(1, 30))
    lo, hi = max((f0 - 0.4) * FS / N, 0.0), (f0 + F - 1 + 0.4) * FS / N
    r = frequencyIndexRange(N, FS, lo, hi)
    F = r[1] - r[0]
    chain = [("l2normalize",), ("l2normalize", "mapminmax"), ("l2normalize", "mapstd"), ("l2normalize", "mapstd", "mapminmax")][int(rng.integers(0, 4))]
    n_out = 1 if rng.random() < 0.7 else int(rng.integers(2, 5))       # several syllables: an output and a threshold each
    net = nets.random_net(rng, F * T, (int(rng.integers(1, 5)),), n_out, transfer=("TanSig", "PureLin"), in_fns=chain,
                          out_fns=[(), ("mapminmax",), ("mapstd",)][int(rng.integers(0, 3))])
    return SyllableDetectorConfig(FS, N, W, W - hop, (lo, hi), T, "linear", [float(t) for t in rng.uniform(-0.5, 0.8, n_out)], net,
                                  window=int(rng.integers(0, 4)), spectrum=_abi.SPECTRUM_POWER, rule=int(rng.integers(0, 2)))


@pytest.mark.parametrize("kernel", ["fused_s_kernel", "fused_r_kernel"])
@pytest.mark.parametrize("seed", range(int(os.environ.get("SYLDET_FUZZ_DRAWS", "24"))))
def test_random_example_class_detector_on_the_register_resident_kernel(oracle_lib, monkeypatch, kernel, seed):
    """Lengths around the kernel's own boundaries (64-frame passes, 2039-evaluation segments, three passes in flight: one,
    two, three passes and their neighbours), several channels, cumulative level steps of up to 50 dB each inside and across
    passes (120 dB between the quietest and the loudest stretch of one recording)."""
    import torch
    util.select_fused(monkeypatch, kernel)
    rng = np.random.default_rng(77000 + seed)
    cfg = draw_example_class(rng)
    hop_ = cfg.windowLength - cfg.windowOverlap
    padded = cfg.windowLength == 256 and cfg.net.layers[0].outputs <= 4          # (the fold kernel's padded ring)
    if kernel == "fused_s_kernel" and (cfg.windowLength % 64 != 0 or (hop_ % 64 == 0 and cfg.windowLength > 128 and not padded)):
        kernel = "fused_r_kernel"      # (windows of 96 samples; hops of 64 / 128 under 192-sample windows or wider layers: the other kernel's padded staging)
    hop = cfg.windowLength - cfg.windowOverlap
    edges = [10, 11, 63, 64, 65, 73, 74, 127, 128, 129, 137, 192, 201, 2047, 2048, 2049, 2057, 2058, 4100]
    frames = max(cfg.timeRange, int(edges[seed % len(edges)] if seed < len(edges) else rng.integers(10, 6000)))
    S = cfg.windowLength + (frames - 1) * hop + int(rng.integers(0, hop))
    C = int(rng.integers(1, 4))
    x = synth.channels(C, S, first=seed * 5, fs=FS) * float(10.0 ** rng.uniform(-3, 1))
    if rng.random() < 0.6:
        # up to four cumulative level steps of up to +-50 dB each, 120 dB between the quietest and the loudest stretch of a
        # recording: every frame carries its own column exponent, and what the pass's sample grid still cannot hold is
        # recomputed from the samples (precision guard, include/syldet.h)
        env = np.ones(S)
        for _ in range(int(rng.integers(1, 5))):
            env[int(rng.integers(0, S)):] *= float(10.0 ** rng.uniform(-2.5, 2.5))
        x = x * np.clip(env, 1e-3, 1e3)[None, :]
    x = x.astype(np.float32)
    o = util.oracle_for(cfg)
    with sd.SyllableDetector(cfg, channels=C) as det:
        det.profile(True)
        out, fl = det.run(torch.from_numpy(x).cuda())
        torch.cuda.synchronize()
        assert util.launched(det) == [kernel]
        out, fl = out.cpu().numpy(), fl.cpu().numpy()
    for c in range(C):
        _, _, w64 = o.run(x[c], po.F64, cfg.rule)
        w32 = o.run(x[c], po.F32, cfg.rule)[0]                   # the fp32 port: the reference's operation order in fp32
        ok = np.isfinite(w64).all(axis=1)
        own = float((np.abs(w32[ok] - w64[ok]) / np.maximum(1.0, np.abs(w64[ok]))).max()) if ok.any() else 0.0
        assert out[c].shape == w64.shape
        assert (np.isfinite(out[c]).all(axis=1) == ok).all(), "NaN/inf evaluations must coincide"
        # 1e-5 (or 4x the fp32 port's own distance from the anchor); where the band holds only a small part of its frames'
        # energy no fp32 transform knows it to 1e-5 of its own norm (util.band_condition), and the bar follows
        flat = max(util.TOL, 4.0 * own)
        kappa = util.band_condition(o, cfg, x[c])
        tol = np.maximum(flat, 2.0 ** -21 * kappa)[ok]
        if ok.any():
            err = (np.abs(out[c][ok] - w64[ok]) / np.maximum(1.0, np.abs(w64[ok]))).max(axis=1)
            own_e = (np.abs(w32[ok] - w64[ok]) / np.maximum(1.0, np.abs(w64[ok]))).max(axis=1)
            own_e = np.array([own_e[max(0, e - cfg.timeRange + 1): e + cfg.timeRange].max() for e in range(len(own_e))])   # (evaluations that share a frame)
            wide = util.widened_evaluations(err, own_e, util.TOL, tol, 2.0 ** -23 * kappa[ok])     # (against 1e-5 itself, not 4x own)
            util.sweep_record("example class", seed, kernel, err.max(), own, flat, (err / tol).max(), "kappa" if (err > flat).any() else "", wide)
            assert not wide or wide["unexplained"] == 0, "beyond the flat bar where fp32 holds it: %s" % wide
            util.assert_outputs_close(out[c][ok], w64[ok], tol)
            util.assert_flags_exact(fl[c][ok], w64[ok], cfg.thresholds, cfg.rule, tol)
        assert not fl[c][~ok].any()


@pytest.mark.parametrize("seed", range(int(os.environ.get("SYLDET_FUZZ_DRAWS_BLOCKS", "16"))))
def test_random_frames_of_four_hops(oracle_lib, seed):
    """kernels_bdft.hip (every block of hop samples transformed once, frames as sliding sums of four or two blocks, the window as
    taps along the bins) on random bands, windows, timeRanges, networks, lengths around its sub-tiles of 16 blocks and tiles of 96
    frames, level steps of up to 100 dB (every block has its own scale)."""
    import torch
    from syllable_detector_swift_amd.config import SyllableDetectorConfig
    rng = np.random.default_rng(31000 + seed)
    N = int(rng.choice([512, 1024]))
    hop = N // 4
    if N == 512 and seed % 3 == 2:
        hop = N // 2                                              # (50 % overlap: a frame is two blocks)
    window = int(rng.choice([_abi.WINDOW_NONE, _abi.WINDOW_HAMMING, _abi.WINDOW_HAMMING, _abi.WINDOW_HANNING]))
    f0 = int(rng.integers(1, N // 2 - 130))
    F = int(rng.integers(33, 122))
    lo, hi = (f0 - 0.4) * FS / N, (f0 + F - 1 + 0.4) * FS / N
    r = frequencyIndexRange(N, FS, lo, hi)
    F = r[1] - r[0]
    T = int(rng.integers(1, 13))
    net = nets.random_net(rng, F * T, (int(rng.integers(1, 5)),), 1, transfer=("TanSig", "PureLin"),
                          in_fns=[("l2normalize",), ("l2normalize", "mapminmax"), ("l2normalize", "mapstd")][int(rng.integers(0, 3))],
                          out_fns=[(), ("mapminmax",)][int(rng.integers(0, 2))])
    cfg = SyllableDetectorConfig(FS, N, N, N - hop, (lo, hi), T, "linear", [float(rng.uniform(-0.5, 0.8))], net, window=window)
    edges = [T, 15, 16, 17, 95, 96, 97, 105, 111, 112, 113, 192, 200, 700, 1300, 2100]
    frames = max(T, int(edges[seed % len(edges)]))
    S = N + (frames - 1) * hop + int(rng.integers(0, hop))
    C = int(rng.integers(1, 4))
    x = synth.channels(C, S, first=seed * 3, fs=FS) * float(10.0 ** rng.uniform(-3, 1))
    env = np.ones(S)
    for _ in range(int(rng.integers(0, 3))):
        env[int(rng.integers(0, S)):] *= float(10.0 ** rng.uniform(-2.5, 2.5))
    x = (x * np.clip(env, 1e-5, 1e3)[None, :]).astype(np.float32)
    o = util.oracle_for(cfg)
    with sd.SyllableDetector(cfg, channels=C) as det:
        det.profile(True)
        out, fl = det.run(torch.from_numpy(x).cuda())
        torch.cuda.synchronize()
        assert util.launched(det) == ["bdft_net_kernel"]
        out, fl = out.cpu().numpy(), fl.cpu().numpy()
    for c in range(C):
        _, _, w64 = o.run(x[c], po.F64, cfg.rule)
        w32 = o.run(x[c], po.F32, cfg.rule)[0]                   # the fp32 port: the reference's operation order in fp32
        ok = np.isfinite(w64).all(axis=1)
        own = float((np.abs(w32[ok] - w64[ok]) / np.maximum(1.0, np.abs(w64[ok]))).max()) if ok.any() else 0.0
        assert (np.isfinite(out[c]).all(axis=1) == ok).all(), "NaN/inf evaluations must coincide"
        flat = max(util.TOL, 4.0 * own)
        kappa = util.band_condition(o, cfg, x[c])
        tol = np.maximum(flat, 2.0 ** -21 * kappa)[ok]
        if ok.any():
            err = (np.abs(out[c][ok] - w64[ok]) / np.maximum(1.0, np.abs(w64[ok]))).max(axis=1)
            own_e = (np.abs(w32[ok] - w64[ok]) / np.maximum(1.0, np.abs(w64[ok]))).max(axis=1)
            own_e = np.array([own_e[max(0, e - cfg.timeRange + 1): e + cfg.timeRange].max() for e in range(len(own_e))])
            wide = util.widened_evaluations(err, own_e, util.TOL, tol, 2.0 ** -23 * kappa[ok])
            util.sweep_record("frames of four hops", seed, "bdft_net_kernel", err.max(), own, flat, (err / tol).max(), "kappa" if (err > flat).any() else "", wide)
            assert not wide or wide["unexplained"] == 0, "beyond the flat bar where fp32 holds it: %s" % wide
            util.assert_outputs_close(out[c][ok], w64[ok], tol)
            util.assert_flags_exact(fl[c][ok], w64[ok], cfg.thresholds, cfg.rule, tol)


@pytest.mark.parametrize("seed", range(int(os.environ.get("SYLDET_FUZZ_DRAWS_WIDE", "16"))))
def test_random_wide_band_detector(oracle_lib, seed):
    """Bands of 33 .. 64 bins under 256-point frames (kernels_fused_s.hip, two row tiles per parity, 4 waves a workgroup): random
    bands, hops, timeRanges, window types, networks of the example class (and a few with other transfer functions / several
    outputs), lengths around the wave segments, level steps of up to 50 dB -- against the fp64 anchor at the flat bar, with the
    same evidence rule as the other sweeps."""
    import torch
    rng = np.random.default_rng(52000 + seed)
    hop = int(rng.choice([32, 48, 68, 84, 100, 116, 124, 132, 132, 136, 140, 160, 180]))
    N = W = 256
    F = int(rng.integers(33, 65))
    f0 = int(rng.integers(0, N // 2 - F + 1))
    lo, hi = max((f0 - 0.4) * FS / N, 0.0), (f0 + F - 1 + 0.4) * FS / N
    r = frequencyIndexRange(N, FS, lo, hi)
    F = r[1] - r[0]
    T = int(rng.integers(1, 13))
    n_out = 1 if rng.random() < 0.7 else int(rng.integers(2, 5))
    exact = rng.random() < 0.6
    net = nets.random_net(rng, F * T, (int(rng.integers(1, 5)),), n_out,
                          transfer=("TanSig", "PureLin") if exact else (["TanSig", "LogSig", "SatLin"][int(rng.integers(0, 3))], ["PureLin", "TanSig"][int(rng.integers(0, 2))]),
                          in_fns=[("l2normalize",), ("l2normalize", "mapminmax"), ("l2normalize", "mapstd"), ("normalize",), ("normalizestd", "mapminmax")][int(rng.integers(0, 3 if exact else 5))],
                          out_fns=[(), ("mapminmax",), ("mapstd",)][int(rng.integers(0, 3))])
    cfg = SyllableDetectorConfig(FS, N, W, W - hop, (lo, hi), T, "linear", [float(t) for t in rng.uniform(-0.5, 0.8, n_out)], net,
                                 window=int(rng.integers(0, 4)), spectrum=_abi.SPECTRUM_POWER, rule=int(rng.integers(0, 2)))
    if not (32 < F <= 64):
        pytest.skip("the band's bins fell outside 33 .. 64")
    edges = [T, 15, 16, 17, 31, 32, 33, 63, 64, 65, 250, 1000, 3000]
    frames = max(T, int(edges[seed % len(edges)]))
    S = W + (frames - 1) * hop + int(rng.integers(0, hop))
    C = int(rng.integers(1, 4))
    x = synth.channels(C, S, first=seed * 11, fs=FS) * float(10.0 ** rng.uniform(-3, 1))
    if rng.random() < 0.6:
        env = np.ones(S)
        for _ in range(int(rng.integers(1, 4))):
            env[int(rng.integers(0, S)):] *= float(10.0 ** rng.uniform(-2.5, 2.5))
        x = x * np.clip(env, 1e-3, 1e3)[None, :]
    x = x.astype(np.float32)
    o = util.oracle_for(cfg)
    with sd.SyllableDetector(cfg, channels=C) as det:
        det.profile(True)
        out, fl = det.run(torch.from_numpy(x).cuda())
        torch.cuda.synchronize()
        names = util.launched(det)
        # (hops that are multiples of 64 have no padded form with two row tiles: the generic engine keeps them)
        assert names == ["fused_s_kernel"] or hop % 64 == 0, names
        out, fl = out.cpu().numpy(), fl.cpu().numpy()
    for c in range(C):
        _, _, w64 = o.run(x[c], po.F64, cfg.rule)
        w32 = o.run(x[c], po.F32, cfg.rule)[0]
        ok = np.isfinite(w64).all(axis=1)
        own = float((np.abs(w32[ok] - w64[ok]) / np.maximum(1.0, np.abs(w64[ok]))).max()) if ok.any() else 0.0
        assert (np.isfinite(out[c]).all(axis=1) == ok).all(), "NaN/inf evaluations must coincide"
        flat = max(util.TOL, 4.0 * own)
        names_in = [f.function for f in cfg.net.inputProcessing]
        kappa = util.band_condition(o, cfg, x[c]) if names_in[:1] == ["l2normalize"] else np.ones(w64.shape[0])
        tol = np.maximum(flat, 2.0 ** -21 * kappa)[ok]
        if ok.any():
            err = (np.abs(out[c][ok] - w64[ok]) / np.maximum(1.0, np.abs(w64[ok]))).max(axis=1)
            own_e = (np.abs(w32[ok] - w64[ok]) / np.maximum(1.0, np.abs(w64[ok]))).max(axis=1)
            own_e = np.array([own_e[max(0, e - cfg.timeRange + 1): e + cfg.timeRange].max() for e in range(len(own_e))])
            wide = util.widened_evaluations(err, own_e, util.TOL, tol, 2.0 ** -23 * kappa[ok] if names_in[:1] == ["l2normalize"] else None)
            util.sweep_record("bands of 33 to 64 bins", seed, names[0], err.max(), own, flat, (err / tol).max(), "kappa" if (err > flat).any() else "", wide)
            assert not wide or wide["unexplained"] == 0, "beyond the flat bar where fp32 holds it: %s" % wide
            util.assert_outputs_close(out[c][ok], w64[ok], tol)
            util.assert_flags_exact(fl[c][ok], w64[ok], cfg.thresholds, cfg.rule, tol)
        assert not fl[c][~ok].any()
