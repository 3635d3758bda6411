"""Pins the CPU oracle (oracle/syldet_oracle.c): analytic known answers, an independent numpy
implementation, the committed golden vectors, and streaming == batch.  No GPU, no product code
beyond plain-data containers."""
import numpy as np
import pytest

import pyoracle as po
import util
from syllable_detector_swift_amd import nets, synth


@pytest.fixture(scope="module")
def sample(oracle_lib):
    cfg = util.sample_net()
    return cfg, util.oracle_for(cfg)


def test_geometry_of_sample_net(sample):
    cfg, o = sample
    # hop 132, bins [12,41), F=29, I=290
    assert (o.g.gap, o.g.overlap, o.g.hop, o.g.f0, o.g.f1, o.g.F, o.g.I, o.g.n_out) == (0, 124, 132, 12, 41, 29, 290, 1)
    assert o.count_frames(2646000) == 20044 and o.count_evals(2646000) == 20035
    assert o.count_frames(255) == 0 and o.count_frames(256) == 1 and o.count_frames(256 + 131) == 1
    assert o.count_frames(256 + 132) == 2 and o.count_evals(256 + 9 * 132 - 1) == 0 and o.count_evals(256 + 9 * 132) == 1


def test_frequency_index_range(oracle_lib):
    import ctypes as C
    f0, f1 = C.c_int(), C.c_int()
    fr = lambda N, fs, lo, hi: (oracle_lib.orc_frequency_index_range(N, fs, lo, hi, C.byref(f0), C.byref(f1)), f0.value, f1.value)
    assert fr(256, 44100.0, 2000.0, 7000.0) == (0, 12, 41)
    assert fr(1024, 44100.0, 2000.0, 7000.0) == (0, 47, 163)
    assert fr(256, 44100.0, 0.0, 1e9)[0:3] == (0, 0, 128)          # clamped to N/2
    assert fr(256, 44100.0, -1.0, 100.0)[0] != 0                    # nil
    assert fr(256, 44100.0, 5000.0, 5000.0)[0] != 0                 # hi <= lo
    assert fr(256, 44100.0, 23000.0, 24000.0)[0] != 0               # f0 >= N/2


@pytest.mark.parametrize("name,formula", [
    ("hamming", lambda a: 0.54 - 0.46 * np.cos(a)),
    ("hanning", lambda a: 0.5 * (1 - np.cos(a))),
    ("blackman", lambda a: 0.42 - 0.5 * np.cos(a) + 0.08 * np.cos(2 * a)),
    ("none", lambda a: np.ones_like(a)),
])
def test_window_tables(sample, name, formula):
    cfg, _ = sample
    c = nets.variant(cfg, window=po.WIN[name])
    w = util.oracle_for(c).window()
    a = 2 * np.pi * np.arange(256) / 256           # periodic, denominator N (vDSP flag 0)
    assert np.array_equal(w, formula(a).astype(np.float32))


@pytest.mark.parametrize("precision,tol", [(po.F64, 1e-9), (po.F32, 2e-5)])
def test_known_answers_pure_tone(sample, precision, tol):
    """A cosine exactly on bin k with a Hamming window: |X[k]| = A*0.54*W/2, neighbours
    A*0.23*W/2, everything else 0; DC input: |X[0]| = A*0.54*W, |X[1]| = A*0.23*W."""
    cfg, o = sample
    n = np.arange(256)
    A, k = 0.7, 20
    spec = o.stft_frame((A * np.cos(2 * np.pi * k * n / 256 + 0.3)).astype(np.float32), precision)
    # the input was rounded to float32, so compare at float32 input accuracy
    assert abs(spec[k] - A * 0.54 * 128) < 1e-4 and abs(spec[k - 1] - A * 0.23 * 128) < 1e-4 and abs(spec[k + 1] - A * 0.23 * 128) < 1e-4
    rest = np.delete(spec, [k - 1, k, k + 1])
    assert rest.max() < 1e-4
    dc = o.stft_frame(np.full(256, A, np.float32), precision)
    assert abs(dc[0] - A * 0.54 * 256) < 1e-4 and abs(dc[1] - A * 0.23 * 256) < 1e-4
    # impulse at n0: |X[k]| = w[n0] for every k
    x = np.zeros(256, np.float32)
    x[37] = 1.0
    imp = o.stft_frame(x, precision)
    assert np.abs(imp - o.window()[37]).max() < (1e-12 if precision == po.F64 else tol)


def test_power_mode_is_square(sample):
    cfg, o = sample
    x = synth.channel(256, 1)
    p = util.oracle_for(nets.variant(cfg, spectrum=1)).stft_frame(x, po.F64)
    np.testing.assert_allclose(p, o.stft_frame(x, po.F64) ** 2, rtol=1e-12)


def test_spectrogram_against_numpy_rfft(sample):
    cfg, o = sample
    x = synth.channel(40000, 2)
    w = o.window().astype(np.float64)
    J = o.count_frames(x.size)
    frames = np.stack([x[j * 132: j * 132 + 256].astype(np.float64) * w for j in range(J)])
    want = np.abs(np.fft.rfft(frames, axis=1))[:, 12:41]
    np.testing.assert_allclose(o.spectrogram(x, po.F64), want, atol=1e-11)
    util.assert_columns_close(o.spectrogram(x, po.F32), want, tol=3e-6)      # the fp32 port vs the anchor


def test_zero_padded_and_gapped_frames_against_numpy(oracle_lib):
    cfg, x, _ = util.load_case("case_chain_normstd_log")     # N=128, W=96, overlap -16 (gap 16)
    o = util.oracle_for(cfg)
    assert (o.g.gap, o.g.hop) == (16, 112)
    w = o.window().astype(np.float64)
    J = o.count_frames(x.size)
    frames = np.stack([x[j * 112 + 16: j * 112 + 16 + 96].astype(np.float64) * w for j in range(J)])
    want = np.abs(np.fft.rfft(frames, n=128, axis=1))[:, o.g.f0:o.g.f1]
    np.testing.assert_allclose(o.spectrogram(x, po.F64), want, atol=1e-11)


def _numpy_net(net, v):
    v = np.asarray(v, np.float64)
    for f in net["inputs"]:
        k = f["function"]
        if k == "l2normalize":
            v = v / np.sqrt((v * v).sum())
        elif k == "normalize":
            mn, mx = v.min(), v.max()
            v = np.full_like(v, -1.0) if mx == mn else 2 * (v - mn) / (mx - mn) - 1
        elif k == "normalizestd":
            v = (v - v.mean()) / v.std()
        else:
            v = (v - f["xOffsets"].astype(np.float64)) * f["gains"].astype(np.float64) + float(f["y"])
    for L in net["layers"]:
        v = L["weights"].astype(np.float64).reshape(L["outputs"], L["inputs"]) @ v + L["biases"].astype(np.float64)
        tf = L["transferFunction"]
        v = np.tanh(v) if tf == "TanSig" else 1 / (1 + np.exp(-v)) if tf == "LogSig" else np.clip(v, 0, 1) if tf == "SatLin" else v
    for f in net["outputs"]:
        v = (v - float(f["y"])) / f["gains"].astype(np.float64) + f["xOffsets"].astype(np.float64)
    return v


@pytest.mark.parametrize("name", util.case_names())
def test_network_against_numpy(oracle_lib, name):
    cfg, x, _ = util.load_case(name)
    net = po.from_config(cfg)
    o = po.Oracle(net)
    cols = o.spectrogram(x, po.F64)
    _, _, o64 = o.run(x, po.F64, cfg.rule)
    T, F = cfg.timeRange, o.g.F
    for e in list(range(0, min(len(o64), 40))) + [len(o64) - 1]:
        v = cols[e:e + T].reshape(-1)
        v = np.log(v) if cfg.spectrogramScaling == "log" else 20 * np.log10(v) if cfg.spectrogramScaling == "db" else v
        np.testing.assert_allclose(o64[e], _numpy_net(net, v), rtol=1e-9, atol=1e-12)


def test_constant_input_closed_form(sample):
    """NeuralNet.test(_:)-style probe (NeuralNet.swift:284-292): a constant vector through
    l2normalize is 1/sqrt(290) everywhere, whatever the constant."""
    cfg, o = sample
    a = o.net_apply(np.full(290, 3.0, np.float32), po.F64)
    b = o.net_apply(np.full(290, 0.25, np.float32), po.F64)
    np.testing.assert_allclose(a, b, rtol=1e-12)
    np.testing.assert_allclose(a, _numpy_net(po.from_config(cfg), np.full(290, 1.0)), rtol=1e-12)


@pytest.mark.parametrize("name", util.case_names())
def test_golden_vectors_pin_the_oracle(oracle_lib, name):
    cfg, x, gold = util.load_case(name)
    o = util.oracle_for(cfg)
    out, fl, o64 = o.run(x, po.F64, cfg.rule)
    assert len(fl) == int(gold["n_evals"][0])
    np.testing.assert_allclose(o.spectrogram(x, po.F64)[:4], gold["columns_head"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(o64[:4096], gold["outputs64"], rtol=1e-11, atol=1e-13)
    assert np.array_equal(fl, gold["flags"])
    assert np.array_equal(o.detections(fl, 0.0), gold["det_0"])
    assert np.array_equal(o.detections(fl, 0.05), gold["det_50ms"])
    # the fp32 port stays within the float tolerance of the anchor and raises the same flags
    out32, fl32, _ = o.run(x, po.F32, cfg.rule)
    util.assert_outputs_close(out32, o64)
    util.assert_flags_exact(fl32, o64, cfg.thresholds, cfg.rule)
    assert float(gold["margin"][0]) > 2e-5


def test_detection_indices_and_debounce(sample):
    cfg, o = sample
    fl = np.zeros(100, np.uint8)
    fl[[0, 1, 2, 50, 51, 99]] = 1
    first = 256 + 9 * 132                      # TrackDetector.swift:39-42
    assert list(o.detections(fl, 0.0)) == [first + e * 132 for e in (0, 1, 2, 50, 51, 99)]
    # debounce 0.005 s = Int(220.5) = 220 samples: e=1 is 132 later (suppressed), e=2 is 264 later (emitted)
    assert list(o.detections(fl, 0.005)) == [first + e * 132 for e in (0, 2, 50, 99)]
    assert list(o.detections(fl, 1.0)) == [first]
    gap = util.oracle_for(util.load_case("case_chain_normstd_log")[0])
    f = np.ones(3, np.uint8)
    assert list(gap.detections(f, 0.0)) == [96 + 112 * 5 + 16 + e * 112 for e in range(3)]


@pytest.mark.parametrize("name", ["case_sample_syllables", "case_chain_normstd_log", "case_chain_mapstd_only"])
def test_streaming_restatement_equals_batch(oracle_lib, name):
    """Driving the two byte rings one frame at a time, in ragged appends, gives exactly the
    batch formulation (frame j at j*hop+gap; evaluation e over columns e..e+T-1)."""
    cfg, x, _ = util.load_case(name)
    x = x[:30000]
    o = util.oracle_for(cfg)
    want, _, _ = o.run(x, po.F32, po.RULE_FIRST)
    s = o.stream(po.F32)
    rng = np.random.default_rng(1)
    got, pos = [], 0
    assert np.array_equal(s.last_outputs(), np.zeros(o.n_out, np.float32))
    while pos < x.size:
        n = int(rng.integers(1, 700))
        assert s.append(x[pos:pos + n]) == 0
        pos += n
        while s.process_new_value():
            got.append(s.last_outputs())
    got = np.array(got).reshape(-1, o.n_out)
    assert got.shape == want.shape and np.array_equal(got, want)


def test_streaming_ring_overflow(sample):
    cfg, o = sample
    s = o.stream(po.F32)
    assert s.append(np.zeros(102400, np.float32)) == 0        # 409600 bytes fill the ring exactly
    assert s.append(np.zeros(1, np.float32)) < 0              # "Insufficient space on buffer."
    assert s.process_new_value()
    assert s.append(np.zeros(1000, np.float32)) == 0


def test_resampler_restatement(oracle_lib):
    r = po.Resampler(48000.0, 44100.0)
    x = np.arange(480, dtype=np.float32)
    y = r.resample(x)
    step = np.float32(48000.0 / 44100.0)
    assert y.size == int(np.float32(480) / step)
    np.testing.assert_allclose(y, np.arange(y.size, dtype=np.float32) * step, rtol=1e-6)   # linear ramp is reproduced
    y2 = r.resample(x + 480)
    assert y2.size in (440, 441)


@pytest.mark.parametrize("N,W,overlap,band", [(256, 256, 124, (2000.0, 7000.0)), (256, 256, 128, (2000.0, 7000.0)), (1024, 1024, 768, (2000.0, 7000.0)),
                                              (512, 384, 100, (500.0, 9000.0))])
def test_spectrogram_against_scipy_signal(oracle_lib, N, W, overlap, band):
    """A third, independent statement of the framing (hop = W - overlap, frames while W samples are left, zero padding to N), the
    periodic window and |X| (round 6): scipy.signal.stft -- its own segmentation and FFT code -- unscaled by the window's sum,
    against the oracle's fp64 columns over the band.  (The reference's |X| is zvabs of vDSP's packed 2 x DFT, halved:
    CircularShortTimeFourierTransform.swift:320-333 -- the DFT's magnitude.)"""
    scipy_signal = pytest.importorskip("scipy.signal")
    from syllable_detector_swift_amd.config import frequencyIndexRange
    base = util.sample_net()
    f0, f1 = frequencyIndexRange(N, base.samplingRate, band[0], band[1])
    # (only the geometry and the transform are used: a one-column network of the band's width keeps the configuration valid)
    cfg = nets.variant(base, fourierLength=N, windowLength=W, windowOverlap=overlap, freqRange=band, timeRange=1,
                       net=nets.random_net(np.random.default_rng(1), f1 - f0, (2,), 1))
    o = util.oracle_for(cfg)
    frames = 58
    x = synth.channel(W + (frames - 1) * (W - overlap) + 33, 6, fs=cfg.samplingRate)
    cols = o.spectrogram(x, po.F64)
    w = o.window().astype(np.float64)
    _, _, Z = scipy_signal.stft(x.astype(np.float64), fs=cfg.samplingRate, window=w, nperseg=W, noverlap=overlap, nfft=N, boundary=None, padded=False,
                                return_onesided=True)
    mag = np.abs(Z) * w.sum()                              # (stft divides by the window's sum)
    assert mag.shape[1] == cols.shape[0] == frames and (o.g.f0, o.g.f1) == (f0, f1)
    want = mag[f0:f1, :].T
    assert np.abs(cols - want).max() <= 1e-10 * max(1.0, want.max())


def test_the_evidence_rule_itself():
    """tests/util.py::widened_evaluations, the one rule of the suite since round 6, on made-up numbers: an error beyond the flat bar
    is explained only where the fp32 port is beyond half the bar on the evaluations that share a frame, or the conditioning floor is;
    anything else is `unexplained` whatever bar was applied."""
    errv = np.array([2e-6, 1.2e-5, 3e-5, 1.1e-5, 4e-5])
    own = np.array([1e-7, 6e-6, 1e-6, 1e-6, 1e-7])          # (already the neighbourhood's maximum)
    floor = np.array([0.0, 0.0, 2e-5, 4e-6, 4.9e-6])
    tol = np.full(5, 1e-4)
    w = util.widened_evaluations(errv, own, util.TOL, tol, floor)
    assert w["evaluations_over_flat_bar"] == 4 and w["unexplained"] == 2      # evaluations 3 and 4: neither the port nor the floor speaks
    assert w["worst_unexplained"]["evaluation"] == 4 and w["worst"]["evaluation"] == 4
    assert util.widened_evaluations(np.array([9e-6, 1e-6]), own[:2], util.TOL, tol[:2], None) is None
    ok = util.widened_evaluations(np.array([2e-5]), np.array([5.1e-6]), util.TOL, np.array([1e-4]), None)
    assert ok["unexplained"] == 0
