"""Which kernel family a batch call takes is decided per batch, not only at create time: a row too long for the fold kernel's
32-bit byte offsets must leave the fused route when the kernel that would take it instead does not hold the contract
(log / dB columns on the pass-scaled kernels) or does not exist for the plan (hops only the fold kernel's ring holds)."""
import numpy as np
import pytest

import spotcheck
import util
from syllable_detector_swift_amd import SyllableDetector, _abi, nets, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant", ["db", "hop176"])
def test_rows_past_2_gib_leave_the_fused_route_where_only_the_fold_kernel_holds_the_contract(oracle_lib, variant):
    import torch
    base = util.sample_net()
    cfg = nets.variant(base, spectrogramScaling="db") if variant == "db" else nets.variant(base, windowOverlap=256 - 176)
    dev = torch.device("cuda", 0)
    S_short, S_long = 1 << 20, (1 << 29) + 4096                  # the long row is past 2 GiB
    with SyllableDetector(cfg, channels=1, device=0) as det:
        assert det.geometry.engine == _abi.ENGINE_FUSED          # AUTO commits these to the fold kernel at create time
        det.profile(True)
        x = synth.channels_on_device(1, S_short, dev, fs=cfg.samplingRate)
        det.run(x)
        torch.cuda.synchronize()
        assert util.launched(det)[0] == "fused_s_kernel"
        del x
        x = synth.channels_on_device(1, S_long, dev, fs=cfg.samplingRate)
        out, fl = det.run(x)
        torch.cuda.synchronize()
        names = util.launched(det)
        assert not any(nm.startswith("fused") and "spectrogram" not in nm for nm in names), names   # (the DFT half as an STFT is fine for linear columns)
        if variant == "db":
            assert not any(nm.startswith("fused") for nm in names), names
        v = spotcheck.check(det, cfg, x, out, fl, [0], width=96)
        assert v["max_error"] <= 1e-5


@pytest.mark.parametrize("lo,hi,hop,T,kind", [(1000.0, 11000.0, 132, 10, "exact"), (2000.0, 7600.0, 132, 10, "exact"), (0.0, 10900.0, 100, 5, "gen"),
                                               (3000.0, 10000.0, 68, 12, "db"), (500.0, 11000.0, 132, 3, "multi")])
def test_bands_of_up_to_64_bins_stay_one_launch(oracle_lib, lo, hi, hop, T, kind, monkeypatch):
    """A band wider than 5.5 kHz at 44.1 kHz under 256-point frames is more than 32 bins: the fold kernel's twice-folded form
    takes up to 64 (two row tiles per parity, 4 waves a workgroup) instead of the two launches of the generic engine
    (frequencyIndexRange, CircularShortTimeFourierTransform.swift:166-191: (1000, 11000) Hz is bins [6, 64))."""
    import torch
    from syllable_detector_swift_amd.config import frequencyIndexRange
    import pyoracle as po
    base = util.sample_net()
    f0, f1 = frequencyIndexRange(256, 44100.0, lo, hi)
    F = f1 - f0
    assert 32 < F <= 64
    rng = np.random.default_rng(int(lo + hi + hop))
    if kind == "exact":
        net = nets.random_net(rng, F * T, (4,), 1)
    elif kind == "gen":
        net = nets.random_net(rng, F * T, (3,), 1, transfer=("LogSig", "SatLin"), in_fns=("normalizestd", "mapstd"))
    elif kind == "db":
        net = nets.random_net(rng, F * T, (4,), 1, in_fns=("l2normalize",))
    else:
        net = nets.random_net(rng, F * T, (2,), 3, in_fns=("l2normalize", "mapminmax"), out_fns=("mapminmax",))
    cfg = nets.variant(base, freqRange=(lo, hi), windowOverlap=256 - hop, timeRange=T, net=net,
                       thresholds=[0.4] * net.layers[-1].outputs, spectrogramScaling="db" if kind == "db" else "linear",
                       rule=_abi.RULE_ANY if kind == "multi" else _abi.RULE_FIRST)
    C = 2
    x = (synth.channels(C, 256 + 2500 * hop + 17, first=21) * np.array([[1.0], [3e-3]])).astype(np.float32)
    x[1, 100000:] *= np.float32(300.0)                            # a level step: every frame has its own scale
    o = util.oracle_for(cfg)
    with SyllableDetector(cfg, channels=C, device=0) as det:
        assert det.geometry.engine == _abi.ENGINE_FUSED and det.geometry.bins == F
        det.profile(True)
        out, fl = det.run(torch.from_numpy(x).cuda())
        torch.cuda.synchronize()
        assert util.launched(det) == ["fused_s_kernel"]
        assert det.fixupStats() == (0, 0)
        out, fl = out.cpu().numpy(), fl.cpu().numpy()
    for c in range(C):
        util.check_with_evidence(o, cfg, x[c], out[c], fl[c])     # (dB columns too: 1e-5, beyond it per-evaluation evidence)
    # ... and with the second fold switched off such a band is the generic engine's (nothing else holds 64 bins)
    monkeypatch.setenv("SYLDET_FUSED_NOFOLD2", "1")
    with SyllableDetector(cfg, channels=C, device=0) as det:
        det.profile(True)
        out2, _ = det.run(torch.from_numpy(x).cuda())
        torch.cuda.synchronize()
        assert not any(nm.startswith("fused_s") for nm, _ in det.lastTimings())
        assert np.abs(out2.cpu().numpy() - out).max() <= (2e-5 if kind != "db" else 2e-4)


@pytest.mark.parametrize("H,n_out,hop,T,chain", [(5, 1, 132, 12, ("l2normalize",)), (8, 1, 128, 10, ("l2normalize",)), (8, 2, 64, 12, ("l2normalize", "mapstd")),
                                                  (9, 1, 132, 10, ("l2normalize",)), (12, 4, 100, 8, ("normalizestd",)),
                                                  (13, 1, 132, 10, ("l2normalize", "mapminmax")), (16, 4, 132, 10, ("l2normalize",)),
                                                  (16, 1, 68, 5, ("normalize", "mapstd"))])
def test_five_to_sixteen_hidden_units_on_the_twice_folded_form(oracle_lib, H, n_out, hop, T, chain, monkeypatch):
    """The reference's network class has no limit on the hidden layer (NeuralNet.swift:155-170: any layer sizes); the fold
    kernel takes up to 16 units as one to four accumulator sets, and 256-point frames under a 256-sample window run twice-folded
    for every set count that keeps its registers (four waves: 9 .. 16 units, and 5 .. 8 where the ring leaves room for four
    waves only).  Both forms against the fp64 anchor, and against each other."""
    import torch
    import pyoracle as po
    base = util.sample_net()
    from syllable_detector_swift_amd.config import frequencyIndexRange
    f0, f1 = frequencyIndexRange(256, base.samplingRate, *base.freqRange)
    F = f1 - f0
    rng = np.random.default_rng(9000 + 16 * H + hop)
    net = nets.random_net(rng, F * T, (H,), n_out, in_fns=chain)
    cfg = nets.variant(base, windowOverlap=256 - hop, timeRange=T, net=net, thresholds=[0.3] * n_out,
                       rule=_abi.RULE_ANY if n_out > 1 else _abi.RULE_FIRST)
    C = 3
    x = (synth.channels(C, 256 + 3100 * hop + 29, first=5) * np.array([[1.0], [2e-3], [40.0]])).astype(np.float32)
    x[1, 150000:] *= np.float32(500.0)
    o = util.oracle_for(cfg)
    xd = torch.from_numpy(x).cuda()
    res = {}
    for form in ("twice", "once"):
        if form == "once":
            monkeypatch.setenv("SYLDET_FUSED_NOFOLD2", "1")
        with SyllableDetector(cfg, channels=C, device=0) as det:
            assert det.geometry.engine == _abi.ENGINE_FUSED
            det.profile(True)
            out, fl = det.run(xd)
            torch.cuda.synchronize()
            names = util.launched(det)
            assert names and all(nm.startswith("fused") for nm in names), names
            res[form] = (out.cpu().numpy(), fl.cpu().numpy(), names)
    for c in range(C):
        _, _, w64 = o.run(x[c], po.F64, cfg.rule)
        w32 = o.run(x[c], po.F32, cfg.rule)[0]
        own = float((np.abs(w32 - w64) / np.maximum(1.0, np.abs(w64))).max())
        tol = max(1e-5, 4 * own)
        for form in res:
            util.assert_outputs_close(res[form][0][c], w64, tol)
            util.assert_flags_exact(res[form][1][c], w64, cfg.thresholds, cfg.rule, tol)


@pytest.mark.parametrize("kind,T,S", [("exact", 10, 256 + 9 * 128), ("exact", 10, 256 + 24 * 128 + 77), ("exact", 10, 700001), ("exact", 1, 40000), ("exact", 12, 90011),
                                      ("gen_db", 10, 120000), ("gen_logsig_3out", 7, 65536)])
def test_hop_128_ring_as_staggered_chunks_and_as_padded_pieces(oracle_lib, kind, T, S, monkeypatch):
    """Hop 128 under a 256-sample window (BASELINE configs[0]'s framing) puts every frame of a tile on the same LDS banks.  Round 3's
    cure pads the ring inside a chunk (two half-wave DMA instructions a chunk, a mirror chunk); round 6's staggers WHOLE chunks over
    the banks (one DMA instruction a chunk, a frame read through two bases; kernels_fused_s.hip, CS8) and is what ships;
    SYLDET_FUSED_PAD128=1 keeps the padded pieces.  Same samples, same arithmetic per frame: the two forms must agree bit for bit,
    and both with the oracle -- one tile, ragged tails, rings that wrap many times, several wave segments, the spectrogram call."""
    import torch
    import pyoracle as po
    base = util.sample_net()
    rng = np.random.default_rng(41)
    if kind == "exact":
        cfg = nets.variant(base, windowOverlap=128, timeRange=T,
                           net=base.net if T == 10 else nets.random_net(rng, 29 * T, (4,), 1, in_fns=("l2normalize", "mapminmax"), out_fns=("mapminmax",)))
    elif kind == "gen_db":
        cfg = nets.variant(base, windowOverlap=128, spectrogramScaling="db",
                           net=nets.random_net(rng, 290, (3,), 1, in_fns=("l2normalize",), out_fns=()), thresholds=[0.1])
    else:
        cfg = nets.variant(base, windowOverlap=128, timeRange=T, net=nets.random_net(rng, 29 * T, (4,), 3, transfer=("LogSig", "TanSig"), in_fns=("l2normalize", "mapstd")),
                           thresholds=[0.1, 0.2, 0.3], rule=_abi.RULE_ANY)
    C = 3
    x = synth.channels(C, S, first=12, fs=cfg.samplingRate)
    x[1] *= np.float32(2e-3)
    x[2, S // 3:] *= np.float32(40.0)
    got = {}
    for form in ("staggered chunks", "padded pieces"):
        if form == "padded pieces":
            monkeypatch.setenv("SYLDET_FUSED_PAD128", "1")
        with SyllableDetector(cfg, channels=C, device=0) as det:
            assert det.geometry.engine == _abi.ENGINE_FUSED
            det.profile(True)
            out, fl = det.run(torch.from_numpy(x).cuda())
            torch.cuda.synchronize()
            assert util.launched(det) == ["fused_s_kernel"]
            cols = det.spectrogram(torch.from_numpy(x).cuda())
            torch.cuda.synchronize()
            got[form] = (out.cpu().numpy(), fl.cpu().numpy(), cols.cpu().numpy())
    monkeypatch.delenv("SYLDET_FUSED_PAD128")
    a, b = got["staggered chunks"], got["padded pieces"]
    assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2], equal_nan=True)
    o = util.oracle_for(cfg)
    for c in range(C):
        util.check_with_evidence(o, cfg, x[c], a[0][c], a[1][c])
        if cfg.spectrogramScaling == "linear":
            util.assert_columns_close(a[2][c], o.spectrogram(x[c], po.F64))
