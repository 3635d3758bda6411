"""The command line tool end to end on the GPU (SyllableDetectorCLI/main.swift:57-131, TrackDetector.swift:45-105):
WAV in, `channel,sample,seconds,out0` lines out, against the oracle on the decoded samples.  Sample numbers and
timestamps must match exactly, outputs to the 1e-5 bar."""
import os
import subprocess

import numpy as np
import pytest

import pyoracle as po
import util
import wavutil
from syllable_detector_swift_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "syllable_detector_swift_amd", "lib", "syllable-detector-cli")
FS = 44100


def run(*args):
    r = subprocess.run([CLI, *args], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    return r.stdout.splitlines()


@pytest.fixture(scope="module")
def net(tmp_path_factory):
    cfg = util.sample_net()
    p = tmp_path_factory.mktemp("net") / "net.txt"
    p.write_text(cfg.toText())
    return cfg, str(p)


def quantise(x):
    return np.clip(np.round(x * 32768.0), -32768, 32767).astype(np.int16)


def expected_events(cfg, x, debounce=0.0):
    """[(sample, seconds-string, output)] for one decoded channel."""
    o = util.oracle_for(cfg)
    _, _, o64 = o.run(x, po.F64, po.RULE_ANY)
    flags = (o64 >= np.asarray(cfg.thresholds)[None, :]).any(axis=1).astype(np.uint8)
    idx = o.detections(flags, debounce)
    first, hop = int(idx[0]) if idx.size else 0, cfg.windowLength - cfg.windowOverlap
    base = cfg.windowLength + hop * (cfg.timeRange - 1) + max(0, -cfg.windowOverlap)
    return [(int(i), repr(int(i) / cfg.samplingRate), o64[(int(i) - base) // hop]) for i in idx], o64


def check_lines(lines, want, channel_of=lambda k: None):
    assert len(lines) == len(want), (lines[:5], want[:5])
    for line, (ch, sample, secs, outs) in zip(lines, want):
        parts = line.split(",")
        assert int(parts[0]) == ch and int(parts[1]) == sample
        assert parts[2] == secs                                 # shortest round-trip digits, like Swift's \(Double)
        got = np.array([float(v) for v in parts[3:]])
        assert got.shape == outs.shape
        assert np.abs(got - outs).max() <= util.TOL * max(1.0, np.abs(outs).max())


@pytest.mark.parametrize("debounce", [None, 0.25])
@pytest.mark.parametrize("chunk", [8192, 0])
def test_stereo_pcm16_file(tmp_path, net, debounce, chunk):
    cfg, net_path = net
    n = 6 * FS
    q = np.stack([quantise(synth.syllable_channel(n, util.template(), seed=21)),
                  quantise(synth.syllable_channel(n, util.template(), seed=22))], axis=1)
    wav = str(tmp_path / "stereo.wav")
    wavutil.write_wav(wav, q, FS, "pcm16")
    x = wavutil.to_float(q, "pcm16")
    per_channel = [expected_events(cfg, x[:, c], debounce or 0.0)[0] for c in range(2)]
    assert sum(len(e) for e in per_channel) >= 4, "fixture should fire a few times"
    want = [(c, s, t, o) for c in range(2) for (s, t, o) in per_channel[c]]
    if chunk:
        want.sort(key=lambda w: ((w[1] - 1) // chunk, w[0], w[1]))     # buffer by buffer, track by track (main.swift:126-130)
    args = ["-n", net_path, "-a", wav, "--chunk", str(chunk)] + (["-d", str(debounce)] if debounce else [])
    check_lines(run(*args), want)
    if debounce:
        assert len(want) < sum(len(expected_events(cfg, x[:, c])[0]) for c in range(2)), "debounce should drop events"


def test_two_files_print_their_names(tmp_path, net):
    cfg, net_path = net
    files, want = [], []
    for k, kind in enumerate(["float32", "pcm24"]):
        x = synth.syllable_channel(3 * FS, util.template(), seed=30 + k).astype(np.float32)
        stored = x[:, None] if kind == "float32" else np.round(x[:, None] * 8388608.0).astype(np.int32)
        p = str(tmp_path / ("f%d.wav" % k))
        wavutil.write_wav(p, stored, FS, kind, extensible=bool(k))
        files.append(p)
        want.append([(0, s, t, o) for (s, t, o) in expected_events(cfg, wavutil.to_float(stored, kind)[:, 0])[0]])
    lines = run("-n", net_path, "-a", files[0], "-a", files[1])
    assert lines[0] == files[0]                                  # main.swift:122-124
    second = lines.index(files[1])
    check_lines(lines[1:second], want[0])
    check_lines(lines[second + 1:], want[1])


def test_other_sampling_rate_goes_through_resampler_linear(tmp_path, net):
    cfg, net_path = net
    x48 = synth.syllable_channel(4 * 48000, util.template(), seed=41, every=24000).astype(np.float32)
    p = str(tmp_path / "r48.wav")
    wavutil.write_wav(p, x48[:, None], 48000, "float32")
    y = po.Resampler(48000.0, cfg.samplingRate).resample(x48)
    ev, _ = expected_events(cfg, y)
    check_lines(run("-n", net_path, "-a", p), [(0, s, t, o) for (s, t, o) in ev])


def test_short_and_silent_files_print_nothing(tmp_path, net):
    cfg, net_path = net
    p1, p2 = str(tmp_path / "short.wav"), str(tmp_path / "silent.wav")
    wavutil.write_wav(p1, np.zeros((100, 1), np.int16), FS, "pcm16")
    wavutil.write_wav(p2, np.zeros((FS, 2), np.int16), FS, "pcm16")
    assert run("-n", net_path, "-a", p1) == []
    assert run("-n", net_path, "-a", p2) == []                   # 0/0 in l2normalize: NaN never detects
