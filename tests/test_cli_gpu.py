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


def convert_rate(x, rate_in, rate_out):
    """What the tool does to a file at another rate: linear interpolation at positions i * rate_in / rate_out, in fp64."""
    n = x.size
    m = int((n - 1) * rate_out / rate_in) + 1
    pos = np.arange(m, dtype=np.float64) * (rate_in / rate_out)
    k = np.minimum(pos.astype(np.int64), n - 1)
    return (x[k].astype(np.float64) + (pos - k) * (x[np.minimum(k + 1, n - 1)].astype(np.float64) - x[k])).astype(np.float32)


def test_other_sampling_rate_is_converted_with_exact_positions(tmp_path, net):
    cfg, net_path = net
    x48 = synth.syllable_channel(4 * 48000, util.template(), seed=41, every=24000).astype(np.float32)
    p = str(tmp_path / "r48.wav")
    wavutil.write_wav(p, x48[:, None], 48000, "float32")
    ev, _ = expected_events(cfg, convert_rate(x48, 48000.0, cfg.samplingRate))
    check_lines(run("-n", net_path, "-a", p), [(0, s, t, o) for (s, t, o) in ev])


def test_long_file_at_another_rate_against_fp64_interpolation(tmp_path, net):
    """Six minutes at 48 kHz: read positions are computed in fp64, so the audio the detector sees is the linear interpolation of
    the file from its first second to its last (an fp32 position ramp would be half a sample off after ~95 s), and so are
    the detections: sample numbers exact, outputs to 1e-5."""
    cfg, net_path = net
    n = 360 * 48000
    x48 = np.zeros(n, np.float32)
    from scipy.signal import resample_poly
    seg = resample_poly(synth.syllable_channel(8 * FS, util.template(), seed=43).astype(np.float64), 160, 147).astype(np.float32)   # 44.1 -> 48 kHz
    for at in (0, 100, 200, 352):                                # syllables at the start, in the middle and in the last seconds
        x48[at * 48000:at * 48000 + seg.size] = seg
    x48 += (0.003 * np.random.default_rng(7).standard_normal(n)).astype(np.float32)
    p = str(tmp_path / "long48.wav")
    wavutil.write_wav(p, x48[:, None], 48000, "float32")
    lines = run("-n", net_path, "-a", p)
    ev, o64 = expected_events(cfg, convert_rate(x48, 48000.0, cfg.samplingRate))
    assert len(ev) >= 8 and ev[-1][0] > 340 * 44100, "fixture should fire in the last seconds too"
    check_lines(lines, [(0, s, t, o) for (s, t, o) in ev])
