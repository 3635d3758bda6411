"""The command line tool without a GPU: options, usage text and exit code (SyllableDetectorCLI/main.swift:19-41),
the WAV reader's header handling, error messages.  Nothing here computes."""
import os
import subprocess

import numpy as np
import pytest

import util
import wavutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "syllable_detector_swift_amd", "lib", "syllable-detector-cli")


def run(*args):
    return subprocess.run([CLI, *args], capture_output=True, text=True, timeout=120)


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.exists(CLI):
        pytest.fail("syllable-detector-cli has not been built: python -c 'import __graft_entry__ as g; g.build()'")


@pytest.mark.parametrize("args", [["-h"], ["--help"], ["--bogus"], [], ["-a", "x.wav"]])
def test_usage_text_and_exit_code(args):
    r = run(*args)
    assert r.returncode == 64                                   # EX_USAGE, main.swift:40
    assert "Path to trained network file." in r.stdout
    assert "\t0,1593298,36.1292063492063,0.918557" in r.stdout  # the example line of main.swift:33
    assert "4. The first neural network output." in r.stdout


def test_missing_option_value():
    r = run("-n")
    assert r.returncode == 64 and "Missing value for --net" in r.stderr


def test_unreadable_network_file(tmp_path):
    r = run("-n", str(tmp_path / "nope.txt"), "-a", "x.wav")
    assert r.returncode == 1
    assert r.stderr.startswith("Unable to load the network configuration:")


@pytest.mark.parametrize("kind,bits,label", [("pcm16", 16, "pcm"), ("pcm8", 8, "pcm"), ("pcm24", 24, "pcm"), ("pcm32", 32, "pcm"),
                                             ("float32", 32, "float"), ("float64", 64, "float")])
@pytest.mark.parametrize("extensible", [False, True])
def test_probe_reads_headers(tmp_path, kind, bits, label, extensible):
    rng = np.random.default_rng(bits)
    a = rng.integers(-100, 100, size=(1237, 3)) if label == "pcm" else rng.standard_normal((1237, 3))
    if kind == "pcm8":
        a = a + 128
    p = str(tmp_path / ("t_%s.wav" % kind))
    wavutil.write_wav(p, a, 22050, kind, extensible)
    r = run("--probe", "-a", p)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip() == "%s: 3 channel(s), 22050.0 Hz, %s %d-bit, 1237 frames" % (p, label, bits)


def test_probe_reports_bad_files(tmp_path):
    bad = tmp_path / "bad.wav"
    bad.write_bytes(b"RIFFxxxxWAVEjunk")
    notwav = tmp_path / "x.txt"
    notwav.write_text("hello world, not audio")
    r = run("--probe", "-a", str(bad), "-a", str(notwav), "-a", str(tmp_path / "missing.wav"))
    assert r.returncode == 1
    lines = r.stderr.strip().splitlines()
    assert lines[0] == "Unable to read %s: no fmt chunk" % bad
    assert lines[1] == "Unable to read %s: not a RIFF/WAVE file" % notwav
    assert lines[2] == "Unable to read %s: cannot open file" % (tmp_path / "missing.wav")


def test_the_one_output_line_the_reference_holds():
    """SyllableDetectorCLI/main.swift:33 shows an event line, `0,1593298,36.1292063492063,0.918557`: channel 0, sample
    1 593 298 at 44.1 kHz, output 0.918557 -- 15 and 6 significant digits, which is what `\\(Double)` / `\\(Float)` print under
    the Swift 4.0 toolchain the project declares (project.pbxproj:593).  `--format swift4` must reproduce it byte for byte
    from (sample, rate, fp32 output); the default prints the shortest digits that round-trip (Swift >= 4.2)."""
    r = run("--format", "swift4", "--format-line", "0", "1593298", "44100", "0.918557")
    assert r.returncode == 0 and r.stdout == "0,1593298,36.1292063492063,0.918557\n"
    r = run("--format-line", "0", "1593298", "44100", "0.918557")
    assert r.returncode == 0 and r.stdout == "0,1593298,%r,%s\n" % (1593298 / 44100, "0.918557")
    # whole numbers get ".0" in both forms; small and large values keep C's exponent form under swift4
    assert run("--format", "swift4", "--format-line", "3", "88200", "44100", "1", "0.5").stdout == "3,88200,2.0,1.0,0.5\n"
    assert run("--format", "swift4", "--format-line", "1", "1", "3", "0.33333334", "1e-7").stdout == "1,1,0.333333333333333,0.333333,1e-07\n"
    assert run("--format", "shortest", "--format-line", "1", "1", "3", "0.33333334").stdout == "1,1,0.3333333333333333,0.33333334\n"
    assert run("--format", "swift5").returncode == 64
