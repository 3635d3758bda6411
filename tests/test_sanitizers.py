"""Sanitizer runs of the HOST side ("ASan/UBSan on the CPU restatement + host shim"; GPU sanitizers are not
available on the pool and nothing here touches a device): the text-format parser and SyllableDetector.init's validation, the
command line tool's WAV reader and argument loop over corpora of truncated, oversized and NaN-laden files under
AddressSanitizer + UndefinedBehaviorSanitizer; the CPU oracle's own test-suite under the same; the streaming front end's
single-producer / single-consumer ring under ThreadSanitizer.  The reference's error surface for the parser:
SyllableDetectorConfig.swift:50-55 (ParseError kinds), :183-189 (lines without exactly one '=' are skipped)."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

import util
import wavutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "tests", "sanitize")
OUT = os.path.join(SAN, "_build")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=86", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=87",
           TSAN_OPTIONS="exitcode=88")


@pytest.fixture(scope="module")
def built():
    r = subprocess.run(["make", "-C", SAN, "-j4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return OUT


def _run(exe, args, env=ENV, timeout=600):
    r = subprocess.run([exe] + list(args), capture_output=True, text=True, timeout=timeout, env=env, errors="replace")
    assert r.returncode not in (86, 87, 88) and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, \
        "sanitizer report from %s:\n%s" % (os.path.basename(exe), r.stderr[-3000:])
    return r


def _config_corpus(tmp_path):
    text = util.sample_net().toText()
    lines = text.split("\n")
    rng = np.random.default_rng(5)
    files = {"good.txt": text, "empty.txt": "", "no_equals.txt": "just words\n# a comment\n\n", "binary.txt": None}
    for k in range(60):                                           # truncated at a random byte
        files["trunc_%02d.txt" % k] = text[: int(rng.integers(0, len(text)))]
    poison = ["nan", "inf", "-inf", "1e400", "-1", "0", "99999999999999999999", "2147483648", "-2147483649", "", " ", "0x10", "1,2,3", "abc", "1e-400", "4.5"]
    keys = [i for i, l in enumerate(lines) if "=" in l]
    for k in range(120):                                          # one value replaced
        L = list(lines)
        i = int(rng.choice(keys))
        key = L[i].split("=")[0]
        L[i] = key + "= " + str(rng.choice(poison))
        files["poison_%03d.txt" % k] = "\n".join(L)
    for k in range(40):                                           # a line dropped / doubled / split by a second '='
        L = list(lines)
        i = int(rng.choice(keys))
        how = k % 4
        if how == 0:
            del L[i]
        elif how == 1:
            L.insert(i, L[i])
        elif how == 2:
            L[i] = L[i] + " = 3"
        else:
            L[i] = L[i].replace(",", ",,", 3)
        files["shape_%02d.txt" % k] = "\n".join(L)
    for k in range(30):                                           # array lengths that disagree with the declared sizes
        L = list(lines)
        i = int(rng.choice([j for j in keys if "," in L[j]]))
        key, val = L[i].split("=", 1)
        vals = val.split(",")
        L[i] = key + "=" + ",".join(vals[: int(rng.integers(0, len(vals)))] if k % 2 == 0 else vals + vals[: int(rng.integers(1, 50))])
        files["length_%02d.txt" % k] = "\n".join(L)
    huge = list(lines)
    for i in keys:
        if huge[i].split("=")[0].strip().endswith(("inputs", "outputs", "Count")):
            huge[i] = huge[i].split("=")[0] + "= 2000000000"
    files["huge_counts.txt"] = "\n".join(huge)
    files["long_line.txt"] = "layer0.weights = " + ",".join(["1.5"] * 400000) + "\n" + text
    files["crlf.txt"] = text.replace("\n", "\r\n")
    files["nul.txt"] = text[:300] + "\0\0\0" + text[300:]
    paths = []
    for name, body in files.items():
        p = tmp_path / name
        if body is None:
            p.write_bytes(bytes(rng.integers(0, 256, 5000, dtype=np.uint8)))
        else:
            p.write_bytes(body.encode("utf-8", "replace"))
        paths.append(str(p))
    paths.append(str(tmp_path / "does_not_exist.txt"))
    paths.append(str(tmp_path))                                   # a directory
    return paths


def test_config_parser_under_asan_and_ubsan(built, tmp_path):
    paths = _config_corpus(tmp_path)
    r = _run(os.path.join(built, "config_asan"), paths)
    assert r.returncode == 0, r.stdout[-500:] + r.stderr[-2000:]
    rows = [tuple(int(v) for v in l.split()) for l in r.stdout.strip().splitlines()]
    assert len(rows) == len(paths)
    assert rows[0] == (0, 0)                                      # the untouched file parses and validates
    known = {0, -20, -21, -22, -23, -1, -7, -11, -30}             # OK, the four ParseError kinds, invalid argument, layer shape (NeuralNet.swift:244-254), out of memory, unsupported
    assert {st for st, _ in rows} <= known, sorted({st for st, _ in rows} - known)
    assert rows[-2][0] == -20                                     # unableToOpenPath
    assert sum(st == 0 for st, _ in rows) < len(rows) // 2        # the corpus does break things


def _wav_corpus(tmp_path):
    rng = np.random.default_rng(9)
    paths = []
    for kind in ("pcm16", "pcm8", "pcm24", "pcm32", "float32", "float64"):
        for ext in (False, True):
            a = rng.integers(-100, 100, size=(257, 3)) if kind.startswith("pcm") else rng.standard_normal((257, 3))
            if kind == "pcm8":
                a = a + 128
            p = str(tmp_path / ("ok_%s_%d.wav" % (kind, ext)))
            wavutil.write_wav(p, a, 22050, kind, ext)
            paths.append(p)
    good = open(paths[0], "rb").read()
    fl = open(paths[8], "rb").read()                              # float32
    variants = {"empty.wav": b"", "riff_only.wav": b"RIFF", "not_wave.wav": b"RIFF\x10\0\0\0WAVX" + good[12:]}
    for k in range(50):
        variants["trunc_%02d.wav" % k] = good[: int(rng.integers(0, len(good)))]
    for k in range(80):                                           # a header byte flipped
        b = bytearray(good if k % 2 == 0 else fl)
        i = int(rng.integers(0, 70))
        b[i] = int(rng.integers(0, 256))
        variants["flip_%02d.wav" % k] = bytes(b)
    fmt_at = good.index(b"fmt ") + 8
    for name, off, code, val in (("zero_channels", 2, "<H", 0), ("many_channels", 2, "<H", 65535), ("zero_rate", 4, "<I", 0), ("zero_align", 12, "<H", 0),
                                 ("zero_bits", 14, "<H", 0), ("odd_bits", 14, "<H", 13), ("huge_bits", 14, "<H", 65535), ("format_7", 0, "<H", 7)):
        b = bytearray(good)
        b[fmt_at + off: fmt_at + off + struct.calcsize(code)] = struct.pack(code, val)
        variants[name + ".wav"] = bytes(b)
    data_at = good.index(b"data") + 4
    for name, val in (("data_huge", 0xFFFFFFFF), ("data_past_end", len(good) * 4), ("data_zero", 0), ("data_odd", 7)):
        b = bytearray(good)
        b[data_at: data_at + 4] = struct.pack("<I", val)
        variants[name + ".wav"] = bytes(b)
    b = bytearray(good)
    b[good.index(b"LIST") + 4: good.index(b"LIST") + 8] = struct.pack("<I", 0xFFFFFFF0)     # a chunk that claims the rest of the address space
    variants["chunk_huge.wav"] = bytes(b)
    nan = np.full((64, 2), np.nan)
    nan[::3] = np.inf
    p = str(tmp_path / "nan_inf.wav")
    wavutil.write_wav(p, nan, 44100, "float32")
    paths.append(p)
    for name, body in variants.items():
        q = tmp_path / name
        q.write_bytes(body)
        paths.append(str(q))
    paths.append(str(tmp_path / "missing.wav"))
    return paths


def test_wav_reader_under_asan_and_ubsan(built, tmp_path):
    paths = _wav_corpus(tmp_path)
    r = _run(os.path.join(built, "wav_asan"), paths)
    assert r.returncode == 0, r.stdout[-500:] + r.stderr[-2000:]
    rows = r.stdout.strip().splitlines()
    assert len(rows) == len(paths)
    for row in rows[:12]:                                         # the twelve well-formed files read whole
        assert row.split()[:4] == ["1", "1", "3", "257"], row
    assert sum(row.split()[1] == "0" for row in rows) > 40        # and the corpus does break things


def test_command_line_tool_under_asan_and_ubsan(built, tmp_path):
    """The tool's own binary: usage, every option's missing value, --probe over the broken WAVs, the number formats."""
    exe = os.path.join(built, "cli_asan")
    env = dict(ENV, ASAN_OPTIONS="detect_leaks=0:exitcode=86")     # (the HIP runtime's start-up allocations are not this tool's)
    paths = _wav_corpus(tmp_path)
    r = _run(exe, ["--probe"] + [a for p in paths for a in ("-a", p)], env)
    assert r.returncode == 1 and r.stdout.count("channel(s)") >= 12
    for args, code in ((["-h"], 64), ([], 64), (["-n"], 64), (["-a"], 64), (["-d"], 64), (["--device"], 64), (["--chunk"], 64), (["--format"], 64),
                       (["--format", "x"], 64), (["--format-line", "1"], 64), (["-n", str(tmp_path / "nope")], 1),
                       (["--format", "swift4", "--format-line", "0", "1593298", "44100", "0.918557"], 0),
                       (["--format-line", "0", "-5", "0", "nan", "inf", "1e39", "-0"], 0),
                       (["-d", "abc", "-n", str(tmp_path / "nope")], 1)):
        r = _run(exe, args, env)
        assert r.returncode == code, (args, r.returncode, r.stderr[-300:])
    # a network file from the parser's corpus of broken ones: an error message and exit code 1, nothing else
    for p in _config_corpus(tmp_path)[1:40]:
        r = _run(exe, ["-n", p], env)
        assert r.returncode in (0, 1), (p, r.returncode)


def test_oracle_suite_under_asan_and_ubsan(built):
    """tests/test_oracle.py once more with the oracle built -fsanitize=address,undefined (the CPU restatement is what every
    parity claim is held against: its own indexing must be clean)."""
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    ubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan.so not found")
    env = dict(ENV, LD_PRELOAD=asan + (":" + ubsan if os.path.isabs(ubsan) and os.path.exists(ubsan) else ""),
               ASAN_OPTIONS="detect_leaks=0:exitcode=86", SYLDET_ORACLE_LIB=os.path.join(built, "libsyldet_oracle_asan.so"))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stdout[-2000:] + r.stderr[-3000:]
    assert " passed" in r.stdout


def test_sample_ring_under_thread_sanitizer(built):
    r = _run(os.path.join(built, "ring_tsan"), [])
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr[-2000:]
    assert "ThreadSanitizer" not in r.stderr
