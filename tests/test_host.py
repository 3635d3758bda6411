"""Host logic of the product on CPU: the C-ABI library loads and exports every declared symbol,
the text-format parser (N1) agrees with an independent reader and reports the reference's error
kinds, SyllableDetector.init's validation, and the loud failure without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import pyoracle as po
import util
import syllable_detector_swift_amd as sd
from syllable_detector_swift_amd import _abi, nets

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "syldet.h")).read()
    declared = set(re.findall(r"\b(syldet_[a-z_0-9]+)\s*\(", header))
    declared -= {"syldet_create_from"}                      # (none; guard against stale names)
    lib = C.CDLL(_abi.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "libsyldet.so does not export %s" % name
    assert declared == set(_abi.SIGNATURES), "ctypes table and header disagree: %s" % (declared ^ set(_abi.SIGNATURES))
    assert lib.syldet_abi_version() == 1


def test_strerror_covers_every_status():
    for st in (0, -1, -2, -3, -4, -5, -6, -7, -8, -9, -10, -11, -20, -21, -22, -23, -30):
        assert _abi.strerror(st) not in ("", "unknown status")
    assert _abi.strerror(-999) == "unknown status"


def _same_net(a, b):
    assert len(a["layers"]) == len(b["layers"]) and len(a["inputs"]) == len(b["inputs"]) and len(a["outputs"]) == len(b["outputs"])
    for k in ("samplingRate", "fourierLength", "windowLength", "windowOverlap", "freqRange", "timeRange", "scaling"):
        assert a[k] == b[k], k
    assert np.array_equal(a["thresholds"], b["thresholds"])
    for x, y in zip(a["layers"], b["layers"]):
        assert (x["inputs"], x["outputs"], x["transferFunction"]) == (y["inputs"], y["outputs"], y["transferFunction"])
        assert np.array_equal(x["weights"], y["weights"]) and np.array_equal(x["biases"], y["biases"])
    for x, y in zip(a["inputs"] + a["outputs"], b["inputs"] + b["outputs"]):
        assert x["function"] == y["function"]
        if "xOffsets" in x:
            assert np.array_equal(x["xOffsets"], y["xOffsets"]) and np.array_equal(x["gains"], y["gains"]) and x["y"] == y["y"]


@pytest.mark.parametrize("name", ["sample"] + util.case_names())
def test_text_format_round_trip_against_independent_reader(tmp_path, name):
    cfg = util.sample_net() if name == "sample" else util.load_case(name)[0]
    text = cfg.toText()
    p = tmp_path / "net.txt"
    p.write_text(text)
    parsed = sd.SyllableDetectorConfig.fromTextFile(str(p))
    _same_net(po.from_config(parsed), po.parse_text(text))
    # %.15g round-trips every float32 exactly
    _same_net(po.from_config(parsed), po.from_config(nets.variant(cfg, window=1, spectrum=0, rule=0)))
    assert (parsed.window, parsed.spectrum, parsed.rule) == (_abi.WINDOW_HAMMING, _abi.SPECTRUM_POWER, _abi.RULE_FIRST)


@pytest.mark.skipif(not os.path.exists("/root/reference/sample.txt"), reason="reference tree not present")
def test_reference_sample_file_parses_to_the_committed_fixture():
    parsed = sd.SyllableDetectorConfig.fromTextFile("/root/reference/sample.txt")
    _same_net(po.from_config(parsed), po.from_config(util.sample_net()))
    g = parsed.geometry()
    assert (g.hop, g.gap, g.f0, g.f1, g.bins, g.inputs, g.outputs, g.first_index) == (132, 0, 12, 41, 29, 290, 1, 1444)


def _write(tmp_path, text):
    p = tmp_path / "cfg.txt"
    p.write_text(text)
    return str(p)


def test_parser_line_rules(tmp_path):
    """Lines without exactly one '=' are skipped (comments, blanks, 'a=b=c'); later keys win; the
    legacy singular `threshold` is accepted; windowLength defaults to fourierLength; CRLF is trimmed."""
    cfg = nets.variant(util.sample_net())
    text = cfg.toText().replace("thresholds = ", "threshold = ")
    text = text.replace("windowLength = 256\n", "")
    text = "junk line\n\n# c = d = e\nsamplingRate = 1.0\n" + text.replace("\n", "\r\n") + "ignored=\n=ignored\n"
    parsed = sd.SyllableDetectorConfig.fromTextFile(_write(tmp_path, text))
    assert parsed.samplingRate == 44100.0 and parsed.windowLength == 256 and parsed.thresholds == [0.442442442442442]


def test_parser_error_kinds(tmp_path):
    good = util.sample_net().toText()
    with pytest.raises(sd.UnableToOpenPath):
        sd.SyllableDetectorConfig.fromTextFile(str(tmp_path / "missing.txt"))
    cases = [
        (lambda t: t.replace("samplingRate = 44100.0\n", ""), sd.MissingValue, "samplingRate"),
        (lambda t: t.replace("fourierLength = 256", "fourierLength = 255"), sd.InvalidValue, "fourierLength"),
        (lambda t: t.replace("fourierLength = 256", "fourierLength = 256.0"), sd.InvalidValue, "fourierLength"),
        (lambda t: t.replace("freqRange = 2000.0, 7000.0", "freqRange = 2000.0"), sd.MismatchedLength, "freqRange"),
        (lambda t: t.replace("thresholds = ", "thresholdz = "), sd.MissingValue, "threshold"),
        (lambda t: t.replace("scaling = linear", "scaling = cubic"), sd.InvalidValue, "scaling"),
        (lambda t: t.replace("layer0.transferFunction = TanSig", "layer0.transferFunction = ReLU"), sd.InvalidValue, "layer0.transferFunction"),
        (lambda t: t.replace("layer1.biases = ", "layer1.biases = 1.0, "), sd.MismatchedLength, "layer1.biases"),
        (lambda t: t.replace("layer1.weights = ", "layer1.weights = x, "), sd.InvalidValue, "layer1.weights"),
        (lambda t: t.replace("processInputs0.function = l2normalize", "processInputs0.function = whiten"), sd.InvalidValue, "processInputs0.function"),
        (lambda t: t.replace("processOutputs0.function = mapminmax", "processOutputs0.function = l2normalize"), sd.InvalidValue, "processOutputs0.function"),
        (lambda t: t.replace("processInputs1.yMin = -1\n", ""), sd.MissingValue, "processInputs1.yMin"),
        (lambda t: t.replace("timeRange = 10", "timeRange = ten"), sd.InvalidValue, "timeRange"),
    ]
    for edit, exc, key in cases:
        bad = edit(good)
        assert bad != good, key
        with pytest.raises(exc) as ei:
            sd.SyllableDetectorConfig.fromTextFile(_write(tmp_path, bad))
        assert key in str(ei.value)


def test_detector_init_validation():
    """The checks SyllableDetector.init / CircularShortTimeFourierTransform.init make with
    fatalError come back as statuses."""
    base = util.sample_net()
    bad = [(dict(windowOverlap=256), _abi.ERR_OVERLAP), (dict(fourierLength=128), _abi.ERR_FFT_SIZE),
           (dict(fourierLength=300), _abi.ERR_FFT_SIZE), (dict(freqRange=(7000.0, 2000.0)), _abi.ERR_FREQ_RANGE),
           (dict(freqRange=(23000.0, 24000.0)), _abi.ERR_FREQ_RANGE), (dict(timeRange=9), _abi.ERR_INPUT_MISMATCH),
           (dict(thresholds=[0.1, 0.2]), _abi.ERR_THRESHOLD_MISMATCH)]
    for change, status in bad:
        with pytest.raises(sd.SyllableDetectorError) as ei:
            nets.variant(base, **change).geometry()
        assert ei.value.status == status, change
    broken = nets.variant(base)
    broken.net.layers[1].inputs = 5
    with pytest.raises(sd.SyllableDetectorError) as ei:
        broken.geometry()
    assert ei.value.status == _abi.ERR_LAYER_SHAPE


def test_frequency_index_range_and_windows_match_the_oracle(oracle_lib):
    assert sd.frequencyIndexRange(256, 44100.0, 2000.0, 7000.0) == (12, 41)
    assert sd.frequencyIndexRange(1024, 44100.0, 2000.0, 7000.0) == (47, 163)
    assert sd.frequencyIndexRange(256, 44100.0, 0.0, 1e9) == (0, 128)
    assert sd.frequencyIndexRange(256, 44100.0, 23000.0, 24000.0) is None
    for w in range(4):
        for n in (96, 256, 1024):
            want = np.zeros(n, np.float32)
            oracle_lib.orc_window(w, n, want.ctypes.data_as(C.POINTER(C.c_float)))
            assert np.array_equal(sd.createWindow(w, n), want)


def test_no_cpu_fallback():
    """Without a gfx950 device the product refuses to construct a detector."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(sd.SyllableDetectorError) as ei:
        sd.SyllableDetector(util.sample_net())
    assert ei.value.status == _abi.ERR_NO_DEVICE


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "syllable_detector_swift_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in text and "syldet_oracle" not in text and "orc_" not in text, f


def test_ingest_entry_points_validate_before_touching_a_device():
    """Resampler / de-interleave argument checks (no GPU needed: they come before any HIP call)."""
    lib = _abi.lib
    h = _abi.Handle()
    assert lib.syldet_resampler_create(0.0, 44100.0, 1, 0, C.byref(h)) == _abi.ERR_INVALID_ARGUMENT
    assert lib.syldet_resampler_create(48000.0, -1.0, 1, 0, C.byref(h)) == _abi.ERR_INVALID_ARGUMENT
    assert lib.syldet_resampler_create(48000.0, 44100.0, 0, 0, C.byref(h)) == _abi.ERR_INVALID_ARGUMENT
    assert lib.syldet_resampler_create(48000.0, 44100.0, 1, 0, None) == _abi.ERR_INVALID_ARGUMENT
    assert lib.syldet_resampler_count(None, 1000) == 0
    assert lib.syldet_resampler_destroy(None) == 0
    assert lib.syldet_resample(None, None, 10, 10, None, 10, None) == _abi.ERR_INVALID_ARGUMENT
    # channel selections outside the interleaved layout
    for total, first, count in [(0, 0, 1), (2, -1, 1), (2, 1, 2), (2, 0, 0)]:
        assert lib.syldet_deinterleave_device(None, 10, total, first, count, None, 10, None) == _abi.ERR_INVALID_ARGUMENT
    assert lib.syldet_deinterleave_device(None, 0, 2, 0, 2, None, 0, None) == 0        # nothing to do
    assert lib.syldet_run_interleaved(None, None, 10, 1, None, None) == _abi.ERR_INVALID_ARGUMENT


def test_resampler_needs_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(sd.SyllableDetectorError) as ei:
        sd.ResamplerLinear(48000.0, 44100.0)
    assert ei.value.status == _abi.ERR_NO_DEVICE


def test_wide_engine_is_declared_and_needs_a_device():
    import torch
    assert _abi.ENGINE_WIDE_BF16 == 3
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from syllable_detector_swift_amd import nets
    with pytest.raises(sd.SyllableDetectorError) as ei:
        sd.SyllableDetector(nets.wide_mlp(util.sample_net()), engine=_abi.ENGINE_WIDE_BF16)
    assert ei.value.status == _abi.ERR_NO_DEVICE


def test_mfma_hazard_checker_sees_a_vector_write_in_front_of_a_hand_placed_mfma(tmp_path):
    """tools/check_mfma_hazards.py guards kernels_fused_r.hip's assembly-statement MFMAs at build time (csrc/Makefile)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("chk", os.path.join(ROOT, "tools", "check_mfma_hazards.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    mfma = "\t;;#ASMSTART\n\tv_mfma_f32_16x16x32_f16 v[0:3], a[8:11], v[20:23], v[0:3]\n\t;;#ASMEND\n"
    ok = tmp_path / "ok.s"
    ok.write_text("\tv_accvgpr_write_b32 a9, v5\n\tv_add_f32_e32 v30, v31, v32\n\tds_read_b64 v[40:41], v3\n" + mfma)
    assert chk.check(str(ok)) == (1, [])
    for writer in ("\tv_accvgpr_write_b32 a9, v5\n", "\tv_mov_b32_e32 v21, v7\n", "\tv_accvgpr_write_b32 a10, v5\n\tv_add_f32_e32 v30, v31, v32\n"):
        bad = tmp_path / "bad.s"
        bad.write_text(writer + mfma)
        seen, found = chk.check(str(bad))
        assert seen == 1 and len(found) == 1
    waited = tmp_path / "waited.s"
    waited.write_text("\tv_mov_b32_e32 v21, v7\n\ts_nop 1\n" + mfma)
    assert chk.check(str(waited)) == (1, [])


def test_operand_selection_checker_refuses_src1_high_into_the_low_half(tmp_path):
    """tools/check_pk_opsel.py (csrc/Makefile runs it on the ISA of every kernel file): a packed fp32 instruction that takes
    src1's high register into its low half loses that half's product beside another wave's matrix instructions on MI355X
    (tools/ubench/pkfma_opsel.hip, MEASUREMENTS R5.1); the same selection on src0 / src2 and op_sel_hi are fine.  Also the two
    checks on functions that issue LDS-DMA through assembly: M0 theirs alone, vmcnt(0) in front of every later barrier."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("chk2", os.path.join(ROOT, "tools", "check_pk_opsel.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    head = "_Z1kv:\n"
    fine = tmp_path / "fine.s"
    fine.write_text(head + "\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel:[1,0,0]\n\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel_hi:[1,0,1]\n"
                    "\tv_pk_mul_f32 v[0:1], v[2:3], v[4:5]\n\tv_fma_mix_f32 v0, v1, v2, v3 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n")
    bad, n_pk, n_dma = chk.check(str(fine))
    assert bad == [] and n_pk == 3 and n_dma == 0
    for line in ("v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel:[0,1,0]", "v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]",
                 "v_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,1] op_sel_hi:[0,1]", "v_fma_mix_f32 v0, v1, v2, v3 op_sel:[0,1,0] op_sel_hi:[0,1,0]"):
        f = tmp_path / "bad.s"
        f.write_text(head + "\t" + line + "\n")
        assert len(chk.check(str(f))[0]) == 1, line
    dma = "\t;;#ASMSTART\n\ts_mov_b32 m0, s7\n\ts_nop 0\n\tbuffer_load_dwordx4 v1, s[0:3], s6 offen lds\n\t;;#ASMEND\n"
    ok = tmp_path / "dma_ok.s"
    ok.write_text(head + "\ts_barrier\n" + dma + "\tv_add_f32_e32 v2, v3, v4\n\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n")
    assert chk.check(str(ok)) == ([], 0, 1)                        # (the barrier in front of the first DMA is the prologue's)
    nowait = tmp_path / "dma_nowait.s"
    nowait.write_text(head + dma + "\tv_add_f32_e32 v2, v3, v4\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n")
    assert len(chk.check(str(nowait))[0]) == 1
    m0 = tmp_path / "dma_m0.s"
    m0.write_text(head + dma + "\ts_mov_b32 m0, s9\n\ts_waitcnt vmcnt(0)\n\ts_barrier\n")
    assert len(chk.check(str(m0))[0]) == 1


def test_header_is_plain_c_and_the_library_refuses_to_run_without_a_device(tmp_path):
    """include/syldet.h compiles as strict C99 (-pedantic -Werror) and a C program links against libsyldet -- the way the
    reference's bridging header binds its one C API (Common/Common-Bridging-Header.h:5).  Without a gfx950 device the
    program must be told SYLDET_ERR_NO_DEVICE: there is no CPU path behind the ABI."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "syllable_detector_swift_amd", "lib")
    exe = str(tmp_path / "header_is_c")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "tests", "c", "header_is_c.c"), "-o", exe, "-L" + lib, "-lsyldet",
                    "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    import torch
    if torch.cuda.is_available():
        pytest.skip("the device run of this program is tests/test_parity_gpu.py::test_c_program_over_the_abi")
    cfg = util.sample_net()
    (tmp_path / "net.txt").write_text(cfg.toText())
    np.zeros(4000, np.float32).tofile(str(tmp_path / "x.f32"))
    r = subprocess.run([exe, str(tmp_path / "net.txt"), str(tmp_path / "x.f32")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "no-device", (r.stdout, r.stderr)


def test_shard_table_of_the_sharded_bank_matches_the_process_per_gpu_table():
    """syldet_shard_table / syldet_shard_evaluations / syldet_shard_samples (the one-process bank's host arithmetic, no device)
    against dist.shard_channels / shard_plane / shard_evaluations / time_shard_samples: 4096 / 8 (BASELINE configs[3]), the
    ragged 4099 / 8, and fewer channels than shards."""
    import ctypes as C
    from syllable_detector_swift_amd import dist
    from syllable_detector_swift_amd.bank import shard_table
    for total, world in ((4096, 8), (4099, 8), (8, 8), (9, 4), (3, 8), (1, 4), (2, 3), (7, 1), (5, 7)):
        got = shard_table(total, world)
        want = [dist.shard_plane(total, world, r) for r in range(world)]
        assert got == want, (total, world)
        if total >= world:
            assert [(f, n) for f, n, _, _ in got] == [dist.shard_channels(total, world, r) for r in range(world)]
            assert sum(n for _, n, _, _ in got) == total
    assert shard_table(4096, 8) == [(512 * r, 512, 0, 1) for r in range(8)]
    assert [n for _, n, _, _ in shard_table(4099, 8)] == [513, 513, 513, 512, 512, 512, 512, 512]
    with pytest.raises(sd.SyllableDetectorError):
        shard_table(0, 4)
    cfg = util.sample_net()
    c, keep = cfg.to_abi()
    for E, parts in ((15877, 8), (5, 8), (100, 3), (0, 2)):
        for part in range(parts):
            f, n = C.c_int64(), C.c_int64()
            assert _abi.lib.syldet_shard_evaluations(E, parts, part, C.byref(f), C.byref(n)) == 0
            assert (f.value, n.value) == dist.shard_evaluations(E, parts, part)
            s0, s1 = C.c_int64(), C.c_int64()
            assert _abi.lib.syldet_shard_samples(C.byref(c), f.value, n.value, C.byref(s0), C.byref(s1)) == 0
            assert (s0.value, s1.value) == dist.time_shard_samples(132, 0, 256, 10, f.value, n.value)
    assert _abi.lib.syldet_shard_evaluations(10, 2, 2, C.byref(f), C.byref(n)) == _abi.ERR_INVALID_ARGUMENT


def test_sharded_bank_refuses_to_exist_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("device runs: tests/test_sharded_gpu.py")
    from syllable_detector_swift_amd.bank import ShardedSyllableDetectorBank
    with pytest.raises(sd.SyllableDetectorError) as e:
        ShardedSyllableDetectorBank(util.sample_net(), 8, [0, 1])
    assert e.value.status == _abi.ERR_NO_DEVICE


def test_sharded_abi_rejects_bad_arguments_without_touching_a_device():
    import ctypes as C
    cfg = util.sample_net()
    c, keep = cfg.to_abi()
    h = _abi.Handle()
    devs = (C.c_int32 * 2)(0, 1)
    lib = _abi.lib
    assert lib.syldet_create_sharded(None, 4, devs, 2, 0, 0, C.byref(h)) == _abi.ERR_INVALID_ARGUMENT
    assert lib.syldet_create_sharded(C.byref(c), 4, None, 2, 0, 0, C.byref(h)) == _abi.ERR_INVALID_ARGUMENT
    assert lib.syldet_create_sharded(C.byref(c), 0, devs, 2, 0, 0, C.byref(h)) == _abi.ERR_INVALID_ARGUMENT
    assert lib.syldet_create_sharded(C.byref(c), 4, devs, 0, 0, 0, C.byref(h)) == _abi.ERR_INVALID_ARGUMENT
    assert lib.syldet_create_sharded(C.byref(c), 4, devs, 2, 0, 7, C.byref(h)) == _abi.ERR_INVALID_ARGUMENT      # unknown exchange
    assert not h.value
    # NULL handles: statuses and zeros, never a crash
    assert lib.syldet_sharded_destroy(None) == 0
    assert lib.syldet_sharded_channels(None) == 0 and lib.syldet_sharded_shards(None) == 0 and lib.syldet_sharded_rccl_ranks(None) == 0
    assert lib.syldet_sharded_run(None, None, 0, 0, None, None) == _abi.ERR_INVALID_ARGUMENT
    assert lib.syldet_sharded_run_device(None, None, 0, None, None, None, None) == _abi.ERR_INVALID_ARGUMENT
    assert lib.syldet_sharded_synchronize(None) == _abi.ERR_INVALID_ARGUMENT
    assert not lib.syldet_sharded_bank(None, 0) and not lib.syldet_sharded_stream(None, 0)
    s = _abi.Shard()
    assert lib.syldet_sharded_shard(None, 0, C.byref(s)) == _abi.ERR_INVALID_ARGUMENT
    assert lib.syldet_shard_table(4, 2, None) == _abi.ERR_INVALID_ARGUMENT
    p = C.c_void_p()
    assert lib.syldet_host_alloc(16, None) == _abi.ERR_INVALID_ARGUMENT and lib.syldet_host_free(None) == 0


def test_shift_fusion_checker_tells_shifted_additions_from_moves(tmp_path):
    """tools/check_dpp_fusion.py (run by csrc/Makefile on kernels_bdft.hip's ISA): the block-transform kernel's sliding sums keep their
    speed only while every lane shift is an operand of its addition."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_dpp_fusion", os.path.join(ROOT, "tools", "check_dpp_fusion.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    head = "_ZN2sd15bdft_net_kernelILi4EEEv: ; @_ZN2sd15bdft_net_kernelILi4EEEv\n"
    fused = tmp_path / "fused.s"
    fused.write_text(head + "\tv_add_f32_dpp v1, v2, v3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" * 200 +
                     "\tv_mov_b32_dpp v4, v5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" * 300 + "\ts_endpgm\n")
    assert mod.check(str(fused))[1] == []
    moves = tmp_path / "moves.s"
    moves.write_text(head + "\tv_mov_b32_dpp v4, v5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_pk_add_f32 v[0:1], v[2:3], v[4:5]\n" * 200 + "\ts_endpgm\n")
    assert len(mod.check(str(moves))[1]) == 1
