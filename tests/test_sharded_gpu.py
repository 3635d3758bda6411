"""The detector bank that spans several GPUs of ONE process (syldet_create_sharded; the reference is one process that owns
every channel: Processor.swift:57-59,128-141, main.swift:86-89,126-130) and the pipelined host-pointer batch call, on the one
GPU of the test box: `devices = [0]` is one RCCL rank (the library's own ncclCommInitAll + ncclAllGather), `devices = [0, 0,
...]` rehearses the shard table, the time-axis split and the copy exchange.  Every result must be the plain bank's, bit for
bit, on the kernels that scale per frame or per hop-aligned block (a shard's evaluations then do not depend on how the bank is
cut); the pass-scaled kernels are held to the oracle's bar."""
import os
import subprocess

import numpy as np
import pytest

import pyoracle as po
import util
from syllable_detector_swift_amd import SyllableDetector, _abi, nets, synth
from syllable_detector_swift_amd.bank import PinnedArray, ShardedSyllableDetectorBank

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    return torch


def _plain(cfg, x, engine=0):
    torch = _torch()
    with SyllableDetector(cfg, channels=x.shape[0], device=0, engine=engine) as det:
        out, fl = det.run(torch.from_numpy(x).cuda())
        torch.cuda.synchronize()
        return out.cpu().numpy(), fl.cpu().numpy()


def _check_bank(cfg, x, devices, exchange=_abi.EXCHANGE_RCCL, want=None):
    torch = _torch()
    C, S = x.shape
    want_out, want_fl = want if want is not None else _plain(cfg, x)
    with ShardedSyllableDetectorBank(cfg, C, devices, exchange=exchange) as bank:
        blocks = bank.scatter(x)
        for rep in range(2):                                   # (twice: the exchange buffers are reused)
            outs, fls, alls = bank.run(blocks, S)
            bank.synchronize()
        for i, s in enumerate(bank.shards):
            _, _, e0, cnt = bank.ranges(i, S)
            rows = slice(s.first_channel, s.first_channel + s.channels)
            assert np.array_equal(outs[i].cpu().numpy(), want_out[rows, e0:e0 + cnt]), "outputs of shard %d" % i
            assert np.array_equal(fls[i].cpu().numpy(), want_fl[rows, e0:e0 + cnt]), "flags of shard %d" % i
            assert np.array_equal(alls[i].cpu().numpy(), want_fl), "gathered flags on shard %d's device" % i
        out_h, fl_h = bank.runHost(x)
        assert np.array_equal(out_h, want_out) and np.array_equal(fl_h, want_fl), "host-pointer call"
        return bank.rcclRanks


def test_one_rank_rccl_bank_equals_the_plain_bank_on_the_syllable_case(oracle_lib):
    """devices = {0}: the library's own RCCL communicator (one rank) carries the exchange; results are the plain bank's and
    the oracle's."""
    cfg, x, gold = util.load_case("case_sample_syllables")
    x = np.stack([x, x[::-1].copy(), 0.5 * x]).astype(np.float32)
    want = _plain(cfg, x)
    assert _check_bank(cfg, x, [0], want=want) == 1
    o = util.oracle_for(cfg)
    _, wfl, w64 = o.run(x[0], po.F64)
    util.assert_outputs_close(want[0][0], w64)
    util.assert_flags_exact(want[1][0], w64, cfg.thresholds, cfg.rule)
    assert want[1][0].sum() > 0


@pytest.mark.parametrize("channels,shards", [(5, 2), (8, 4), (7, 3), (2, 3), (1, 4), (3, 8)])
def test_shard_tables_on_one_device(channels, shards):
    """Ragged channel blocks (5 / 2, 7 / 3), equal ones (8 / 4) and the time-axis split with its halo (2 / 3: one channel cut
    in two, one whole; 1 / 4; 3 / 8), all shards on device 0 with the copy exchange."""
    cfg = util.sample_net()
    x = synth.channels(channels, 30000 + 137 * channels, first=40)
    assert _check_bank(cfg, x, [0] * shards) == 0


def test_time_sharded_bank_on_other_engines():
    """The halo is geometry, not a property of one kernel: 1024-point frames on the block-transform kernel and a chain the
    generic engine takes, cut along time."""
    for cfg in (nets.config3(), nets.variant(util.sample_net(), spectrogramScaling="db")):
        hop = cfg.windowLength - cfg.windowOverlap
        x = synth.channels(2, cfg.windowLength + 700 * hop + 5, first=7, fs=cfg.samplingRate)
        _check_bank(cfg, x, [0] * 5)


def test_time_sharded_bank_on_a_pass_scaled_kernel_stays_inside_the_bar(oracle_lib, monkeypatch):
    """Bit for bit holds for the kernels that scale per frame or per hop-aligned block.  The pass-scaled fused kernels (here the
    register-resident-basis kernel under SYLDET_FUSED_NOFOLD) pick a block scale per pass, so a time shard -- whose passes start
    elsewhere -- agrees with the whole run to a few 1e-7 only: held to the oracle at the 1e-5 bar like any other run."""
    torch = _torch()
    monkeypatch.setenv("SYLDET_FUSED_NOFOLD", "1")
    cfg, x, gold = util.load_case("case_sample_syllables")
    x = x[None, :90000].astype(np.float32)
    S = x.shape[1]
    o = util.oracle_for(cfg)
    _, wfl, w64 = o.run(x[0], po.F64)
    with ShardedSyllableDetectorBank(cfg, 1, [0, 0, 0]) as bank:
        outs, fls, alls = bank.run(bank.scatter(x), S)
        bank.synchronize()
        for i in range(3):
            _, _, e0, cnt = bank.ranges(i, S)
            util.assert_outputs_close(outs[i].cpu().numpy()[0], w64[e0:e0 + cnt])
            util.assert_flags_exact(fls[i].cpu().numpy()[0], w64[e0:e0 + cnt], cfg.thresholds, cfg.rule)
            util.assert_flags_exact(alls[i].cpu().numpy()[0], w64, cfg.thresholds, cfg.rule)


@pytest.mark.parametrize("devices,exchange", [([0], _abi.EXCHANGE_RCCL), ([0, 0, 0], _abi.EXCHANGE_PEER_COPY)])
def test_exchange_of_one_batch_runs_beside_the_kernels_of_the_next(devices, exchange):
    """Batches queued back to back with no synchronisation between them (the exchange of batch i is on streams of its own, two
    sets of buffers in turn, and batch i + 1's kernels do not wait for it): five batches of different audio, every batch's own
    results and gathered flags the plain bank's; then the same with ONE set of result tensors reused by every batch, where the
    last batch's must be what is left."""
    torch = _torch()
    cfg = util.sample_net()
    C, S = 5, 52000
    xs = [synth.channels(C, S, first=100 + 10 * k) for k in range(5)]
    want = [_plain(cfg, x) for x in xs]
    with ShardedSyllableDetectorBank(cfg, C, devices, exchange=exchange) as bank:
        blocks = [bank.scatter(x) for x in xs]
        got = [bank.run(b, S) for b in blocks]                    # (nothing waits in between)
        bank.synchronize()
        for k, (outs, fls, alls) in enumerate(got):
            for i, s in enumerate(bank.shards):
                _, _, e0, cnt = bank.ranges(i, S)
                rows = slice(s.first_channel, s.first_channel + s.channels)
                assert np.array_equal(outs[i].cpu().numpy(), want[k][0][rows, e0:e0 + cnt]), "batch %d outputs of shard %d" % (k, i)
                assert np.array_equal(alls[i].cpu().numpy(), want[k][1]), "batch %d gathered flags on shard %d" % (k, i)
        outs, fls, alls = got[0]
        for b in blocks:
            bank.run(b, S, outputs=outs, flags=fls, flags_all=alls)
        bank.synchronize()
        for i in range(len(bank.shards)):
            assert np.array_equal(alls[i].cpu().numpy(), want[-1][1])
        # the streams a caller can order its own work on
        for i in range(len(bank.shards)):
            compute, exch = bank.streams(i)
            assert compute and exch and compute != exch


def test_launcher_threads_and_the_inline_form_give_the_same_bits(monkeypatch):
    """A bank of several shards queues a batch on one persistent launcher thread per shard (round 6); SYLDET_SHARDED_INLINE=1
    keeps round 5's form, shard after shard on the caller's thread.  Both against the plain bank, bit for bit, ragged blocks and
    the time-axis split; a stream of batches through prepared calls (the benchmark's loop)."""
    torch = _torch()
    cfg = util.sample_net()
    for channels, shards in ((7, 3), (2, 5)):
        x = synth.channels(channels, 36000 + 91 * channels, first=21)
        want = _plain(cfg, x)
        for inline in (False, True):
            if inline:
                monkeypatch.setenv("SYLDET_SHARDED_INLINE", "1")
            else:
                monkeypatch.delenv("SYLDET_SHARDED_INLINE", raising=False)
            with ShardedSyllableDetectorBank(cfg, channels, [0] * shards) as bank:
                assert bank.launcherThreads == (0 if inline else shards)
            assert _check_bank(cfg, x, [0] * shards, want=want) == 0
    monkeypatch.delenv("SYLDET_SHARDED_INLINE", raising=False)
    with ShardedSyllableDetectorBank(cfg, 1, [0]) as bank:
        assert bank.launcherThreads == 0                        # one shard: the caller's thread
    # prepared calls: 40 batches back to back, two flags sets in turn, nothing waits in between
    C, S = 6, 50000
    x = synth.channels(C, S, first=33)
    want_out, want_fl = _plain(cfg, x)
    with ShardedSyllableDetectorBank(cfg, C, [0, 0, 0, 0]) as bank:
        bank.connect()
        blocks = bank.scatter(x)
        outs, fls, alls = bank.run(blocks, S)
        fls_b = [torch.empty_like(f) for f in fls]
        calls = (bank.prepare(blocks, S, outs, fls, alls), bank.prepare(blocks, S, outs, fls_b, alls))
        for k in range(40):
            calls[k & 1]()
        bank.synchronize()
        for i, s in enumerate(bank.shards):
            rows = slice(s.first_channel, s.first_channel + s.channels)
            assert np.array_equal(outs[i].cpu().numpy(), want_out[rows]) and np.array_equal(fls_b[i].cpu().numpy(), want_fl[rows])
            assert np.array_equal(fls[i].cpu().numpy(), want_fl[rows]) and np.array_equal(alls[i].cpu().numpy(), want_fl)
        with pytest.raises(ValueError):
            bank.prepare(blocks, S, outs, [f[:, 1:] for f in fls], alls)


def test_a_host_without_rccl_falls_back_in_the_same_process(monkeypatch):
    """SYLDET_RCCL_FAIL=1 makes ncclCommInitAll's step fail: connect() raises (and so would the first gathering batch), the
    same process makes the bank again with the copy exchange, results are the plain bank's."""
    from syllable_detector_swift_amd.config import SyllableDetectorError as SyldetError
    cfg = util.sample_net()
    x = synth.channels(3, 30000, first=8)
    want = _plain(cfg, x)
    monkeypatch.setenv("SYLDET_RCCL_FAIL", "1")
    with ShardedSyllableDetectorBank(cfg, 3, [0]) as bank:
        with pytest.raises(SyldetError):
            bank.connect()
        with pytest.raises(SyldetError):
            bank.run(bank.scatter(x), 30000)
    assert _check_bank(cfg, x, [0], exchange=_abi.EXCHANGE_PEER_COPY, want=want) == 0
    monkeypatch.delenv("SYLDET_RCCL_FAIL")
    assert _check_bank(cfg, x, [0], want=want) == 1


def _bench(args, env=None, launcher=None, timeout=600):
    """bench.py as the driver runs it (a child process; its one stdout line parsed)."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e.update(env or {})
    cmd = (launcher or [sys.executable]) + [os.path.join(root, "bench.py")] + args
    r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=timeout, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_multi_gpu_branch_rehearsed_on_one_device():
    """`bench.py --devices 0,0,0,0,0,0,0,0`: the whole single-process N > 1 branch -- eight shards, launcher threads, the exchange
    (by copies: RCCL refuses a device listed twice), per-device timings, gathered flags on every device, the oracle check of the
    first and last shard on the timed step and on planted syllables, JSON emission with `summary` as its last key."""
    line = _bench(["--devices", "0,0,0,0,0,0,0,0", "--channels", "4", "--log2-samples", "16", "--steps", "4", "--warmup", "1", "--preroll", "3"])
    assert line["shards"] == 8 and line["n_gpus"] == 1 and line["launcher"] == "single-process" and line["launcher_threads"] == 8
    assert line["exchange"] == "peer_copy" and line["rccl_ranks"] == 0 and line["rccl_error"] is None
    assert line["config"]["channels_per_gpu"] == [4] * 8 and line["gathered_flags_shape"][0] == 32
    assert line["gathered_flags_identical_on_every_device"] is True and line["verified"] is True
    assert len(line["roofline"]["kernel_ms_per_device"]) == 8 and line["roofline"]["frac"] > 0
    assert line["verify_planted"]["detections"] > 0 and set(line["verify"]) == {"shard0", "shard7"}
    assert list(line)[-1] == "summary" and line["summary"]["exchange"] == "peer_copy"
    assert line["value"] == pytest.approx(32 * line["config"]["frames_per_channel"] / (line["ms_per_step"] * 1e-3), rel=1e-6)


def test_bench_single_process_falls_back_when_rccl_does_not_come_up():
    """One shard on device 0 is one RCCL rank; with SYLDET_RCCL_FAIL=1 the communicator step fails, and bench.py makes the bank
    again in the same process with the copy exchange and says so in the line."""
    args = ["--devices", "0", "--channels", "6", "--log2-samples", "16", "--steps", "3", "--warmup", "1", "--preroll", "2"]
    ok = _bench(args)
    assert ok["exchange"] == "rccl" and ok["rccl_ranks"] == 1 and ok["rccl_error"] is None and ok["verified"] is True
    fb = _bench(args, env={"SYLDET_RCCL_FAIL": "1"})
    assert fb["exchange"] == "peer_copy" and fb["rccl_ranks"] == 0 and "SYLDET_RCCL_FAIL" in fb["rccl_error"] and fb["verified"] is True
    assert fb["verify_planted"]["detections"] == ok["verify_planted"]["detections"] > 0


def test_bench_process_per_gpu_branch_rehearsed_with_two_ranks_on_one_device():
    """The launcher the driver uses (torch.distributed.run, one process per rank), two ranks on the box's one GPU: the gloo
    control plane, the agreement on the exchange, the host-staged fallback (SYLDET_BENCH_NO_RCCL: RCCL refuses two ranks on one
    device anyway), barriers, max-over-ranks timing, rank 0's line."""
    import sys
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29641"]
    line = _bench(["--gpus", "2", "--channels", "6", "--log2-samples", "16", "--steps", "3", "--warmup", "1", "--preroll", "2"],
                  env={"SYLDET_BENCH_NO_RCCL": "1"}, launcher=launcher)
    assert line["n_gpus"] == 2 and line["exchange"] == "gloo_host" and line["rccl_ranks"] == 0 and "SYLDET_BENCH_NO_RCCL" in line["rccl_error"]
    assert line["config"]["total_channels"] == 12 and line["gathered_flags_shape"][0] == 12 and line["verified"] is True
    assert line["verify_planted"]["detections"] > 0 and list(line)[-1] == "summary"
    assert line["value"] == pytest.approx(12 * line["config"]["frames_per_channel"] / (line["ms_per_step"] * 1e-3), rel=1e-6)


def test_bench_one_rank_through_the_rccl_group_of_the_process_per_gpu_launcher():
    """`bench.py --force-gather` under torch.distributed.run with ONE rank: the branch the driver's N > 1 runs take when RCCL comes
    up -- gloo control plane, the RCCL group of its own with its probe, dist.PipelinedFlagGather on a side stream -- on the shard shape's
    smaller cousin; the line says which exchange carried the flags."""
    import sys
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29643"]
    line = _bench(["--gpus", "1", "--force-gather", "--channels", "8", "--log2-samples", "17", "--steps", "3", "--warmup", "1", "--preroll", "2", "--no-cpu-baseline"],
                  launcher=launcher)
    assert line["n_gpus"] == 1 and line["exchange"] == "rccl" and line["rccl_ranks"] == 1 and line["rccl_error"] is None
    assert line["gathered_flags_shape"][0] == 8 and line["verified"] is True and line["verify_planted"]["detections"] > 0
    assert "side stream" in line["config"]["sharding"] and list(line)[-1] == "summary"


# More than one GPU on the box: the multi-rank RCCL exchange itself (ncclCommInitAll over distinct devices, the grouped
# all-gather issued from one thread, the ragged padded_rows layout).  Defined only where it can run: the builder's and the
# driver's test boxes have ONE GPU, where this has never executed (README says so).
if _torch().cuda.device_count() >= 2:
    @pytest.mark.parametrize("channels,n_dev", [(5, 2), (7, 3), (1, 2), (3, 2)])
    def test_multi_rank_rccl_exchange_on_distinct_devices(channels, n_dev):
        torch = _torch()
        if torch.cuda.device_count() < n_dev:
            n_dev = torch.cuda.device_count()
        cfg = util.sample_net()
        x = synth.channels(channels, 40000 + 131 * channels, first=60)
        assert _check_bank(cfg, x, list(range(n_dev)), exchange=_abi.EXCHANGE_RCCL) == n_dev
        _check_bank(cfg, x, list(range(n_dev)), exchange=_abi.EXCHANGE_PEER_COPY)


def test_configs3_shard_shape_through_the_one_rank_bank():
    """BASELINE configs[3]'s per-GPU shape (512 channels x 2^21 samples) through the sharded handle with devices = {0}:
    bit-identical to the plain bank, gathered flags through the library's RCCL group."""
    torch = _torch()
    cfg = util.sample_net()
    C, S = 512, 1 << 21
    dev = torch.device("cuda", 0)
    x = synth.channels_on_device(C, S, dev)
    with SyllableDetector(cfg, channels=C, device=0) as det:
        want_out, want_fl = det.run(x)
        torch.cuda.synchronize()
    with ShardedSyllableDetectorBank(cfg, C, [0]) as bank:
        outs, fls, alls = bank.run([x], S)
        bank.synchronize()
        assert bank.rcclRanks == 1
        assert torch.equal(outs[0], want_out) and torch.equal(fls[0], want_fl) and torch.equal(alls[0], want_fl)


def test_c_program_over_the_sharded_abi(tmp_path):
    """tests/c/sharded_bank.c (strict C99): plain bank vs the sharded handle, host and device calls, identical bits -- as one
    RCCL rank and as three shards of two channels (time-axis split) with the copy exchange."""
    exe = os.path.join(os.path.dirname(_abi.LIB_PATH), "sharded_bank")
    if not os.path.exists(exe):
        pytest.skip("sharded_bank not built (run __graft_entry__.build())")
    cfg, x, gold = util.load_case("case_sample_syllables")
    x = np.stack([x[:60000], x[20000:80000]]).astype(np.float32)
    (tmp_path / "net.txt").write_text(cfg.toText())
    x.tofile(str(tmp_path / "x.f32"))
    for devices, ranks in (("0", 1), ("0,0,0", 0)):
        r = subprocess.run([exe, str(tmp_path / "net.txt"), str(tmp_path / "x.f32"), "2", devices], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        E, det, rk, word = r.stdout.strip().splitlines()[-1].split()      # (RCCL prints a version banner to stdout with its first communicator)
        assert word == "identical" and int(rk) == ranks and int(det) > 0
        assert int(E) == (60000 - 256) // 132 + 1 - 9


# ---- the host-pointer batch call as a pipeline along time -------------------------------------------------------------

@pytest.mark.parametrize("kind", ["fold", "generic", "config3"])
def test_pipelined_host_call_equals_the_device_call_across_stage_seams(kind, monkeypatch):
    """syldet_run cut into many stages (SYLDET_HOST_CHUNK_BYTES forces seams every few hundred evaluations) == the one-shot
    device call, bit for bit, where the kernels scale per frame or per hop-aligned block; pageable and pinned buffers."""
    torch = _torch()
    cfg, engine = {"fold": (util.sample_net(), 0), "generic": (util.sample_net(), 1), "config3": (nets.config3(), 0)}[kind]
    hop = cfg.windowLength - cfg.windowOverlap
    C = 3
    S = cfg.windowLength + 2500 * hop + 77
    x = synth.channels(C, S, first=3, fs=cfg.samplingRate)
    x[1] *= np.float32(1e-3)
    want_out, want_fl = _plain(cfg, x, engine)
    E = want_out.shape[1]
    for chunk in (C * 4 * hop * 300, C * 4 * hop * 997, 1 << 30):
        monkeypatch.setenv("SYLDET_HOST_CHUNK_BYTES", str(chunk))
        with SyllableDetector(cfg, channels=C, device=0, engine=engine) as det:
            out, fl = det.runHost(x)
            assert np.array_equal(out, want_out) and np.array_equal(fl, want_fl), "pageable buffers, stages of %d bytes" % chunk
            px, po_, pf = PinnedArray(x.shape, np.float32), PinnedArray(want_out.shape, np.float32), PinnedArray(want_fl.shape, np.uint8)
            px.array[:] = x
            po_.array[:] = -1
            pf.array[:] = 7
            det.runHost(px.array, outputs=po_.array, flags=pf.array)
            assert np.array_equal(po_.array, want_out) and np.array_equal(pf.array, want_fl), "pinned buffers, stages of %d bytes" % chunk
            # a strided source (rows of a longer recording) and results only partly asked for
            wide = np.zeros((C, S + 50), np.float32)
            wide[:, :S] = x
            out2 = np.zeros_like(want_out)
            _abi.lib.syldet_run(det._h, wide.ctypes.data_as(_abi.c_float_p), S, S + 50, out2.ctypes.data_as(_abi.c_float_p), None)
            assert np.array_equal(out2, want_out)
            for p in (px, po_, pf):
                p.free()
    assert E > 2000


def test_pipelined_host_call_on_a_pass_scaled_kernel_stays_inside_the_bar(oracle_lib, monkeypatch):
    """The pass-scaled kernels (here the register-resident-basis kernel under SYLDET_FUSED_NOFOLD) agree between tilings to a
    few 1e-7, not to the bit: a staged run is held to the oracle like any other."""
    monkeypatch.setenv("SYLDET_FUSED_NOFOLD", "1")
    cfg, x, gold = util.load_case("case_sample_syllables")
    x = x[:90000]
    monkeypatch.setenv("SYLDET_HOST_CHUNK_BYTES", str(4 * 132 * 150))
    with SyllableDetector(cfg, channels=1, device=0) as det:
        out, fl = det.runHost(x[None, :])
    o = util.oracle_for(cfg)
    _, wfl, w64 = o.run(x, po.F64)
    util.assert_outputs_close(out[0], w64)
    util.assert_flags_exact(fl[0], w64, cfg.thresholds, cfg.rule)


def test_sharded_bank_edge_cases():
    """Fewer evaluations than time shards (a shard with nothing to do), recordings too short for one evaluation, results only
    partly asked for, the library's own flag buffers when the caller keeps none."""
    import ctypes as C
    torch = _torch()
    cfg = util.sample_net()
    hop, W, T = 132, 256, cfg.timeRange
    # 3 evaluations over 4 time shards of one channel: one shard computes nothing
    S = W + (T - 1 + 2) * hop + 5
    x = synth.channels(1, S, first=9)
    want_out, want_fl = _plain(cfg, x)
    assert want_out.shape[1] == 3
    with ShardedSyllableDetectorBank(cfg, 1, [0, 0, 0, 0]) as bank:
        counts = [bank.ranges(i, S)[3] for i in range(4)]
        assert sorted(counts) == [0, 1, 1, 1]
        blocks = bank.scatter(x)
        outs, fls, alls = bank.run(blocks, S)
        bank.synchronize()
        for i in range(4):
            _, _, e0, cnt = bank.ranges(i, S)
            assert np.array_equal(outs[i].cpu().numpy(), want_out[:, e0:e0 + cnt])
            assert np.array_equal(alls[i].cpu().numpy(), want_fl)
        out_h, fl_h = bank.runHost(x)
        assert np.array_equal(out_h, want_out) and np.array_equal(fl_h, want_fl)
        # too short for one evaluation: nothing to do, nothing touched
        short = synth.channels(1, W + (T - 2) * hop, first=1)
        o2, f2 = bank.runHost(short)
        assert o2.shape == (1, 0, 1) and f2.shape == (1, 0)
    # flags only / outputs only through the host call; the device call without per-shard flag buffers (the exchange uses the library's own)
    x = synth.channels(5, 40000, first=2)
    want_out, want_fl = _plain(cfg, x)
    with ShardedSyllableDetectorBank(cfg, 5, [0, 0]) as bank:
        E = want_out.shape[1]
        fl = np.zeros((5, E), np.uint8)
        assert _abi.lib.syldet_sharded_run(bank._h, x.ctypes.data_as(_abi.c_float_p), x.shape[1], x.shape[1], None, fl.ctypes.data_as(_abi.c_uint8_p)) == 0
        assert np.array_equal(fl, want_fl)
        out = np.zeros_like(want_out)
        assert _abi.lib.syldet_sharded_run(bank._h, x.ctypes.data_as(_abi.c_float_p), x.shape[1], x.shape[1], out.ctypes.data_as(_abi.c_float_p), None) == 0
        assert np.array_equal(out, want_out)
        blocks = bank.scatter(x)
        alls = [torch.empty((5, E), dtype=torch.uint8, device="cuda") for _ in range(2)]
        arr = lambda ts: (C.c_void_p * 2)(*[t.data_ptr() for t in ts])
        strides = (C.c_int64 * 2)(*[int(b.stride(0)) for b in blocks])
        assert _abi.lib.syldet_sharded_run_device(bank._h, arr(blocks), x.shape[1], strides, None, None, arr(alls)) == 0
        bank.synchronize()
        assert all(np.array_equal(a.cpu().numpy(), want_fl) for a in alls)
        # a misaligned gathered-flags pointer is refused, not written through
        bad = (C.c_void_p * 2)(alls[0].data_ptr() + 1, alls[1].data_ptr())
        assert _abi.lib.syldet_sharded_run_device(bank._h, arr(blocks), x.shape[1], strides, None, None, bad) == _abi.ERR_INVALID_ARGUMENT


@pytest.mark.parametrize("switch", ["SYLDET_FUSED_NOFOLD2", "SYLDET_WIDE_WG16"])
def test_ab_switches_give_the_same_results(oracle_lib, monkeypatch, switch):
    """The forms kept behind switches for A/B runs (the once-folded fold kernel; the wide GEMM as one 16-wave workgroup a CU)
    stay correct: same oracle, same bars."""
    torch = _torch()
    if switch == "SYLDET_FUSED_NOFOLD2":
        cfg, x, gold = util.load_case("case_sample_syllables")
        x = x[:90000]
        engine, tol = 0, 1e-5
    else:
        cfg = nets.wide_mlp(util.sample_net())
        x = synth.channel(70000, 4)
        engine, tol = _abi.ENGINE_WIDE_BF16, 1e-2
    o = util.oracle_for(cfg)
    _, wfl, w64 = o.run(x, po.F64)
    outs = []
    for on in (False, True):
        if on:
            monkeypatch.setenv(switch, "1")
        with SyllableDetector(cfg, channels=1, device=0, engine=engine) as det:
            out, fl = det.run(torch.from_numpy(x[None, :]).cuda())
            torch.cuda.synchronize()
            out, fl = out.cpu().numpy()[0], fl.cpu().numpy()[0]
        util.assert_outputs_close(out, w64, tol)
        util.assert_flags_exact(fl, w64, cfg.thresholds, cfg.rule, tol)
        outs.append(out)
    assert np.abs(outs[0] - outs[1]).max() <= 2 * tol
