"""N > 1 path on CPU: channel sharding + the single flag gather, world_size 2 over gloo.
The per-rank flags come from the CPU oracle (this is test infrastructure; the product computes them
with the HIP engine), so the test pins the sharding arithmetic and the collective's layout."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pyoracle as po
import util
from syllable_detector_swift_amd import synth
from syllable_detector_swift_amd.dist import gather_flags, shard_channels


def test_shard_channels_partition():
    for total in (1, 2, 7, 8, 64, 4096, 4099):
        for world in (1, 2, 3, 8):
            got = [shard_channels(total, world, r) for r in range(world)]
            assert sum(c for _, c in got) == total
            nxt = 0
            for first, count in got:
                assert first == nxt and count in (total // world, total // world + 1)
                nxt += count
    assert shard_channels(4096, 8, 3) == (1536, 512)          # BASELINE config 4: 512 channels per GPU
    with pytest.raises(ValueError):
        shard_channels(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total, S, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = util.sample_net()
        o = util.oracle_for(cfg)
        first, count = shard_channels(total, world, rank)
        tpl = util.template()
        local = np.stack([o.run(synth.syllable_channel(S, tpl, seed=300 + first + c), po.F64)[1] for c in range(count)])
        full = gather_flags(torch.from_numpy(local), total)
        q.put((rank, full.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [4, 5])          # equal shards / ragged shards
def test_flag_gather_world_size_2(oracle_lib, total):
    S = 20000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, S, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = util.sample_net()
    o = util.oracle_for(cfg)
    want = np.stack([o.run(synth.syllable_channel(S, util.template(), seed=300 + c), po.F64)[1] for c in range(total)])
    assert want.sum() > 0
    for rank in (0, 1):
        assert np.array_equal(got[rank], want)


# ---- fewer channels than ranks: the time axis is sharded, with a halo ----------------------------------------
def test_shard_plane_and_evaluation_ranges_partition():
    from syllable_detector_swift_amd.dist import shard_evaluations, shard_plane, time_shard_samples
    for total in (1, 2, 3, 5, 8):
        for world in (1, 2, 3, 8):
            plan = [shard_plane(total, world, r) for r in range(world)]
            if total >= world:
                assert [(f, c) for f, c, _, _ in plan] == [shard_channels(total, world, r) for r in range(world)]
                assert all(p == (f, c, 0, 1) for p, (f, c, _, _) in zip(plan, plan))
                continue
            # every channel is covered by parts 0 .. parts-1 exactly once, in rank order
            seen = {}
            for ch, count, part, parts in plan:
                assert count == 1 and parts in (world // total, world // total + 1)
                seen.setdefault(ch, []).append((part, parts))
            assert sorted(seen) == list(range(total))
            for ch, lst in seen.items():
                assert [p for p, _ in lst] == list(range(lst[0][1]))
    assert shard_plane(1, 8, 5) == (0, 1, 5, 8)
    assert shard_plane(3, 8, 7) == (2, 1, 1, 2)                       # channels 0 and 1 by three ranks each, channel 2 by two
    for E in (0, 1, 7, 100, 8189):
        for parts in (1, 2, 3, 8):
            spans = [shard_evaluations(E, parts, p) for p in range(parts)]
            assert sum(n for _, n in spans) == E
            nxt = 0
            for e0, n in spans:
                assert e0 == nxt and n in (E // parts, E // parts + 1)
                nxt += n
    # sample.txt geometry: hop 132, no gap, 256-sample frames, 8 columns: neighbours overlap by (T - 1) hop + W - hop
    hop, gap, W, T = 132, 0, 256, 8
    a0, a1 = time_shard_samples(hop, gap, W, T, 0, 50)
    b0, b1 = time_shard_samples(hop, gap, W, T, 50, 50)
    assert (a0, b0) == (0, 50 * hop) and a1 - b0 == (T - 1) * hop + W - hop
    assert (a1 - a0 - gap - W) // hop + 1 - T + 1 == 50               # a run over the slice has exactly its 50 evaluations
    with pytest.raises(ValueError):
        shard_plane(0, 2, 0)


def _time_worker(rank, world, port, total, S, q):
    from syllable_detector_swift_amd.dist import gather_time_shards, shard_evaluations, shard_plane, time_shard_samples
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = util.sample_net()
        o = util.oracle_for(cfg)
        g = cfg.geometry()
        ch, count, part, parts = shard_plane(total, world, rank)
        x = synth.syllable_channel(S, util.template(), seed=300 + ch)
        E = o.count_evals(S)
        e0, n = shard_evaluations(E, parts, part)
        s0, s1 = time_shard_samples(g.hop, g.gap, cfg.windowLength, cfg.timeRange, e0, n)
        out, fl, _ = o.run(x[s0:s1], po.F64)                     # this rank's slice only: its range plus the halo
        assert fl.shape[0] == n
        full = gather_time_shards(torch.from_numpy(np.ascontiguousarray(fl)).reshape(1, n), total, E)
        q.put((rank, full.numpy(), np.asarray(out), e0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 1), (3, 2)])            # one channel over two ranks / two channels over three
def test_time_sharded_flags_world_size_2_and_3(oracle_lib, world, total):
    S = 20011                                                    # (a ragged last range)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_time_worker, args=(r, world, port, total, S, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = util.sample_net()
    o = util.oracle_for(cfg)
    whole = [o.run(synth.syllable_channel(S, util.template(), seed=300 + c), po.F64) for c in range(total)]
    want = np.stack([w[1] for w in whole])
    assert want.sum() > 0
    from syllable_detector_swift_amd.dist import shard_plane
    for rank, full, out, e0 in got:
        assert np.array_equal(full, want)
        ch = shard_plane(total, world, rank)[0]
        # a slice's outputs are the whole run's, number for number: the halo is exactly what the last windows reach into
        assert np.array_equal(out, np.asarray(whole[ch][0])[e0: e0 + out.shape[0]])
