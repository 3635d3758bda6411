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
