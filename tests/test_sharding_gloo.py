"""CPU test of the N>1 path: world_size-2 gloo processes run the sharding + MIN(=AND) reduce host
logic that bench.py / the engine use with RCCL on the GPUs."""
import os
import socket

import numpy as np
import pytest

from sylow_amd import sharding


def test_shard_bounds_cover_exactly():
    for n in (0, 1, 7, 8, 1 << 20, (1 << 20) + 5):
        for world in (1, 2, 3, 4, 8):
            spans = [sharding.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_bounds(10, 2, 2)


def _worker(rank, world, port, bad_index, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1001
    truth = np.ones(n, dtype=np.uint8)
    if bad_index is not None:
        truth[bad_index] = 0
    msgs = [bytes([i % 256]) for i in range(n)]
    pk = np.arange(n * 16, dtype=np.uint64).reshape(n, 16)
    sig = np.arange(n * 8, dtype=np.uint64).reshape(n, 8)

    def fake_verify(pk_s, msgs_s, sig_s):          # stands in for engine.bls_verify on this rank's GPU
        lo = int(pk_s[0, 0]) // 16 if len(pk_s) else 0
        assert len(pk_s) == len(msgs_s) == len(sig_s)
        return truth[lo:lo + len(msgs_s)]

    flags, (lo, hi), ok = sharding.verify_sharded(fake_verify, pk, msgs, sig, dist)
    full = sharding.gather_flags(flags, n, dist)
    # the reduce bench.py uses on its device flag word, and its MAX-over-ranks clock
    import torch
    word = sharding.and_reduce_(torch.tensor([int(flags.all())], dtype=torch.int32), dist)
    slowest = sharding.max_over_ranks(1.0 + rank, dist)
    assert int(word.item()) == int(ok) and slowest == 2.0 and sharding.collective_device(dist).type == "cpu"
    q.put((rank, lo, hi, int(ok), bool(np.array_equal(full, truth))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bad_index", [None, 3, 900])
def test_two_rank_and_reduce(bad_index):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bad_index, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(0, 501), (501, 1001)]
    expect = 1 if bad_index is None else 0
    assert all(r[3] == expect for r in res)          # every rank sees the global AND, wherever the bad flag lives
    assert all(r[4] for r in res)                    # and the gathered flag vector is the planted pattern
