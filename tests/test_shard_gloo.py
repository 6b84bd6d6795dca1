"""world_size-2 gloo (CPU) test of the candidate sharding + energy all-gather."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from acousticswarms_speech_amd.shard import ShardedScorer, shard_bounds, shard_groups


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_score(_mix, offs):
    """Deterministic stand-in for SpotModel.shift_and_score: a function of the offsets only."""
    o = np.asarray(offs, dtype=np.float64)
    return np.stack([np.abs(o).sum(1) + 1.0, np.sqrt((o ** 2).sum(1) + 1.0)], axis=1)


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        offs = np.random.default_rng(3).integers(-100, 100, size=(n, 6)).astype(np.int32)
        sc = ShardedScorer(_fake_score)
        seen = []

        def local(mix, o):
            seen.append(len(o))
            return _fake_score(mix, o)
        sc.local_score = local
        full = sc.score(None, offs, device="cpu")
        q.put((rank, full, seen))
    finally:
        dist.destroy_process_group()


def _run(n, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res, key=lambda t: t[0])


def test_bounds_and_groups():
    assert shard_bounds(10, 4) == [0, 3, 6, 8, 10]
    assert shard_bounds(2, 4) == [0, 1, 2, 2, 2]
    owners = shard_groups([38, 12, 20, 31, 15, 22], 2)
    assert sorted(sum(owners, [])) == list(range(6))
    loads = [sum([38, 12, 20, 31, 15, 22][i] for i in o) for o in owners]
    assert abs(loads[0] - loads[1]) <= 12


def test_two_rank_all_gather_matches_single_process():
    for n in (7, 64, 1):           # ragged, even, fewer candidates than ranks
        offs = np.random.default_rng(3).integers(-100, 100, size=(n, 6)).astype(np.int32)
        want = _fake_score(None, offs)
        res = _run(n)
        b = shard_bounds(n, 2)
        for rank, full, seen in res:
            np.testing.assert_array_equal(full, want)          # every rank holds every energy
            assert seen == [b[rank + 1] - b[rank]]             # and scored only its own shard


def test_single_process_passthrough():
    offs = np.arange(30).reshape(5, 6)
    np.testing.assert_array_equal(ShardedScorer(_fake_score).score(None, offs), _fake_score(None, offs))
