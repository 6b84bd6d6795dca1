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


# ---------------------------------------------------------------- whole search, two ranks
def _pipeline_worker(rank, world, port, q):
    """Every rank runs MicArray's four stages on the same mixture through ShardedSpotModel
    (surrogate scorer, gloo): candidate evaluations are sharded, results must equal the
    single-process reference trace (fixture g10)."""
    import io
    from contextlib import redirect_stdout
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from acousticswarms_speech_amd.mic_array import MicArray
        from acousticswarms_speech_amd.shard import ShardedSpotModel
        from tests.golden.make_golden_search import ROI, scene_in_roi
        from tests.golden.surrogate import SurrogateSpot
        gdir = os.path.join(os.path.dirname(__file__), "golden")
        g7 = np.load(os.path.join(gdir, "g7_srp_map.npz"))
        mics, _spk, mix = scene_in_roi()
        with redirect_stdout(io.StringIO()):
            ma = MicArray(mics, Spk_Range=ROI)
            node = ma.SRP_node
            node.SRP_Map_WINDOW_new = lambda signal, window=36000: node.set_map(g7["srp_map"])
            inner = SurrogateSpot()
            spot = ShardedSpotModel(inner)
            mix_t = torch.from_numpy(mix)
            p1, _ = ma.Apply_SRP_PHAT(mix_t)
            p2 = ma.Spotform_Big_Patch(mix_t, p1, spot)
            kept = [int(np.flatnonzero([x is p for x in p1])[0]) for p in p2]
            pairs = ma.Spotform_Small_Patch_Parallel(mix_t, p2, spot)
            _audio, final, spot_times, _ = ma.Clustering_new(pairs)
        q.put((rank, kept, [p[3] for p in pairs], np.array([p[2] for p in pairs]),
               np.stack([p[0].center_pos() for p in pairs]), [p[3] for p in final],
               np.stack([p[0].center_pos() for p in final]), spot_times, sum(n for n, _s in inner.calls)))
    finally:
        dist.destroy_process_group()


def test_two_rank_search_matches_reference_trace():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g10_stage_trace.npz"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    evaluated = 0
    for rank, kept, names, power, centre, fnames, fcentre, spot_times, n_eval in res:
        assert kept == g["kept"].tolist()
        assert names == g["pair_names"].tolist()
        np.testing.assert_allclose(power, g["pair_power"], rtol=1e-5)
        np.testing.assert_allclose(centre, g["pair_center"], atol=1e-6)
        assert fnames == g["final_names"].tolist()
        np.testing.assert_allclose(fcentre, g["final_center"], atol=1e-6)
        assert spot_times == int(g["spot_times"])
        evaluated += n_eval
    # the two ranks together evaluated every candidate exactly once
    assert evaluated == int(sum(c[0] for c in g["calls"].tolist()))


# ---------------------------------------------------------------- batch of mixtures, two ranks
def _batch_setup():
    import io
    from contextlib import redirect_stdout
    from acousticswarms_speech_amd.joint import JointModel
    from tests.golden.make_golden_search import ROI, scene_in_roi
    from tests.golden.surrogate import SurrogateSpot
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    g7 = np.load(os.path.join(gdir, "g7_srp_map.npz"))
    mics, _spk, mix = scene_in_roi()
    jm = JointModel(SurrogateSpot())
    with redirect_stdout(io.StringIO()):
        jm.setup(mics, ROI)
    node = jm.Mic_processor.SRP_node
    node.SRP_Map_WINDOW_new = lambda signal, window=36000: node.set_map(g7["srp_map"])   # no GPU here
    mixes = [torch.from_numpy(mix * gain) for gain in (1.0, 0.6, 0.35)]
    return jm, mixes


def _batch_worker(rank, world, port, q):
    import io
    from contextlib import redirect_stdout
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from acousticswarms_speech_amd.shard import localize_batch
        jm, mixes = _batch_setup()
        with redirect_stdout(io.StringIO()):
            out = localize_batch(jm, mixes)
        q.put((rank, [(r["centres"], r["powers"], r["names"], r["spot_times"]) for r in out],
               len(jm.spot_model.calls)))
    finally:
        dist.destroy_process_group()


def test_two_rank_mixture_batch_matches_single_process():
    import io
    from contextlib import redirect_stdout
    from acousticswarms_speech_amd.shard import localize_batch
    jm, mixes = _batch_setup()
    with redirect_stdout(io.StringIO()):
        want = localize_batch(jm, mixes)                     # no process group: plain loop
    assert len(want) == 3 and len(want[0]["names"]) >= 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_batch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for _rank, out, _n in res:
        assert len(out) == 3
        for got, w in zip(out, want):
            np.testing.assert_array_equal(got[0], w["centres"])
            np.testing.assert_array_equal(got[1], w["powers"])
            assert got[2] == w["names"] and got[3] == w["spot_times"]
    # rank 0 ran two mixtures, rank 1 one (2 spot calls per mixture: coarse + fine)
    assert [n for _r, _o, n in res] == [4, 2]


# ---------------------------------------------------------------- ranks that disagree
def _diverging_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        offs = np.random.default_rng(3).integers(-100, 100, size=(9, 6)).astype(np.int32)
        if rank == 1:
            offs = offs.copy()
            offs[4, 2] += 1                       # one candidate differs on this rank
        try:
            ShardedScorer(_fake_score).score(None, offs, device="cpu")
            q.put((rank, "no error"))
        except RuntimeError as e:
            q.put((rank, str(e)))
    finally:
        dist.destroy_process_group()


def test_diverging_candidate_lists_raise_on_every_rank():
    """The fixed-shape energy exchange assumes identical candidate lists on all ranks; a
    divergence must be detected (fingerprint all-gather) instead of mis-assembling energies."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_diverging_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _rank, msg in res:
        assert "ranks disagree" in msg


def _deal_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from acousticswarms_speech_amd.shard import ShardedSpotModel

        class _Inner:                                     # only the plumbing of the wrapper is exercised
            pass
        sm = ShardedSpotModel(_Inner(), device="cpu")
        weights = [900, 120, 4000, 333, 50, 2100, 7]      # 1 cm grid points of seven coarse patches
        true_sizes = [31, 12, 38, 20, 12, 35, 12]         # what subdividing them would give
        owners = sm.deal_groups(weights)
        mine = owners[rank]
        sizes = sm.gather_sizes({g: true_sizes[g] for g in mine}, len(weights))
        bounds = [0]
        for n in sizes:
            bounds.append(bounds[-1] + n)
        local = np.concatenate([np.stack([np.full(true_sizes[g], g + 0.5), np.arange(true_sizes[g], dtype=np.float64)], 1)
                                for g in mine]) if mine else np.zeros((0, 2))
        full = sm.all_gather_groups(local, mine, bounds, owners=owners)
        q.put((rank, owners, sizes, full))
    finally:
        dist.destroy_process_group()


def test_weight_dealt_groups_sizes_and_energy_gather():
    """The fine stage of one rank per GPU: coarse patches dealt by a weight known without subdividing them
    (deal_groups), the sizes of the subdivisions exchanged afterwards (gather_sizes), and the energy all-gather
    assembled with that deal (all_gather_groups(owners=...)): every rank ends with the same full table in the
    global candidate order."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_deal_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, owners0, sizes0, full0), (_, owners1, sizes1, full1) = res
    assert owners0 == owners1 and sorted(owners0[0] + owners0[1]) == list(range(7))
    assert sizes0 == sizes1 == [31, 12, 38, 20, 12, 35, 12]
    np.testing.assert_array_equal(full0, full1)
    want = np.concatenate([np.stack([np.full(n, g + 0.5), np.arange(n, dtype=np.float64)], 1) for g, n in enumerate(sizes0)])
    np.testing.assert_array_equal(full0, want)
