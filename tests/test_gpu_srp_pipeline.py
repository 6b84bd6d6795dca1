"""GPU: SRP-PHAT map kernels against the reference's map (fixture g7) and the whole
search pipeline (device SRP map -> coarse -> fine -> clustering) against the reference's
stage trace (fixture g10, surrogate scorer).  Needs an MI355X."""
import io
import os
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

from tests.golden.make_golden_search import ROI, scene_in_roi
from tests.golden.surrogate import SurrogateSpot

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _log(msg):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "diag_srp.txt"), "a") as f:
        f.write(msg + "\n")
    print(msg)


@pytest.fixture(scope="module")
def mic_array():
    from acousticswarms_speech_amd.mic_array import MicArray
    mics, spk, mix = scene_in_roi()
    with redirect_stdout(io.StringIO()):
        ma = MicArray(mics, Spk_Range=ROI, device="cuda")
    return ma, mics, spk, mix


def test_srp_map_matches_reference(mic_array, golden):
    ma, _, _, mix = mic_array
    g7 = golden("g7_srp_map")
    node = ma.SRP_node
    node.reset()
    node.SRP_Map_WINDOW_new(mix, window=24000)
    got, ref = node.SRP_map.astype(np.float64), g7["srp_map"]
    err = np.abs(got - ref).max()
    _log(f"SRP map: max abs err {err:.3e} (map max {ref.max():.4f}), rel l2 {np.linalg.norm(got-ref)/np.linalg.norm(ref):.3e}")
    # fp32 DFT-GEMM + fp32 sincos against the reference's complex64 x float64 table
    np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-5)
    assert abs(node.MAX_POWER - float(g7["max_power"])) < 1e-4
    # odd length (not a multiple of 4) goes through the padded path
    node.SRP_Map_WINDOW_new(mix[:, :47999], window=24000)
    from oracle import srp_ref
    from acousticswarms_speech_amd.mic_array import FREQ_BINS, N_FFT
    want = srp_ref.srp_map(mix[:, :47999], 24000, N_FFT, FREQ_BINS, node.tau, node.omega)
    np.testing.assert_allclose(node.SRP_map, want, rtol=2e-4, atol=2e-5)


def test_pipeline_trace_with_device_srp(mic_array, golden):
    """JointModel.forward end to end: SRP map from the HIP kernels, surrogate scorer."""
    from acousticswarms_speech_amd.joint import JointModel
    ma, mics, spk, mix = mic_array
    g = golden("g10_stage_trace")
    spot = SurrogateSpot()
    jm = JointModel(spot, None, device="cuda")
    with redirect_stdout(io.StringIO()):
        jm.setup(mics, ROI)
        patches, audio_loc, audio, _, _, spot_times = jm.forward(torch.from_numpy(mix))
    _log(f"pipeline: calls {spot.calls}, final {[p[3] for p in patches]}, times {np.round(jm.times, 3)}")
    assert audio is None and audio_loc.shape == (len(patches), mix.shape[1])
    assert spot.calls == [tuple(c) for c in g["calls"].tolist()]
    assert [p[3] for p in patches] == g["final_names"].tolist()
    np.testing.assert_allclose(np.stack([p[0].center_pos() for p in patches]), g["final_center"], atol=1e-6)
    assert spot_times == int(g["spot_times"])
    assert all(t >= 0 for t in jm.times) and jm.times[0] > 0


def test_pipeline_with_hip_spot_model(mic_array):
    """Same pipeline with the real HIP spot model (seeded random weights): exercises the
    product path of every stage; a random network finds no talkers, so the reference's
    empty-result convention must come back."""
    from acousticswarms_speech_amd.config import FULL
    from acousticswarms_speech_amd.joint import JointModel
    from acousticswarms_speech_amd.spot import SpotModel
    from acousticswarms_speech_amd.weights import make_spot_state_dict
    ma, mics, spk, mix = mic_array
    spot = SpotModel(FULL, make_spot_state_dict(FULL, 5), batch_size=32).to("cuda")
    jm = JointModel(spot, None, device="cuda")
    with redirect_stdout(io.StringIO()):
        jm.setup(mics, ROI)
        out = jm.forward(torch.from_numpy(mix[:, :24000 * 2]))
    _log(f"pipeline(HIP spot, random weights): n_final={len(out[0])} times={np.round(jm.times, 3)}")
    assert len(out) == 6


def test_fine_stage_device_resident_equals_host_path(mic_array, golden):
    """Spotform_Small_Patch_Parallel with the HIP model: the device-resident path (energies +
    SI-SDR on the GPU, only cluster heads copied) must produce the same clusters as the
    reference-style host loops on the downloaded waveforms."""
    from acousticswarms_speech_amd.config import FULL
    from acousticswarms_speech_amd.spot import SpotModel
    from acousticswarms_speech_amd.weights import make_spot_state_dict
    ma, mics, spk, mix = mic_array
    g7 = golden("g7_srp_map")
    spot = SpotModel(FULL, make_spot_state_dict(FULL, 5), batch_size=32, precision="f16x3").to("cuda")

    class HostOnly:                       # hides the resident entry point -> reference host loops
        def __init__(self, m):
            self.m = m

        def shift_and_sep(self, *a, **k):
            return self.m.shift_and_sep(*a, **k)

    mix_t = torch.from_numpy(mix[:, :24000 * 2])
    node = ma.SRP_node
    with redirect_stdout(io.StringIO()):
        node.set_map(g7["srp_map"])
        coarse = node.local_source_adaptive()[:6]
        ma.Relative_Threshold = 0.0
        import copy
        a = ma.Spotform_Small_Patch_Parallel(mix_t, copy.deepcopy(coarse), spot)
        b = ma.Spotform_Small_Patch_Parallel(mix_t, copy.deepcopy(coarse), HostOnly(spot))
    _log(f"fine stage: resident {len(a)} pairs, host {len(b)} pairs")
    assert [p[3] for p in a] == [p[3] for p in b]
    np.testing.assert_allclose([p[2] for p in a], [p[2] for p in b], rtol=1e-5)
    for pa, pb in zip(a, b):
        np.testing.assert_array_equal(pa[4]["audio_offset"], pb[4]["audio_offset"])
        np.testing.assert_allclose(pa[4]["localization_offset"], pb[4]["localization_offset"], rtol=1e-5, atol=1e-6)
        # the two paths batch the candidates differently (chunks of coarse patches vs one call), and the
        # GEMM tile shape -- hence the GroupNorm partial-sum order -- follows the batch: 1e-6-level noise
        np.testing.assert_allclose(pa[1], pb[1], rtol=0, atol=5e-6 * max(1.0, np.abs(pb[1]).max()))


def test_search_hip_vs_oracle_north_star_tolerance(mic_array):
    """The north-star tolerance, end to end: the complete search (SRP-PHAT -> coarse -> fine ->
    clustering) run twice on the same mixture with the same seeded FULL weights -- once with the
    HIP spot model in its default f16x3 arithmetic, once with the CPU oracle behind the
    reference's shift_and_sep surface -- must pick the same talkers, place them within 2 cm
    and return waveforms whose SI-SDR against the oracle's is far beyond the 0.1 dB budget
    (>= 60 dB, i.e. a 1e-6 relative perturbation of any SI-SDR computed from them)."""
    from acousticswarms_speech_amd.config import FULL
    from acousticswarms_speech_amd.hostdsp import si_sdr
    from acousticswarms_speech_amd.joint import JointModel
    from acousticswarms_speech_amd.spot import SpotModel
    from acousticswarms_speech_amd.weights import make_spot_state_dict
    from oracle import spot_ref
    ma, mics, spk, mix = mic_array
    sd = make_spot_state_dict(FULL, 5)
    T = 24000
    mix_t = torch.from_numpy(mix[:, :T].copy())

    class OracleSpot:                      # the reference surface, computed by oracle/spot_ref.py on the CPU
        def __init__(self):
            self.calls = []

        def shift_and_sep(self, m, patch_list, Strict=0, save_input=False):
            self.calls.append((len(patch_list), Strict))
            if len(patch_list) == 0:
                return np.empty((0, m.shape[1]), dtype=np.float32)
            return spot_ref.shift_and_sep(sd, FULL, m, [p.sample_offset for p in patch_list], strict=Strict,
                                          batch_size=8)

    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    res = {}
    for name, spot in (("hip", SpotModel(FULL, sd, batch_size=32, precision="f16x3").to("cuda")),
                       ("oracle", OracleSpot())):
        jm = JointModel(spot, None, device="cuda")
        with redirect_stdout(io.StringIO()):
            jm.setup(mics, ROI)
            patches, audio_loc, _audio, _, _, spot_times = jm.forward(mix_t)
        res[name] = (patches, audio_loc, spot_times, list(jm.times))
    (ph, ah, nh, th), (po, ao, no, to) = res["hip"], res["oracle"]
    _log(f"north-star: talkers hip={len(ph)} oracle={len(po)}, spot calls {nh}/{no}, "
         f"stage s hip={np.round(th, 3)} oracle={np.round(to, 1)}")
    assert nh == no and len(ph) == len(po) and len(ph) >= 1
    assert [p[3] for p in ph] == [p[3] for p in po]                     # same candidates survive, same order
    err_cm = [100 * float(np.linalg.norm(a[0].center_pos() - b[0].center_pos())) for a, b in zip(ph, po)]
    sdr = [si_sdr(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)) for a, b in zip(ah, ao)]
    _log(f"north-star: position error cm {np.round(err_cm, 4)}, SI-SDR(hip, oracle) dB {np.round(sdr, 1)}")
    assert max(err_cm) <= 2.0
    assert min(sdr) >= 60.0


def test_global_clustering_device_equals_host(mic_array):
    """Clustering_new with the SI-SDR matrix and the segment-wise tensor computed on the GPU
    (asw_pair_sisdr / asw_segment_sisdr) makes the decisions of the reference's host loops."""
    from acousticswarms_speech_amd.config import FULL
    from acousticswarms_speech_amd.hostdsp import split_wav, split_wise_sisdr
    from acousticswarms_speech_amd.mic_array import MicArray
    from acousticswarms_speech_amd.spot import SpotModel
    from acousticswarms_speech_amd.weights import make_spot_state_dict
    mics, _spk, mix = scene_in_roi()
    spot = SpotModel(FULL, make_spot_state_dict(FULL, 5), batch_size=32, precision="f16x3").to("cuda")
    with redirect_stdout(io.StringIO()):
        ma = MicArray(mics, Spk_Range=ROI, device="cuda")
        mix_t = torch.from_numpy(mix[:, :24000].copy())
        p1, _ = ma.Apply_SRP_PHAT(mix_t)
        p2 = ma.Spotform_Big_Patch(mix_t, p1, spot)
        pairs = ma.Spotform_Small_Patch_Parallel(mix_t, p2, spot)
        assert ma._device_scorer is spot and len(pairs) > 10
        _a, final_dev, _n, _w = ma.Clustering_new(pairs)
        ma._device_scorer = None
        _a, final_host, _n, _w = ma.Clustering_new(pairs)
    assert [p[3] for p in final_dev] == [p[3] for p in final_host]
    # the kernel itself against the host statement, segment by segment
    waves = np.stack([np.asarray(p[1], dtype=np.float32) for p in pairs[:12]])
    segs = [split_wav(w) for w in waves]
    got, cnt = spot.segment_sisdr(torch.from_numpy(waves).cuda(), segs)
    worst = 0.0
    for i in range(len(waves)):
        for j in range(len(waves)):
            if cnt[i] == 0:
                continue
            want = np.array(split_wise_sisdr(waves[i], waves[j], segs[i]))
            worst = max(worst, float(np.abs(got[i, j, :cnt[i]] - want).max()))
            assert np.all(np.isnan(got[i, j, cnt[i]:]))
    _log(f"global clustering: {len(pairs)} candidates -> {len(final_dev)} talkers on both paths; "
         f"segment SI-SDR device vs host max |diff| {worst:.2e} dB")
    assert worst < 1e-3


# ---------------------------------------------------------------- two ranks, real HIP model
def _sharded_hip_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from acousticswarms_speech_amd.config import FULL
        from acousticswarms_speech_amd.mic_array import MicArray
        from acousticswarms_speech_amd.shard import ShardedSpotModel
        from acousticswarms_speech_amd.spot import SpotModel
        from acousticswarms_speech_amd.weights import make_spot_state_dict
        mics, _spk, mix = scene_in_roi()
        inner = SpotModel(FULL, make_spot_state_dict(FULL, 5), batch_size=32, precision="f16x3").to("cuda")
        spot = ShardedSpotModel(inner, device="cpu")          # both ranks share the one GPU here: gloo carries the collectives
        evaluated = []
        orig = inner.shift_and_sep_device

        def counting(mix_d, offs, *a, **kw):
            evaluated.append(int(offs.shape[0]))
            return orig(mix_d, offs, *a, **kw)
        inner.shift_and_sep_device = counting
        with redirect_stdout(io.StringIO()):
            ma = MicArray(mics, Spk_Range=ROI, device="cuda")
            mix_t = torch.from_numpy(mix[:, :24000].copy())
            p1, _ = ma.Apply_SRP_PHAT(mix_t)
            p2 = ma.Spotform_Big_Patch(mix_t, p1, spot)
            pairs = ma.Spotform_Small_Patch_Parallel(mix_t, p2, spot)
            _audio, final, spot_times, _ = ma.Clustering_new(pairs) if len(pairs) else ([], [], 0, None)
        q.put((rank, len(p2), [p[3] for p in pairs], np.array([p[2] for p in pairs]),
               np.array([p[0].center_pos() for p in pairs]), [p[3] for p in final], spot_times, sum(evaluated),
               None if not hasattr(ma, "fine_energies") else ma.fine_energies))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_search_with_hip_model(mic_array):
    """The multi-GPU search path with the real HIP model: two ranks (sharing the one GPU of this
    box, collectives over gloo) shard the coarse candidates and the fine-stage groups, all-gather
    energies and output tuples, and must end with the same output as one process."""
    import socket
    import torch.multiprocessing as mp
    from acousticswarms_speech_amd.config import FULL
    from acousticswarms_speech_amd.mic_array import MicArray
    from acousticswarms_speech_amd.spot import SpotModel
    from acousticswarms_speech_amd.weights import make_spot_state_dict
    mics, _spk, mix = scene_in_roi()
    spot = SpotModel(FULL, make_spot_state_dict(FULL, 5), batch_size=32, precision="f16x3").to("cuda")
    with redirect_stdout(io.StringIO()):
        ma = MicArray(mics, Spk_Range=ROI, device="cuda")
        mix_t = torch.from_numpy(mix[:, :24000].copy())
        p1, _ = ma.Apply_SRP_PHAT(mix_t)
        p2 = ma.Spotform_Big_Patch(mix_t, p1, spot)
        pairs = ma.Spotform_Small_Patch_Parallel(mix_t, p2, spot)
        _a, final, spot_times, _ = ma.Clustering_new(pairs)
    want = (len(p2), [p[3] for p in pairs], np.array([p[2] for p in pairs]),
            np.array([p[0].center_pos() for p in pairs]), [p[3] for p in final], spot_times)
    del spot
    torch.cuda.empty_cache()

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_hip_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    total_eval = 0
    for rank, n2, names, power, centre, fnames, st, n_eval, fine_en in res:
        assert n2 == want[0] and names == want[1] and fnames == want[4] and st == want[5]
        np.testing.assert_allclose(power, want[2], rtol=1e-5)
        np.testing.assert_allclose(centre, want[3], atol=1e-9)
        assert fine_en is not None and fine_en.ndim == 2 and fine_en.shape[1] == 2
        total_eval += n_eval
    np.testing.assert_array_equal(res[0][8], res[1][8])                  # every rank holds every fine energy
    _log(f"two-rank sharded search: pairs {len(want[1])}, final {len(want[4])}, candidates evaluated per rank "
         f"{[r[7] for r in res]} of {want[5]}")
    assert total_eval == want[5]                                        # each candidate evaluated exactly once


def test_evaluation_record_with_hip_model(tmp_path):
    """The evaluation harness end to end on the GPU path: a sample directory in the reference's
    format -> JointModel with the HIP spot model -> result record + result_<sample>.json."""
    import json
    from acousticswarms_speech_amd import evalkit
    from acousticswarms_speech_amd.config import FULL
    from acousticswarms_speech_amd.joint import JointModel
    from acousticswarms_speech_amd.scenes import make_scene
    from acousticswarms_speech_amd.spot import SpotModel
    from acousticswarms_speech_amd.weights import make_spot_state_dict
    sc = make_scene(1001, 3, 7, 24000)
    evalkit.write_scene_dir(sc, str(tmp_path / "ds" / "00000"))
    jm = JointModel(SpotModel(FULL, make_spot_state_dict(FULL, 5), batch_size=64, precision="f16x3").to("cuda"),
                    None, device="cuda")
    with redirect_stdout(io.StringIO()):
        summary = evalkit.evaluate_dataset(jm, str(tmp_path / "ds"), str(tmp_path / "res"))
    rec = json.load(open(tmp_path / "res" / "result_00000.json"))
    assert summary["tp"] + summary["fn"] == 3 and summary["tp"] == len(rec["pred"])
    assert len(rec["gt"]) == 3 and len(rec["mic_pos"]) == 7
    assert summary["fp"] == len(rec["false_positive"])
    _log(f"evaluation record: tp={summary['tp']} fp={summary['fp']} fn={summary['fn']} (random weights)")
