"""Evaluation harness: matching equals the reference's exhaustive search over
permutations (restated here as the brute-force checker), the sample-directory format round-trips,
and the per-sample record carries the reference's fields."""
import io
import itertools
import json
import os
from contextlib import redirect_stdout

import numpy as np

from acousticswarms_speech_amd import evalkit
from acousticswarms_speech_amd.hostdsp import si_sdr


def _brute_force(wav_gt, wav_pred, pos_gt, pos_pred, acceptable_range=1, accept_sisdr=-15):
    """sep/eval/eval_model.py:18-59 as written there: score every permutation."""
    n_gt, n_pred = pos_gt.shape[0], pos_pred.shape[0]
    n = max(n_gt, n_pred)
    neg = np.ones((n, n)) * 10000
    dis = np.ones((n, n)) * 10000
    for i in range(n_gt):
        for j in range(n_pred):
            dis[i, j] = np.linalg.norm(pos_gt[i][:2] - pos_pred[j][:2])
            neg[i, j] = -si_sdr(wav_pred[j], wav_gt[i])
    best, best_in, best_err = None, -1, 10000
    for perm in itertools.permutations(range(n)):
        losses, paired = [], []
        for a, b in enumerate(perm):
            if dis[a, b] < acceptable_range and neg[a, b] < -accept_sisdr:
                losses.append(neg[a, b] + dis[a, b])
                paired.append((b, a))
        err = np.mean(losses) if losses else np.inf
        if len(losses) > best_in or (len(losses) == best_in and err < best_err):
            best, best_in, best_err = paired, len(losses), err
    return best


def test_matching_equals_exhaustive_search():
    rng = np.random.default_rng(0)
    for trial in range(40):
        n_gt, n_pred = rng.integers(1, 5), rng.integers(1, 6)
        wav_gt = rng.standard_normal((n_gt, 400))
        pos_gt = rng.uniform(-2, 2, (n_gt, 3))
        # predictions: noisy copies of some talkers (sometimes misplaced), plus spurious ones
        src = rng.integers(0, n_gt, n_pred)
        wav_pred = wav_gt[src] + rng.uniform(0.05, 3.0, (n_pred, 1)) * rng.standard_normal((n_pred, 400))
        pos_pred = pos_gt[src] + rng.uniform(0.0, 0.9, (n_pred, 1)) * rng.standard_normal((n_pred, 3))
        want = _brute_force(wav_gt, wav_pred, pos_gt, pos_pred)
        got = evalkit.find_best_permutation(wav_gt, wav_pred, pos_gt, pos_pred)
        assert sorted(got) == sorted(want), (trial, got, want)
        assert [g for _p, g in got] == sorted(g for _p, g in got)        # ground-truth order
    assert evalkit.find_best_permutation(np.zeros((0, 10)), np.zeros((2, 10)), np.zeros((0, 3)), np.zeros((2, 3))) == []


def test_sample_directory_round_trip_and_record(tmp_path):
    from acousticswarms_speech_amd.joint import JointModel
    from acousticswarms_speech_amd.scenes import make_scene
    from tests.golden.make_golden_search import ROI, scene_in_roi
    from tests.golden.surrogate import SurrogateSpot
    sc = make_scene(1001, 3, 7, 6000)
    evalkit.write_scene_dir(sc, str(tmp_path / "00000"))
    meta, mix, gt = evalkit.get_items(str(tmp_path / "00000"))
    np.testing.assert_array_equal(mix, sc.mix.astype(np.float32))
    np.testing.assert_array_equal(gt, sc.sources.astype(np.float32))
    _m, micp, _v, voicep, off_gt, roi = evalkit.preprocess_metadata(meta)
    np.testing.assert_allclose(micp, sc.mic_positions)
    np.testing.assert_allclose(voicep, sc.speaker_positions)
    np.testing.assert_array_equal(off_gt, np.round(sc.tdoa_samples()).T)
    assert roi[-1] == sc.speaker_range[-1] + 0.02

    # one full sample through the pipeline (surrogate scorer, CPU): the record has the reference's fields
    mics, spk, mixr = scene_in_roi()
    g7 = np.load(os.path.join(os.path.dirname(__file__), "golden", "g7_srp_map.npz"))
    meta = {"ROI": list(ROI[:5]) + [ROI[5] - 0.02]}
    for m in range(mics.shape[0]):
        meta[f"mic{m:02d}"] = {"position": mics[m].tolist()}
    for s in range(spk.shape[0]):
        meta[f"voice{s:02d}"] = {"position": spk[s].tolist()}
    jm = JointModel(SurrogateSpot())
    orig_setup = jm.setup

    def setup(mic_positions, speaker_range, **kw):                    # no GPU here: reuse the fixture's SRP map
        orig_setup(mic_positions, speaker_range, **kw)
        node = jm.Mic_processor.SRP_node
        node.SRP_Map_WINDOW_new = lambda signal, window=36000: node.set_map(g7["srp_map"])
    jm.setup = setup
    gt = np.stack([mixr[0]] * spk.shape[0]).astype(np.float32)         # placeholder ground truth waveforms
    with redirect_stdout(io.StringIO()):
        rec, tp, fp, fn = evalkit.evaluate_sample(jm, meta, mixr, gt)
    assert set(rec) == {"mic_pos", "speaker_pos", "gt", "pred", "false_positive", "est_offsets", "perm"}
    assert tp + fn == spk.shape[0] and tp == len(rec["pred"]) and fp == len(rec["false_positive"])
    for p in rec["pred"]:
        assert {"voice_id", "shifts", "pos", "sample_err", "dis_err", "si_snr_in", "si_snri", "si_snr_in_old",
                "si_snri_old"} <= set(p)
    json.dumps(rec)                                                    # serialisable as the reference writes it
