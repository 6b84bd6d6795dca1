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


def test_bss_eval_sdr_properties():
    """BSS-eval SDR (restated; mir_eval absent -> parity unpinned): invariant to a short FIR
    filtering of the target, equal to the SNR for white noise, and the interference term uses the
    other references."""
    from scipy.signal import lfilter
    rng = np.random.default_rng(3)
    ref = rng.standard_normal((2, 6000))
    # a filtered copy of the target is "distortion-free" up to the allowed filter (what is left is
    # the two-sample tail the truncated FIR output lacks: ~ -50 dB at this length); a pure gain is exact
    est = np.stack([lfilter([0.9, 0.3, -0.2], [1.0], ref[0]), 0.5 * ref[1]])
    sdr = evalkit.bss_eval_sdr(ref, est, flen=64)
    assert sdr[0] > 40 and sdr[1] > 100
    # additive white noise at 10 dB SNR
    noise = rng.standard_normal(6000)
    noise *= np.linalg.norm(ref[0]) / np.linalg.norm(noise) / np.sqrt(10.0)
    sdr = evalkit.bss_eval_sdr(ref[:1], (ref[0] + noise)[None], flen=64)
    assert abs(sdr[0] - 10.0) < 0.5
    # leakage of the other talker counts as distortion
    sdr = evalkit.bss_eval_sdr(ref, np.stack([ref[0] + 0.1 * ref[1], ref[1]]), flen=64)
    assert 18.0 < sdr[0] < 22.0 and sdr[1] > 100
    ins, outs, isi, osi = evalkit.compute_metrics(np.stack([ref[0] + ref[1]] * 2), ref + 0.01 * rng.standard_normal(ref.shape), ref)
    assert np.all(np.asarray(outs) > np.asarray(ins) + 20) and np.all(np.asarray(osi) > np.asarray(isi) + 20)


class _Sched:                                     # stands in for the lr scheduler the reference pickles into state.pt
    pass


def test_experiment_directory_loading(tmp_path):
    """load_model_from_exp (sep/helpers/utils.py:165-215): description.json -> network, best /
    last checkpoint selection, nothing unpickled."""
    import torch
    from acousticswarms_speech_amd.config import SEP_SMALL, SMALL
    from acousticswarms_speech_amd.experiment import config_from_description, load_model_from_exp
    from acousticswarms_speech_amd.weights import make_sep_state_dict, make_spot_state_dict
    # the reference's own description files
    kind, cfg = config_from_description({"model_name": "SpeakerLocalization", "model_params": {
        "n_mics": 7, "channels": 64, "growth": 2, "encoder_channels": 2048, "stride_list": [2, 2, 4, 4, 4],
        "kernel_size": 7, "residual_dilation_factor": 7}})
    assert kind == "spot" and cfg.stride_list == (2, 2, 4, 4, 4) and cfg.encoder_channels == 2048
    kind, cfg = config_from_description({"model_name": "SpeakerSeparation", "model_params": {
        "n_mics": 7, "max_speakers": 5, "channels": 64, "growth": 2, "encoder_channels": 4096}})
    assert kind == "sep" and cfg.max_speakers == 5 and cfg.stride_list == (2, 2, 4, 4) and cfg.encoder_channels == 4096
    # spot experiment with two checkpoints and a loadable state.pt -> 'best' picks epoch 1
    exp = tmp_path / "loc"
    (exp / "checkpoints").mkdir(parents=True)
    params = {"n_mics": 7, "kernel_size": 7, "stride_list": list(SMALL.stride_list), "channels": SMALL.channels, "growth": 2,
              "encoder_channels": SMALL.encoder_channels, "ffw_dim": SMALL.ffw_dim}
    (exp / "description.json").write_text(json.dumps({"model_name": "SpeakerLocalization", "model_params": params}))
    for ep in (0, 1, 2):
        sd = {k: torch.from_numpy(v) for k, v in make_spot_state_dict(SMALL, seed=100 + ep).items()}
        torch.save(sd, exp / "checkpoints" / f"loc_{ep}.pt")
    torch.save({"val_losses": [0.9, 0.2, 0.5]}, exp / "checkpoints" / "state.pt")
    with redirect_stdout(io.StringIO()):
        m = load_model_from_exp(str(exp), mode="best")
    np.testing.assert_array_equal(m._sd["preproc.weight"], make_spot_state_dict(SMALL, seed=101)["preproc.weight"])
    with redirect_stdout(io.StringIO()):
        m = load_model_from_exp(str(exp), mode="last")
    np.testing.assert_array_equal(m._sd["preproc.weight"], make_spot_state_dict(SMALL, seed=102)["preproc.weight"])
    # a state.pt the safe loader refuses (the reference pickles its scheduler object into it): 'best' must not
    # silently become 'last'; the caller names the epoch or opts into the fallback
    import pytest
    torch.save({"val_losses": [0.9, 0.2, 0.5], "lr_sched": _Sched()}, exp / "checkpoints" / "state.pt")
    with pytest.raises(RuntimeError, match="best_epoch"), redirect_stdout(io.StringIO()):
        load_model_from_exp(str(exp), mode="best")
    with redirect_stdout(io.StringIO()):
        m = load_model_from_exp(str(exp), mode="best", best_epoch=1)
    np.testing.assert_array_equal(m._sd["preproc.weight"], make_spot_state_dict(SMALL, seed=101)["preproc.weight"])
    with redirect_stdout(io.StringIO()) as out:
        m = load_model_from_exp(str(exp), mode="best", fallback_to_last=True)
    assert "WARNING" in out.getvalue()
    np.testing.assert_array_equal(m._sd["preproc.weight"], make_spot_state_dict(SMALL, seed=102)["preproc.weight"])
    with pytest.raises(RuntimeError, match="growth"):
        config_from_description({"model_name": "SpeakerLocalization", "model_params": {"growth": 1.5}})
    # separation experiment without state.pt: falls back to 'last'; 'new' loads nothing
    exp2 = tmp_path / "sepexp"
    (exp2 / "checkpoints").mkdir(parents=True)
    sparams = {"n_mics": 7, "max_speakers": 5, "stride_list": list(SEP_SMALL.stride_list), "channels": 64, "growth": 2,
               "encoder_channels": SEP_SMALL.encoder_channels, "ffw_dim": SEP_SMALL.ffw_dim,
               "bottleneck_layers": SEP_SMALL.bottleneck_layers, "bottleneck_ksize": SEP_SMALL.bottleneck_ksize}
    (exp2 / "description.json").write_text(json.dumps({"model_name": "SpeakerSeparation", "model_params": sparams}))
    torch.save({k: torch.from_numpy(v) for k, v in make_sep_state_dict(SEP_SMALL, seed=7).items()},
               exp2 / "checkpoints" / "sepexp_4.pt")
    with redirect_stdout(io.StringIO()) as out:
        m2 = load_model_from_exp(str(exp2), mode="best")
    assert "WARNING" in out.getvalue() and m2._sd is not None and m2.cfg == SEP_SMALL
    with redirect_stdout(io.StringIO()):
        assert load_model_from_exp(str(exp2), mode="new")._sd is None
    # a wrong key set is rejected (strict)
    bad = {k: torch.from_numpy(v) for k, v in make_sep_state_dict(SEP_SMALL, seed=7).items()}
    bad.pop("preproc.bias")
    torch.save(bad, exp2 / "checkpoints" / "sepexp_9.pt")
    import pytest
    with pytest.raises(RuntimeError), redirect_stdout(io.StringIO()):
        load_model_from_exp(str(exp2), mode="last")
