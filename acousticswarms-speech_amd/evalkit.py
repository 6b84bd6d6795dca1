"""Evaluation harness of the search: the reference's sample-directory format,
prediction <-> ground-truth matching, localisation / separation scores and the per-sample result
record of ``sep/eval/eval_model.py``.

Follows sep/eval/eval_model.py:18-91 (matching, metadata preprocessing), :129-249 (result
record) and sep/eval/get_items.py:10-46 (directory layout written by
datasets/generate_dataset.py:633-699: ``metadata.json`` + ``micNN_mixed.wav`` +
``mic00_voiceNN.wav``).  Two third-party metrics the reference calls are absent here and are
NOT restated: mir_eval's BSS-eval SDR (fields ``si_snr_in_mir`` / ``si_snri_mir`` are None) and
asteroid's metric wrapper, whose SI-SDR is the plain scale-invariant SDR of
``sep/helpers/eval_utils.py:11-39`` (``hostdsp.si_sdr``) -- "parity unpinned" for the wrapper.
"""
import json
import os

import numpy as np

from .hostdsp import si_sdr
from .patch import FS, SPEED_OF_SOUND

NO_MATCH = 10000.0


def find_best_permutation(wav_gt, wav_pred, pos_gt, pos_pred, acceptable_range=1, accept_sisdr=-15):
    """Pairs (prediction index, ground-truth index), in ground-truth order, that maximise the
    number of inliers (xy distance < acceptable_range AND SI-SDR > accept_sisdr) and, among
    those, minimise the mean of (distance - SI-SDR) over the inlier pairs
    (sep/eval/eval_model.py:18-59).

    The reference scores every permutation of max(n_gt, n_pred) items; the same optimum is the
    maximum-cardinality, minimum-cost matching of the inlier graph, found here with the
    Hungarian method (identical result up to exact ties, and usable beyond ~9 items)."""
    from scipy.optimize import linear_sum_assignment
    pos_gt, pos_pred = np.asarray(pos_gt, dtype=np.float64), np.asarray(pos_pred, dtype=np.float64)
    n_gt, n_pred = pos_gt.shape[0], pos_pred.shape[0]
    if n_gt == 0 or n_pred == 0:
        return []
    loss = np.full((n_gt, n_pred), NO_MATCH)
    inlier = np.zeros((n_gt, n_pred), dtype=bool)
    for i in range(n_gt):
        for j in range(n_pred):
            dis = np.linalg.norm(pos_gt[i][:2] - pos_pred[j][:2])
            neg = -si_sdr(np.asarray(wav_pred[j]), np.asarray(wav_gt[i]))
            loss[i, j] = neg + dis
            inlier[i, j] = dis < acceptable_range and neg < -accept_sisdr
    if not inlier.any():
        return []
    lmin = float(loss[inlier].min())
    span = float(loss[inlier].max()) - lmin + 1.0
    big = span * (min(n_gt, n_pred) + 1)                     # one more inlier always beats any cost change
    cost = np.where(inlier, (loss - lmin) - big, 0.0)
    rows, cols = linear_sum_assignment(cost)
    return [(int(j), int(i)) for i, j in zip(rows, cols) if inlier[i, j]]


def preprocess_metadata(metadata):
    """(mic names, mic_positions [M,3], voice names, voice_positions [S,3],
    sample_offsets_gt [M-1,S] rounded TDoA in samples, speaker range with +2 cm on z max)
    (sep/eval/eval_model.py:61-91)."""
    mics = sorted(k for k in metadata if k.startswith("mic"))
    mic_positions = np.array([metadata[k]["position"] for k in mics], dtype=np.float64)
    voices = sorted(k for k in metadata if k.startswith("voice"))
    voice_positions = np.array([metadata[k]["position"][:3] for k in voices], dtype=np.float64).reshape(-1, 3)
    gt = np.zeros((mic_positions.shape[0] - 1, len(voices)))
    for j in range(len(voices)):
        for i in range(1, mic_positions.shape[0]):
            d = np.linalg.norm(voice_positions[j] - mic_positions[i]) - np.linalg.norm(voice_positions[j] - mic_positions[0])
            gt[i - 1, j] = int(np.round(d / SPEED_OF_SOUND * FS))
    roi = list(metadata["ROI"])
    roi[-1] += 0.02
    return mics, mic_positions, voices, voice_positions, gt, roi


# ---- sample directories --------------------------------------------------------------------
def _read_wav(path):
    from scipy.io import wavfile
    _sr, x = wavfile.read(path)
    if x.dtype == np.int16:
        return x.astype(np.float32) / 32768.0
    if x.dtype == np.int32:
        return x.astype(np.float32) / 2147483648.0
    return x.astype(np.float32)


def write_scene_dir(scene, path):
    """Write a ``scenes.Scene`` in the reference's on-disk sample format (float32 wav)."""
    from scipy.io import wavfile
    os.makedirs(path, exist_ok=True)
    meta = {"ROI": [float(v) for v in scene.speaker_range], "real": False}
    for m in range(scene.mic_positions.shape[0]):
        meta[f"mic{m:02d}"] = {"position": [float(v) for v in scene.mic_positions[m]]}
        wavfile.write(os.path.join(path, f"mic{m:02d}_mixed.wav"), scene.fs, scene.mix[m].astype(np.float32))
    for s in range(scene.speaker_positions.shape[0]):
        meta[f"voice{s:02d}"] = {"position": [float(v) for v in scene.speaker_positions[s]]}
        wavfile.write(os.path.join(path, f"mic00_voice{s:02d}.wav"), scene.fs, scene.sources[s].astype(np.float32))
    with open(os.path.join(path, "metadata.json"), "w") as f:
        json.dump(meta, f, indent=1)


def get_items(path):
    """(metadata, mixture [M,T] float32, ground truth at mic 0 [S,T]) (sep/eval/get_items.py:10-46)."""
    with open(os.path.join(path, "metadata.json")) as f:
        metadata = json.load(f)
    mics = sorted(k for k in metadata if k.startswith("mic"))
    mix = np.stack([_read_wav(os.path.join(path, f"{m}_mixed.wav")) for m in mics])
    voices = sorted(k for k in metadata if k.startswith("voice"))
    gt = []
    for v in voices:
        den = os.path.join(path, f"{mics[0]}_{v}_denoised.wav")
        gt.append(_read_wav(den if os.path.exists(den) else os.path.join(path, f"{mics[0]}_{v}.wav")))
    return metadata, mix, np.stack(gt) if gt else np.zeros((0, mix.shape[1]), dtype=np.float32)


# ---- separation metrics ------------------------------------------------------------------------
def _project(reference_sources, estimated_source, flen):
    """Least-squares projection of ``estimated_source`` on the span of the references delayed by
    0..flen-1 samples (BSS-eval "mtifilt" decomposition, Vincent et al. 2006), solved through
    FFT-computed auto / cross-correlations and one (nsrc*flen)-square linear system."""
    from scipy.linalg import toeplitz
    from scipy.signal import fftconvolve
    nsrc, nsampl = reference_sources.shape
    ref = np.hstack((reference_sources, np.zeros((nsrc, flen - 1))))
    est = np.hstack((estimated_source, np.zeros(flen - 1)))
    n_fft = int(2 ** np.ceil(np.log2(nsampl + flen - 1.0)))
    sf = np.fft.fft(ref, n=n_fft, axis=1)
    sef = np.fft.fft(est, n=n_fft)
    G = np.zeros((nsrc * flen, nsrc * flen))
    for i in range(nsrc):
        for j in range(i, nsrc):
            ssf = np.real(np.fft.ifft(sf[i] * np.conj(sf[j])))
            ss = toeplitz(np.hstack((ssf[0], ssf[-1:-flen:-1])), r=ssf[:flen])
            G[i * flen:(i + 1) * flen, j * flen:(j + 1) * flen] = ss
            G[j * flen:(j + 1) * flen, i * flen:(i + 1) * flen] = ss.T
    D = np.zeros(nsrc * flen)
    for i in range(nsrc):
        ssef = np.real(np.fft.ifft(sf[i] * np.conj(sef)))
        D[i * flen:(i + 1) * flen] = np.hstack((ssef[0], ssef[-1:-flen:-1]))
    try:
        C = np.linalg.solve(G, D).reshape(flen, nsrc, order="F")
    except np.linalg.LinAlgError:
        C = np.linalg.lstsq(G, D, rcond=None)[0].reshape(flen, nsrc, order="F")
    sproj = np.zeros(nsampl + flen - 1)
    for i in range(nsrc):
        sproj += fftconvolve(C[:, i], ref[i])[:nsampl + flen - 1]
    return sproj


def bss_eval_sdr(reference_sources, estimated_sources, flen: int = 512):
    """Signal-to-distortion ratio of estimate j against reference j with a 512-tap
    time-invariant distortion filter allowed -- the quantity the reference takes from
    ``mir_eval.separation.bss_eval_sources(..., compute_permutation=False)``
    (sep/eval/get_items.py:50-52).  mir_eval is absent from the image and unpinned upstream; this
    is a restatement of the published BSS-eval v3 algorithm: "parity unpinned"."""
    ref = np.atleast_2d(np.asarray(reference_sources, dtype=np.float64))
    est = np.atleast_2d(np.asarray(estimated_sources, dtype=np.float64))
    if ref.shape != est.shape:
        raise ValueError(f"reference {ref.shape} and estimate {est.shape} shapes differ")
    out = np.zeros(ref.shape[0])
    for j in range(ref.shape[0]):
        nsampl = est.shape[1]
        s_true = np.hstack((ref[j], np.zeros(flen - 1)))
        e_spat = _project(ref[j:j + 1], est[j], flen) - s_true
        e_interf = _project(ref, est[j], flen) - s_true - e_spat
        e_artif = -s_true - e_spat - e_interf
        e_artif[:nsampl] += est[j]
        num, den = np.sum((s_true + e_spat) ** 2), np.sum((e_interf + e_artif) ** 2)
        out[j] = np.inf if den == 0 else 10 * np.log10(num / den)
    return out


def compute_metrics(input_signal, est_signal, gt):
    """(input_sdr, output_sdr, input_sisdr, output_sisdr) per matched talker, as
    sep/eval/get_items.py:46-70 with permute=False: BSS-eval SDR and SI-SDR of the unprocessed
    reference-microphone signal and of the separated output against the ground truth."""
    gt = np.asarray(gt, dtype=np.float64)
    inp = np.asarray(input_signal, dtype=np.float64)
    est = np.asarray(est_signal, dtype=np.float64)
    input_sdr, output_sdr = bss_eval_sdr(gt, inp), bss_eval_sdr(gt, est)
    input_sisdr = [si_sdr(inp[i], gt[i]) for i in range(gt.shape[0])]
    output_sisdr = [si_sdr(est[i], gt[i]) for i in range(gt.shape[0])]
    return input_sdr, output_sdr, input_sisdr, output_sisdr


# ---- one sample --------------------------------------------------------------------------------
def evaluate_sample(model, metadata, mix, gt):
    """Run ``model`` (a ``JointModel``) on one sample and build the reference's result record
    (sep/eval/eval_model.py:129-236).  Returns (record, tp, fp, fn)."""
    import torch
    _mics, mic_positions, _voices, gt_pos, offsets_gt, roi = preprocess_metadata(metadata)
    model.setup(mic_positions=mic_positions, speaker_range=roi)
    patches, audio_loc, audio, _, _, _ = model(torch.from_numpy(np.ascontiguousarray(mix, dtype=np.float32)))
    n_out = len(patches)
    est_pos = np.array([p[0].center_pos() for p in patches]).reshape(-1, 3)
    est_off = [np.asarray(p[4]["localization_offset"]) for p in patches]
    audio_loc = np.asarray(audio_loc).reshape(n_out, -1) if n_out else np.zeros((0, mix.shape[1]))
    sep_audio = audio_loc if audio is None else np.asarray(audio)     # joint decoder output when plugged in
    perm = find_best_permutation(gt, sep_audio, gt_pos, est_pos, acceptable_range=1) if n_out else []
    rec = {"mic_pos": mic_positions.tolist(), "speaker_pos": gt_pos.tolist(), "gt": [], "pred": [],
           "false_positive": [], "est_offsets": np.array(est_off).tolist(), "perm": perm}
    tp, fn, fp = len(perm), gt.shape[0] - len(perm), n_out - len(perm)
    for s in range(gt_pos.shape[0]):
        rec["gt"].append({"sample": offsets_gt[:, s].tolist(), "pos": gt_pos[s].tolist()})
    unmatched = list(range(n_out))
    if perm:
        # matched pairs in the reference's order (eval_model.py:162-187): separated output (joint
        # decoder when attached), localisation-stage output, ground truth, unprocessed mic 0
        pa = np.array(perm)
        ref_sig = np.repeat(mix[0:1].astype(np.float64), len(perm), axis=0)
        in_sdr, out_sdr, in_sisdr_l, out_sisdr_l = compute_metrics(ref_sig, sep_audio[pa[:, 0]], gt[pa[:, 1]])
    for i, (out_id, s) in enumerate(perm):
        unmatched.remove(out_id)
        in_sisdr = in_sisdr_l[i]
        rec["pred"].append({
            "voice_id": s, "shifts": est_off[out_id].tolist(), "pos": est_pos[out_id].tolist(),
            "sample_err": float(np.mean(np.abs(est_off[out_id] - offsets_gt[:, s]))),
            "dis_err": float(np.linalg.norm(est_pos[out_id][:2] - gt_pos[s][:2])),
            "si_snr_in_mir": float(in_sdr[i]), "si_snri_mir": float(out_sdr[i] - in_sdr[i]),   # BSS-eval SDR (restated)
            "si_snr_in": in_sisdr,
            "si_snri": out_sisdr_l[i] - in_sisdr,
            "si_snr_in_old": in_sisdr,
            "si_snri_old": si_sdr(audio_loc[out_id].astype(np.float64), gt[s].astype(np.float64)) - in_sisdr})
    for rid in unmatched:
        rec["false_positive"].append({"pos": est_pos[rid].tolist(),
                                      "sample": np.asarray(patches[rid][4]["audio_offset"]).tolist()})
    return rec, tp, fp, fn


def evaluate_dataset(model, dataset_dir, results_folder=None):
    """Every sample directory of ``dataset_dir``; writes ``result_<sample>.json`` like the
    reference and returns overall (tp, fp, fn, precision, recall)."""
    tot = np.zeros(3, dtype=np.int64)
    for name in sorted(d for d in os.listdir(dataset_dir) if os.path.isdir(os.path.join(dataset_dir, d))):
        metadata, mix, gt = get_items(os.path.join(dataset_dir, name))
        rec, tp, fp, fn = evaluate_sample(model, metadata, mix, gt)
        tot += (tp, fp, fn)
        if results_folder is not None:
            os.makedirs(results_folder, exist_ok=True)
            with open(os.path.join(results_folder, f"result_{name}.json"), "w") as f:
                json.dump(rec, f, indent=4)
    tp, fp, fn = (int(v) for v in tot)
    return {"tp": tp, "fp": fp, "fn": fn, "precision": tp / max(tp + fp, 1), "recall": tp / max(tp + fn, 1)}
