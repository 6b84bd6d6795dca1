"""Small host-side DSP helpers of the search path (numpy only).

Rows a-J / a-R of SURVEY.md §8: windowed-RMS energy, SI-SDR, voiced-segment
splitting.  The reference obtains its STFT framing from pyroomacoustics 0.5.0 and
its RMS/split from librosa (both absent offline and un-pinned by any reference
test): ``stft_frames``, ``frame_rms`` and ``nonsilent_intervals`` restate the
published behaviour of those third-party calls -- "parity unpinned" for the
framing itself; everything built on top of them is pinned by
fixtures g7-g10.
"""
import math

import numpy as np

MIN_ERR = 1e-8            # sep/helpers/eval_utils.py:9


# ---- third-party call sites, restated ------------------------------------
def stft_frames(x: np.ndarray, nfft: int, hop: int) -> np.ndarray:
    """pyroomacoustics.transform.stft.analysis(x, nfft, hop) as called at
    sep/Traditional_SP/SRP_Prunning.py:404-409: rectangular window, no padding,
    frames at multiples of ``hop``, one-sided FFT, single precision for a
    float32 input.  Returns [n_frames, nfft//2+1]."""
    n = (x.shape[0] - nfft) // hop + 1
    idx = np.arange(nfft)[None, :] + hop * np.arange(n)[:, None]
    X = np.fft.rfft(x[idx], axis=1)
    return X.astype(np.complex64 if x.dtype == np.float32 else np.complex128)


def frame_rms(y: np.ndarray, frame_length: int = 1024, hop_length: int = 256) -> np.ndarray:
    """librosa.feature.rms(y=, frame_length=, hop_length=) (centered, zero padded):
    returns [1, 1 + len(y)//hop].  The samples are squared once and the frames are a strided
    view of that array (no gather, no per-frame re-squaring)."""
    pad = frame_length // 2
    yp = np.pad(y, (pad, pad), mode="constant")
    sq = yp * yp
    frames = np.lib.stride_tricks.sliding_window_view(sq, frame_length)[::hop_length]
    return np.sqrt(np.mean(frames, axis=1))[None, :]


def nonsilent_intervals(y, top_db=60, ref=np.max, frame_length=2048, hop_length=512, rms=None):
    """librosa.effects.split: frames whose RMS is within ``top_db`` of ``ref``.  ``rms`` may
    carry frame_rms(y, frame_length, hop_length) when the caller already has it."""
    rms = frame_rms(y, frame_length, hop_length)[0] if rms is None else rms[0]
    amin = 1e-5
    ref_value = np.abs(ref(rms)) if callable(ref) else np.abs(ref)
    db = 10.0 * np.log10(np.maximum(amin ** 2, rms ** 2)) - 10.0 * np.log10(max(amin ** 2, ref_value ** 2))
    ns = db > -top_db
    edges = [np.flatnonzero(np.diff(ns.astype(int))) + 1]
    if ns[0]:
        edges.insert(0, np.array([0]))
    if ns[-1]:
        edges.append(np.array([len(ns)]))
    e = np.concatenate(edges) * hop_length
    e = np.minimum(e, y.shape[-1])
    return e.reshape((-1, 2))


# ---- reference helpers ----------------------------------------------------
def si_sdr(est: np.ndarray, ref: np.ndarray) -> float:
    """Scale-invariant SDR, sep/helpers/eval_utils.py:11-39."""
    rss = np.dot(ref, ref)
    a = np.dot(ref, est) / rss
    target = a * ref
    resid = est - target
    return 10 * math.log10((target ** 2).sum() / ((resid ** 2).sum() + MIN_ERR))


def split_wav(wav: np.ndarray, top_db: float = 18):
    """Voiced segments of 1000..4000 samples, sep/helpers/eval_utils.py:43-70."""
    lo, hi = 1000, 4000
    rms = frame_rms(wav, 1024, 256)
    peak = np.amax(rms)
    if peak < 0.04:
        iv = nonsilent_intervals(wav, top_db=top_db, ref=0.04, frame_length=1024, hop_length=256, rms=rms)
    else:
        iv = nonsilent_intervals(wav, top_db=top_db, frame_length=1024, hop_length=256, rms=rms)
    segs = []
    for a, b in iv:
        n = b - a
        if n < lo:
            continue
        if n > hi:
            k = n // hi
            for i in range(k):
                segs.append([a + i * hi, b if i >= k - 1 else a + (i + 1) * hi])
        else:
            segs.append([a, b])
    return segs


def split_wise_sisdr(est, ref, segments):
    """sep/helpers/eval_utils.py:73-82."""
    assert len(segments) > 0
    return [si_sdr(est[a:b], ref[a:b]) for a, b in segments]


def max_avg_power(x: np.ndarray, window_size: int = 12000) -> float:
    """max over start index of sqrt(mean(x^2 over a forward window, zero padded));
    sep/helpers/local_utils_3d.py:13-17 (value only)."""
    sq = (x ** 2).astype(np.float64)
    c = np.concatenate([[0.0], np.cumsum(sq)])
    n = sq.shape[0]
    hi = np.minimum(np.arange(n) + window_size, n)
    return float(np.sqrt(np.abs((c[hi] - c[:n]) / window_size)).max())
