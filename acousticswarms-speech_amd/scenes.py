"""Seeded synthetic scenes (inputs only; no arithmetic of the hot path).

pyroomacoustics and the Zenodo datasets are unavailable offline, so benchmarks
and parity tests run on free-field (optionally first-order image-source)
mixtures whose geometry follows the reference's dataset recipe
(datasets/generate_dataset.py:341-376 desk + robots fanned from mic 0,
:512-578 talker placement, :19-20 amplitudes).
"""
from dataclasses import dataclass

import numpy as np

SPEED_OF_SOUND = 343.0   # sep/helpers/constants.py:7
FS = 48000               # sep/helpers/constants.py:8


@dataclass
class Scene:
    mic_positions: np.ndarray      # [M,3]
    speaker_positions: np.ndarray  # [S,3]
    speaker_range: list            # [xmin,xmax,ymin,ymax,zmin,zmax]
    mix: np.ndarray                # [M,T] float32
    sources: np.ndarray            # [S,T] float32 (as received at mic 0)
    fs: int

    def tdoa_samples(self) -> np.ndarray:
        """Ground-truth sample offsets [S, M-1] relative to mic 0."""
        d = np.linalg.norm(self.speaker_positions[:, None, :] - self.mic_positions[None], axis=2)
        return (d[:, 1:] - d[:, :1]) / SPEED_OF_SOUND * self.fs


def _speech_like(rng, T, fs):
    """Band-limited (80 Hz-6 kHz, ~1/sqrt(f)) Gaussian noise with a 2-4 Hz
    syllabic envelope."""
    n = rng.standard_normal(T)
    spec = np.fft.rfft(n)
    f = np.fft.rfftfreq(T, 1.0 / fs)
    shape = np.zeros_like(f)
    band = (f >= 80) & (f <= 6000)
    shape[band] = 1.0 / np.sqrt(f[band])
    x = np.fft.irfft(spec * shape, T)
    rate = rng.uniform(2.0, 4.0)
    ph = rng.uniform(0, 2 * np.pi)
    t = np.arange(T) / fs
    env = np.clip(np.sin(2 * np.pi * rate * t + ph) + 0.3, 0.0, None) ** 2
    x = x * env
    return x / (np.abs(x).max() + 1e-12)


def _frac_delay(x, delay_samples):
    """Exact fractional delay by linear phase in the frequency domain."""
    T = x.shape[0]
    X = np.fft.rfft(x)
    k = np.arange(X.shape[0])
    return np.fft.irfft(X * np.exp(-2j * np.pi * k * delay_samples / T), T)


def desk_mics(rng, n_mics=7):
    """mic0 at the desk edge centre, the others fanned towards the desk edges
    (generate_dataset.py:341-376, simplified: fixed radius fractions)."""
    dx = rng.uniform(1.2, 2.0)
    dy = rng.uniform(0.6, 1.2)
    mics = [np.array([0.0, 0.0, 0.02])]
    ang = np.linspace(0, np.pi, n_mics - 1) - np.pi / 2
    for i, a in enumerate(ang):
        # distance to the desk boundary along direction a, minus 4 cm
        cx, cy = np.cos(a), np.sin(a)
        lim = []
        if abs(cy) > 1e-9:
            lim.append((dx / 2) / abs(cy))
        if cx > 1e-9:
            lim.append(dy / cx)
        r = max(min(lim) - 0.04, 0.15) * rng.uniform(0.6, 1.0)
        mics.append(np.array([r * cy, r * cx, 0.02]))
    if n_mics > 7:
        # second concentric fan for the 16-mic stress configuration
        mics = mics[:1]
        ang = np.linspace(0, np.pi, n_mics - 1) - np.pi / 2
        for i, a in enumerate(ang):
            r = (0.25 if i % 2 == 0 else 0.5) * rng.uniform(0.9, 1.1)
            mics.append(np.array([r * np.sin(a), r * np.cos(a), 0.02]))
    return np.stack(mics), (dx, dy)


def make_scene(seed: int, n_speakers: int = 3, n_mics: int = 7, T: int = 48000,
               fs: int = FS, reverb: bool = False, noise_std: float = 1e-3, mic_positions=None) -> Scene:
    """``mic_positions`` fixes the array geometry (several mixtures of one recording session);
    by default every seed draws its own desk and robot positions."""
    rng = np.random.default_rng(1000 + seed if seed < 1000 else seed)
    mics, (dx, dy) = desk_mics(rng, n_mics)
    if mic_positions is not None:
        mics = np.asarray(mic_positions, dtype=np.float64)
    roi = [-2.2, 2.2, 0.3, 4.0, 0.1, 0.9]
    spk = []
    tries = 0
    while len(spk) < n_speakers and tries < 10000:
        tries += 1
        p = np.array([rng.uniform(roi[0] + 0.1, roi[1] - 0.1),
                      rng.uniform(max(roi[2], dy + 0.25), roi[3] - 0.1),
                      rng.uniform(0.1, 0.8)])
        if all(np.linalg.norm(p - q) >= 0.51 for q in spk):
            spk.append(p)
    spk = np.stack(spk)
    mix = np.zeros((n_mics, T))
    srcs = np.zeros((n_speakers, T))
    for s in range(n_speakers):
        x = _speech_like(rng, T, fs) * rng.uniform(0.2, 0.5)
        d = np.linalg.norm(spk[s] - mics, axis=1)
        images = [(spk[s], 1.0)]
        if reverb:
            for axis, wall, g in ((0, -3.0, 0.5), (0, 3.0, 0.5), (1, 5.0, 0.45), (2, 2.4, 0.4)):
                q = spk[s].copy()
                q[axis] = 2 * wall - q[axis]
                images.append((q, g))
        for pos, g in images:
            dd = np.linalg.norm(pos - mics, axis=1)
            for m in range(n_mics):
                sig = _frac_delay(x, (dd[m] - d[0]) / SPEED_OF_SOUND * fs) * (g * min(1.0, 1.0 / dd[m]))
                mix[m] += sig
                if m == 0 and g == 1.0:
                    srcs[s] = sig
    mix += noise_std * rng.standard_normal(mix.shape)
    return Scene(mics, spk, roi, mix.astype(np.float32), srcs.astype(np.float32), fs)


def random_offsets(seed: int, n: int, n_pairs: int = 6, max_abs: int = 140) -> np.ndarray:
    """Seeded integer TDoA candidates [n, n_pairs] (benchmark fan-out)."""
    rng = np.random.default_rng(seed)
    return rng.integers(-max_abs, max_abs + 1, size=(n, n_pairs)).astype(np.int32)
