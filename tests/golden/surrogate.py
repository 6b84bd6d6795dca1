"""Model-free surrogate with the ``shift_and_sep`` surface (ours, not from the reference).

Used to pin the SEARCH ORCHESTRATION (candidate ordering, thresholds, clustering)
independently of network numerics and of the unavailable checkpoints (SURVEY.md §8c,
fixture g10): align the channels by the candidate's integer TDoA, keep the
time-frequency bins where the aligned channels are coherent, resynthesise mic 0.
A talker at the candidate's TDoA survives, others are attenuated -- which is what the
trained spot network does, crudely.
"""
import numpy as np


class SurrogateSpot(object):
    def __init__(self, nfft=512, hop=256):
        self.nfft, self.hop = nfft, hop
        self.win = np.hanning(nfft + 1)[:-1]
        self.calls = []

    def _stft(self, x):
        n = 1 + (x.shape[-1] - self.nfft) // self.hop
        idx = np.arange(self.nfft)[None, :] + self.hop * np.arange(n)[:, None]
        return np.fft.rfft(x[..., idx] * self.win, axis=-1)

    def _istft(self, X, T):
        frames = np.fft.irfft(X, self.nfft, axis=-1) * self.win
        y = np.zeros(T)
        norm = np.zeros(T)
        for i in range(frames.shape[0]):
            s = i * self.hop
            y[s:s + self.nfft] += frames[i]
            norm[s:s + self.nfft] += self.win ** 2
        return y / np.maximum(norm, 1e-3)

    def shift_and_sep(self, input_channels, patch_list, Strict=0, save_input=False):
        mix = input_channels.numpy() if hasattr(input_channels, "numpy") else np.asarray(input_channels)
        mix = mix.astype(np.float64)
        M, T = mix.shape
        p = 12.0 if Strict == 1 else 6.0
        out = np.zeros((len(patch_list), T), dtype=np.float32)
        self.calls.append((len(patch_list), Strict))
        for n, patch in enumerate(patch_list):
            off = np.rint(np.asarray(patch.sample_offset, dtype=np.float64)).astype(int)
            al = np.stack([mix[0]] + [np.roll(mix[m + 1], -off[m]) for m in range(M - 1)])
            X = self._stft(al)
            coh = np.abs(X.mean(0)) / (np.abs(X).mean(0) + 1e-9)
            out[n] = self._istft(coh ** p * X[0], T).astype(np.float32)
        return out

    def to(self, device=None):
        return self
