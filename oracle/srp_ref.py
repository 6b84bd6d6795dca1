"""ORACLE (test infrastructure, never on the product path).

numpy restatement of SRP_PHAT.SRP_Map_WINDOW_torch
(sep/Traditional_SP/SRP_Prunning.py:387-434) on the [G,M] propagation delays instead of
the reference's pre-multiplied [G,bins,pairs] table (algebraically identical:
v_i conj(v_j) = exp(j w (tau_i - tau_j)), :230-246,368-381).  The STFT framing is this
repo's restatement of pyroomacoustics 0.5.0's analysis() (hostdsp.stft_frames; third-party,
"parity unpinned" for the framing itself).  Pinned by fixture g7 (reference-generated).
"""
import numpy as np


def stft_frames(x, nfft, hop):
    n = (x.shape[0] - nfft) // hop + 1
    idx = np.arange(nfft)[None, :] + hop * np.arange(n)[:, None]
    X = np.fft.rfft(x[idx], axis=1)
    return X.astype(np.complex64 if x.dtype == np.float32 else np.complex128)


def cross_spectra(signal, window, nfft, freq_bins, tol=1e-8):
    """Per window: PHAT-normalised, frame-averaged cross-spectrum of every pair i<j -> list of [bins, P]."""
    M, T = signal.shape
    step = window // 2
    ii, jj = np.triu_indices(M, k=1)
    out = []
    for j in range(0, T // step - 1):
        if j * step + window > T:
            break
        win = signal[:, j * step:j * step + window]
        X = np.array([stft_frames(x, nfft, nfft // 4).T for x in win])         # [M, bins, frames]
        a = np.abs(X)
        a[a < tol] = tol
        pX = X / a
        F = pX.shape[2]
        sel = pX[:, freq_bins, :]                                               # [M, nb, F]
        cc = np.einsum("ikf,jkf->kij", sel, np.conj(sel)) / F                   # [nb, M, M]
        out.append(cc[:, ii, jj])
    return out


def srp_map(signal, window, nfft, freq_bins, tau, omega, tol=1e-8, chunk=512):
    """max over windows (starting from zeros) of mean_{bin,pair} Re(CC * exp(j w (tau_i - tau_j)))."""
    M = signal.shape[0]
    ii, jj = np.triu_indices(M, k=1)
    ccs = cross_spectra(signal, window, nfft, np.asarray(freq_bins), tol)
    G = tau.shape[0]
    best = np.zeros(G)
    dt = tau[:, ii] - tau[:, jj]                                                # [G, P]
    for g0 in range(0, G, chunk):
        steer = np.exp(1j * omega[None, :, None] * dt[g0:g0 + chunk, None, :])  # [g, nb, P]
        for cc in ccs:
            r = np.real(cc[None].astype(np.complex128) * steer).sum((1, 2)) / len(freq_bins) / len(ii)
            best[g0:g0 + chunk] = np.maximum(best[g0:g0 + chunk], r)
    return best
