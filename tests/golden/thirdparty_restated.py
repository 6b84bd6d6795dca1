"""Stand-ins handed to the REFERENCE (generator side only) for the three absent
third-party calls on the search path.  They are this repo's restatements
(acousticswarms_speech_amd.hostdsp), not the third-party originals: fixtures that
flow through them (g7, g8, g10) pin everything except that framing."""
from acousticswarms_speech_amd import hostdsp as _h


def pra_stft_analysis(x, L, hop, win=None, zp_back=0, zp_front=0):
    return _h.stft_frames(x, L, hop)


def librosa_rms(y=None, frame_length=2048, hop_length=512, **_):
    return _h.frame_rms(y, frame_length, hop_length)


def librosa_split(y, top_db=60, ref=None, frame_length=2048, hop_length=512, **_):
    import numpy as np
    return _h.nonsilent_intervals(y, top_db=top_db, ref=np.max if ref is None else ref,
                                  frame_length=frame_length, hop_length=hop_length)
