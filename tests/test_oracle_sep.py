"""CPU: the joint-separation oracle (oracle/sep_ref.py) against the fixtures produced by the
reference's own Network (tests/golden/make_golden_sep.py; speechbrain's two classes restated, see
that file), and the evaluation matcher against the reference's find_best_permutation (g12)."""
import numpy as np
import torch

from acousticswarms_speech_amd import evalkit
from acousticswarms_speech_amd.config import SEP_FULL, SEP_SMALL, sep_param_shapes
from acousticswarms_speech_amd.scenes import make_scene
from acousticswarms_speech_amd.weights import make_sep_state_dict
from oracle import sep_ref


def _snr(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return 10 * np.log10(np.sum(want ** 2) / max(np.sum((got - want) ** 2), 1e-300))


def test_param_count_of_the_full_network():
    # 33.75 M parameters + the 256-entry inv_freq buffer of RelPosEncXL
    assert sum(int(np.prod(s)) for _n, s in sep_param_shapes(SEP_FULL)) == 33752961


def test_forward_small_matches_reference(golden):
    g = golden("g11a_sep_forward_small")
    sd = make_sep_state_dict(SEP_SMALL, 31)
    for t in (2048, 2100):
        rng = np.random.default_rng(500 + t)
        x = torch.from_numpy(rng.standard_normal((2, 21, t)).astype(np.float32))
        y = sep_ref.sep_forward(sd, SEP_SMALL, x, 3).numpy()
        want = g[f"y_t{t}"]
        assert y.shape == want.shape == (2, SEP_SMALL.max_speakers, t)
        assert np.all(y[:, 3:] == 0) and np.all(want[:, 3:] == 0)          # rows padded to max_speakers
        assert _snr(y, want) > 100, _snr(y, want)


def test_forward_with_different_speaker_counts_matches_reference(golden):
    """g11d: the reference's Network.forward with 3 / 1 / 2 and 2 / 3 speakers per item (speakers_to_batches /
    batches_to_speakers, :236-268): a missing speaker is a zero sequence in every inter-speaker layer and leaves as
    the bare output_decoder bias; rows beyond the largest count are zero."""
    g = golden("g11d_sep_forward_ragged")
    sd = make_sep_state_dict(SEP_SMALL, 31)
    for name, t in (("a", 2100), ("b", 2048)):
        counts = [int(c) for c in g[f"counts_{name}"]]
        rng = np.random.default_rng(900 + t)
        x = torch.from_numpy(rng.standard_normal((len(counts), 21, t)).astype(np.float32))
        y = sep_ref.sep_forward(sd, SEP_SMALL, x, counts).numpy()
        want = g[f"y_{name}"]
        assert y.shape == want.shape == (len(counts), SEP_SMALL.max_speakers, t)
        assert _snr(y, want) > 100, _snr(y, want)
        bias = float(sd["output_decoder.bias"][0])
        for b, c in enumerate(counts):
            assert np.all(want[b, c:3] == np.float32(bias)) and np.all(y[b, c:3] == np.float32(bias))
        assert np.all(y[:, 3:] == 0) and np.all(want[:, 3:] == 0)


def test_infer_sample_small_matches_reference(golden):
    g = golden("g11b_sep_infer_small")
    sd = make_sep_state_dict(SEP_SMALL, 31)
    mix = torch.from_numpy(make_scene(4, 3, 7, 4000).mix)
    for i in range(3):
        samples = g[f"samples{i}"]
        y = sep_ref.infer_sample(sd, SEP_SMALL, mix, list(samples))
        assert y.shape == g[f"y{i}"].shape == (len(samples), 4000)
        assert _snr(y, g[f"y{i}"]) > 100, (i, _snr(y, g[f"y{i}"]))


def test_infer_sample_full_matches_reference(golden):
    g = golden("g11c_sep_infer_full")
    sd = make_sep_state_dict(SEP_FULL, 9)
    mix = torch.from_numpy(make_scene(6, 3, 7, 9600).mix)
    y = sep_ref.infer_sample(sd, SEP_FULL, mix, list(g["samples"]))
    assert _snr(y, g["y"]) > 90, _snr(y, g["y"])


def test_matcher_reproduces_reference(golden):
    g = golden("g12_best_permutation")
    for k in range(int(g["n_cases"])):
        got = evalkit.find_best_permutation(g[f"wav_gt{k}"].astype(np.float64), g[f"wav_pred{k}"].astype(np.float64),
                                            g[f"pos_gt{k}"], g[f"pos_pred{k}"])
        want = [tuple(r) for r in g[f"best{k}"].tolist()]
        assert sorted(got) == sorted(want), (k, got, want)
