"""Golden vectors of the joint separation network (fixtures g11*) and of the evaluation
matcher (g12), produced by running the REFERENCE's own code from /root/reference:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_sep.py [--only g11a,...]

``sep/training/SpeakerSeparation/network.py`` imports two classes from speechbrain
(ConformerEncoder, RelPosEncXL), which is absent from the image and unpinned upstream.  They
are supplied by THIS repo's restatement of the published speechbrain definitions
(tests/golden/thirdparty_restated.py); everything else -- Network.__init__/forward/infer_sample,
Encoder/Decoder, speakers_to_batches / batches_to_speakers, the inter-speaker
nn.TransformerEncoderLayer, the mask path -- is the reference's code, run unmodified.  The
fixtures therefore pin all of that and pin oracle/sep_ref.py to the restated Conformer; the
Conformer arithmetic against speechbrain itself stays "parity unpinned".

Weights are this repo's seeded generator (weights.make_sep_state_dict), loaded with
``load_state_dict(strict=True)``, so only seeds, inputs' seeds and outputs are stored.
"""
import argparse
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from tests.golden.make_golden import _save, _stub, install_stubs  # noqa: E402


def install_sep_stubs():
    from tests.golden import thirdparty_restated as tp
    install_stubs()
    _stub("speechbrain.lobes.models.transformer.Conformer", ConformerEncoder=tp.ConformerEncoder)
    _stub("speechbrain.nnet.attention", RelPosEncXL=tp.RelPosEncXL)


def _ref_sep_network(cfg, seed):
    from sep.training.SpeakerSeparation.network import Network
    from acousticswarms_speech_amd.weights import make_sep_state_dict
    net = Network(device="cpu", n_mics=cfg.n_mics, max_speakers=cfg.max_speakers, kernel_size=cfg.kernel_size,
                  stride_list=list(cfg.stride_list), channels=cfg.channels, growth=cfg.growth,
                  encoder_channels=cfg.encoder_channels, encoder_kernel_size=cfg.encoder_kernel_size,
                  encoder_stride=cfg.encoder_stride, residual_layers=cfg.residual_layers,
                  residual_dilation_factor=cfg.residual_dilation_factor, num_head=cfg.num_head, ffw_dim=cfg.ffw_dim,
                  bottleneck_layers=cfg.bottleneck_layers, bottleneck_ksize=cfg.bottleneck_ksize)
    sd = {k: torch.from_numpy(v) for k, v in make_sep_state_dict(cfg, seed).items()}
    net.load_state_dict(sd, strict=True)
    net.eval()
    return net


SMALL_SAMPLES = [
    [[0, 0, 0, 0, 0, 0], [5, -9, 14, -22, 31, -40]],
    [[3.4, -1.6, 2.5, -2.5, 7.49, -7.51], [131, -131, 7, -7, 64, -64], [-40, 30, -20, 10, -5, 2]],
    [[0, 0, 0, 0, 0, 0], [1, 1, 1, 1, 1, 1], [-1, -1, -1, -1, -1, -1], [2500, -2500, 3999, -3999, 4000, -4100],
     [17, -3, 8, 0, -12, 40], [60, 50, 40, 30, 20, 10]],
]
FULL_SAMPLES = [[12, -20, 33, -41, 57, -8], [-25.5, 14.2, 3.7, -64.0, 90.49, 11.5], [0, 0, 0, 0, 0, 0]]


def g11a():
    """Network.forward (normalised input), SEP_SMALL: B=2 items of S=3 speakers, t multiple /
    non-multiple of the stride product; fewer speakers than max_speakers (padded rows)."""
    from acousticswarms_speech_amd.config import SEP_SMALL
    net = _ref_sep_network(SEP_SMALL, seed=31)
    out = {}
    for t in (2048, 2100):
        rng = np.random.default_rng(500 + t)
        x = torch.from_numpy(rng.standard_normal((2, 3 * 7, t)).astype(np.float32))
        with torch.no_grad():
            out[f"y_t{t}"] = net(x, torch.tensor([[3], [3]])).numpy()
    _save("g11a_sep_forward_small", **out)


def g11d():
    """Network.forward with DIFFERENT speaker counts per batch item (3, 1 and 2 of a 3-wide stack, and 2 / 3):
    speakers_to_batches / batches_to_speakers (:236-268) drop and re-insert the missing speakers, whose zero rows take
    part in the inter-speaker attention of every bottleneck layer, and whose outputs are the bare output_decoder bias."""
    from acousticswarms_speech_amd.config import SEP_SMALL
    net = _ref_sep_network(SEP_SMALL, seed=31)
    out = {}
    for name, counts, t in (("a", [3, 1, 2], 2100), ("b", [2, 3], 2048)):
        rng = np.random.default_rng(900 + t)
        x = torch.from_numpy(rng.standard_normal((len(counts), 3 * 7, t)).astype(np.float32))
        with torch.no_grad():
            out[f"y_{name}"] = net(x, torch.tensor(counts).view(-1, 1)).numpy()
        out[f"counts_{name}"] = np.array(counts, dtype=np.int64)
    _save("g11d_sep_forward_ragged", **out)


def g11b():
    """Network.infer_sample, SEP_SMALL, on a seeded scene: 2, 3 and 6 (> max_speakers) speakers,
    fractional offsets (rounded half-to-even by the reference) and offsets beyond the clip length."""
    from acousticswarms_speech_amd.config import SEP_SMALL
    from acousticswarms_speech_amd.scenes import make_scene
    net = _ref_sep_network(SEP_SMALL, seed=31)
    mix = torch.from_numpy(make_scene(4, 3, 7, 4000).mix)
    out = {}
    for i, samples in enumerate(SMALL_SAMPLES):
        out[f"y{i}"] = net.infer_sample(mix.clone(), [np.array(s, dtype=np.float64) for s in samples])
        out[f"samples{i}"] = np.array(samples, dtype=np.float64)
    _save("g11b_sep_infer_small", **out)


def g11c():
    """Network.infer_sample with the FULL separation network (33.75 M parameters), T = 9600, three
    speakers; output + per-block activation probes captured with forward hooks."""
    from acousticswarms_speech_amd.config import SEP_FULL
    from acousticswarms_speech_amd.scenes import make_scene
    net = _ref_sep_network(SEP_FULL, seed=9)
    mix = torch.from_numpy(make_scene(6, 3, 7, 9600).mix)
    taps = {}

    def hook(name):
        def f(_m, _i, o):
            taps[name] = (o[0] if isinstance(o, tuple) else o).detach()
        return f
    hs = [net.bottleneck.register_forward_hook(hook("bottleneck"))]
    for i, b in enumerate(net.encoder.module_list):
        hs.append(b.register_forward_hook(hook(f"enc{i}")))
    for i, b in enumerate(net.decoder.module_list):
        hs.append(b.register_forward_hook(hook(f"dec{i}")))
    for l, layer in enumerate(net.bottleneck.module_list):
        hs.append(layer["intra"].register_forward_hook(hook(f"intra{l}")))
        hs.append(layer["inter"].register_forward_hook(hook(f"inter{l}")))
    y = net.infer_sample(mix.clone(), [np.array(s, dtype=np.float64) for s in FULL_SAMPLES])
    for h in hs:
        h.remove()
    arrs = {"y": y, "samples": np.array(FULL_SAMPLES, dtype=np.float64)}
    for k, v in taps.items():
        v = v.reshape(-1, *v.shape[-2:])
        arrs[f"{k}_shape"] = np.array(v.shape)
        arrs[f"{k}_l2"] = v.pow(2).sum((1, 2)).sqrt().numpy()
        idx = np.linspace(0, v.shape[-1] - 1, 8).astype(np.int64)
        arrs[f"{k}_probe"] = v[:, ::max(1, v.shape[1] // 16), :][:, :, idx].numpy()
    _save("g11c_sep_infer_full", **arrs)


def g12():
    """find_best_permutation (sep/eval/eval_model.py:18-59) on seeded inputs."""
    from sep.eval.eval_model import find_best_permutation
    rng = np.random.default_rng(12)
    cases = {}
    for trial in range(24):
        n_gt, n_pred = int(rng.integers(1, 5)), int(rng.integers(1, 6))
        # inputs are float32-representable so the fixture can store them in half the space
        wav_gt = rng.standard_normal((n_gt, 256)).astype(np.float32).astype(np.float64)
        pos_gt = rng.uniform(-2, 2, (n_gt, 3))
        src = rng.integers(0, n_gt, n_pred)
        wav_pred = wav_gt[src] + rng.uniform(0.05, 3.0, (n_pred, 1)) * rng.standard_normal((n_pred, 256))
        wav_pred = wav_pred.astype(np.float32).astype(np.float64)
        pos_pred = pos_gt[src] + rng.uniform(0.0, 0.9, (n_pred, 1)) * rng.standard_normal((n_pred, 3))
        best = find_best_permutation(wav_gt, wav_pred, pos_gt, pos_pred)
        cases[f"wav_gt{trial}"], cases[f"wav_pred{trial}"] = wav_gt.astype(np.float32), wav_pred.astype(np.float32)
        cases[f"pos_gt{trial}"], cases[f"pos_pred{trial}"] = pos_gt, pos_pred
        cases[f"best{trial}"] = np.array(best, dtype=np.int64).reshape(-1, 2)
    _save("g12_best_permutation", n_cases=np.array(24), **cases)


ALL = {"g11a": g11a, "g11b": g11b, "g11c": g11c, "g11d": g11d, "g12": g12}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    install_sep_stubs()
    torch.manual_seed(0)
    for name in ([n for n in a.only.split(",") if n] or list(ALL)):
        ALL[name]()
