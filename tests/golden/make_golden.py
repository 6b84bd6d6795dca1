"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own
code (imported read-only from /root/reference) on seeded inputs and weights.

Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--only g1,g2,...]

The reference imports a number of packages that are absent here and unused on the
hot path (librosa, soundfile, torchaudio, ...).  They are replaced by inert stub
modules so that the module-level ``import`` statements succeed; no stub supplies
arithmetic, with two documented exceptions used only by the search-stage
fixtures (g7/g8/g10): ``pyroomacoustics.transform.stft.analysis`` and
``librosa.feature.rms`` / ``librosa.effects.split`` are OUR restatements (the
third-party originals are absent), so those fixtures pin everything except that
third-party framing ("parity unpinned" for the framing itself).

Fixtures store only seeds + outputs (+ small probes); weights and inputs are
regenerated from the seeds by acousticswarms_speech_amd.{weights,scenes}.
"""
import argparse
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402


# --------------------------------------------------------------------------
# inert stubs for absent, unused third-party imports
# --------------------------------------------------------------------------
def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    from tests.golden import thirdparty_restated as tp
    for n in ["soundfile", "noisereduce", "torchaudio", "mir_eval", "cv2", "seaborn", "opuslib",
              "speechbrain", "speechbrain.lobes", "speechbrain.lobes.models",
              "speechbrain.lobes.models.transformer",
              "speechbrain.lobes.models.transformer.Conformer",
              "speechbrain.nnet", "speechbrain.nnet.attention"]:
        _stub(n)
    _stub("asteroid")
    _stub("asteroid.losses")
    _stub("asteroid.losses.sdr", SingleSrcNegSDR=type("SingleSrcNegSDR", (), {"__init__": lambda s, *a, **k: None}))
    _stub("asteroid.metrics", get_metrics=lambda *a, **k: {})
    lib = _stub("librosa")
    lib.feature = _stub("librosa.feature", rms=tp.librosa_rms)
    lib.effects = _stub("librosa.effects", split=tp.librosa_split)
    pra = _stub("pyroomacoustics")
    pra.transform = _stub("pyroomacoustics.transform")
    pra.transform.stft = _stub("pyroomacoustics.transform.stft", analysis=tp.pra_stft_analysis)


# --------------------------------------------------------------------------
def _save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def _ref_network(cfg, seed):
    """Reference Network with OUR seeded weights loaded strictly."""
    from sep.training.SpeakerLocalization.network import Network
    from acousticswarms_speech_amd.weights import make_spot_state_dict
    net = Network(n_mics=cfg.n_mics, kernel_size=cfg.kernel_size, stride_list=list(cfg.stride_list),
                  channels=cfg.channels, growth=cfg.growth, encoder_channels=cfg.encoder_channels,
                  encoder_kernel_size=cfg.encoder_kernel_size, encoder_stride=cfg.encoder_stride,
                  residual_layers=cfg.residual_layers,
                  residual_dilation_factor=cfg.residual_dilation_factor, num_head=cfg.num_head,
                  ffw_dim=cfg.ffw_dim, num_transformer_layers=cfg.num_transformer_layers)
    sd = {k: torch.from_numpy(v) for k, v in make_spot_state_dict(cfg, seed).items()}
    net.load_state_dict(sd, strict=True)
    net.eval()
    return net


G1_OFFSETS = np.array([[0, 0, 0, 0, 0, 0], [1, -1, 2, -2, 3, -3], [131, -131, 7, -7, 64, -64],
                       [-1, -1, -1, -1, -1, -1], [2500, -2500, 4799, -4799, 2400, -2400],
                       [5, 9, -14, 22, -31, 40], [-140, 140, -139, 139, -138, 138],
                       [3, 1, 4, 1, 5, 9]], dtype=np.int64)
PROBES = np.array([0, 1, 2, 3, 17, 255, 256, 1023, 2047, 2399, 2400, 2401, 3333, 4095, 4798, 4799])


def g1():
    """roll_by_gather + normalize_input (JointModel/network.py:12-25,80-90)."""
    from sep.training.JointModel.network import roll_by_gather
    from sep.training.SpeakerLocalization.network import normalize_input
    from acousticswarms_speech_amd.scenes import make_scene
    mix = torch.from_numpy(make_scene(0, 2, 7, 4800).mix)
    data = torch.zeros((len(G1_OFFSETS), 7, 4800))
    for j, off in enumerate(G1_OFFSETS):
        shifts = torch.round(-torch.Tensor([0, *off]).unsqueeze(1)).long()
        data[j] = roll_by_gather(mix, 1, shifts)
    dn, mu, sg = normalize_input(data)
    _save("g1_shift_norm", offsets=G1_OFFSETS, probes=PROBES, rolled_probe=data[:, :, PROBES].numpy(),
          norm_probe=dn[:, :, PROBES].numpy(), mean=mu.flatten().numpy(), std=sg.flatten().numpy(),
          norm_l2=dn.pow(2).sum(-1).sqrt().numpy(), norm_sum=dn.sum(-1).numpy())


def _net_inputs(seed, B, M, T):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.standard_normal((B, M, T)).astype(np.float32))


def g2():
    """Spot Network.forward, tiny config, both windows, T multiple / non-multiple of 256."""
    from acousticswarms_speech_amd.config import TINY
    net = _ref_network(TINY, seed=11)
    out = {}
    for T in (4800, 5000):
        x = _net_inputs(100 + T, 2, 7, T)
        for wi, w in enumerate(([1.0, 0.0], [0.0, 1.0])):
            wemb = torch.tensor([w, w])
            with torch.no_grad():
                out[f"y_T{T}_w{wi}"] = net(x, wemb).numpy()
    # mixed / non-one-hot embedding row (general Network.forward contract)
    x = _net_inputs(777, 2, 7, 4800)
    wemb = torch.tensor([[1.0, 0.0], [0.25, 0.75]])
    with torch.no_grad():
        out["y_mixed"] = net(x, wemb).numpy()
    _save("g2_spot_tiny", **out)


def g2b():
    """Spot Network.forward, SMALL config (the smallest the MFMA tiles accept), for the
    GPU parity tests: both windows, T = 4800 and 5000."""
    from acousticswarms_speech_amd.config import SMALL
    net = _ref_network(SMALL, seed=21)
    out = {}
    for T in (4800, 5000):
        x = _net_inputs(200 + T, 3, 7, T)
        for wi, w in enumerate(([1.0, 0.0], [0.0, 1.0])):
            with torch.no_grad():
                out[f"y_T{T}_w{wi}"] = net(x, torch.tensor([w] * 3)).numpy()
    _save("g2b_spot_small", **out)


def g4b():
    """DataParallelSpotModel.shift_and_sep with the FULL network (47.27 M parameters) on a
    3-talker scene, T = 6000, 5 candidates, Strict 0/1 -- the GPU path's end-to-end pin."""
    from sep.training.JointModel.network import DataParallelSpotModel
    from acousticswarms_speech_amd.config import FULL
    from acousticswarms_speech_amd.scenes import make_scene
    net = _ref_network(FULL, seed=5)
    model = DataParallelSpotModel(net, use_fp16=False, batch_size=4)
    mix = torch.from_numpy(make_scene(2, 3, 7, 6000).mix)
    offs = np.array([[0, 0, 0, 0, 0, 0], [4, -8, 12, -16, 20, -24], [-33, 45, -57, 61, -73, 87],
                     [100, -100, 50, -50, 25, -25], [-139, 96, -72, 48, -24, 12]], dtype=np.float64)
    patches = [_P(o) for o in offs]
    y0 = model.shift_and_sep(mix, patches, Strict=0)
    y1 = model.shift_and_sep(mix, patches, Strict=1)
    _save("g4b_shift_and_sep_full", offsets=offs, y_strict0=y0, y_strict1=y1)


def g3():
    """Spot Network.forward, FULL config (47.27 M params), T=12288, B=2; output +
    per-block activation probes captured with forward hooks."""
    from acousticswarms_speech_amd.config import FULL
    net = _ref_network(FULL, seed=5)
    x = _net_inputs(31, 2, 7, 12288)
    wemb = torch.tensor([[0.0, 1.0], [0.0, 1.0]])
    taps = {}

    def hook(name):
        def f(_m, _i, o):
            taps[name] = o.detach()
        return f
    hs = [net.preproc.register_forward_hook(hook("preproc")),
          net.bottleneck.register_forward_hook(hook("bottleneck"))]
    for i, b in enumerate(net.encoder.module_list):
        hs.append(b.register_forward_hook(hook(f"enc{i}")))
    for i, b in enumerate(net.decoder.module_list):
        hs.append(b.register_forward_hook(hook(f"dec{i}")))
    with torch.no_grad():
        y = net(x, wemb).numpy()
    for h in hs:
        h.remove()
    arrs = {"y": y}
    for k, v in taps.items():
        T = v.shape[-1]
        idx = np.linspace(0, T - 1, 16).astype(np.int64)
        arrs[f"{k}_l2"] = v.pow(2).sum((1, 2)).sqrt().numpy()
        arrs[f"{k}_probe"] = v[:, :, idx].numpy()
        arrs[f"{k}_idx"] = idx
    _save("g3_spot_full", **arrs)


class _P:  # minimal stand-in for the reference Patch at the shift_and_sep boundary
    def __init__(self, off):
        self.sample_offset = np.asarray(off)


def g4():
    """DataParallelSpotModel.shift_and_sep (JointModel/network.py:37-104), tiny net,
    6 candidates, Strict 0 and 1, batch_size 4 (so the ragged last batch is hit)."""
    from sep.training.JointModel.network import DataParallelSpotModel
    from acousticswarms_speech_amd.config import TINY
    from acousticswarms_speech_amd.scenes import make_scene
    net = _ref_network(TINY, seed=11)
    model = DataParallelSpotModel(net, use_fp16=False, batch_size=4)
    mix = torch.from_numpy(make_scene(1, 3, 7, 4800).mix)
    offs = np.array([[0, 0, 0, 0, 0, 0], [4, -8, 12, -16, 20, -24], [-3, 5, -7, 11, -13, 17],
                     [100, -100, 50, -50, 25, -25], [1.4, -1.6, 2.5, 3.5, -0.5, 0.49],
                     [-120, 96, -72, 48, -24, 12]], dtype=np.float64)
    patches = [_P(o) for o in offs]
    y0 = model.shift_and_sep(mix, patches, Strict=0)
    y1 = model.shift_and_sep(mix, patches, Strict=1)
    _save("g4_shift_and_sep", offsets=offs, y_strict0=y0, y_strict1=y1)


def g5():
    """max_avg_power + stage power loop (local_utils_3d.py:13-17,349-354)."""
    from sep.helpers.local_utils_3d import max_avg_power
    d = np.load(os.path.join(HERE, "g4_shift_and_sep.npz"))
    rows = []
    for y in (d["y_strict0"], d["y_strict1"]):
        for i in range(y.shape[0]):
            x = y[i] - np.mean(y[i])
            p2, _ = max_avg_power(x)
            p2s, _ = max_avg_power(x, window_size=1000)
            rows.append([np.sum(x ** 2), p2, p2s])
    rng = np.random.default_rng(9)
    z = (rng.standard_normal(30000) * np.hanning(30000)).astype(np.float32)
    zp, _ = max_avg_power(z)
    _save("g5_energies", rows=np.array(rows, dtype=np.float64), z_power2=np.float64(zp))


def g9():
    """si_sdr / check_sisnr_win / weight_mean_pos (eval_utils.py:11-39, Mic_Array.py:18-47)."""
    from sep.helpers.eval_utils import si_sdr
    from sep.Mic_Array import check_sisnr_win, weight_mean_pos
    from sep.Traditional_SP.Patch_3D import Patch
    rng = np.random.default_rng(4)
    a = rng.standard_normal((5, 3000)).astype(np.float32)
    a[1] = 0.7 * a[0] + 0.1 * a[1]
    a[3] = -a[2]
    S = np.array([[si_sdr(a[i], a[j]) for j in range(5)] for i in range(5)])
    wins = [[-1.0, -3.0], [-3.0, -8.0], [-1.5, -6.9], [5.0], [-2.0, -2.0]]
    cw = np.array([check_sisnr_win(w) for w in wins])
    cw2 = np.array([check_sisnr_win(w, SISNR_THRESHOLD=-1, SISNR_THRESHOLD2=-5) for w in wins])
    patches = [Patch(rng.integers(-20, 20, 6).astype(float), [4] * 6, None, rng.uniform(0, 3, 3))
               for _ in range(6)]
    powers = [5.0, 4.5, 1.0, 3.9, 3.7, 0.2]
    pos, off = weight_mean_pos(patches, powers, [0, 1, 2, 3, 4])
    _save("g9_sisdr", sig_seed=np.int64(4), S=S, check_win=cw, check_win2=cw2,
          wm_offsets=np.stack([p.sample_offset for p in patches]),
          wm_peaks=np.stack([p.peak_pos for p in patches]), wm_powers=np.array(powers),
          wm_pos=pos, wm_off=off)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    install_stubs()
    torch.set_num_threads(8)
    from tests.golden import make_golden_search as mgs
    todo = {"g1": g1, "g2": g2, "g2b": g2b, "g3": g3, "g4": g4, "g4b": g4b, "g5": g5, "g9": g9}
    todo.update(mgs.GENERATORS)
    sel = [s for s in args.only.split(",") if s] or list(todo)
    for k in sel:
        print("==", k)
        todo[k]()


if __name__ == "__main__":
    main()
