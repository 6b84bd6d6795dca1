"""GPU: the joint separation network (SURVEY.md §8 a-S, f-1) -- every new kernel against a plain
PyTorch fp32 CPU statement of the same op, and the whole network (forward / infer_sample) against
the fixtures produced by the reference's own ``Network`` (g11a-c) and against the oracle.
Needs an MI355X."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _log(msg):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "diag_sep.txt"), "a") as f:
        f.write(msg + "\n")
    print(msg)


def _relerr(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _snr(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return 10 * np.log10(np.sum(want ** 2) / max(np.sum((got - want) ** 2), 1e-300))


def _rand(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.fixture(scope="module")
def ops():
    from acousticswarms_speech_amd import ops as o
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return o


# ---------------------------------------------------------------- kernels
def test_joint_shift_stats(ops):
    from acousticswarms_speech_amd.scenes import make_scene
    from oracle import spot_ref
    mix = torch.from_numpy(make_scene(4, 3, 7, 4000).mix)
    offs = np.array([[0, 0, 0, 0, 0, 0], [131, -131, 7, -7, 64, -64], [2500, -2500, 3999, -3999, 4000, -4100],
                     [17, -3, 8, 0, -12, 40]], dtype=np.int32)
    for S in (1, 4):
        data = torch.cat([spot_ref.roll_channels(mix, o, circular=False) for o in offs[:S]], dim=0).unsqueeze(0)
        _dn, mu, sg = spot_ref.normalize_input(data)
        mean, std = ops.joint_shift_stats(mix.cuda(), torch.from_numpy(offs[:S]).cuda())
        assert mean.shape == (S,) and torch.all(mean == mean[0]) and torch.all(std == std[0])
        # double accumulation vs torch's float reductions: 2e-6 relative
        np.testing.assert_allclose(mean[0].item(), mu.item(), rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(std[0].item(), sg.item(), rtol=2e-6)


@pytest.mark.parametrize("N", [128, 512, 1024])
def test_add_layernorm2(ops, N):
    rows = 37
    x, y = _rand(rows, N, seed=1), _rand(rows, N, seed=2)
    g, b = 1 + 0.1 * _rand(N, seed=3), 0.1 * _rand(N, seed=4)
    s, o = ops.add_layernorm2(x.cuda(), y.cuda(), 0.5, g.cuda(), b.cuda(), eps=1e-5, want_sum=True)
    assert torch.equal(s.cpu(), x + 0.5 * y) or _relerr(s.cpu(), x + 0.5 * y) < 1e-7
    assert _relerr(o.cpu(), F.layer_norm(x + 0.5 * y, (N,), g, b, 1e-5)) < 2e-6
    _s, o2 = ops.add_layernorm2(x.cuda(), None, 0.0, g.cuda(), b.cuda(), eps=1e-6, act=2)
    want = F.layer_norm(x, (N,), g, b, 1e-6)
    assert _relerr(o2.cpu(), want * torch.sigmoid(want)) < 2e-6


def test_glu_and_dwconv(ops):
    raw = _rand(50, 256, seed=5)
    assert _relerr(ops.glu_rows(raw.cuda()).cpu(), F.glu(raw, dim=1)) < 1e-6
    for d, K, L in ((512, 31, 75), (128, 7, 40)):
        u = _rand(3, L, d, seed=6)
        w, bias = _rand(d, 1, K, seed=7, scale=0.3), _rand(d, seed=8, scale=0.1)
        g, b = 1 + 0.1 * _rand(d, seed=9), 0.1 * _rand(d, seed=10)
        h = F.conv1d(u.transpose(1, 2), w, bias, padding=(K - 1) // 2, groups=d).transpose(1, 2)
        h = F.layer_norm(h, (d,), g, b, 1e-5)
        want = h * torch.sigmoid(h)
        got = ops.dwconv_ln_swish(u.cuda(), w.cuda(), bias.cuda(), g.cuda(), b.cuda())
        assert _relerr(got.cpu(), want) < 2e-6


@pytest.mark.parametrize("L,d,H", [(75, 128, 8), (200, 256, 8), (130, 512, 8), (64, 512, 8), (750, 512, 8)])
def test_relpos_attention(ops, L, d, H):
    """RelPosMHAXL core against the pad-and-reshape rel_shift statement."""
    B, hd = 2, d // H
    qkv = _rand(B, L, 3 * d, seed=11, scale=0.7)
    P = _rand(2 * L - 1, d, seed=12, scale=0.7)
    bu, bv = _rand(d, seed=13, scale=0.3), _rand(d, seed=14, scale=0.3)
    scale = 1.0 / math.sqrt(d)
    q, k, v = [t.view(B, L, H, hd) for t in qkv.split(d, dim=-1)]
    qu = (q + bu.view(1, 1, H, hd)).transpose(1, 2)
    qv = (q + bv.view(1, 1, H, hd)).transpose(1, 2)
    ac = torch.matmul(qu * scale, k.permute(0, 2, 3, 1))
    bd = torch.matmul(qv * scale, P.view(1, -1, H, hd).permute(0, 2, 3, 1))
    bd = F.pad(bd, (1, 0)).view(B, H, -1, L)[:, :, 1:].view(B, H, L, 2 * L - 1)[..., :L]      # rel_shift
    att = torch.softmax(ac + bd, dim=-1)
    want = torch.matmul(att, v.transpose(1, 2)).transpose(1, 2).reshape(B, L, d)
    got = ops.relpos_attention(qkv.cuda(), P.cuda(), bu.cuda(), bv.cuda(), H, scale)
    rel = _relerr(got.cpu(), want)
    _log(f"relpos attention L={L} d={d}: rel {rel:.2e}")
    assert rel < 3e-6


@pytest.mark.parametrize("S", [1, 3, 6, 27])
def test_inter_attention(ops, S):
    NB, L, d, H = 2, 33, 512, 8
    hd = d // H
    qkv = _rand(NB, S, L, 3 * d, seed=15, scale=0.7)
    q, k, v = [t.permute(0, 2, 1, 3).reshape(NB * L, S, H, hd).transpose(1, 2) for t in qkv.split(d, dim=-1)]
    att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), dim=-1)
    want = (att @ v).transpose(1, 2).reshape(NB, L, S, d).permute(0, 2, 1, 3)
    got = ops.inter_attention(qkv.cuda(), H)
    assert _relerr(got.cpu(), want) < 2e-6


def test_swish_epilogue(ops):
    x = _rand(1, 200, 128, seed=16)
    w, b = _rand(256, 128, seed=17, scale=0.1), _rand(256, seed=18, scale=0.1)
    out, _ = ops.convgemm(x.cuda(), w.cuda(), 200, 256, 128, bias=b.cuda(), relu=2)
    h = F.linear(x, w, b)
    assert _relerr(out.cpu(), h * torch.sigmoid(h)) < 2e-6


# ---------------------------------------------------------------- the network
def _model(cfg, seed, precision="f32"):
    from acousticswarms_speech_amd.sep import SepModel
    from acousticswarms_speech_amd.weights import make_sep_state_dict
    sd = make_sep_state_dict(cfg, seed)
    return SepModel(cfg, sd, precision=precision).to("cuda"), sd


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_forward_small_vs_reference_golden(golden, precision):
    from acousticswarms_speech_amd.config import SEP_SMALL
    g = golden("g11a_sep_forward_small")
    model, _sd = _model(SEP_SMALL, 31, precision)
    for t in (2048, 2100):
        rng = np.random.default_rng(500 + t)
        x = torch.from_numpy(rng.standard_normal((2, 21, t)).astype(np.float32))
        y = model(x, torch.tensor([[3], [3]])).cpu().numpy()
        want = g[f"y_t{t}"]
        assert y.shape == want.shape and np.all(y[:, 3:] == 0)
        snr = _snr(y, want)
        _log(f"sep forward SMALL {precision} t={t}: {snr:.1f} dB vs reference")
        assert snr > 80.0
    with pytest.raises(RuntimeError):
        model(x, torch.tensor([[3], [3], [3]]))               # one count per batch item


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_forward_with_different_speaker_counts_vs_reference_golden(golden, precision):
    """g11d: 3 / 1 / 2 and 2 / 3 speakers per item through the reference's own Network.forward
    (speakers_to_batches / batches_to_speakers, SpeakerSeparation/network.py:236-268) -- the HIP path runs every item
    as wide as the largest count, zeroes the missing speakers' rows before every inter-speaker layer and writes the
    bare decoder bias for them; the channel blocks of missing speakers in the input are ignored."""
    from acousticswarms_speech_amd.config import SEP_SMALL
    from oracle import sep_ref
    g = golden("g11d_sep_forward_ragged")
    model, sd = _model(SEP_SMALL, 31, precision)
    for name, t in (("a", 2100), ("b", 2048)):
        counts = [int(c) for c in g[f"counts_{name}"]]
        rng = np.random.default_rng(900 + t)
        x = torch.from_numpy(rng.standard_normal((len(counts), 21, t)).astype(np.float32))
        y = model(x, torch.tensor(counts).view(-1, 1)).cpu().numpy()
        want = g[f"y_{name}"]
        assert y.shape == want.shape
        snr = _snr(y, want)
        snr_o = _snr(y, sep_ref.sep_forward(sd, SEP_SMALL, x, counts).numpy())
        _log(f"sep forward ragged counts={counts} {precision}: {snr:.1f} dB vs reference, {snr_o:.1f} dB vs oracle")
        assert snr > 80.0 and snr_o > 80.0
        bias = np.float32(sd["output_decoder.bias"][0])
        for b, c in enumerate(counts):
            assert np.all(y[b, c:3] == bias)
        assert np.all(y[:, 3:] == 0)
        # what sits in a missing speaker's channel block does not matter
        x2 = x.clone()
        for b, c in enumerate(counts):
            x2[b, c * 7:] = 123.0
        np.testing.assert_array_equal(model(x2, torch.tensor(counts).view(-1, 1)).cpu().numpy(), y)


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_infer_sample_small_vs_reference_golden(golden, precision):
    from acousticswarms_speech_amd.config import SEP_SMALL
    from acousticswarms_speech_amd.scenes import make_scene
    g = golden("g11b_sep_infer_small")
    model, _sd = _model(SEP_SMALL, 31, precision)
    mix = torch.from_numpy(make_scene(4, 3, 7, 4000).mix)
    for i in range(3):
        y = model.infer_sample(mix, list(g[f"samples{i}"]))
        assert y.shape == g[f"y{i}"].shape and y.dtype == np.float32
        snr = _snr(y, g[f"y{i}"])
        _log(f"sep infer_sample SMALL {precision} case {i} (S={y.shape[0]}): {snr:.1f} dB vs reference")
        assert snr > 80.0
    assert model.infer_sample(mix, []).shape == (0, 4000)

    class P:
        def __init__(self, o):
            self.sample_offset = o
    y2 = model.infer(mix, [P(o) for o in g["samples1"]])
    np.testing.assert_array_equal(y2, model.infer_sample(mix, list(g["samples1"])))


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_infer_sample_full_vs_reference_golden(golden, precision):
    """FULL separation network (33.75 M parameters): output and every block tap."""
    from acousticswarms_speech_amd.config import SEP_FULL
    from acousticswarms_speech_amd.scenes import make_scene
    g = golden("g11c_sep_infer_full")
    model, _sd = _model(SEP_FULL, 9, precision)
    mix = torch.from_numpy(make_scene(6, 3, 7, 9600).mix)
    y = model.infer_sample(mix, list(g["samples"]))
    worst = 1e9
    for name in [f"enc{i}" for i in range(4)] + [f"intra{l}" for l in range(3)] + [f"inter{l}" for l in range(3)] + \
            ["bottleneck"] + [f"dec{i}" for i in range(4)]:
        shp = tuple(int(v) for v in g[f"{name}_shape"])                  # reference layout [rows, C, T] (or [rows, T, C])
        tap = model.get_tap(name).cpu()
        if name.startswith("intra"):
            ref_l2 = g[f"{name}_l2"]                                      # conformer output [B*S, L, d]
            got = tap.view(shp[0], shp[1], shp[2])
        elif name.startswith("inter"):
            # reference hook sees [N*L, S, d]; ours is [S, L, d]
            got = tap.view(shp[1], shp[0], shp[2]).permute(1, 0, 2).contiguous()
            ref_l2 = g[f"{name}_l2"]
        elif name == "bottleneck":
            got = tap.view(shp[0], shp[2], shp[1]).permute(0, 2, 1)       # reference [N*S? , d, L] flattened
            ref_l2 = g[f"{name}_l2"]
        else:
            got = tap.view(shp[0], shp[2], shp[1]).permute(0, 2, 1)       # channels-last -> [rows, C, T]
            ref_l2 = g[f"{name}_l2"]
        l2 = got.double().pow(2).sum((1, 2)).sqrt().numpy()
        rel = float(np.abs(l2 - ref_l2).max() / np.abs(ref_l2).max())
        worst = min(worst, -20 * math.log10(max(rel, 1e-12)))
        assert rel < 1e-3, (name, rel)
    snr = _snr(y, g["y"])
    _log(f"sep infer_sample FULL {precision}: {snr:.1f} dB vs reference; worst tap norm agreement {worst:.0f} dB")
    assert snr > 80.0


def test_infer_full_size_vs_oracle_and_properties():
    """configs[2] shape: 5 speakers, T = 48 000, FULL network in the bench arithmetic against the
    CPU oracle; plus properties that hold at any size: permuting the speakers permutes the rows
    (joint statistics and the inter-speaker attention are permutation-equivariant), a second call
    is bit-identical."""
    from acousticswarms_speech_amd.config import SEP_FULL
    from acousticswarms_speech_amd.scenes import make_scene
    from oracle import sep_ref
    model, sd = _model(SEP_FULL, 9, "f16x3")
    sc = make_scene(1010, 5, 7, 48000, reverb=True)
    mix = torch.from_numpy(sc.mix)
    offs = list(sc.tdoa_samples())
    y = model.infer_sample(mix, offs)
    assert y.shape == (5, 48000) and np.all(np.isfinite(y))
    np.testing.assert_array_equal(y, model.infer_sample(mix, offs))
    perm = [3, 0, 4, 1, 2]
    yp = model.infer_sample(mix, [offs[i] for i in perm])
    assert _snr(yp, y[perm]) > 80.0
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    want = sep_ref.infer_sample(sd, SEP_FULL, mix, offs)
    snr = _snr(y, want)
    _log(f"sep infer_sample FULL f16x3, 5 speakers, T=48000: {snr:.1f} dB vs oracle; permutation {_snr(yp, y[perm]):.1f} dB")
    assert snr > 80.0


def test_joint_model_runs_the_separation_stage():
    """JointModel.forward with both networks: stage 5 is timed and returns one row per talker."""
    import io
    from contextlib import redirect_stdout
    from acousticswarms_speech_amd.config import FULL, SEP_FULL
    from acousticswarms_speech_amd.joint import JointModel
    from acousticswarms_speech_amd.scenes import make_scene
    from acousticswarms_speech_amd.spot import SpotModel
    from acousticswarms_speech_amd.weights import make_spot_state_dict
    sep, _sd = _model(SEP_FULL, 9, "f16x3")
    spot = SpotModel(FULL, make_spot_state_dict(FULL, 5), batch_size=64, precision="f16x3").to("cuda")
    sc = make_scene(1010, 5, 7, 24000, reverb=True)
    jm = JointModel(spot, sep, device="cuda")
    with redirect_stdout(io.StringIO()):
        jm.setup(sc.mic_positions, sc.speaker_range)
        patches, audio_loc, audio, _d0, _d1, _n = jm.forward(torch.from_numpy(sc.mix))
    assert len(patches) >= 1 and audio is not None
    assert audio.shape == (len(patches), 24000) == audio_loc.shape and np.all(np.isfinite(audio))
    assert jm.times[4] > 0
    want = sep.infer(torch.from_numpy(sc.mix), [p[0] for p in patches])
    np.testing.assert_array_equal(audio, want)
    _log(f"joint model: {len(patches)} talkers, stage times {np.round(jm.times, 4)}")


def test_infer_reference_native_length_property():
    """T = 144 000 (3 s at the reference's native 48 kHz, bottleneck length 2250): the two
    arithmetic modes of the FULL separation network agree to fp32-class accuracy and the f16x3
    range guard stays silent."""
    from acousticswarms_speech_amd import ops
    from acousticswarms_speech_amd.config import SEP_FULL
    from acousticswarms_speech_amd.scenes import make_scene
    model, _sd = _model(SEP_FULL, 9, "f32")
    sc = make_scene(1011, 2, 7, 144000)
    mix = torch.from_numpy(sc.mix)
    offs = list(sc.tdoa_samples())
    y32 = model.infer_sample(mix, offs)
    ops.f16x3_overflow_count(reset=True)
    model.set_precision("f16x3")
    y16 = model.infer_sample(mix, offs)
    assert y32.shape == (2, 144000) and np.all(np.isfinite(y32))
    snr = _snr(y16, y32)
    _log(f"sep infer_sample FULL T=144000, 2 speakers: f16x3 vs f32 {snr:.1f} dB")
    assert snr > 80.0 and ops.f16x3_overflow_count(reset=True) == 0
