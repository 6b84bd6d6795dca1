"""Parity of the fused 64-channel residual stacks (csrc/resstack.hip, asw_resstack64_f16x3): the layers'
intermediate tensors stay in LDS, accumulators are transposed (a lane owns one time row), LayerNorm runs in
registers.  References: the torch fp32 statement of DilatedResidualLayer / DilatedResidualSequence
(sep/training/SpeakerLocalization/network.py:50-82) and the per-layer HIP kernels (asw_convgemm_f32).
Bar: relative L2 <= 2e-5 against torch fp32 (the f16x3 bar of tests/test_gpu_f16x3.py), <= 3e-6 against the
per-layer HIP path (same arithmetic; only summation order and the residual's hi + lo read differ).
Needs an MI355X."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = 64


def _log(msg):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "diag_resstack.txt"), "a") as f:
        f.write(msg + "\n")
    print(msg)


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm())


def _layers(dils, K, seed=0):
    out = []
    for i, d in enumerate(dils):
        w = _rand(C, C, K, seed=seed + 10 * i + 1, scale=1.0 / math.sqrt(C * K))
        b = _rand(C, seed=seed + 10 * i + 2, scale=0.1)
        g = 1 + _rand(C, seed=seed + 10 * i + 3, scale=0.1)
        be = _rand(C, seed=seed + 10 * i + 4, scale=0.1)
        out.append((w, b, g, be, d))
    return out


def _torch_stack(x, layers, K):
    """x [B][C][T] -> [B][T][C]"""
    for w, b, g, be, d in layers:
        y = F.relu(F.conv1d(x, w, b, dilation=d, padding=d * (K - 1) // 2)) + x
        x = F.layer_norm(y.transpose(1, 2), (C,), g, be, 1e-5).transpose(1, 2)
    return x.transpose(1, 2)


def _hip_layers(ops, xc, layers, K):
    """the per-layer HIP kernels (halo-staged resconv16), f16x3"""
    T = xc.shape[1]
    cur = xc
    for w, b, g, be, d in layers:
        cur, _ = ops.convgemm(cur, ops.pack_conv_weight(w).cuda(), T, C, C, taps=K, dil=d, pad=d * (K - 1) // 2,
                              bias=b.cuda(), relu=True, resid=cur, ln=(g.cuda(), be.cuda()), precision="f16x3")
    return cur


def _dev_layers(ops, layers):
    return [(ops.pack_conv_weight(w).cuda(), b.cuda(), g.cuda(), be.cuda(), d) for w, b, g, be, d in layers]


@pytest.mark.parametrize("dils,K,T", [((1,), 7, 1000), ((1,), 7, 37), ((7,), 7, 777), ((49,), 7, 900), ((49,), 7, 100),
                                      ((49,), 7, 4900), ((49,), 7, 3000), ((49,), 7, 13000), ((7,), 7, 3000),
                                      ((1, 7), 7, 1000), ((1, 7), 7, 214), ((1, 7), 7, 215), ((1, 7), 7, 5), ((1, 7), 7, 3010),
                                      ((1, 2, 4), 5, 1500), ((1, 2), 5, 233), ((2, 4), 5, 700), ((1, 7, 49), 7, 600)])
def test_resstack_vs_torch_and_per_layer_kernels(dils, K, T):
    """single layers (contiguous and polyphase), the spot pair (1, 7), the separation network's (1, 2, 4) with
    5 taps; ragged lengths around the tile size (214 finished rows per workgroup for the pair)."""
    from acousticswarms_speech_amd import ops
    B = 3
    x = _rand(B, C, T, seed=3)
    layers = _layers(dils, K)
    want = _torch_stack(x, layers, K)
    xc = x.transpose(1, 2).contiguous().cuda()
    if len(dils) == 3 and dils[2] == 49:
        # the wide third layer cannot share the launch: the call must say so, the host splits the stack
        with pytest.raises(RuntimeError, match="fuse fewer layers|exceeds LDS"):
            ops.resstack(xc, _dev_layers(ops, layers), taps=K)
        mid = ops.resstack(xc, _dev_layers(ops, layers[:2]), taps=K)
        got = ops.resstack(mid, _dev_layers(ops, layers[2:]), taps=K)
    else:
        got = ops.resstack(xc, _dev_layers(ops, layers), taps=K)
    r = _rel(got.cpu(), want)
    r2 = _rel(got.cpu(), _hip_layers(ops, xc, layers, K).cpu())
    _log(f"resstack dils={dils} K={K} T={T}: rel vs torch {r:.3e}, vs per-layer HIP {r2:.3e}")
    assert torch.isfinite(got).all()
    assert r < 2e-5 and r2 < 3e-6


def test_resstack_single_pass_f16_mode():
    from acousticswarms_speech_amd import ops
    x = _rand(2, C, 800, seed=5)
    layers = _layers((1, 7), 7, seed=50)
    want = _torch_stack(x, layers, 7)
    got = ops.resstack(x.transpose(1, 2).contiguous().cuda(), _dev_layers(ops, layers), taps=7, precision="f16")
    r = _rel(got.cpu(), want)
    _log(f"resstack (1,7) single-pass f16: rel {r:.3e}")
    assert r < 2e-3


def test_resstack_groupnorm_glu_on_load():
    """first layer fed by the un-normalised output of a transposed convolution: GroupNorm(2) + GLU while the rows are
    staged (asw_resstack_args.glu_raw) == asw_gn_glu followed by the stack, bit for bit in the staged operand."""
    from acousticswarms_speech_amd import ops
    B, T = 2, 1300
    raw = _rand(B, T, 2 * C, seed=7).cuda()
    gg, gb = (1 + _rand(2 * C, seed=8, scale=0.1)).cuda(), _rand(2 * C, seed=9, scale=0.1).cuda()
    # partial statistics as a GEMM would have written them: one slot per item
    st = torch.zeros(B, 1, 4, device="cuda")
    st[:, 0, 0] = raw[:, :, :C].sum((1, 2)); st[:, 0, 1] = (raw[:, :, :C] ** 2).sum((1, 2))
    st[:, 0, 2] = raw[:, :, C:].sum((1, 2)); st[:, 0, 3] = (raw[:, :, C:] ** 2).sum((1, 2))
    x = ops.gn_glu(raw, st, gg, gb)
    mr = ops.gn_finalize(st, T, C)
    layers = _dev_layers(ops, _layers((1, 7), 7, seed=70))
    a = ops.resstack(x, layers, taps=7)
    b = ops.resstack(None, layers, taps=7, glu=(raw, mr, gg, gb))
    _log(f"resstack glu-on-load: max abs diff {float((a - b).abs().max()):.3e}")
    assert torch.equal(a, b)
    # the encoder's form: the normalised rows are also written out once (the skip connection), each by the tile that owns them
    for dils, Tn in (((1, 7), T), ((1,), 701), ((1, 7), 48128)):
        rawn = _rand(B, Tn, 2 * C, seed=17).cuda()
        stn = torch.zeros(B, 1, 4, device="cuda")
        stn[:, 0, 0] = rawn[:, :, :C].sum((1, 2)); stn[:, 0, 1] = (rawn[:, :, :C] ** 2).sum((1, 2))
        stn[:, 0, 2] = rawn[:, :, C:].sum((1, 2)); stn[:, 0, 3] = (rawn[:, :, C:] ** 2).sum((1, 2))
        xn, mrn = ops.gn_glu(rawn, stn, gg, gb), ops.gn_finalize(stn, Tn, C)
        ly = _dev_layers(ops, _layers(dils, 7, seed=71))
        side = torch.full((B, Tn, C), float("nan"), device="cuda")
        c = ops.resstack(None, ly, taps=7, glu=(rawn, mrn, gg, gb), glu_out=side)
        assert torch.equal(c, ops.resstack(xn, ly, taps=7))
        d = float((side - xn).abs().max())
        _log(f"resstack glu side output dils={dils} T={Tn}: max abs diff to asw_gn_glu {d:.3e}, bit-equal {torch.equal(side, xn)}")
        assert torch.isfinite(side).all() and d <= 1e-6


def test_resstack_full_size_properties():
    """BASELINE size (T = 48 128 rows, the spot network's full-rate level): finite, equal to the per-layer kernels,
    and independent of the batch split (run-to-run and item-to-item bit identity)."""
    from acousticswarms_speech_amd import ops
    B, T, K = 4, 48128, 7
    x = _rand(B, T, C, seed=11).cuda()
    x[1] = x[0]
    layers = _layers((1, 7), K, seed=90)
    dl = _dev_layers(ops, layers)
    got = ops.resstack(x, dl, taps=K)
    again = ops.resstack(x, dl, taps=K)
    assert torch.isfinite(got).all() and torch.equal(got, again) and torch.equal(got[0], got[1])
    ref = _hip_layers(ops, x, layers, K)
    r = _rel(got.cpu(), ref.cpu())
    got49 = ops.resstack(got, _dev_layers(ops, _layers((49,), K, seed=95)), taps=K)
    ref49 = _hip_layers(ops, got, _layers((49,), K, seed=95), K)
    r49 = _rel(got49.cpu(), ref49.cpu())
    _log(f"resstack full size: pair rel vs per-layer HIP {r:.3e}; dil 49 {r49:.3e}")
    assert r < 3e-6 and r49 < 3e-6
