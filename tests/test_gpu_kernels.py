"""Per-kernel parity: every C-ABI kernel against a plain PyTorch fp32 CPU statement of
the same reference op (tolerances written at each assert).  Needs an MI355X."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG = os.path.join(ROOT, "gpurun_out", "diag_kernels.txt")


def _log(msg):
    os.makedirs(os.path.dirname(DIAG), exist_ok=True)
    with open(DIAG, "a") as f:
        f.write(msg + "\n")
    print(msg)


def _relerr(a, b):
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30)), float((a - b).abs().max())


@pytest.fixture(scope="module")
def ops():
    from acousticswarms_speech_amd import ops as o
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return o


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# ---------------------------------------------------------------- shift / normalise
def test_shift_stats_and_preproc(ops, golden):
    from acousticswarms_speech_amd.scenes import make_scene
    from oracle import spot_ref
    g = golden("g1_shift_norm")
    mix = torch.from_numpy(make_scene(0, 2, 7, 4800).mix)
    offs = torch.from_numpy(g["offsets"].astype(np.int32))
    mean, std = ops.shift_stats(mix.cuda(), offs.cuda())
    # double accumulation vs torch's float reductions: 2e-6 relative
    np.testing.assert_allclose(mean.cpu().numpy(), g["mean"], rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(std.cpu().numpy(), g["std"], rtol=2e-6)
    w, b = _rand(64, 7, seed=1, scale=0.4), _rand(64, seed=2, scale=0.1)
    T_pad = 4864
    # use the reference's own mean/std so the element-wise part can be compared tightly
    x0, refn = ops.shift_norm_preproc(mix.cuda(), offs.cuda(), torch.from_numpy(g["mean"]).cuda(),
                                      torch.from_numpy(g["std"]).cuda(), w.cuda(), b.cuda(), T_pad)
    data = torch.stack([spot_ref.roll_channels(mix, o) for o in g["offsets"]])
    dn, _, _ = spot_ref.normalize_input(data)
    dn = F.pad(dn, (T_pad - 4800, 0))
    want = F.conv1d(dn, w.unsqueeze(-1), b).transpose(1, 2)
    # the shifted / quantised / normalised reference channel is element-wise: bit exact
    assert torch.equal(refn.cpu(), dn[:, 0])
    rel, mx = _relerr(x0.cpu(), want)
    _log(f"preproc rel={rel:.3e} max={mx:.3e}")
    assert rel < 1e-6
    # zero-filled (joint decoder) shift variant
    mean2, std2 = ops.shift_stats(mix.cuda(), offs.cuda(), circular=False)
    data2 = torch.stack([spot_ref.roll_channels(mix, o, circular=False) for o in g["offsets"]])
    _, mu2, sg2 = spot_ref.normalize_input(data2)
    np.testing.assert_allclose(mean2.cpu().numpy(), mu2.flatten().numpy(), rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(std2.cpu().numpy(), sg2.flatten().numpy(), rtol=2e-6)


# ---------------------------------------------------------------- conv-as-GEMM
@pytest.mark.parametrize("C,T,dil", [(64, 1000, 1), (64, 700, 7), (64, 900, 49), (128, 520, 7),
                                     (256, 300, 49), (512, 200, 1), (512, 130, 7)])
def test_residual_layer(ops, C, T, dil):
    """DilatedResidualLayer: conv(k=7,dil) -> ReLU -> +x -> LayerNorm(C) (network.py:57-68)."""
    B, K = 2, 7
    x = _rand(B, C, T, seed=3)
    w = _rand(C, C, K, seed=4, scale=1.0 / math.sqrt(C * K))
    b, g, be = _rand(C, seed=5, scale=0.1), 1 + _rand(C, seed=6, scale=0.1), _rand(C, seed=7, scale=0.1)
    y = F.relu(F.conv1d(x, w, b, dilation=dil, padding=3 * dil)) + x
    want = F.layer_norm(y.transpose(1, 2), (C,), g, be, 1e-5)
    xc = x.transpose(1, 2).contiguous().cuda()
    out, _ = ops.convgemm(xc, ops.pack_conv_weight(w).cuda(), T, C, C, taps=K, dil=dil, pad=3 * dil,
                          bias=b.cuda(), relu=True, resid=xc, ln=(g.cuda(), be.cuda()))
    rel, mx = _relerr(out.cpu(), want)
    _log(f"res C={C} T={T} dil={dil}: rel={rel:.3e} max={mx:.3e}")
    assert rel < 2e-6 and mx < 5e-5      # fp32 fmaf chain vs oneDNN fp32: ordering only


@pytest.mark.parametrize("Cin,Cout,T,s", [(64, 64, 1000, 2), (64, 128, 600, 2), (128, 256, 512, 4),
                                          (512, 1024, 96, 4)])
def test_down_conv_groupnorm_glu(ops, Cin, Cout, T, s):
    """EncoderBlock tail: gate * x -> Conv1d(stride) -> GroupNorm(2) -> GLU (network.py:101-113)."""
    B, K = 2, 7
    x = _rand(B, Cin, T, seed=8)
    w = _rand(2 * Cout, Cin, K, seed=9, scale=1.0 / math.sqrt(Cin * K))
    b = _rand(2 * Cout, seed=10, scale=0.1)
    gate = 0.5 + _rand(Cin, seed=11, scale=0.2)
    gg, gb = 1 + _rand(2 * Cout, seed=12, scale=0.1), _rand(2 * Cout, seed=13, scale=0.1)
    raw = F.conv1d(gate.view(1, -1, 1) * x, w, b, stride=s, padding=K // 2)
    want = F.glu(F.group_norm(raw, 2, gg, gb, 1e-5), dim=1).transpose(1, 2)
    To = raw.shape[-1]
    wt = ops.pack_conv_weight(w * gate.view(1, -1, 1)).cuda()
    xc = x.transpose(1, 2).contiguous().cuda()
    r, st = ops.convgemm(xc, wt, To, 2 * Cout, Cin, taps=K, stride=s, pad=K // 2, bias=b.cuda(),
                         stats_chan_mod=2 * Cout)
    rel0, _ = _relerr(r.cpu(), raw.transpose(1, 2))
    out = ops.gn_glu(r, st, gg.cuda(), gb.cuda())
    rel, mx = _relerr(out.cpu(), want)
    _log(f"down Cin={Cin} Cout={Cout} s={s}: raw rel={rel0:.3e} glu rel={rel:.3e} max={mx:.3e}")
    assert rel0 < 2e-6 and rel < 5e-6


@pytest.mark.parametrize("Cin,Cout,T,s", [(1024, 512, 50, 4), (256, 128, 300, 4), (128, 64, 700, 2),
                                          (64, 64, 900, 2)])
def test_up_conv_groupnorm_glu(ops, Cin, Cout, T, s):
    """DecoderBlock head: (x+skip) -> ConvTranspose1d(k=s,stride=s) -> gate -> GN(2) -> GLU
    (network.py:180-197)."""
    B = 2
    x, skip = _rand(B, Cin, T, seed=14), _rand(B, Cin, T, seed=15)
    w = _rand(Cin, 2 * Cout, s, seed=16, scale=1.0 / math.sqrt(Cin))
    b = _rand(2 * Cout, seed=17, scale=0.1)
    gate = 0.5 + _rand(2 * Cout, seed=18, scale=0.2)
    gg, gb = 1 + _rand(2 * Cout, seed=19, scale=0.1), _rand(2 * Cout, seed=20, scale=0.1)
    raw = gate.view(1, -1, 1) * F.conv_transpose1d(x + skip, w, b, stride=s)
    want = F.glu(F.group_norm(raw, 2, gg, gb, 1e-5), dim=1).transpose(1, 2)
    co2 = 2 * Cout
    wt = (w * gate.view(1, -1, 1)).permute(2, 1, 0).reshape(s * co2, Cin).contiguous().cuda()
    bb = (b * gate).repeat(s).contiguous().cuda()
    xc = x.transpose(1, 2).contiguous().cuda()
    sc = skip.transpose(1, 2).contiguous().cuda()
    r, st = ops.convgemm(xc, wt, T, s * co2, Cin, bias=bb, stats_chan_mod=co2, A2=sc)
    r = r.view(B, T * s, co2)
    rel0, _ = _relerr(r.cpu(), raw.transpose(1, 2))
    out = ops.gn_glu(r, st, gg.cuda(), gb.cuda())
    rel, mx = _relerr(out.cpu(), want)
    _log(f"up Cin={Cin} Cout={Cout} s={s}: raw rel={rel0:.3e} glu rel={rel:.3e} max={mx:.3e}")
    assert rel0 < 2e-6 and rel < 5e-6


@pytest.mark.parametrize("d,f,L,B", [(128, 128, 19, 3), (1024, 1024, 47, 2)])
def test_linear_residual_layernorm(ops, d, f, L, B):
    """Transformer linears with fused residual + LayerNorm (post-norm, network.py:254)."""
    x = _rand(B * L, f, seed=21)
    res = _rand(B * L, d, seed=22)
    w, b = _rand(d, f, seed=23, scale=1 / math.sqrt(f)), _rand(d, seed=24, scale=0.1)
    g, be = 1 + _rand(d, seed=25, scale=0.1), _rand(d, seed=26, scale=0.1)
    want = F.layer_norm(res + F.linear(x, w, b), (d,), g, be, 1e-5)
    out, _ = ops.convgemm(x.cuda().view(1, B * L, f), w.cuda(), B * L, d, f, bias=b.cuda(),
                          resid=res.cuda().view(1, B * L, d), ln=(g.cuda(), be.cuda()), B=1)
    rel, mx = _relerr(out.cpu(), want)
    _log(f"linear+LN d={d}: rel={rel:.3e} max={mx:.3e}")
    assert rel < 2e-6
    want2 = F.relu(F.linear(res, _rand(f, d, seed=27, scale=1 / math.sqrt(d)), None))
    out2, _ = ops.convgemm(res.cuda().view(1, B * L, d), _rand(f, d, seed=27, scale=1 / math.sqrt(d)).cuda(),
                           B * L, f, d, relu=True, B=1)
    rel, _ = _relerr(out2.cpu(), want2)
    assert rel < 2e-6


@pytest.mark.parametrize("L,d,nhead,B", [(19, 128, 8, 3), (188, 1024, 8, 2), (300, 1024, 8, 1), (150, 1024, 8, 2),
                                         (192, 1024, 8, 1), (33, 1024, 8, 2), (47, 512, 4, 2), (563, 1024, 8, 1), (700, 1024, 8, 1)])
@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_attention(ops, L, d, nhead, B, precision):
    """f32: exact fp32 MFMA; f16x3: both products on the f16 MFMA with split operands (head_dim 128, L <= 352; other
    shapes run the f32 kernels), the softmax in fp32 either way: same bar."""
    qkv = _rand(B, L, 3 * d, seed=28)
    hd = d // nhead
    q, k, v = qkv.split(d, dim=-1)
    q = q.view(B, L, nhead, hd).transpose(1, 2)
    k = k.view(B, L, nhead, hd).transpose(1, 2)
    v = v.view(B, L, nhead, hd).transpose(1, 2)
    want = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), -1) @ v).transpose(1, 2).reshape(B, L, d)
    out = ops.attention(qkv.cuda(), nhead, precision=precision)
    rel, mx = _relerr(out.cpu(), want)
    _log(f"attention {precision} L={L} d={d}: rel={rel:.3e} max={mx:.3e}")
    assert rel < 3e-6


@pytest.mark.parametrize("rows,N", [(5, 256), (1203, 1024), (64, 2048), (1, 768)])
def test_add_layernorm(ops, rows, N):
    """norm(x + resid) of the post-norm transformer block, rows wider than a fused GEMM tile."""
    x, r = _rand(rows, N, seed=40, scale=2.0) + 0.3, _rand(rows, N, seed=41)
    g, b = _rand(N, seed=42) * 0.2 + 1.0, _rand(N, seed=43, scale=0.1)
    want = F.layer_norm(x + r, (N,), g, b, 1e-5)
    out = ops.add_layernorm(x.cuda(), r.cuda(), g.cuda(), b.cuda())
    rel, mx = _relerr(out.cpu(), want)
    _log(f"add_layernorm rows={rows} N={N}: rel={rel:.3e} max={mx:.3e}")
    assert rel < 1e-6
    with pytest.raises(RuntimeError):
        ops.add_layernorm(x[:, :100].contiguous().cuda(), r[:, :100].contiguous().cuda(), g[:100].cuda(), b[:100].cuda())


def test_mask_path(ops):
    """reference_bypass * mask_encoder -> output_decoder -> trim (network.py:397-405)."""
    B, C, E, EK, ES, Tp, t = 2, 64, 256, 33, 16, 2048, 1900
    x = _rand(B, C, Tp, seed=30)
    ref = _rand(B, 1, Tp, seed=31)
    wb, bb = _rand(E, 1, EK, seed=32, scale=0.2), _rand(E, seed=33, scale=0.1)
    wm, bm = _rand(E, C, EK, seed=34, scale=1 / math.sqrt(C * EK)), _rand(E, seed=35, scale=0.1)
    wd, bd = _rand(E, 1, EK, seed=36, scale=1 / math.sqrt(E)), 0.05
    y = F.relu(F.conv1d(ref, wb, bb, stride=ES, padding=EK // 2))
    mask = F.relu(F.conv1d(x, wm, bm, stride=ES, padding=EK // 2))
    lat = y * mask
    full = F.conv_transpose1d(lat, wd, torch.tensor([bd]), stride=EK // 2)
    want = full[..., 9:-8][..., -t:][:, 0]
    Fr = lat.shape[-1]
    RL = EK // 2 + Tp + 64 + 64
    refx = torch.zeros(B, RL)
    refx[:, EK // 2:EK // 2 + Tp] = ref[:, 0]
    wbp = torch.zeros(E, 64)
    wbp[:, :EK] = wb[:, 0]
    Y, _ = ops.convgemm(refx.cuda(), wbp.cuda(), Fr, E, 64, bias=bb.cuda(), relu=True, a_row_stride=ES,
                        a_batch_stride=RL, a_len=RL, B=B)
    rel, _ = _relerr(Y.cpu(), y.transpose(1, 2))
    _log(f"bypass rel={rel:.3e}")
    assert rel < 2e-6
    xc = x.transpose(1, 2).contiguous().cuda()
    Lm, _ = ops.convgemm(xc, ops.pack_conv_weight(wm).cuda(), Fr, E, C, taps=EK, stride=ES, pad=EK // 2,
                         bias=bm.cuda(), relu=True, mul=Y, out=Y)
    rel, _ = _relerr(Lm.cpu(), lat.transpose(1, 2))
    _log(f"latent rel={rel:.3e}")
    assert rel < 3e-6
    wdp = torch.zeros(64, E)
    wdp[:EK] = wd[:, 0].t()
    D, _ = ops.convgemm(Lm, wdp.cuda(), Fr, 64, E)
    out = ops.overlap_add_unnorm(D, EK, EK // 2, t, 9, 8, bd)
    rel, mx = _relerr(out.cpu(), want)
    _log(f"decode rel={rel:.3e} max={mx:.3e}")
    assert rel < 3e-6
    mean, std = torch.tensor([0.1, -0.2]), torch.tensor([0.5, 2.0])
    out2 = ops.overlap_add_unnorm(D, EK, EK // 2, t, 9, 8, bd, mean.cuda(), std.cuda())
    rel, _ = _relerr(out2.cpu(), want * std.view(-1, 1) + mean.view(-1, 1))
    assert rel < 3e-6


@pytest.mark.parametrize("B,E,Tp,t", [(2, 256, 2048, 1900), (3, 512, 8192 + 16 * 37, 8000)])
def test_mask_path_fused(ops, B, E, Tp, t):
    """The same mask path in one launch (asw_mask_path_f16x3: bypass and decoder taps in the mask
    encoder's epilogue, partial taps per 256-channel column tile) against torch fp32; ragged last row
    tile, two column tiles."""
    C, EK, ES = 64, 33, 16
    x = _rand(B, C, Tp, seed=40)
    ref = _rand(B, 1, Tp, seed=41)
    wb, bb = _rand(E, 1, EK, seed=42, scale=0.2), _rand(E, seed=43, scale=0.1)
    wm, bm = _rand(E, C, EK, seed=44, scale=1 / math.sqrt(C * EK)), _rand(E, seed=45, scale=0.1)
    wd, bd = _rand(E, 1, EK, seed=46, scale=1 / math.sqrt(E)), 0.05
    y = F.relu(F.conv1d(ref, wb, bb, stride=ES, padding=EK // 2))
    mask = F.relu(F.conv1d(x, wm, bm, stride=ES, padding=EK // 2))
    lat = y * mask
    full = F.conv_transpose1d(lat, wd, torch.tensor([bd]), stride=EK // 2)
    want = full[..., 9:-8][..., -t:][:, 0]
    Fr = lat.shape[-1]
    RL = EK // 2 + Tp + 64 + 64
    refx = torch.zeros(B, RL)
    refx[:, EK // 2:EK // 2 + Tp] = ref[:, 0]
    xc = x.transpose(1, 2).contiguous().cuda()
    parts = ops.mask_path(xc, refx.cuda(), ES, wm, bm.cuda(), wb, bb.cuda(), wd, Fr, ES, EK // 2)
    assert parts.shape == (E // 256, B, Fr, 64)
    taps_want = torch.einsum("bef,ej->bfj", lat, wd[:, 0])
    rel, _ = _relerr(parts.sum(0)[..., :EK].cpu(), taps_want)
    _log(f"fused taps rel={rel:.3e}")
    assert rel < 3e-6
    out = ops.overlap_add_parts(parts, EK, EK // 2, t, 9, 8, bd)
    rel, mx = _relerr(out.cpu(), want)
    _log(f"fused decode rel={rel:.3e} max={mx:.3e}")
    assert rel < 3e-6


@pytest.mark.parametrize("n,T", [(5, 1003), (17, 4099), (70, 2500)])
def test_tiled_sisdr_kernels_on_ragged_shapes(n, T):
    """pair_sisdr / segment_sisdr (tiled Gram kernels: 16 x 16 pairs, one estimate x 64 references, 64-sample
    chunks) where neither n nor T nor the segment bounds fill a tile, against the host si_sdr of the same rows
    (float32 numpy arithmetic: 1e-3 dB); a row compared with itself hits the residual clamp."""
    from acousticswarms_speech_amd.config import SMALL
    from acousticswarms_speech_amd.hostdsp import si_sdr
    from acousticswarms_speech_amd.spot import SpotModel
    from acousticswarms_speech_amd.weights import make_spot_state_dict
    rng = np.random.default_rng(100 + n)
    base = rng.standard_normal((3, T)).astype(np.float32)
    a = (rng.standard_normal((n, 3)).astype(np.float32) @ base + 0.2 * rng.standard_normal((n, T)).astype(np.float32))
    m = SpotModel(SMALL, make_spot_state_dict(SMALL, 1), batch_size=4).to("cuda")
    waves = torch.from_numpy(a).cuda()
    S = m.pair_sisdr(waves)
    segs = []
    for i in range(n):
        k = int(rng.integers(0, 4))
        cuts = np.sort(rng.choice(np.arange(1, T), size=2 * k, replace=False)) if k else np.zeros(0, dtype=int)
        segs.append([(int(cuts[2 * j]), int(cuts[2 * j + 1])) for j in range(k)])
    G, cnt = m.segment_sisdr(waves, segs)
    worst = 0.0
    for i in range(n):
        for j in range(n):
            if i == j:
                assert S[i, i] > 100.0 and np.isfinite(S[i, i])
                continue
            worst = max(worst, abs(S[i, j] - si_sdr(a[i], a[j])))
            for k, (lo, hi) in enumerate(segs[i]):
                worst = max(worst, abs(G[i, j, k] - si_sdr(a[i, lo:hi], a[j, lo:hi])))
        assert cnt[i] == len(segs[i]) and np.all(np.isnan(G[i, :, len(segs[i]):]))
    _log(f"tiled si-sdr n={n} T={T}: worst difference {worst:.2e} dB")
    assert worst < 1e-3


def test_energies_and_sisdr(ops):
    from oracle import spot_ref
    rng = np.random.default_rng(5)
    y = (rng.standard_normal((6, 30011)) * np.hanning(30011) * 0.03 + 0.01).astype(np.float32)
    y[3, :] *= 0.0
    y[3, 100:200] = 0.5
    want = spot_ref.candidate_energies(y, 12000)
    got = ops.energies(torch.from_numpy(y).cuda(), 12000).cpu().numpy()
    _log(f"energies max rel {np.abs(got / np.maximum(want, 1e-30) - 1).max():.3e}")
    np.testing.assert_allclose(got, want, rtol=2e-6)          # fp32 numpy sums vs double accumulation
    got2 = ops.energies(torch.from_numpy(y).cuda(), 1000).cpu().numpy()
    np.testing.assert_allclose(got2, spot_ref.candidate_energies(y, 1000), rtol=2e-6)
    a = rng.standard_normal((5, 20000)).astype(np.float32)
    a[1] = 0.7 * a[0] + 0.1 * a[1]
    a[3] = -a[2] + 0.3 * a[3]
    S = ops.pair_sisdr(torch.from_numpy(a).cuda()).cpu().numpy()
    for i in range(5):
        for j in range(5):
            if i != j:
                assert abs(S[i, j] - spot_ref.si_sdr(a[i], a[j])) < 1e-3, (i, j)   # dB
