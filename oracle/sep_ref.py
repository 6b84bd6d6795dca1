"""ORACLE (test infrastructure, never on the product path).

CPU restatement of the joint separation network ("separation by localization"):
per-speaker zero-filled integer shift -> joint int16 quantise + normalise -> U-Net encoder ->
bottleneck of {Conformer layer per speaker, inter-speaker transformer layer per time step} ->
U-Net decoder -> masked latent -> output decoder -> un-normalise.  Plain functions over a
state dict (torch CPU functional ops), each citing the reference lines it follows
(sep/training/SpeakerSeparation/network.py unless another file is named).

Pinned by tests/golden/g11_*.npz, produced by the reference's own ``Network`` (forward /
infer_sample) with seeded weights.  The reference builds its bottleneck from speechbrain
(ConformerEncoder, RelPosEncXL; requirements.txt:14, unpinned, absent from the image): on
the generator side those two classes are this repo's restatement of the published speechbrain
definitions (tests/golden/thirdparty_restated.py), so the fixtures pin everything the
reference itself wrote and pin this file to that restatement -- the Conformer arithmetic
against speechbrain itself is "parity unpinned".

Only tests/, __graft_entry__.smoke() and bench.py's baseline legs may import this module.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .spot_ref import _t, normalize_input, roll_channels, unnormalize_input


def _residual_sequence(x, sd, prefix, cfg):
    """DilatedResidualLayer x residual_layers (:50-82): conv(k, dil=f**i, pad=(dil*(k-1)+1)//2)
    -> ReLU -> +x -> LayerNorm over channels."""
    K = cfg.kernel_size
    for j in range(cfg.residual_layers):
        d = cfg.residual_dilation_factor ** j
        p = f"{prefix}.res.seq.{j}"
        y = F.conv1d(x, _t(sd, p + ".conv.weight"), _t(sd, p + ".conv.bias"), dilation=d, padding=(d * (K - 1) + 1) // 2)
        y = F.relu(y) + x
        c = y.shape[1]
        x = F.layer_norm(y.transpose(1, 2), (c,), _t(sd, p + ".norm.weight"), _t(sd, p + ".norm.bias"), 1e-5).transpose(1, 2)
    return x


def _encoder_block(x, sd, i, stride, cfg):
    """EncoderBlock.forward (:103-111): no window gate in this network."""
    p = f"encoder.module_list.{i}"
    x = _residual_sequence(x, sd, p, cfg)
    x = F.conv1d(x, _t(sd, p + ".conv1.weight"), _t(sd, p + ".conv1.bias"), stride=stride, padding=cfg.kernel_size // 2)
    x = F.group_norm(x, 2, _t(sd, p + ".norm1.weight"), _t(sd, p + ".norm1.bias"), 1e-5)
    return F.glu(x, dim=1)


def _decoder_block(x, skip, sd, i, stride, cfg):
    """DecoderBlock.forward (:192-202)."""
    p = f"decoder.module_list.{i}"
    x = x + skip
    x = F.conv_transpose1d(x, _t(sd, p + ".upsample.conv.weight"), _t(sd, p + ".upsample.conv.bias"), stride=stride)
    x = F.group_norm(x, 2, _t(sd, p + ".norm1.weight"), _t(sd, p + ".norm1.bias"), 1e-5)
    x = F.glu(x, dim=1)
    return _residual_sequence(x, sd, p, cfg)


# --------------------------------------------------------------------------------------------
# bottleneck (:270-321)
# --------------------------------------------------------------------------------------------
def rel_pos_table(L: int, d: int, inv_freq: torch.Tensor) -> torch.Tensor:
    """RelPosEncXL: rows m = 0 .. 2L-2 stand for relative positions L-1 .. -(L-1); even
    columns sin, odd columns cos of |position| * inv_freq (the published table uses the same
    sinusoid for past and future and cos is even, so row m depends on |L-1-m| only)."""
    pos = (torch.arange(2 * L - 1, dtype=torch.float32) - (L - 1)).abs().unsqueeze(-1)
    ang = pos * inv_freq.to(torch.float32)
    pe = torch.zeros((2 * L - 1, d), dtype=torch.float32)
    pe[:, 0::2] = torch.sin(ang)
    pe[:, 1::2] = torch.cos(ang)
    return pe


def _ffn(x, sd, p):
    """ffn_module = Sequential(LayerNorm, PositionalwiseFeedForward(Linear, Swish, Linear))."""
    d = x.shape[-1]
    h = F.layer_norm(x, (d,), _t(sd, p + ".0.weight"), _t(sd, p + ".0.bias"), 1e-5)
    h = F.linear(h, _t(sd, p + ".1.ffn.0.weight"), _t(sd, p + ".1.ffn.0.bias"))
    h = h * torch.sigmoid(h)
    return F.linear(h, _t(sd, p + ".1.ffn.3.weight"), _t(sd, p + ".1.ffn.3.bias"))


def _rel_pos_mha(x, sd, p, nhead, pe):
    """RelPosMHAXL self-attention.  in_proj has no bias and its output is cut per head into
    (q, k, v) thirds; scores = ((q+u) k^T + shift((q+v) P^T)) / sqrt(embed_dim) where
    P = linear_pos(pe) and shift picks P row (L-1) + j - i for query i, key j."""
    B, L, d = x.shape
    hd = d // nhead
    qkv = F.linear(x, _t(sd, p + ".in_proj_weight")).view(B, L, nhead, 3 * hd)
    q, k, v = qkv[..., :hd], qkv[..., hd:2 * hd], qkv[..., 2 * hd:]
    u = _t(sd, p + ".pos_bias_u").reshape(nhead, hd)          # .view(1,1,H,hd) of the stored [hd,H] tensor
    w = _t(sd, p + ".pos_bias_v").reshape(nhead, hd)
    P = F.linear(pe, _t(sd, p + ".linear_pos.weight")).view(2 * L - 1, nhead, hd)
    scale = 1.0 / math.sqrt(d)
    ac = torch.einsum("blhc,bmhc->bhlm", (q + u) * scale, k)
    bd_full = torch.einsum("blhc,phc->bhlp", (q + w) * scale, P)           # [B,H,L,2L-1]
    idx = (L - 1) + torch.arange(L).view(1, L) - torch.arange(L).view(L, 1)  # [i,j] -> (L-1)+j-i
    bd = torch.gather(bd_full, 3, idx.view(1, 1, L, L).expand(B, nhead, L, L))
    att = torch.softmax(ac + bd, dim=-1)
    ctx = torch.einsum("bhlm,bmhc->blhc", att, v).reshape(B, L, d)
    return F.linear(ctx, _t(sd, p + ".out_proj.weight"), _t(sd, p + ".out_proj.bias"))


def _conv_module(x, sd, p, ksize):
    """ConvolutionModule: LayerNorm -> pointwise Conv1d(d, 2d) -> GLU -> depthwise Conv1d(k,
    pad (k-1)/2) -> LayerNorm -> Swish -> Linear."""
    d = x.shape[-1]
    h = F.layer_norm(x, (d,), _t(sd, p + ".layer_norm.weight"), _t(sd, p + ".layer_norm.bias"), 1e-5).transpose(1, 2)
    h = F.glu(F.conv1d(h, _t(sd, p + ".bottleneck.0.weight"), _t(sd, p + ".bottleneck.0.bias")), dim=1)
    h = F.conv1d(h, _t(sd, p + ".conv.weight"), _t(sd, p + ".conv.bias"), padding=(ksize - 1) // 2, groups=d).transpose(1, 2)
    h = F.layer_norm(h, (d,), _t(sd, p + ".after_conv.0.weight"), _t(sd, p + ".after_conv.0.bias"), 1e-5)
    h = h * torch.sigmoid(h)
    return F.linear(h, _t(sd, p + ".after_conv.2.weight"), _t(sd, p + ".after_conv.2.bias"))


def conformer_layer(x, sd, p, cfg, pe):
    """ConformerEncoder(num_layers=1) = one ConformerEncoderLayer + final LayerNorm(eps 1e-6):
    x + FFN/2 -> +MHA(norm1) -> +conv module -> norm2(x + FFN/2) -> norm.  x: [B*S, L, d]."""
    d = x.shape[-1]
    c = p + ".layers.0"
    x = x + 0.5 * _ffn(x, sd, c + ".ffn_module1")
    h = F.layer_norm(x, (d,), _t(sd, c + ".norm1.norm.weight"), _t(sd, c + ".norm1.norm.bias"), 1e-5)
    x = x + _rel_pos_mha(h, sd, c + ".mha_layer", cfg.num_head, pe)
    x = x + _conv_module(x, sd, c + ".convolution_module", cfg.bottleneck_ksize)
    x = x + 0.5 * _ffn(x, sd, c + ".ffn_module2")
    x = F.layer_norm(x, (d,), _t(sd, c + ".norm2.norm.weight"), _t(sd, c + ".norm2.norm.bias"), 1e-5)
    return F.layer_norm(x, (d,), _t(sd, p + ".norm.norm.weight"), _t(sd, p + ".norm.norm.bias"), 1e-6)


def inter_layer(x, sd, p, nhead):
    """nn.TransformerEncoderLayer(batch_first=True) defaults: post-norm, ReLU, eps 1e-5.
    x: [rows, S, d] -- the S speakers at one time step are the sequence (:311-316)."""
    R, S, d = x.shape
    hd = d // nhead
    qkv = F.linear(x, _t(sd, p + ".self_attn.in_proj_weight"), _t(sd, p + ".self_attn.in_proj_bias"))
    q, k, v = qkv.split(d, dim=-1)
    q = q.view(R, S, nhead, hd).transpose(1, 2)
    k = k.view(R, S, nhead, hd).transpose(1, 2)
    v = v.view(R, S, nhead, hd).transpose(1, 2)
    att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
    ctx = (att @ v).transpose(1, 2).reshape(R, S, d)
    sa = F.linear(ctx, _t(sd, p + ".self_attn.out_proj.weight"), _t(sd, p + ".self_attn.out_proj.bias"))
    x = F.layer_norm(x + sa, (d,), _t(sd, p + ".norm1.weight"), _t(sd, p + ".norm1.bias"), 1e-5)
    h = F.relu(F.linear(x, _t(sd, p + ".linear1.weight"), _t(sd, p + ".linear1.bias")))
    h = F.linear(h, _t(sd, p + ".linear2.weight"), _t(sd, p + ".linear2.bias"))
    return F.layer_norm(x + h, (d,), _t(sd, p + ".norm2.weight"), _t(sd, p + ".norm2.bias"), 1e-5)


def _to_batches(x, counts):
    """speakers_to_batches (:236-247): x [B, S, ...] -> the first counts[b] speakers of every item, stacked."""
    return torch.cat([x[b, :c] for b, c in enumerate(counts)], 0)


def _to_speakers(x, counts):
    """batches_to_speakers (:250-268): stacked speakers -> [B, max(counts), ...], missing speakers zero."""
    m, out, pos = max(counts), [], 0
    for c in counts:
        blk = x[pos:pos + c]
        if c < m:
            blk = torch.cat([blk, torch.zeros((m - c,) + tuple(blk.shape[1:]), dtype=blk.dtype)], 0)
        out.append(blk.unsqueeze(0))
        pos += c
    return torch.cat(out, 0)


def bottleneck(x, sd, cfg, taps=None, counts=None):
    """BottleNeck.forward (:296-321): x [N, S, d, L]; counts[b] <= S speakers present per item (default: all)."""
    N, S, d, L = x.shape
    counts = [S] * N if counts is None else list(counts)
    pe = rel_pos_table(L, d, _t(sd, "bottleneck.pe_single.inv_freq"))
    for l in range(cfg.bottleneck_layers):
        p = f"bottleneck.module_list.{l}"
        h = _to_batches(x, counts).transpose(1, 2)                      # speakers_to_batches + transpose (:303-305)
        h = conformer_layer(h, sd, p + ".intra", cfg, pe)
        h = _to_speakers(h, counts)                                     # [N, S, L, d], missing speakers zero (:309)
        if taps is not None:
            taps[f"intra{l}"] = h
        h = h.permute(0, 2, 1, 3).reshape(N * L, S, d)                  # (:311-314): rows (n,t), sequence = speakers
        h = inter_layer(h, sd, p + ".inter.layers.0", cfg.num_head)
        x = h.reshape(N, L, S, d).permute(0, 2, 3, 1)
        if taps is not None:
            taps[f"inter{l}"] = x.permute(0, 1, 3, 2)
    return x


# --------------------------------------------------------------------------------------------
# Network.forward (:418-490)
# --------------------------------------------------------------------------------------------
def sep_forward(sd, cfg, mix: torch.Tensor, n_speakers, taps: dict = None) -> torch.Tensor:
    """mix [B, S*M, t] (already normalised) -> [B, max(S, max_speakers), t].  n_speakers: one count for every item,
    or a sequence of per-item counts (then S = the largest; only the first counts[b] channel blocks of item b are read)."""
    with torch.no_grad():
        B, SM, t_in = mix.shape
        M = cfg.n_mics
        counts = [int(n_speakers)] * B if np.isscalar(n_speakers) else [int(v) for v in np.asarray(n_speakers).reshape(-1)]
        S = max(counts)
        assert len(counts) == B and SM % M == 0 and SM >= S * M and min(counts) >= 1
        T = cfg.padded_length(t_in)
        mix = F.pad(mix, (T - t_in, 0))
        ref = mix[:, 0:1]                                              # first channel of the stack (:430)
        x = _to_batches(mix.reshape(B, SM // M, M, T), counts)         # (:441-442)
        x = F.conv1d(x, _t(sd, "preproc.weight"), _t(sd, "preproc.bias"))
        skips = [x]
        for i, s in enumerate(cfg.stride_list):
            x = _encoder_block(x, sd, i, s, cfg)
            skips.append(x)
            if taps is not None:
                taps[f"enc{i}"] = x
        x = bottleneck(_to_speakers(x, counts), sd, cfg, taps, counts)  # (:451-455)
        x = _to_batches(x, counts)                                     # (:458)
        if taps is not None:
            taps["bottleneck"] = x
        for i, (_ci, _co, s) in enumerate(cfg.dec_channels()):
            x = _decoder_block(x, skips[-(i + 1)], sd, i, s, cfg)
            if taps is not None:
                taps[f"dec{i}"] = x
        EK, ES = cfg.encoder_kernel_size, cfg.encoder_stride
        y = F.relu(F.conv1d(ref, _t(sd, "reference_bypass.weight"), _t(sd, "reference_bypass.bias"), stride=ES, padding=EK // 2))
        mask = F.relu(F.conv1d(x, _t(sd, "mask_encoder.weight"), _t(sd, "mask_encoder.bias"), stride=ES, padding=EK // 2))
        E, Fr = mask.shape[1], mask.shape[2]
        mask = _to_speakers(mask, counts)                              # [B, S, E, Fr], a missing speaker's mask is zero (:470)
        lat = (y.unsqueeze(1) * mask).reshape(B * S, E, Fr)            # (:466-477)
        out = F.conv_transpose1d(lat, _t(sd, "output_decoder.weight"), _t(sd, "output_decoder.bias"), stride=EK // 2)
        out = out.reshape(B, S, -1)[..., 9:-8]
        if S < cfg.max_speakers:
            out = F.pad(out, (0, 0, 0, cfg.max_speakers - S))
        return out[..., -t_in:]


def infer_sample(sd, cfg, mix: torch.Tensor, sample_list) -> np.ndarray:
    """Network.infer_sample (:496-548): mix [M,T], sample_list S x (M-1) -> ndarray [S,T]."""
    mix = mix.to(torch.float32)
    S = len(sample_list)
    # :507 rounds with np.round on the float64 offsets (half to even), not through a float32 tensor
    rounded = [np.round(np.asarray(o, dtype=np.float64)).astype(int) for o in sample_list]
    data = torch.cat([roll_channels(mix, o, circular=False) for o in rounded], dim=0).unsqueeze(0)
    dn, mu, sg = normalize_input(data)
    y = unnormalize_input(sep_forward(sd, cfg, dn, S), mu, sg)
    return y[0].numpy()[:S]
