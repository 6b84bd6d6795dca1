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


# ---------------------------------------------------------------------------------------------
# speechbrain stand-ins for the joint separation network's bottleneck
# (sep/training/SpeakerSeparation/network.py:8-9,285,290).  speechbrain is absent from the image
# and unpinned in the reference's requirements.txt, so these nn.Modules are THIS repo's
# restatement of the published definitions (speechbrain.lobes.models.transformer.Conformer:
# ConformerEncoder / ConformerEncoderLayer / ConvolutionModule; speechbrain.nnet.attention:
# RelPosEncXL / RelPosMHAXL / PositionalwiseFeedForward; speechbrain.nnet.normalization.LayerNorm;
# speechbrain.nnet.activations.Swish) with the same parameter names.  Fixtures that flow through
# them pin the reference's own wiring (U-Net, speaker/batch reshapes, inter-speaker attention,
# mask path, infer_sample) and pin the oracle to these modules -- not the Conformer arithmetic to
# speechbrain itself: "parity unpinned" for that.
# ---------------------------------------------------------------------------------------------
import math as _math

import torch as _torch
import torch.nn as _nn
import torch.nn.functional as _F


class Swish(_nn.Module):
    def forward(self, x):
        return x * _torch.sigmoid(x)


class SbLayerNorm(_nn.Module):
    """speechbrain.nnet.normalization.LayerNorm: wraps torch's module as ``.norm``."""

    def __init__(self, input_size, eps=1e-5):
        super().__init__()
        self.norm = _nn.LayerNorm(input_size, eps=eps)

    def forward(self, x):
        return self.norm(x)


class RelPosEncXL(_nn.Module):
    """Transformer-XL sinusoidal table for relative positions L-1 .. -(L-1): [1, 2L-1, d]."""

    def __init__(self, emb_dim):
        super().__init__()
        self.emb_dim = emb_dim
        inv_freq = _torch.exp(_torch.arange(0, emb_dim, 2, dtype=_torch.float32) * -(_math.log(10000.0) / emb_dim))
        self.register_buffer("inv_freq", inv_freq)

    def forward(self, x):
        seq_len = x.size(1)
        with _torch.no_grad():
            tot_pe = _torch.zeros((2, seq_len, self.emb_dim), dtype=x.dtype).to(x)
            pe_past, pe_future = tot_pe[0], tot_pe[1]
            positions = _torch.arange(0, seq_len, dtype=x.dtype).to(x).unsqueeze(-1)
            sinusoids = _torch.sin(positions * self.inv_freq)
            pe_past[:, 0::2] = sinusoids
            pe_past[:, 1::2] = _torch.cos(positions * self.inv_freq)
            pe_future[:, 0::2] = sinusoids                       # same for past and future
            pe_future[:, 1::2] = _torch.cos(-positions * self.inv_freq)
            pe_past = _torch.flip(pe_past, (0,)).unsqueeze(0)
            pe_future = pe_future[1:].unsqueeze(0)
            return _torch.cat([pe_past, pe_future], dim=1)


class RelPosMHAXL(_nn.Module):
    def __init__(self, embed_dim, num_heads):
        super().__init__()
        self.embed_dim, self.num_heads, self.head_dim = embed_dim, num_heads, embed_dim // num_heads
        self.in_proj_weight = _nn.Parameter(_torch.empty(3 * embed_dim, embed_dim))
        self.out_proj = _nn.Linear(embed_dim, embed_dim)
        self.linear_pos = _nn.Linear(embed_dim, embed_dim, bias=False)
        self.pos_bias_u = _nn.Parameter(_torch.empty(self.head_dim, self.num_heads))
        self.pos_bias_v = _nn.Parameter(_torch.empty(self.head_dim, self.num_heads))
        _nn.init.xavier_uniform_(self.in_proj_weight)
        _nn.init.xavier_uniform_(self.pos_bias_u)
        _nn.init.xavier_uniform_(self.pos_bias_v)
        self.scale = 1 / _math.sqrt(self.embed_dim)             # embed_dim, not head_dim

    @staticmethod
    def rel_shift(x):
        b, h, qlen, pos_len = x.size()
        x = _F.pad(x, pad=(1, 0))
        x = x.view(b, h, -1, qlen)
        x = x[:, :, 1:].view(b, h, qlen, pos_len)
        return x[..., : pos_len // 2 + 1]

    def forward(self, query, key, value, pos_embs):
        bsz = query.shape[0]
        H, hd = self.num_heads, self.head_dim
        query, key, value = _F.linear(query, self.in_proj_weight).view(bsz, -1, H, hd * 3).chunk(3, dim=-1)
        p_k = self.linear_pos(pos_embs).view(1, -1, H, hd)
        q_u = (query + self.pos_bias_u.view(1, 1, H, hd)).transpose(1, 2)
        q_v = (query + self.pos_bias_v.view(1, 1, H, hd)).transpose(1, 2)
        matrix_ac = _torch.matmul(q_u * self.scale, key.permute(0, 2, 3, 1))
        matrix_bd = self.rel_shift(_torch.matmul(q_v * self.scale, p_k.permute(0, 2, 3, 1)))
        attn = _F.softmax(matrix_ac + matrix_bd, -1, dtype=_torch.float32)
        x = _torch.matmul(attn, value.transpose(1, 2))
        x = x.transpose(1, 2).contiguous().view(bsz, -1, hd * H)
        return self.out_proj(x), attn


class PositionalwiseFeedForward(_nn.Module):
    def __init__(self, d_ffn, input_size):
        super().__init__()
        self.ffn = _nn.Sequential(_nn.Linear(input_size, d_ffn), Swish(), _nn.Dropout(0.0), _nn.Linear(d_ffn, input_size))

    def forward(self, x):
        return self.ffn(x)


class ConvolutionModule(_nn.Module):
    def __init__(self, input_size, kernel_size=31):
        super().__init__()
        self.layer_norm = _nn.LayerNorm(input_size)
        self.bottleneck = _nn.Sequential(_nn.Conv1d(input_size, 2 * input_size, kernel_size=1), _nn.GLU(dim=1))
        self.conv = _nn.Conv1d(input_size, input_size, kernel_size=kernel_size, padding=(kernel_size - 1) // 2,
                               groups=input_size)
        self.after_conv = _nn.Sequential(_nn.LayerNorm(input_size), Swish(), _nn.Linear(input_size, input_size),
                                         _nn.Dropout(0.0))

    def forward(self, x):
        out = self.layer_norm(x).transpose(1, 2)
        out = self.conv(self.bottleneck(out)).transpose(1, 2)
        return self.after_conv(out)


class ConformerEncoderLayer(_nn.Module):
    def __init__(self, d_model, d_ffn, nhead, kernel_size=31):
        super().__init__()
        self.mha_layer = RelPosMHAXL(d_model, nhead)
        self.convolution_module = ConvolutionModule(d_model, kernel_size)
        self.ffn_module1 = _nn.Sequential(_nn.LayerNorm(d_model), PositionalwiseFeedForward(d_ffn, d_model), _nn.Dropout(0.0))
        self.ffn_module2 = _nn.Sequential(_nn.LayerNorm(d_model), PositionalwiseFeedForward(d_ffn, d_model), _nn.Dropout(0.0))
        self.norm1 = SbLayerNorm(d_model)
        self.norm2 = SbLayerNorm(d_model)

    def forward(self, x, pos_embs=None):
        x = x + 0.5 * self.ffn_module1(x)
        skip = x
        x = self.norm1(x)
        x, attn = self.mha_layer(x, x, x, pos_embs=pos_embs)
        x = x + skip
        x = x + self.convolution_module(x)
        x = self.norm2(x + 0.5 * self.ffn_module2(x))
        return x, attn


class ConformerEncoder(_nn.Module):
    def __init__(self, num_layers, d_model, d_ffn, nhead, kernel_size=31, **_unused):
        super().__init__()
        self.layers = _nn.ModuleList([ConformerEncoderLayer(d_model, d_ffn, nhead, kernel_size) for _ in range(num_layers)])
        self.norm = SbLayerNorm(d_model, eps=1e-6)

    def forward(self, src, pos_embs=None):
        if pos_embs is None:
            raise ValueError("RelPosMHAXL needs positional embeddings")
        out, attns = src, []
        for layer in self.layers:
            out, a = layer(out, pos_embs=pos_embs)
            attns.append(a)
        return self.norm(out), attns
