"""ORACLE (test infrastructure, never on the product path).

CPU restatement of the reference's per-candidate hot loop: integer circular
shift -> int16 quantise + normalise -> spot network forward -> un-normalise ->
energies.  Written as plain functions over a state dict (torch CPU functional
ops / numpy), each citing the reference lines it follows.  Pinned against golden
vectors produced by the reference itself (tests/golden/make_golden.py,
tests/test_oracle_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# shift / normalise  (sep/training/JointModel/network.py:12-25,80-83;
#                     sep/training/SpeakerLocalization/network.py:28-47)
# --------------------------------------------------------------------------
def roll_channels(mix: torch.Tensor, sample_offset, circular: bool = True) -> torch.Tensor:
    """out[0] = mix[0]; out[m,t] = mix[m,(t + round(offset[m-1])) mod T].

    JointModel/network.py:81-83 builds shifts = -round([0,*offset]) and
    roll_by_gather reads index (t - shift) mod T.  torch.round is half-to-even.
    ``circular=False`` is the joint decoder's zero-filled variant
    (SpeakerSeparation/network.py:510-522)."""
    M, T = mix.shape
    off = torch.round(torch.tensor([0.0, *[float(o) for o in sample_offset]])).long()
    assert off.shape[0] == M
    idx = torch.arange(T).view(1, T) + off.view(M, 1)
    out = torch.gather(mix, 1, idx % T)
    if not circular:
        out = torch.where((idx >= 0) & (idx < T), out, torch.zeros_like(out))
    return out


def normalize_input(data: torch.Tensor):
    """SpeakerLocalization/network.py:28-40 (data [B,M,T])."""
    data = torch.round(data * 32768.0) / 32768.0
    ref = data.mean(1)
    means = ref.mean(1).view(-1, 1, 1)
    stds = ref.std(1).view(-1, 1, 1)          # unbiased (N-1)
    return (data - means) / stds, means, stds


def unnormalize_input(data, means, stds):
    """SpeakerLocalization/network.py:42-47."""
    return data * stds + means


# --------------------------------------------------------------------------
# spot network forward (SpeakerLocalization/network.py:50-405)
# --------------------------------------------------------------------------
def _t(sd, key):
    v = sd[key]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))


def _residual_sequence(x, sd, prefix, cfg):
    """DilatedResidualLayer x residual_layers (network.py:50-82):
    conv(dil=f**i, pad=3*dil) -> ReLU -> +x -> LayerNorm over channels."""
    K = cfg.kernel_size
    for j in range(cfg.residual_layers):
        d = cfg.residual_dilation_factor ** j
        p = f"{prefix}.res.seq.{j}"
        y = F.conv1d(x, _t(sd, p + ".conv.weight"), _t(sd, p + ".conv.bias"),
                     dilation=d, padding=(d * (K - 1) + 1) // 2)
        y = F.relu(y) + x
        c = y.shape[1]
        x = F.layer_norm(y.transpose(1, 2), (c,), _t(sd, p + ".norm.weight"),
                         _t(sd, p + ".norm.bias"), 1e-5).transpose(1, 2)
    return x


def _gate(sd, prefix, window_embedding):
    """embed1 = Conv1d(2->C,k=1) applied to the [B,2,1] window one-hot
    (network.py:101,186): gate[b,c] = W[c,:]·w[b,:] + bias[c]."""
    w = _t(sd, prefix + ".embed1.weight")[:, :, 0]            # [C,2]
    return (window_embedding @ w.t() + _t(sd, prefix + ".embed1.bias")).unsqueeze(2)


def _encoder_block(x, sd, i, stride, cfg, wemb):
    """EncoderBlock.forward (network.py:98-113)."""
    p = f"encoder.module_list.{i}"
    x = _residual_sequence(x, sd, p, cfg)
    x = _gate(sd, p, wemb) * x
    x = F.conv1d(x, _t(sd, p + ".conv1.weight"), _t(sd, p + ".conv1.bias"),
                 stride=stride, padding=cfg.kernel_size // 2)
    x = F.group_norm(x, 2, _t(sd, p + ".norm1.weight"), _t(sd, p + ".norm1.bias"), 1e-5)
    return F.glu(x, dim=1)


def _decoder_block(x, skip, sd, i, stride, cfg, wemb):
    """DecoderBlock.forward (network.py:180-200)."""
    p = f"decoder.module_list.{i}"
    x = x + skip
    x = F.conv_transpose1d(x, _t(sd, p + ".upsample.conv.weight"),
                           _t(sd, p + ".upsample.conv.bias"), stride=stride)
    x = _gate(sd, p, wemb) * x
    x = F.group_norm(x, 2, _t(sd, p + ".norm1.weight"), _t(sd, p + ".norm1.bias"), 1e-5)
    x = F.glu(x, dim=1)
    return _residual_sequence(x, sd, p, cfg)


def _transformer_layer(x, sd, p, nhead):
    """nn.TransformerEncoderLayer defaults (network.py:254): post-norm, ReLU,
    eps 1e-5, no dropout in eval.  x: [B, L, d] (batch-first restatement of the
    reference's (L, B, d) layout, network.py:261-263)."""
    B, L, d = x.shape
    hd = d // nhead
    qkv = F.linear(x, _t(sd, p + ".self_attn.in_proj_weight"), _t(sd, p + ".self_attn.in_proj_bias"))
    q, k, v = qkv.split(d, dim=-1)
    q = q.view(B, L, nhead, hd).transpose(1, 2)
    k = k.view(B, L, nhead, hd).transpose(1, 2)
    v = v.view(B, L, nhead, hd).transpose(1, 2)
    att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
    ctx = (att @ v).transpose(1, 2).reshape(B, L, d)
    sa = F.linear(ctx, _t(sd, p + ".self_attn.out_proj.weight"), _t(sd, p + ".self_attn.out_proj.bias"))
    x = F.layer_norm(x + sa, (d,), _t(sd, p + ".norm1.weight"), _t(sd, p + ".norm1.bias"), 1e-5)
    h = F.relu(F.linear(x, _t(sd, p + ".linear1.weight"), _t(sd, p + ".linear1.bias")))
    h = F.linear(h, _t(sd, p + ".linear2.weight"), _t(sd, p + ".linear2.bias"))
    return F.layer_norm(x + h, (d,), _t(sd, p + ".norm2.weight"), _t(sd, p + ".norm2.bias"), 1e-5)


def spot_forward(sd, cfg, mix: torch.Tensor, window_embedding: torch.Tensor,
                 taps: dict = None) -> torch.Tensor:
    """Network.forward (network.py:363-405).  mix [B,M,t] (already normalised),
    window_embedding [B,2] -> [B,1,t].  ``taps`` (optional dict) receives
    intermediate activations for layer-wise parity tests."""
    with torch.no_grad():
        t_in = mix.shape[-1]
        T = cfg.padded_length(t_in)
        mix = F.pad(mix, (T - t_in, 0))
        ref = mix[:, 0:1]
        x = F.conv1d(mix, _t(sd, "preproc.weight"), _t(sd, "preproc.bias"))
        if taps is not None:
            taps["preproc"] = x
        skips = [x]
        for i, s in enumerate(cfg.stride_list):
            x = _encoder_block(x, sd, i, s, cfg, window_embedding)
            skips.append(x)
            if taps is not None:
                taps[f"enc{i}"] = x
        h = x.permute(0, 2, 1)
        for l in range(cfg.num_transformer_layers):
            h = _transformer_layer(h, sd, f"bottleneck.transf.layers.{l}", cfg.num_head)
        x = h.permute(0, 2, 1)
        if taps is not None:
            taps["bottleneck"] = x
        for i, (_, _, s) in enumerate(cfg.dec_channels()):
            x = _decoder_block(x, skips[-(i + 1)], sd, i, s, cfg, window_embedding)
            if taps is not None:
                taps[f"dec{i}"] = x
        EK, ES = cfg.encoder_kernel_size, cfg.encoder_stride
        y = F.relu(F.conv1d(ref, _t(sd, "reference_bypass.weight"), _t(sd, "reference_bypass.bias"),
                            stride=ES, padding=EK // 2))
        mask = F.relu(F.conv1d(x, _t(sd, "mask_encoder.weight"), _t(sd, "mask_encoder.bias"),
                               stride=ES, padding=EK // 2))
        if taps is not None:
            taps["latent"] = y * mask
        x = F.conv_transpose1d(y * mask, _t(sd, "output_decoder.weight"),
                               _t(sd, "output_decoder.bias"), stride=EK // 2)
        x = x[..., 9:-8]
        return x[..., -t_in:]


def shift_and_sep(sd, cfg, mix: torch.Tensor, offsets, strict: int = 0,
                  batch_size: int = 128) -> np.ndarray:
    """DataParallelSpotModel.shift_and_sep (JointModel/network.py:37-104) on CPU:
    offsets is a sequence of per-candidate sample_offset vectors [M-1]."""
    mix = mix.to(torch.float32)
    N, T = len(offsets), mix.shape[-1]
    out = np.zeros((N, T), dtype=np.float32)
    w = torch.tensor([1.0, 0.0] if strict == 1 else [0.0, 1.0])
    for i in range(0, N, batch_size):
        chunk = offsets[i:i + batch_size]
        data = torch.stack([roll_channels(mix, o) for o in chunk])
        dn, mu, sg = normalize_input(data)
        y = spot_forward(sd, cfg, dn, w.expand(len(chunk), 2))
        out[i:i + len(chunk)] = unnormalize_input(y, mu, sg)[:, 0].numpy()
    return out


# --------------------------------------------------------------------------
# energies (sep/helpers/local_utils_3d.py:13-17,349-354; sep/Mic_Array.py:290-295)
# --------------------------------------------------------------------------
def max_avg_power(x: np.ndarray, window_size: int = 12000) -> float:
    """max_i sqrt(|mean(x^2[i:i+W])|) with zero padding on the right: the
    uniform_filter1d(size=W, mode='constant', origin=-W//2) of the reference is a
    forward-looking window.  x**2 stays in x's dtype, the
    running mean is float64, as scipy does for a float32 input."""
    sq = (x ** 2).astype(np.float64)
    c = np.concatenate([[0.0], np.cumsum(sq)])
    n = sq.shape[0]
    hi = np.minimum(np.arange(n) + window_size, n)
    win = (c[hi] - c[:n]) / window_size
    return float(np.sqrt(np.abs(win)).max())


def candidate_energies(y: np.ndarray, window_size: int = 12000) -> np.ndarray:
    """Per-candidate (power, power2) exactly as the stage loops compute them
    (local_utils_3d.py:349-354): mean-removal in float32, power = sum x^2 (fp32),
    power2 = max_avg_power."""
    out = np.zeros((y.shape[0], 2), dtype=np.float64)
    for i in range(y.shape[0]):
        x = y[i] - np.mean(y[i])
        out[i, 0] = np.sum(x ** 2)
        out[i, 1] = max_avg_power(x, window_size)
    return out


def si_sdr(est: np.ndarray, ref: np.ndarray) -> float:
    """sep/helpers/eval_utils.py:11-39 (scaling=True)."""
    rss = np.dot(ref, ref)
    a = np.dot(ref, est) / rss
    e_true = a * ref
    e_res = est - e_true
    return 10 * math.log10((e_true ** 2).sum() / ((e_res ** 2).sum() + 1e-8))
