"""Thin tensor-level wrappers over the individual C-ABI kernels (include/asw_hip.h).

Used by the parity tests (each kernel against the oracle) and by the host code of the
SRP-PHAT and clustering stages.  Tensors are torch CUDA tensors used purely as device
buffers; every function launches on torch's current stream.
"""
from ctypes import byref

import torch

from .native import ConvGemmArgs, MaskPathArgs, ResStackArgs, check, current_stream, lib, ptr


def _f32(t):
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), "need contiguous float32 CUDA tensor"
    return t


def pack_conv_weight(w: torch.Tensor) -> torch.Tensor:
    """torch Conv1d weight [N, Cin, K] -> Wt [N, K*Cin] (k-major, channel fastest)."""
    N, Cin, K = w.shape
    return w.permute(0, 2, 1).reshape(N, K * Cin).contiguous()


def convgemm(A, Wt, M_out, N, Cin, taps=1, stride=1, dil=1, pad=0, bias=None, relu=False, resid=None, mul=None,
             ln=None, stats_chan_mod=0, A2=None, B=None, a_row_stride=None, a_batch_stride=None, a_len=None,
             out=None, ln_eps=1e-5, precision="f32", use_fragments=True, glu=None, glu_out=None):
    """out[b][r][n] per include/asw_hip.h:asw_convgemm_f32.  Returns (out, stats|None)."""
    _f32(A); _f32(Wt)
    if B is None:
        B = A.shape[0]
    a_row_stride = Cin if a_row_stride is None else a_row_stride
    if a_batch_stride is None:
        a_batch_stride = A[0].numel()
    if a_len is None:
        a_len = a_batch_stride
    if out is None:
        out = torch.empty((B, M_out, N), dtype=torch.float32, device=A.device)
    stats = None
    if stats_chan_mod:
        tiles = lib().asw_convgemm_stats_tiles(M_out, N)
        stats = torch.zeros((B, tiles, 4), dtype=torch.float32, device=A.device)
    a = ConvGemmArgs()
    a.A, a.A2, a.Wt = A.data_ptr(), (A2.data_ptr() if A2 is not None else None), Wt.data_ptr()
    a.bias = bias.data_ptr() if bias is not None else None
    a.resid = resid.data_ptr() if resid is not None else None
    a.mul = mul.data_ptr() if mul is not None else None
    a.ln_gamma = ln[0].data_ptr() if ln is not None else None
    a.ln_beta = ln[1].data_ptr() if ln is not None else None
    a.out = out.data_ptr()
    a.stats = stats.data_ptr() if stats is not None else None
    a.B, a.M_out, a.N, a.Cin, a.taps, a.stride, a.dil, a.pad = B, M_out, N, Cin, taps, stride, dil, pad
    a.a_row_stride, a.a_batch_stride, a.a_len = a_row_stride, a_batch_stride, a_len
    a.chan_mod, a.relu, a.ln_eps = stats_chan_mod, int(relu), ln_eps
    if glu is not None:                      # (raw [B][M_out][2N], mr [B][4], gamma [2N], beta [2N]): GroupNorm + GLU on load
        a.glu_raw, a.glu_mr, a.glu_gamma, a.glu_beta = (_f32(t).data_ptr() for t in glu)
        if glu_out is not None:
            a.glu_out = _f32(glu_out).data_ptr()
    keep = None
    if precision in ("f16x3", "f16"):
        hi, lo, shift = split_weights_f16(Wt)
        keep = (hi, lo)
        a.precision, a.w_shift, a.Wt_hi, a.Wt_lo = (1 if precision == "f16x3" else 2), shift, hi.data_ptr(), lo.data_ptr()
        if use_fragments and N % 32 == 0 and (taps * Cin) % 16 == 0:
            fh, fl, sh2 = pack_fragments_f16(Wt, N, taps * Cin)
            assert sh2 == shift
            keep = keep + (fh, fl)
            a.Wf_hi, a.Wf_lo = fh.data_ptr(), fl.data_ptr()
    check(lib().asw_convgemm_f32(byref(a), current_stream()))
    torch.cuda.current_stream().synchronize() if keep is not None else None
    return out, stats


def resstack(x, layers, taps=7, precision="f16x3", eps=1e-5, glu=None, out=None, glu_out=None):
    """A stack of 1..3 64-channel residual layers in one launch (include/asw_hip.h:asw_resstack64_f16x3).
    layers: [(Wt [64][taps*64], bias, gamma, beta, dil), ...]; x [B][T][64] or None with glu = (raw, mr, gamma, beta)."""
    src = x if glu is None else glu[0]
    B, T = src.shape[0], src.shape[1]
    if out is None:
        out = torch.empty((B, T, 64), dtype=torch.float32, device=src.device)
    a = ResStackArgs()
    a.x = _f32(x).data_ptr() if x is not None else None
    a.out = out.data_ptr()
    a.B, a.T, a.C, a.taps, a.n_layers = B, T, 64, taps, len(layers)
    a.precision, a.ln_eps = (1 if precision == "f16x3" else 2), eps
    keep = []
    for i, (Wt, bias, gamma, beta, dil) in enumerate(layers):
        fh, fl, sh = pack_fragments_f16(_f32(Wt), 64, taps * 64)
        keep.append((fh, fl))
        d = a.layer[i]
        d.Wf_hi, d.Wf_lo, d.w_shift, d.dil = fh.data_ptr(), fl.data_ptr(), sh, dil
        d.bias, d.ln_gamma, d.ln_beta = _f32(bias).data_ptr(), _f32(gamma).data_ptr(), _f32(beta).data_ptr()
    if glu is not None:
        a.glu_raw, a.glu_mr, a.glu_gamma, a.glu_beta = (_f32(t).data_ptr() for t in glu)
        if glu_out is not None:
            a.glu_out = _f32(glu_out).data_ptr()
    check(lib().asw_resstack64_f16x3(byref(a), current_stream()))
    torch.cuda.current_stream().synchronize()
    return out


def pack_fragments_f16(Wt, N, K):
    """fp32 device Wt[N][K] -> fragment-major (hi, lo) device tensors + shift (asw_pack_fragments_f16)."""
    import ctypes
    import numpy as np
    w = np.ascontiguousarray(Wt.detach().cpu().numpy(), dtype=np.float32).reshape(N, K)
    hi = np.empty(w.size, dtype=np.uint16)
    lo = np.empty(w.size, dtype=np.uint16)
    sh = ctypes.c_int32()
    check(lib().asw_pack_fragments_f16(ctypes.c_void_p(w.ctypes.data), N, K, ctypes.c_void_p(hi.ctypes.data),
                                       ctypes.c_void_p(lo.ctypes.data), byref(sh)))
    dev = Wt.device
    return (torch.from_numpy(hi.view(np.int16)).to(dev), torch.from_numpy(lo.view(np.int16)).to(dev), sh.value)


def split_weights_f16(Wt):
    """fp32 device weights -> (hi, lo) fp16-bit uint16 device tensors + power-of-two shift,
    through the library's own host splitter (asw_split_weights_f16)."""
    import ctypes
    import numpy as np
    w = np.ascontiguousarray(Wt.detach().cpu().numpy(), dtype=np.float32)
    hi = np.empty(w.size, dtype=np.uint16)
    lo = np.empty(w.size, dtype=np.uint16)
    sh = ctypes.c_int32()
    check(lib().asw_split_weights_f16(ctypes.c_void_p(w.ctypes.data), w.size, ctypes.c_void_p(hi.ctypes.data),
                                      ctypes.c_void_p(lo.ctypes.data), byref(sh)))
    dev = Wt.device
    return (torch.from_numpy(hi.view(np.int16)).to(dev), torch.from_numpy(lo.view(np.int16)).to(dev), sh.value)


def gn_glu(raw, stats, gamma, beta, eps=1e-5):
    B, T, C2 = raw.shape
    C = C2 // 2
    out = torch.empty((B, T, C), dtype=torch.float32, device=raw.device)
    check(lib().asw_gn_glu(ptr(_f32(raw)), ptr(_f32(stats)), stats.shape[1], ptr(_f32(gamma)), ptr(_f32(beta)),
                           B, T, C, eps, ptr(out), current_stream()))
    return out


def gn_finalize(stats, T, C, eps=1e-5):
    """Partial GroupNorm sums [B][n][4] -> (mean0, rstd0, mean1, rstd1) per batch item (asw_gn_finalize)."""
    B = stats.shape[0]
    mr = torch.empty((B, 4), dtype=torch.float32, device=stats.device)
    check(lib().asw_gn_finalize(ptr(_f32(stats)), stats.shape[1], B, T, C, eps, ptr(mr), current_stream()))
    return mr


def add_layernorm(x, resid, gamma, beta, eps=1e-5):
    rows, N = x.shape
    out = torch.empty_like(x)
    check(lib().asw_add_layernorm(ptr(_f32(x)), ptr(_f32(resid)), ptr(_f32(gamma)), ptr(_f32(beta)), rows, N, eps,
                                  ptr(out), current_stream()))
    return out


def attention(qkv, nhead, precision="f32"):
    B, L, d3 = qkv.shape
    d = d3 // 3
    ctx = torch.empty((B, L, d), dtype=torch.float32, device=qkv.device)
    check(lib().asw_attention_prec(ptr(_f32(qkv)), B, L, d, nhead, {"f32": 0, "f16x3": 1, "f16": 2}[precision], ptr(ctx),
                                   current_stream()))
    return ctx


def shift_stats(mix, offsets, circular=True):
    M, T = mix.shape
    N = offsets.shape[0]
    mean = torch.empty((N,), dtype=torch.float32, device=mix.device)
    std = torch.empty((N,), dtype=torch.float32, device=mix.device)
    check(lib().asw_shift_stats(ptr(_f32(mix)), M, T, ptr(offsets), N, int(circular), ptr(mean), ptr(std),
                                current_stream()))
    return mean, std


def shift_norm_preproc(mix, offsets, mean, std, w, b, T_pad, circular=True):
    M, T = mix.shape
    N = offsets.shape[0]
    C = w.shape[0]
    x0 = torch.empty((N, T_pad, C), dtype=torch.float32, device=mix.device)
    refn = torch.zeros((N, T_pad), dtype=torch.float32, device=mix.device)
    check(lib().asw_shift_norm_preproc(ptr(_f32(mix)), M, T, T_pad, ptr(offsets), N, int(circular), ptr(mean),
                                       ptr(std), ptr(_f32(w)), ptr(_f32(b)), C, ptr(x0), ptr(refn), T_pad,
                                       current_stream()))
    return x0, refn


def overlap_add_unnorm(D, taps, hop, t, trim_left, trim_right, bias, mean=None, std=None):
    B, F, ldd = D.shape
    out = torch.empty((B, t), dtype=torch.float32, device=D.device)
    check(lib().asw_overlap_add_unnorm(ptr(_f32(D)), B, F, ldd, taps, hop, t, trim_left, trim_right, float(bias),
                                       ptr(mean), ptr(std), ptr(out), current_stream()))
    return out


def overlap_add_parts(Dp, taps, hop, t, trim_left, trim_right, bias, mean=None, std=None):
    """Dp [nparts][B][F][ldd] partial tap products (mask_path) -> out [B][t]."""
    nparts, B, F, ldd = Dp.shape
    out = torch.empty((B, t), dtype=torch.float32, device=Dp.device)
    check(lib().asw_overlap_add_parts(ptr(_f32(Dp)), nparts, B, F, ldd, taps, hop, t, trim_left, trim_right,
                                      float(bias), ptr(mean), ptr(std), ptr(out), current_stream()))
    return out


def mask_path(x, ref, ref_hop, enc_w, enc_b, byp_w, byp_b, dec_w, frames, stride, pad):
    """Fused mask path (asw_mask_path_f16x3).  x [B][Tp][C] channels-last activations, ref [B][RL] padded
    reference rows (frame f reads ref[b][f*ref_hop + k]), enc_w [E][C][EK], byp_w [E][1][EKb], dec_w
    [E][1][EKd] torch-layout weights.  Returns the partial tap products [E/256][B][frames][64]."""
    B, Tp, C = x.shape
    E, _, EK = enc_w.shape
    dev = x.device
    Wt = pack_conv_weight(enc_w).to(dev)
    fh, fl, sh = pack_fragments_f16(Wt, E, EK * C)
    wb = torch.zeros((E, 48), dtype=torch.float32)
    wb[:, :byp_w.shape[-1]] = byp_w[:, 0].cpu()
    bh, bl, bsh = pack_fragments_f16(wb.to(dev), E, 48)
    wd = torch.zeros((64, E), dtype=torch.float32)
    wd[:dec_w.shape[-1]] = dec_w[:, 0].t().cpu()
    dh, dl, dsh = pack_fragments_f16(wd.to(dev), 64, E)
    parts = torch.zeros((E // 256, B, frames, 64), dtype=torch.float32, device=dev)
    m = MaskPathArgs()
    a = m.enc
    a.A, a.bias = _f32(x).data_ptr(), (_f32(enc_b).data_ptr() if enc_b is not None else None)
    a.B, a.M_out, a.N, a.Cin, a.taps, a.stride, a.dil, a.pad = B, frames, E, C, EK, stride, 1, pad
    a.a_row_stride, a.a_batch_stride, a.a_len = C, Tp * C, Tp * C
    a.precision, a.w_shift, a.Wf_hi, a.Wf_lo = 1, sh, fh.data_ptr(), fl.data_ptr()
    m.ref, m.ref_batch_stride, m.ref_len, m.ref_hop = _f32(ref).data_ptr(), ref.shape[1], ref.shape[1], ref_hop
    m.byp_k, m.byp_taps, m.byp_shift = 48, byp_w.shape[-1], bsh
    m.byp_hi, m.byp_lo = bh.data_ptr(), bl.data_ptr()
    m.byp_bias = _f32(byp_b).data_ptr() if byp_b is not None else None
    m.dec_hi, m.dec_lo, m.dec_shift, m.dec_taps = dh.data_ptr(), dl.data_ptr(), dsh, dec_w.shape[-1]
    m.taps = parts.data_ptr()
    check(lib().asw_mask_path_f16x3(byref(m), current_stream()))
    torch.cuda.current_stream().synchronize()
    return parts


def energies(y, window=12000):
    B, T = y.shape
    scratch = torch.empty((B, T + 1), dtype=torch.float64, device=y.device)
    out = torch.empty((B, 2), dtype=torch.float64, device=y.device)
    check(lib().asw_energies(ptr(_f32(y)), B, T, window, ptr(scratch), ptr(out), current_stream()))
    return out


def pair_sisdr(y):
    n, T = y.shape
    out = torch.empty((n, n), dtype=torch.float64, device=y.device)
    check(lib().asw_pair_sisdr(ptr(_f32(y)), n, T, ptr(out), current_stream()))
    return out


# ---- kernels of the joint separation network (csrc/sep_kernels.hip) --------------------------
def joint_shift_stats(mix, offsets):
    """mean / unbiased std of the all-channel average of the S*M zero-fill shifted channels."""
    M, T = mix.shape
    S = offsets.shape[0]
    scratch = torch.empty((lib().asw_joint_shift_stats_scratch_doubles(),), dtype=torch.float64, device=mix.device)
    mean = torch.empty((S,), dtype=torch.float32, device=mix.device)
    std = torch.empty((S,), dtype=torch.float32, device=mix.device)
    check(lib().asw_joint_shift_stats(ptr(_f32(mix)), M, T, ptr(offsets), S, ptr(scratch), ptr(mean), ptr(std),
                                      current_stream()))
    return mean, std


def add_layernorm2(x, y=None, alpha=1.0, gamma=None, beta=None, eps=1e-5, act=0, want_sum=False, want_ln=True):
    rows, N = x.shape
    s = torch.empty_like(x) if want_sum else None
    o = torch.empty_like(x) if want_ln else None
    check(lib().asw_add_layernorm2(ptr(_f32(x)), ptr(y), float(alpha), ptr(gamma), ptr(beta), rows, N, eps, act, ptr(s),
                                   ptr(o), current_stream()))
    return s, o


def glu_rows(raw):
    rows, C2 = raw.shape
    out = torch.empty((rows, C2 // 2), dtype=torch.float32, device=raw.device)
    check(lib().asw_glu_rows(ptr(_f32(raw)), rows, C2 // 2, ptr(out), current_stream()))
    return out


def dwconv_ln_swish(u, w, bias, gamma, beta, eps=1e-5):
    """u [BS, L, d]; w torch depthwise weight [d, 1, K]."""
    BS, L, d = u.shape
    K = w.shape[-1]
    wT = w.reshape(d, K).t().contiguous()
    out = torch.empty_like(u)
    check(lib().asw_dwconv_ln_swish(ptr(_f32(u)), ptr(_f32(wT)), ptr(_f32(bias)), ptr(_f32(gamma)), ptr(_f32(beta)),
                                    BS, L, d, K, eps, ptr(out), current_stream()))
    return out


def relpos_attention(qkv, P, bias_u, bias_v, nhead, scale):
    BS, L, d3 = qkv.shape
    d = d3 // 3
    ctx = torch.empty((BS, L, d), dtype=torch.float32, device=qkv.device)
    check(lib().asw_relpos_attention(ptr(_f32(qkv)), ptr(_f32(P)), ptr(_f32(bias_u)), ptr(_f32(bias_v)), BS, L, d, nhead,
                                     float(scale), ptr(ctx), current_stream()))
    return ctx


def inter_attention(qkv, nhead):
    NB, S, L, d3 = qkv.shape
    d = d3 // 3
    ctx = torch.empty((NB, S, L, d), dtype=torch.float32, device=qkv.device)
    check(lib().asw_inter_attention(ptr(_f32(qkv)), NB, S, L, d, nhead, ptr(ctx), current_stream()))
    return ctx


def f16x3_overflow_count(reset=True) -> int:
    """Threads of f16x3 launches that produced an activation beyond the fp16 range since the last
    reset (asw_f16x3_overflow_count); 0 means no later GEMM clipped its input."""
    import ctypes
    c = ctypes.c_int32()
    check(lib().asw_f16x3_overflow_count(int(reset), byref(c)))
    return int(c.value) & 0xFFFFFFFF
