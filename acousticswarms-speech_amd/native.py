"""ctypes binding of libasw_hip.so (the C ABI declared in include/asw_hip.h).

The library is the only implementation of the hot path: there is no CPU or
PyTorch fallback.  ``lib()`` raises if the shared object is missing, and every
wrapper raises ``RuntimeError`` with ``asw_last_error()`` on a non-zero status.
"""
import ctypes
import os
import subprocess
import sys
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64,
                    c_long, c_size_t, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ASW_LIB_PATH") or os.path.join(_HERE, "libasw_hip.so")   # env: A/B builds only
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["asw_common.cpp", "convgemm.hip", "resstack.hip", "downconv.hip", "prep_kernels.hip", "misc_kernels.hip", "attention_mfma.hip", "srp_kernels.hip",
           "search_host.cpp", "sep_kernels.hip", "spot_model.hip", "sep_model.hip"]
HEADERS = ["asw_common.h", "model_common.h", "mfma_util.h"]
OPS_PATH = os.path.join(_HERE, "libasw_torch_ops.so")      # TORCH_LIBRARY(asw, ...) adapters over the C ABI
OPS_SOURCE = "torch_ops.cpp"


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into libasw_hip.so (in-tree): one object per source
    under build/ (only the stale ones, in parallel), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(os.path.dirname(_HERE), "include", "asw_hip.h")]
    hdr_time = max(os.path.getmtime(h) for h in hdrs)
    objdir = os.path.join(_HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs, objs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        objs.append(op)
        if force or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), hdr_time):
            jobs.append(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-c", sp, "-o", op])
    if not jobs and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(o) for o in objs):
        return LIB_PATH

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    run(["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB_PATH] + objs)
    return LIB_PATH


def build_torch_ops(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/torch_ops.cpp (host code only: TORCH_LIBRARY registration + argument checks)
    against this interpreter's torch headers and link it to libasw_hip.so, in-tree."""
    import torch
    src = os.path.join(CSRC, OPS_SOURCE)
    hdr = os.path.join(os.path.dirname(_HERE), "include", "asw_hip.h")
    if not force and os.path.exists(OPS_PATH) and os.path.getmtime(OPS_PATH) >= max(
            os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(LIB_PATH)):
        return OPS_PATH
    ti = os.path.dirname(torch.__file__)
    cmd = ["hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-w",
           f"-I{ti}/include", f"-I{ti}/include/torch/csrc/api/include", "-I/opt/rocm/include", src, "-o", OPS_PATH,
           f"-L{_HERE}", "-lasw_hip", f"-L{ti}/lib", "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip",
           "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{ti}/lib"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return OPS_PATH


class SpotConfigC(Structure):
    _fields_ = [("n_mics", c_int32), ("kernel_size", c_int32), ("depth", c_int32),
                ("stride_list", c_int32 * 8), ("channels", c_int32), ("growth", c_int32),
                ("encoder_channels", c_int32), ("encoder_kernel_size", c_int32),
                ("encoder_stride", c_int32), ("residual_layers", c_int32),
                ("residual_dilation_factor", c_int32), ("num_head", c_int32), ("ffw_dim", c_int32),
                ("num_transformer_layers", c_int32)]

    @classmethod
    def from_config(cls, cfg):
        s = cls()
        s.n_mics, s.kernel_size, s.depth = cfg.n_mics, cfg.kernel_size, cfg.depth
        for i, v in enumerate(cfg.stride_list):
            s.stride_list[i] = v
        s.channels, s.growth = cfg.channels, int(cfg.growth)
        s.encoder_channels, s.encoder_kernel_size = cfg.encoder_channels, cfg.encoder_kernel_size
        s.encoder_stride, s.residual_layers = cfg.encoder_stride, cfg.residual_layers
        s.residual_dilation_factor, s.num_head = cfg.residual_dilation_factor, cfg.num_head
        s.ffw_dim, s.num_transformer_layers = cfg.ffw_dim, cfg.num_transformer_layers
        return s


class SepConfigC(Structure):
    _fields_ = [("n_mics", c_int32), ("max_speakers", c_int32), ("kernel_size", c_int32), ("depth", c_int32),
                ("stride_list", c_int32 * 8), ("channels", c_int32), ("growth", c_int32),
                ("encoder_channels", c_int32), ("encoder_kernel_size", c_int32), ("encoder_stride", c_int32),
                ("residual_layers", c_int32), ("residual_dilation_factor", c_int32), ("num_head", c_int32),
                ("ffw_dim", c_int32), ("bottleneck_layers", c_int32), ("bottleneck_ksize", c_int32)]

    @classmethod
    def from_config(cls, cfg):
        s = cls()
        s.n_mics, s.max_speakers, s.kernel_size, s.depth = cfg.n_mics, cfg.max_speakers, cfg.kernel_size, cfg.depth
        for i, v in enumerate(cfg.stride_list):
            s.stride_list[i] = v
        s.channels, s.growth = cfg.channels, int(cfg.growth)
        s.encoder_channels, s.encoder_kernel_size = cfg.encoder_channels, cfg.encoder_kernel_size
        s.encoder_stride, s.residual_layers = cfg.encoder_stride, cfg.residual_layers
        s.residual_dilation_factor, s.num_head = cfg.residual_dilation_factor, cfg.num_head
        s.ffw_dim, s.bottleneck_layers, s.bottleneck_ksize = cfg.ffw_dim, cfg.bottleneck_layers, cfg.bottleneck_ksize
        return s


class ConvGemmArgs(Structure):
    _fields_ = [("A", c_void_p), ("A2", c_void_p), ("Wt", c_void_p), ("bias", c_void_p),
                ("resid", c_void_p), ("mul", c_void_p), ("ln_gamma", c_void_p), ("ln_beta", c_void_p),
                ("out", c_void_p), ("stats", c_void_p),
                ("B", c_int32), ("M_out", c_int32), ("N", c_int32), ("Cin", c_int32), ("taps", c_int32),
                ("stride", c_int32), ("dil", c_int32), ("pad", c_int32), ("a_row_stride", c_int32),
                ("a_batch_stride", c_int64), ("a_len", c_int64), ("chan_mod", c_int32), ("relu", c_int32),
                ("ln_eps", c_float), ("precision", c_int32), ("w_shift", c_int32), ("Wt_hi", c_void_p),
                ("Wt_lo", c_void_p), ("Wf_hi", c_void_p), ("Wf_lo", c_void_p), ("stats_stride", c_int32),
                ("glu_raw", c_void_p), ("glu_mr", c_void_p), ("glu_gamma", c_void_p), ("glu_beta", c_void_p),
                ("glu_out", c_void_p)]


class MaskPathArgs(Structure):
    _fields_ = [("enc", ConvGemmArgs), ("ref", c_void_p), ("ref_batch_stride", c_int64), ("ref_len", c_int64),
                ("ref_hop", c_int32), ("byp_k", c_int32), ("byp_taps", c_int32), ("byp_shift", c_int32),
                ("byp_hi", c_void_p), ("byp_lo", c_void_p), ("byp_bias", c_void_p), ("dec_hi", c_void_p),
                ("dec_lo", c_void_p), ("dec_shift", c_int32), ("dec_taps", c_int32), ("taps", c_void_p)]


class ResLayerDesc(Structure):
    _fields_ = [("Wf_hi", c_void_p), ("Wf_lo", c_void_p), ("bias", c_void_p), ("ln_gamma", c_void_p),
                ("ln_beta", c_void_p), ("dil", c_int32), ("w_shift", c_int32)]


class ResStackArgs(Structure):
    _fields_ = [("x", c_void_p), ("out", c_void_p), ("B", c_int32), ("T", c_int32), ("C", c_int32), ("taps", c_int32),
                ("n_layers", c_int32), ("precision", c_int32), ("ln_eps", c_float), ("layer", ResLayerDesc * 3),
                ("glu_raw", c_void_p), ("glu_mr", c_void_p), ("glu_gamma", c_void_p), ("glu_beta", c_void_p),
                ("glu_out", c_void_p)]


# name -> (restype, argtypes); must list every symbol declared in include/asw_hip.h
SIGNATURES = {
    "asw_last_error": (c_char_p, []),
    "asw_abi_version": (c_int, []),
    "asw_profile_enable": (c_int, [c_int]),
    "asw_profile_report": (c_int, [c_char_p, c_size_t]),
    "asw_spot_create": (c_int, [POINTER(SpotConfigC), POINTER(c_void_p)]),
    "asw_spot_destroy": (None, [c_void_p]),
    "asw_spot_set_param": (c_int, [c_void_p, c_char_p, c_void_p, c_size_t]),
    "asw_spot_finalize": (c_int, [c_void_p]),
    "asw_spot_set_batch": (c_int, [c_void_p, c_int]),
    "asw_spot_set_precision": (c_int, [c_void_p, c_int]),
    "asw_spot_set_lanes": (c_int, [c_void_p, c_int]),
    "asw_split_weights_f16": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, POINTER(c_int32)]),
    "asw_pack_fragments_f16": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, POINTER(c_int32)]),
    "asw_spot_shift_and_sep": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int,
                                       c_void_p, c_void_p, c_int, c_void_p]),
    "asw_spot_shift_and_sep_multi": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                             c_void_p, c_void_p, c_int, c_void_p]),
    "asw_spot_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, POINTER(c_float), c_void_p, c_void_p]),
    "asw_spot_set_fused_mask": (c_int, [c_void_p, c_int]),
    "asw_spot_get_tap": (c_int, [c_void_p, c_char_p, c_void_p, c_size_t, POINTER(c_size_t), c_void_p]),
    "asw_sep_create": (c_int, [POINTER(SepConfigC), POINTER(c_void_p)]),
    "asw_sep_destroy": (None, [c_void_p]),
    "asw_sep_set_param": (c_int, [c_void_p, c_char_p, c_void_p, c_size_t]),
    "asw_sep_finalize": (c_int, [c_void_p]),
    "asw_sep_set_precision": (c_int, [c_void_p, c_int]),
    "asw_sep_infer": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "asw_sep_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "asw_sep_forward_counts": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "asw_sep_get_config": (c_int, [c_void_p, POINTER(SepConfigC)]),
    "asw_sep_get_tap": (c_int, [c_void_p, c_char_p, c_void_p, c_size_t, POINTER(c_size_t), c_void_p]),
    "asw_joint_shift_stats": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "asw_joint_shift_stats_scratch_doubles": (c_int, []),
    "asw_add_layernorm2": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_int, c_float, c_int,
                                   c_void_p, c_void_p, c_void_p]),
    "asw_glu_rows": (c_int, [c_void_p, c_long, c_int, c_void_p, c_void_p]),
    "asw_dwconv_ln_swish": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                    c_float, c_void_p, c_void_p]),
    "asw_relpos_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                     c_void_p, c_void_p]),
    "asw_inter_attention": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "asw_shift_stats": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "asw_shift_norm_preproc": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_long, c_void_p]),
    "asw_shift_stats_multi": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "asw_shift_norm_preproc_multi": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_long, c_void_p]),
    "asw_pad_preproc": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                c_void_p, c_long, c_void_p]),
    "asw_convgemm_f32": (c_int, [POINTER(ConvGemmArgs), c_void_p]),
    "asw_mask_path_f16x3": (c_int, [POINTER(MaskPathArgs), c_void_p]),
    "asw_resstack64_f16x3": (c_int, [POINTER(ResStackArgs), c_void_p]),
    "asw_convgemm_stats_tiles": (c_int, [c_int, c_int]),
    "asw_f16x3_overflow_count": (c_int, [c_int, POINTER(c_int32)]),
    "asw_gn_glu": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p,
                           c_void_p]),
    "asw_gn_finalize": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "asw_attention": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "asw_attention_prec": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "asw_overlap_add_unnorm": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                       c_void_p, c_void_p, c_void_p, c_void_p]),
    "asw_overlap_add_parts": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                      c_void_p, c_void_p, c_void_p, c_void_p]),
    "asw_energies": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "asw_pair_sisdr": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "asw_center_rows": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "asw_segment_sisdr": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "asw_add_layernorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p]),
    "asw_search_area": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_double, c_double,
                                POINTER(c_int), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                POINTER(c_void_p)]),
    "asw_free": (None, [c_void_p]),
    "asw_cube_select": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                c_void_p, c_int64, POINTER(c_int64)]),
    "asw_cube_select_planes": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                c_void_p, c_int64, POINTER(c_int64)]),
    "asw_srp_frames": (c_int, [c_int, c_int, c_int]),
    "asw_srp_cross_spectra": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                      c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "asw_srp_map": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                            c_void_p, c_void_p, c_void_p]),
}

_lib = None


def lib():
    """Load libasw_hip.so (fail loudly when it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the hot path.")
        # torch first: its bundled HIP runtime must be the one libasw_hip.so binds to, otherwise
        # two libamdhip64 instances end up in the process and device pointers cannot be shared.
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


_ops = None


def torch_ops():
    """``torch.ops.asw`` -- the PyTorch-ROCm custom ops the host classes call (csrc/torch_ops.cpp).
    Loads libasw_hip.so first, then the adapter library; raises when either has not been built."""
    global _ops
    if _ops is None:
        lib()
        if not os.path.exists(OPS_PATH):
            raise RuntimeError(f"{OPS_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`.  "
                               "The hot path is exposed only through these custom ops; there is no fallback.")
        import torch
        torch.ops.load_library(OPS_PATH)
        _ops = torch.ops.asw
    return _ops


def check(status: int):
    if status != 0:
        msg = lib().asw_last_error()
        raise RuntimeError(f"libasw_hip: status {status}: {msg.decode() if msg else ''}")


def ptr(t):
    """Device/host pointer of a torch tensor (or None)."""
    return None if t is None else c_void_p(t.data_ptr())


def current_stream():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def to_device(a, device):
    """Small host array -> device tensor through page-locked memory (no blocking copy call)."""
    import torch
    src = torch.from_numpy(a)
    host = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
    host.copy_(src)
    return host.to(device, non_blocking=True)


def to_host(t):
    """Device tensor -> numpy array backed by page-locked memory from torch's caching host allocator: the
    [N, T] results of the reference-compatible calls (136 MB for one fine stage) cross PCIe at DMA rate and
    the runtime has nothing to pin or un-pin per call."""
    import torch
    if not t.is_cuda:
        return t.numpy()
    host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    host.copy_(t)
    return host.numpy()
