"""Host-side mirror of the joint separation network's call surface over libasw_hip.so.

``SepModel`` keeps the names, argument meaning and return types of the reference ``Network``
(sep/training/SpeakerSeparation/network.py:323-548): ``infer(input_channels, patch_list)``,
``infer_sample(input_channels, sample_list)`` -> ndarray [S, T], and ``forward(mix,
num_speakers)`` on already normalised input.  It is what ``JointModel.sep_model`` holds
(sep/training/JointModel/network.py:122,201-209).  All arithmetic runs in the HIP library,
reached through ``torch.ops.asw.sep_infer`` / ``sep_forward``; PyTorch is used for device
memory and streams only.  There is no CPU path.
"""
from collections import OrderedDict
from ctypes import byref, c_size_t, c_void_p

import numpy as np

from . import native
from .config import SEP_FULL, SepConfig, sep_param_shapes

MAX_SEQUENCES = 64       # S (or B*S) per library call


def rounded_offsets(sample_list, n_pairs: int) -> np.ndarray:
    """np.round on the float64 offsets, as :507 does (half to even), -> int32 [S, M-1]."""
    if len(sample_list) == 0:
        return np.zeros((0, n_pairs), dtype=np.int32)
    offs = np.stack([np.asarray(s, dtype=np.float64) for s in sample_list])
    if offs.ndim != 2 or offs.shape[1] != n_pairs:
        raise RuntimeError(f"speaker has {offs.shape[-1]} offsets, mixture has {n_pairs + 1} channels")
    return np.round(offs).astype(np.int32)


class SepModel:
    PRECISIONS = {"f32": 0, "f16x3": 1, "f16": 2}

    def __init__(self, cfg: SepConfig = SEP_FULL, state_dict=None, precision: str = "f32"):
        if precision not in self.PRECISIONS:
            raise RuntimeError(f"precision must be one of {list(self.PRECISIONS)}")
        self.cfg = cfg
        self.n_mics = cfg.n_mics
        self.max_n_speaker = cfg.max_speakers
        self.precision = precision
        self.device = None
        self._h = None
        self._sd = None
        if state_dict is not None:
            self.load_state_dict(state_dict)

    # ---- weights -------------------------------------------------------------
    def load_state_dict(self, sd, strict: bool = True):
        """Reference-format state dict (numpy arrays or torch tensors), strict by default."""
        want = OrderedDict(sep_param_shapes(self.cfg))
        clean = OrderedDict()
        for k, v in sd.items():
            a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
            clean[k] = np.ascontiguousarray(a, dtype=np.float32)
        missing = [k for k in want if k not in clean]
        extra = [k for k in clean if k not in want]
        if strict and (missing or extra):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:3]} unexpected {extra[:3]}")
        for k, shp in want.items():
            if k in clean and tuple(clean[k].shape) != tuple(shp):
                raise RuntimeError(f"size mismatch for {k}: {tuple(clean[k].shape)} vs {tuple(shp)}")
        self._sd = clean
        if self._h is not None:
            import torch
            with torch.cuda.device(self.device):
                self._upload()
        return self

    def _upload(self):
        L = native.lib()
        for k, a in self._sd.items():
            native.check(L.asw_sep_set_param(self._h, k.encode(), c_void_p(a.ctypes.data), a.size))
        native.check(L.asw_sep_finalize(self._h))

    def to(self, device=None):
        import torch
        if device is None:
            return self
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("SepModel runs only on an MI355X (device 'cuda'); there is no CPU path")
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: the separation network has no CPU fallback")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        L = native.lib()
        with torch.cuda.device(device):
            if self._h is not None:
                L.asw_sep_destroy(self._h)
                self._h = None
            h = c_void_p()
            cc = native.SepConfigC.from_config(self.cfg)
            native.check(L.asw_sep_create(byref(cc), byref(h)))
            self._h = h
            native.check(L.asw_sep_set_precision(self._h, self.PRECISIONS[self.precision]))
            self.device = device
            if self._sd is not None:
                self._upload()
        return self

    def eval(self):
        return self

    def set_precision(self, precision: str):
        if precision not in self.PRECISIONS:
            raise RuntimeError(f"precision must be one of {list(self.PRECISIONS)}")
        self.precision = precision
        if self._h is not None:
            native.check(native.lib().asw_sep_set_precision(self._h, self.PRECISIONS[precision]))

    def __del__(self):
        try:
            if self._h is not None:
                native.lib().asw_sep_destroy(self._h)
        except Exception:
            pass

    def _need(self):
        if self._h is None:
            raise RuntimeError("SepModel.to('cuda') must be called before inference")
        if self._sd is None:
            raise RuntimeError("SepModel has no weights: call load_state_dict()")

    # ---- device-level entry (tensors stay on the GPU) --------------------------
    def infer_device(self, mix_dev, offsets_dev):
        """mix_dev [M,T] float32 cuda, offsets_dev [S,M-1] int32 cuda (rounded) -> [S,T] float32 cuda."""
        import torch
        self._need()
        M, T = mix_dev.shape
        S = offsets_dev.shape[0]
        if S > MAX_SEQUENCES:
            raise RuntimeError(f"{S} speakers in one call; the library takes at most {MAX_SEQUENCES}")
        assert mix_dev.dtype == torch.float32 and mix_dev.is_contiguous() and mix_dev.is_cuda
        assert offsets_dev.dtype == torch.int32 and offsets_dev.is_contiguous() and offsets_dev.is_cuda
        return native.torch_ops().sep_infer(self._h.value, mix_dev, offsets_dev)

    # ---- reference call surface ---------------------------------------------------
    def infer(self, input_channels, patch_list) -> np.ndarray:
        """:492-494."""
        return self.infer_sample(input_channels, [p.sample_offset for p in patch_list])

    def infer_sample(self, input_channels, sample_list) -> np.ndarray:
        """:496-548: input_channels (M x T), sample_list (S x (M-1)) -> ndarray [S, T] float32."""
        import torch
        self._need()
        mix = torch.as_tensor(input_channels)
        offs = rounded_offsets(sample_list, mix.shape[0] - 1)
        if offs.shape[0] == 0:
            return np.empty((0, mix.shape[-1]), dtype=np.float32)
        mix_d = mix.to(self.device, dtype=torch.float32).contiguous()
        off_d = native.to_device(offs, self.device)
        return native.to_host(self.infer_device(mix_d, off_d))

    def forward(self, mix, num_speakers):
        """Network.forward (:418-490): mix [B, S*M, t] already normalised, num_speakers [B,1]
        -> device tensor [B, max(max(S), max_speakers), t].  Items may hold different numbers of speakers: like the
        reference (speakers_to_batches / batches_to_speakers, :236-268) only the first num_speakers[b] blocks of M
        channels of item b are read, a missing speaker takes part in the inter-speaker attention as a zero sequence,
        its output row is the bare output_decoder bias, and the stack is as wide as the largest count."""
        import torch
        self._need()
        ns = np.asarray(torch.as_tensor(num_speakers).cpu()).reshape(-1).astype(np.int64)
        mix = torch.as_tensor(mix).to(self.device, dtype=torch.float32).contiguous()
        B, SM, t = mix.shape
        if ns.shape[0] != B or B < 1:
            raise RuntimeError(f"num_speakers holds {ns.shape[0]} entries for {B} batch items")
        S = int(ns.max())
        if ns.min() < 1 or SM % self.n_mics != 0 or SM < S * self.n_mics:
            raise RuntimeError(f"mix has {SM} channels, expected at least max(num_speakers)*n_mics = {S * self.n_mics}")
        if B * S > MAX_SEQUENCES:
            raise RuntimeError(f"{B * S} sequences in one call; the library takes at most {MAX_SEQUENCES}")
        if np.all(ns == S) and SM == S * self.n_mics:
            return native.torch_ops().sep_forward(self._h.value, mix, S, self.n_mics, self.max_n_speaker)
        if SM != S * self.n_mics:
            mix = mix[:, :S * self.n_mics].contiguous()       # blocks beyond the largest count are never read (:244)
        return native.torch_ops().sep_forward_counts(self._h.value, mix, [int(v) for v in ns], self.n_mics, self.max_n_speaker)

    __call__ = forward

    def get_tap(self, name: str, shape=None):
        """Intermediate activation of the last call (channels-last), for parity tests."""
        import torch
        n = c_size_t()
        L = native.lib()
        native.check(L.asw_sep_get_tap(self._h, name.encode(), None, 0, byref(n), None))
        buf = torch.empty((n.value,), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            native.check(L.asw_sep_get_tap(self._h, name.encode(), native.ptr(buf), n.value, byref(n),
                                           native.current_stream()))
        return buf if shape is None else buf.view(*shape)
