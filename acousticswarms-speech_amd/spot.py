"""Host-side mirror of the reference's spot-model call surface over libasw_hip.so.

``SpotModel`` keeps the names, argument meaning and return types of
``DataParallelSpotModel.shift_and_sep`` (sep/training/JointModel/network.py:27-104) and
``Network.forward`` (sep/training/SpeakerLocalization/network.py:363-405); all
arithmetic runs in the HIP library, reached through the PyTorch-ROCm custom ops
``torch.ops.asw.*`` (csrc/torch_ops.cpp, thin adapters over the C ABI).  PyTorch is used only
for device memory and streams.  ``shift_and_score`` is the additive energies-only fast path:
waveforms stay on the GPU, two doubles per candidate come back.
"""
from collections import OrderedDict
from ctypes import byref, c_size_t, c_void_p

import numpy as np

from . import native
from .config import FULL, SpotConfig, spot_param_shapes


def offsets_from_patches(patch_list, n_pairs: int) -> np.ndarray:
    """round(sample_offset) per candidate as int32 [N, M-1]
    (JointModel/network.py:81-82: the offsets pass through a float32 torch.Tensor before
    torch.round, which is round-half-even == np.rint on the float32 value)."""
    if len(patch_list) == 0:
        return np.zeros((0, n_pairs), dtype=np.int32)
    offs = np.stack([np.asarray(getattr(p, "sample_offset", p), dtype=np.float64) for p in patch_list])
    if offs.shape[1] != n_pairs:
        raise RuntimeError(f"candidate has {offs.shape[1]} offsets, mixture has {n_pairs + 1} channels")
    return np.rint(offs.astype(np.float32)).astype(np.int32)


class SpotModel:
    PRECISIONS = {"f32": 0, "f16x3": 1, "f16": 2}

    def __init__(self, cfg: SpotConfig = FULL, state_dict=None, batch_size: int = 32, precision: str = "f32",
                 lanes: int = 1):
        """precision: "f32" = exact fp32 MFMA; "f16x3" = split-operand half MFMA with fp32
        accumulation (~21-bit operands, 5.3x the f32 matrix rate), see csrc/convgemm.hip."""
        if precision not in self.PRECISIONS:
            raise RuntimeError(f"precision must be one of {list(self.PRECISIONS)}")
        self.cfg = cfg
        self.precision = precision
        self.batch_size = batch_size
        self.lanes = int(lanes)             # 2: consecutive internal batches on two HIP streams (asw_spot_set_lanes)
        self.device = None
        self._h = None
        self._sd = None
        self.last_waveforms = None          # device tensor [N,T] of the latest shift_and_score
        if state_dict is not None:
            self.load_state_dict(state_dict)

    # ---- weights -------------------------------------------------------------
    def load_state_dict(self, sd, strict: bool = True):
        """Reference-format state dict (numpy arrays or torch tensors)."""
        import contextlib
        want = OrderedDict(spot_param_shapes(self.cfg))
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
            with (torch.cuda.device(self.device) if self.device is not None else contextlib.nullcontext()):
                self._upload()
        return self

    def _upload(self):
        L = native.lib()
        for k, a in self._sd.items():
            native.check(L.asw_spot_set_param(self._h, k.encode(), c_void_p(a.ctypes.data), a.size))
        native.check(L.asw_spot_finalize(self._h))

    def to(self, device=None):
        """Create the device-resident model (weights uploaded once; C1 of SURVEY.md §2.2)."""
        import torch
        if device is None:
            return self
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("SpotModel runs only on an MI355X (device 'cuda'); there is no CPU path")
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: the spot hot path has no CPU fallback")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        L = native.lib()
        with torch.cuda.device(device):            # the handle lives on this device; the process default is untouched
            if self._h is not None:
                L.asw_spot_destroy(self._h)
                self._h = None
            h = c_void_p()
            cc = native.SpotConfigC.from_config(self.cfg)
            native.check(L.asw_spot_create(byref(cc), byref(h)))
            self._h = h
            native.check(L.asw_spot_set_batch(self._h, int(self.batch_size)))
            native.check(L.asw_spot_set_precision(self._h, self.PRECISIONS[self.precision]))
            native.check(L.asw_spot_set_lanes(self._h, self.lanes))
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
            native.check(native.lib().asw_spot_set_precision(self._h, self.PRECISIONS[precision]))

    def set_lanes(self, lanes: int):
        self.lanes = int(lanes)
        if self._h is not None:
            native.check(native.lib().asw_spot_set_lanes(self._h, self.lanes))

    def set_batch_size(self, b: int):
        self.batch_size = int(b)
        if self._h is not None:
            native.check(native.lib().asw_spot_set_batch(self._h, int(b)))

    def __del__(self):
        try:
            if self._h is not None:
                native.lib().asw_spot_destroy(self._h)
        except Exception:
            pass

    def _need(self):
        if self._h is None:
            raise RuntimeError("SpotModel.to('cuda') must be called before inference")
        if self._sd is None:
            raise RuntimeError("SpotModel has no weights: call load_state_dict()")

    # ---- device-level entry (tensors stay on the GPU) --------------------------
    def shift_and_sep_device(self, mix_dev, offsets_dev, strict: int = 0, want_wave: bool = True,
                             want_energy: bool = False, window: int = 12000, circular: bool = True):
        """mix_dev [M,T] float32 cuda, offsets_dev [N,M-1] int32 cuda ->
        (wave [N,T] float32 cuda | None, energy [N,2] float64 cuda | None)."""
        import torch
        self._need()
        M, T = mix_dev.shape
        N = offsets_dev.shape[0]
        assert mix_dev.dtype == torch.float32 and mix_dev.is_contiguous() and mix_dev.is_cuda
        assert offsets_dev.dtype == torch.int32 and offsets_dev.is_contiguous() and offsets_dev.is_cuda
        # torch.ops.asw.spot_shift_and_sep: runs on the tensors' device and torch's current stream
        wave, en = native.torch_ops().spot_shift_and_sep(self._h.value, mix_dev, offsets_dev, int(strict), bool(circular),
                                                         bool(want_wave), bool(want_energy), int(window))
        return (wave if want_wave else None), (en if want_energy else None)

    def shift_and_sep_device_multi(self, mix_stack, offsets_dev, mix_index_dev, strict: int = 0, want_wave: bool = True,
                                   want_energy: bool = False, window: int = 12000, circular: bool = True):
        """The same call over candidates of several mixtures (torch.ops.asw.spot_shift_and_sep_multi):
        mix_stack [K,M,T] float32 cuda, offsets_dev [N,M-1] int32, mix_index_dev [N] int32 with values in [0,K)
        (the caller builds it on the host and guarantees the range)."""
        import torch
        self._need()
        assert mix_stack.dim() == 3 and mix_stack.dtype == torch.float32 and mix_stack.is_contiguous() and mix_stack.is_cuda
        assert offsets_dev.dtype == torch.int32 and offsets_dev.is_contiguous() and offsets_dev.is_cuda
        assert mix_index_dev.dtype == torch.int32 and mix_index_dev.is_contiguous() and mix_index_dev.is_cuda
        wave, en = native.torch_ops().spot_shift_and_sep_multi(self._h.value, mix_stack, offsets_dev, mix_index_dev,
                                                               int(strict), bool(circular), bool(want_wave),
                                                               bool(want_energy), int(window))
        return (wave if want_wave else None), (en if want_energy else None)

    # ---- reference call surface ---------------------------------------------------
    def shift_and_sep(self, input_channels, patch_list, Strict: int = 0, save_input: bool = False) -> np.ndarray:
        """Drop-in for DataParallelSpotModel.shift_and_sep: returns ndarray [N,T] float32."""
        import torch
        self._need()
        if save_input:
            raise RuntimeError("save_input=True is a debugging path of the reference that materialises every "
                               "shifted mixture; it is not provided (the shifted tensor never exists here)")
        mix = torch.as_tensor(input_channels)
        T = mix.shape[-1]
        if len(patch_list) == 0:
            return np.empty((0, T), dtype=np.float32)
        offs = offsets_from_patches(patch_list, mix.shape[0] - 1)
        mix_d = mix.to(self.device, dtype=torch.float32).contiguous()
        off_d = torch.from_numpy(offs).to(self.device)
        wave, _ = self.shift_and_sep_device(mix_d, off_d, Strict, want_wave=True)
        return native.to_host(wave)

    def shift_and_score(self, input_channels, patch_list, Strict: int = 0, window: int = 12000,
                        keep_waveforms: bool = True) -> np.ndarray:
        """Energies-only fast path: ndarray [N,2] float64 = (power, power2) of every
        mean-removed candidate output (local_utils_3d.py:349-354 / Mic_Array.py:290-295)."""
        import torch
        self._need()
        mix = torch.as_tensor(input_channels)
        if len(patch_list) == 0:
            self.last_waveforms = None
            return np.empty((0, 2), dtype=np.float64)
        offs = offsets_from_patches(patch_list, mix.shape[0] - 1)
        mix_d = mix.to(self.device, dtype=torch.float32).contiguous()
        off_d = torch.from_numpy(offs).to(self.device)
        wave, en = self.shift_and_sep_device(mix_d, off_d, Strict, want_wave=keep_waveforms, want_energy=True,
                                             window=window)
        self.last_waveforms = wave
        return en.cpu().numpy()

    def shift_and_sep_resident(self, input_channels, patch_list, Strict: int = 0, window: int = 12000,
                               device_energies: bool = False):
        """Device-resident variant for the fine stage: returns (waves, energies) where ``waves``
        is a CUDA tensor [N,T] of the MEAN-REMOVED candidate outputs (sep/Mic_Array.py:291) that
        stays on the GPU, and ``energies`` the host ndarray [N,2] = (power, power2).  Only the
        energies (and later the few cluster heads) cross PCIe.  With ``device_energies`` the
        energies stay on the GPU too and nothing in the call waits for the device (the offsets go
        up through pinned memory), so the caller can overlap host work with it."""
        import torch
        self._need()
        mix = torch.as_tensor(input_channels)
        offs = offsets_from_patches(patch_list, mix.shape[0] - 1)
        mix_d = mix.to(self.device, dtype=torch.float32).contiguous()
        if device_energies:
            off_d = torch.from_numpy(offs).pin_memory().to(self.device, non_blocking=True)
        else:
            off_d = torch.from_numpy(offs).to(self.device)
        wave, en = self.shift_and_sep_device(mix_d, off_d, Strict, want_wave=True, want_energy=True, window=window)
        native.torch_ops().center_rows_(wave)
        return wave, (en if device_energies else en.cpu().numpy())

    def pair_sisdr(self, waves):
        """SI-SDR matrix S[i][j] = si_sdr(est=waves[i], ref=waves[j]) computed on the GPU
        (sep/helpers/eval_utils.py:11-39); returns a host ndarray [n,n] float64."""
        return native.torch_ops().pair_sisdr(waves.contiguous()).cpu().numpy()

    def segment_sisdr(self, waves, segments):
        """Segment-wise SI-SDR tensor S[i][j][k] = si_sdr(waves[i][seg_k(i)], waves[j][seg_k(i)])
        on the GPU (split_wise_sisdr, sep/helpers/eval_utils.py:73-82): ``segments[i]`` is the
        list of [start, end) of waveform i (split_wav).  Returns (host ndarray [n,n,kmax]
        float64, counts [n]); entries beyond a waveform's segment count are NaN."""
        import torch
        n, T = waves.shape
        cnt = np.array([len(s) for s in segments], dtype=np.int32)
        kmax = max(1, int(cnt.max()) if n else 1)
        seg = np.zeros((n, kmax, 2), dtype=np.int32)
        for i, sl in enumerate(segments):
            for k, (a, b) in enumerate(sl):
                if not (0 <= a <= b <= T):
                    raise RuntimeError(f"segment [{a},{b}) of waveform {i} lies outside [0,{T}]")
                seg[i, k] = (a, b)
        out = native.torch_ops().segment_sisdr(waves.contiguous(), torch.from_numpy(seg).to(waves.device),
                                               torch.from_numpy(cnt).to(waves.device))
        return out.cpu().numpy(), cnt

    def forward(self, mix, window_embedding):
        """Network.forward: mix [B,M,t] (already normalised), window_embedding [B,2] -> [B,1,t]
        (device tensor).  Rows are grouped by identical embedding because the window gate
        is folded into the convolution weights."""
        import torch
        self._need()
        mix = torch.as_tensor(mix).to(self.device, dtype=torch.float32).contiguous()
        wemb = torch.as_tensor(window_embedding, dtype=torch.float32).cpu().numpy().reshape(mix.shape[0], 2)
        B, M, t = mix.shape
        out = torch.empty((B, 1, t), dtype=torch.float32, device=self.device)
        keys = {}
        for i, row in enumerate(map(tuple, wemb)):
            keys.setdefault(row, []).append(i)
        ops = native.torch_ops()
        for row, idx in keys.items():
            whole = len(idx) == B
            y = ops.spot_forward(self._h.value, mix if whole else mix[idx].contiguous(), float(row[0]), float(row[1]))
            if whole:
                return y.view(B, 1, t)
            out.view(B, t)[idx] = y
        return out

    __call__ = forward

    def set_fused_mask(self, on: bool = True):
        """f16x3 mode: run the mask path (bypass, mask encoder, product, decoder taps) as one launch
        (default) or as three GEMMs (the "latent" tap exists only then)."""
        self._need()
        native.check(native.lib().asw_spot_set_fused_mask(self._h, int(bool(on))))
        return self

    def get_tap(self, name: str, shape=None):
        """Intermediate activation of the last forward (channels-last), for parity tests."""
        import torch
        n = c_size_t()
        L = native.lib()
        native.check(L.asw_spot_get_tap(self._h, name.encode(), None, 0, byref(n), None))
        buf = torch.empty((n.value,), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            native.check(L.asw_spot_get_tap(self._h, name.encode(), native.ptr(buf), n.value, byref(n),
                                            native.current_stream()))
        return buf if shape is None else buf.view(*shape)
