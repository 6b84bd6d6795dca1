"""Deterministic spot-network weights + reference-format state-dict I/O.

Pretrained checkpoints are not available offline, so parity and
benchmarks run on seeded weights.  The generator is counter-based (one Philox
stream per tensor, keyed by seed and tensor index) so any subset can be
reproduced on any box without storing the 47 M parameters.  Scales are chosen
so activations stay O(1) through the ~40 normalised layers.

State-dict key layout: reference spot ``Network``
(sep/training/SpeakerLocalization/network.py:305-349; SURVEY.md §8 a-N).
"""
import math
from collections import OrderedDict

import numpy as np

from .config import SepConfig, SpotConfig, sep_param_shapes, spot_param_shapes


def _tensor(seed: int, index: int, shape, kind: str) -> np.ndarray:
    rng = np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFF, index]))
    n = rng.standard_normal(size=shape, dtype=np.float32)
    if kind == "norm_w":
        return (1.0 + 0.1 * n).astype(np.float32)
    if kind == "bias":
        return (0.05 * n).astype(np.float32)
    if kind == "embed_w":          # window gate: keep gates near 1, distinct per window
        return (0.5 + 0.25 * n).astype(np.float32)
    if kind == "embed_b":
        return (0.5 + 0.1 * n).astype(np.float32)
    # conv / linear weight: fan-in scaling
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
    if kind == "convT":            # ConvTranspose1d weight is [in, out, k]
        fan_in = shape[0]
    return (n * (1.0 / math.sqrt(max(fan_in, 1)))).astype(np.float32)


def _kind(name: str) -> str:
    if ".embed1.weight" in name:
        return "embed_w"
    if ".embed1.bias" in name:
        return "embed_b"
    if ".norm" in name and name.endswith("weight"):
        return "norm_w"
    if name.endswith("bias"):
        return "bias"
    if "upsample.conv.weight" in name or name == "output_decoder.weight":
        return "convT"
    return "weight"


def make_spot_state_dict(cfg: SpotConfig, seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Seeded float32 numpy state dict with the reference's key names."""
    sd = OrderedDict()
    for i, (name, shape) in enumerate(spot_param_shapes(cfg)):
        sd[name] = _tensor(seed, i, shape, _kind(name))
    return sd


def make_sep_state_dict(cfg: SepConfig, seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Seeded float32 state dict of the joint separation network (reference key names for the
    U-Net / mask path, published speechbrain names for the Conformer, config.sep_param_shapes)."""
    import math as _m
    sd = OrderedDict()
    for i, (name, shape) in enumerate(sep_param_shapes(cfg)):
        if name.endswith("pe_single.inv_freq"):
            d = 2 * shape[0]              # RelPosEncXL: exp(arange(0, d, 2) * -(ln 10000 / d)), float32
            sd[name] = np.exp(np.arange(0, d, 2, dtype=np.float32) * np.float32(-(_m.log(10000.0) / d))).astype(np.float32)
            continue
        kind = _kind(name)
        if "pos_bias_" in name:
            kind = "bias"
        elif name.endswith("in_proj_weight") or name.endswith("linear_pos.weight"):
            kind = "weight"
        elif name.endswith("weight") and len(shape) == 1:      # every LayerNorm / GroupNorm scale
            kind = "norm_w"
        sd[name] = _tensor(seed + 7919, i, shape, kind)
    return sd


def count_params(cfg: SpotConfig) -> int:
    return sum(int(np.prod(s)) for _, s in spot_param_shapes(cfg))


def load_reference_checkpoint(path: str, cfg: SpotConfig):
    """Load a reference-format ``<exp>_<epoch>.pt`` state dict without executing
    anything from the file (``weights_only=True``; sep/helpers/utils.py:196-198 is
    the loader this replaces).  Returns a float32 numpy state dict and checks
    every key/shape against ``spot_param_shapes``."""
    import torch
    raw = torch.load(path, map_location="cpu", weights_only=True)
    want = dict(spot_param_shapes(cfg))
    missing = [k for k in want if k not in raw]
    extra = [k for k in raw if k not in want]
    if missing or extra:
        raise RuntimeError(f"state dict mismatch: missing={missing[:4]} extra={extra[:4]}")
    out = OrderedDict()
    for k, shp in want.items():
        t = raw[k]
        if tuple(t.shape) != tuple(shp):
            raise RuntimeError(f"{k}: shape {tuple(t.shape)} != {tuple(shp)}")
        out[k] = t.detach().to(torch.float32).numpy()
    return out
