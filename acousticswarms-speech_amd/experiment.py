"""Experiment-directory loading (sep/helpers/utils.py:165-215, ``load_model_from_exp``):
``<exp>/description.json`` names the network (``model_name``: SpeakerLocalization ->
``SpotModel``, SpeakerSeparation -> ``SepModel``) and its ``model_params``; checkpoints are
``<exp>/checkpoints/<exp>_<epoch>.pt`` (or ``<exp>/<experiment_name>/...``) with the training
state in ``state.pt``.

Difference to the reference, on purpose: nothing is unpickled.  Checkpoints are read with
``torch.load(..., weights_only=True)``; ``state.pt`` holds a pickled scheduler object beside
``val_losses`` (sep/training/train.py:218-226), so when the safe loader refuses it the 'best'
mode falls back to 'last' with the reference's own warning instead of executing the file.
"""
import glob
import json
import os

import numpy as np

from .config import SepConfig, SpotConfig

_SPOT_KEYS = ("n_mics", "kernel_size", "stride_list", "channels", "growth", "encoder_channels", "encoder_kernel_size",
              "encoder_stride", "residual_layers", "residual_dilation_factor", "num_head", "ffw_dim",
              "num_transformer_layers")
_SEP_KEYS = ("n_mics", "max_speakers", "kernel_size", "stride_list", "channels", "growth", "encoder_channels",
             "encoder_kernel_size", "encoder_stride", "residual_layers", "residual_dilation_factor", "num_head", "ffw_dim",
             "bottleneck_layers", "bottleneck_ksize")
# constructor defaults of the reference networks that differ from the dataclass defaults
_SEP_CTOR_DEFAULTS = {"max_speakers": 6}


def config_from_description(desc: dict):
    """-> ("spot" | "sep", config) from a description.json dictionary."""
    name, params = desc["model_name"], dict(desc.get("model_params", {}))
    if name == "SpeakerLocalization":
        kind, cls, keys, base = "spot", SpotConfig, _SPOT_KEYS, {}
    elif name == "SpeakerSeparation":
        kind, cls, keys, base = "sep", SepConfig, _SEP_KEYS, dict(_SEP_CTOR_DEFAULTS)
    else:
        raise RuntimeError(f"model_name {name!r}: only SpeakerLocalization and SpeakerSeparation are on the path")
    unknown = [k for k in params if k not in keys and k not in ("device", "rescale")]
    if unknown:
        raise RuntimeError(f"description.json: unknown model_params {unknown}")
    base.update({k: v for k, v in params.items() if k in keys})
    if "stride_list" in base:
        base["stride_list"] = tuple(int(v) for v in base["stride_list"])
    if "growth" in base:
        # the reference widens by int(growth * channels); the device networks take whole-number factors only
        if float(base["growth"]) != int(base["growth"]):
            raise RuntimeError(f"description.json: growth={base['growth']} -- only whole-number channel growth is supported")
        base["growth"] = int(base["growth"])
    return kind, cls(**base)


def _safe_load(path):
    import torch
    return torch.load(path, map_location="cpu", weights_only=True)


def load_model_from_exp(exp_dir: str, mode: str = "best", precision: str = "f32", best_epoch=None,
                        fallback_to_last: bool = False, **model_kwargs):
    """-> SpotModel / SepModel with the experiment's weights (not yet moved to a device).
    mode: 'best' (argmin of state.pt's val_losses), 'last' (highest epoch), 'new' (no weights).

    The reference's state.pt also holds a pickled scheduler object (train.py:219-226), which the safe loader
    (weights_only=True) refuses -- and nothing from a checkpoint file is ever unpickled here.  For such an
    experiment 'best' needs the epoch from the caller (`best_epoch`, e.g. read off the training log) or an
    explicit `fallback_to_last=True`; silently evaluating another checkpoint than the reference would is not
    an option.  A missing state.pt falls back to 'last' with a warning, as the reference does (utils.py:186-190)."""
    with open(os.path.join(exp_dir, "description.json"), "rb") as f:
        desc = json.load(f)
    if "experiment_name" in desc:
        exp_name = desc["experiment_name"]
        ckpt_dir = exp_name
    else:
        exp_name = os.path.basename(exp_dir.strip("/"))
        ckpt_dir = "checkpoints"
    kind, cfg = config_from_description(desc)
    if kind == "spot":
        from .spot import SpotModel
        model = SpotModel(cfg, None, precision=precision, **model_kwargs)
    else:
        from .sep import SepModel
        model = SepModel(cfg, None, precision=precision, **model_kwargs)
    if mode == "best":
        import pickle
        state_path = os.path.join(exp_dir, ckpt_dir, "state.pt")
        best = None if best_epoch is None else int(best_epoch)
        if best is not None:
            pass
        elif not os.path.exists(state_path):
            print("[WARNING] Could not find experiment state dict, using load mode 'last' instead")
        else:
            try:
                state = _safe_load(state_path)
            except pickle.UnpicklingError as e:  # pickled objects inside: refused, never executed
                if not fallback_to_last:
                    raise RuntimeError(
                        f"{state_path} cannot be read without unpickling ({e}); pass best_epoch=<epoch> or "
                        "fallback_to_last=True to choose the checkpoint explicitly") from e
                print("[WARNING] state.pt is not loadable without unpickling; using load mode 'last' as requested")
            else:
                best = int(np.argmin(np.asarray(state["val_losses"], dtype=np.float64)))
        if best is None:
            mode = "last"
        else:
            model.load_state_dict(_safe_load(os.path.join(exp_dir, ckpt_dir, f"{exp_name}_{best}.pt")), strict=True)
            print("Loaded best checkpoint", best)
    if mode == "last":
        ckpts = glob.glob(os.path.join(exp_dir, ckpt_dir, f"{exp_name}_*.pt"))
        ckpts = sorted(ckpts, key=lambda c: -int(c[c.rfind("_") + 1:-len(".pt")]))
        if ckpts:
            model.load_state_dict(_safe_load(ckpts[0]), strict=True)
            print("Loaded last checkpoint", ckpts[0])
        else:
            print("[WARNING] Provided experiment has no pretrained checkpoint, using default parameters instead")
    elif mode not in ("best", "new"):
        raise RuntimeError("mode must be 'best', 'last' or 'new'")
    return model
