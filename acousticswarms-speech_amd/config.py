"""Spot-network hyper-parameters and the derived per-level shape table.

Mirrors the constructor arguments of the reference spot ``Network``
(sep/training/SpeakerLocalization/network.py:268-292) and the values in
experiments/localization/description.json:5-13.  Only shapes live here; no
arithmetic of the path.
"""
from dataclasses import dataclass
from typing import List, Tuple


@dataclass(frozen=True)
class SpotConfig:
    n_mics: int = 7
    kernel_size: int = 7
    stride_list: Tuple[int, ...] = (2, 2, 4, 4, 4)
    channels: int = 64
    growth: int = 2
    encoder_channels: int = 2048
    encoder_kernel_size: int = 33
    encoder_stride: int = 16
    residual_layers: int = 3
    residual_dilation_factor: int = 7
    num_head: int = 8
    ffw_dim: int = 1024
    num_transformer_layers: int = 2

    # ---- derived -------------------------------------------------------
    @property
    def depth(self) -> int:
        return len(self.stride_list)

    @property
    def stride_product(self) -> int:
        p = 1
        for s in self.stride_list:
            p *= s
        return p

    def enc_channels(self) -> List[Tuple[int, int]]:
        """(in, out) channel pair of every encoder block (network.py:135-148)."""
        out, cin, ch = [], self.channels, self.channels
        for _ in range(self.depth):
            out.append((cin, ch))
            cin, ch = ch, int(self.growth * ch)
        return out

    def dec_channels(self) -> List[Tuple[int, int, int]]:
        """(in, out, stride) of every decoder block in *execution* order
        (network.py:221-231: blocks are inserted at the front)."""
        blocks, cin, ch = [], self.channels, self.channels
        for i in range(self.depth):
            blocks.insert(0, (ch, cin, self.stride_list[i]))
            cin, ch = ch, int(self.growth * ch)
        return blocks

    @property
    def bottleneck_channels(self) -> int:
        return self.enc_channels()[-1][1]

    def padded_length(self, t: int) -> int:
        """network.py:377: left-pad to a multiple of the stride product."""
        sp = self.stride_product
        return ((t - 1) // sp + 1) * sp

    def latent_frames(self, t_pad: int) -> int:
        k, s = self.encoder_kernel_size, self.encoder_stride
        return (t_pad + 2 * (k // 2) - k) // s + 1


FULL = SpotConfig()
# Reduced configuration used by the CPU-only oracle tests (fixture G2).
TINY = SpotConfig(channels=8, encoder_channels=64, ffw_dim=32)
# Smallest configuration the MFMA tiles accept (channel widths multiples of 64):
# two levels, bottleneck width 128 (head_dim 16).  Used by fast GPU parity tests.
SMALL = SpotConfig(stride_list=(2, 4), channels=64, encoder_channels=128, ffw_dim=128)


def spot_param_shapes(cfg: SpotConfig):
    """Ordered (name, shape) list of the reference spot ``Network`` state dict
    (SURVEY.md §8 a-N; network.py:305-349)."""
    K = cfg.kernel_size
    shapes = [("preproc.weight", (cfg.channels, cfg.n_mics, 1)),
              ("preproc.bias", (cfg.channels,))]

    def res(prefix, c):
        for j in range(cfg.residual_layers):
            shapes.extend([
                (f"{prefix}.res.seq.{j}.conv.weight", (c, c, K)),
                (f"{prefix}.res.seq.{j}.conv.bias", (c,)),
                (f"{prefix}.res.seq.{j}.norm.weight", (c,)),
                (f"{prefix}.res.seq.{j}.norm.bias", (c,)),
            ])

    for i, (cin, cout) in enumerate(cfg.enc_channels()):
        p = f"encoder.module_list.{i}"
        res(p, cin)
        shapes.extend([
            (f"{p}.conv1.weight", (2 * cout, cin, K)),
            (f"{p}.conv1.bias", (2 * cout,)),
            (f"{p}.norm1.weight", (2 * cout,)),
            (f"{p}.norm1.bias", (2 * cout,)),
            (f"{p}.embed1.weight", (cin, 2, 1)),
            (f"{p}.embed1.bias", (cin,)),
        ])
    for i, (cin, cout, s) in enumerate(cfg.dec_channels()):
        p = f"decoder.module_list.{i}"
        shapes.extend([
            (f"{p}.upsample.conv.weight", (cin, 2 * cout, s)),
            (f"{p}.upsample.conv.bias", (2 * cout,)),
            (f"{p}.norm1.weight", (2 * cout,)),
            (f"{p}.norm1.bias", (2 * cout,)),
        ])
        res(p, cout)
        shapes.extend([
            (f"{p}.embed1.weight", (2 * cout, 2, 1)),
            (f"{p}.embed1.bias", (2 * cout,)),
        ])
    E, EK = cfg.encoder_channels, cfg.encoder_kernel_size
    shapes.extend([
        ("reference_bypass.weight", (E, 1, EK)), ("reference_bypass.bias", (E,)),
        ("mask_encoder.weight", (E, cfg.channels, EK)), ("mask_encoder.bias", (E,)),
        ("output_decoder.weight", (E, 1, EK)), ("output_decoder.bias", (1,)),
    ])
    d, f = cfg.bottleneck_channels, cfg.ffw_dim
    for l in range(cfg.num_transformer_layers):
        p = f"bottleneck.transf.layers.{l}"
        shapes.extend([
            (f"{p}.self_attn.in_proj_weight", (3 * d, d)),
            (f"{p}.self_attn.in_proj_bias", (3 * d,)),
            (f"{p}.self_attn.out_proj.weight", (d, d)),
            (f"{p}.self_attn.out_proj.bias", (d,)),
            (f"{p}.linear1.weight", (f, d)), (f"{p}.linear1.bias", (f,)),
            (f"{p}.linear2.weight", (d, f)), (f"{p}.linear2.bias", (d,)),
            (f"{p}.norm1.weight", (d,)), (f"{p}.norm1.bias", (d,)),
            (f"{p}.norm2.weight", (d,)), (f"{p}.norm2.bias", (d,)),
        ])
    return shapes
