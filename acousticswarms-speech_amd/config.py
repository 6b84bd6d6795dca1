"""Hyper-parameters of the two networks on the path and their derived shape tables.

``SpotConfig`` mirrors the constructor arguments of the reference spot ``Network``
(sep/training/SpeakerLocalization/network.py:268-292, experiments/localization/
description.json:5-13); ``SepConfig`` those of the joint separation ``Network``
(sep/training/SpeakerSeparation/network.py:324-341, experiments/separation/
description.json:4-10).  Only shapes live here; no arithmetic of the path.
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


# ------------------------------------------------------------------------------------------
# joint separation network (sep/training/SpeakerSeparation/network.py:323-416)
# ------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class SepConfig:
    n_mics: int = 7
    max_speakers: int = 5            # experiments/separation/description.json:6 (constructor default is 6)
    kernel_size: int = 5
    stride_list: Tuple[int, ...] = (2, 2, 4, 4)
    channels: int = 64
    growth: int = 2
    encoder_channels: int = 4096
    encoder_kernel_size: int = 33
    encoder_stride: int = 16
    residual_layers: int = 3
    residual_dilation_factor: int = 2
    num_head: int = 8
    ffw_dim: int = 1024
    bottleneck_layers: int = 3
    bottleneck_ksize: int = 31

    depth = SpotConfig.depth
    stride_product = SpotConfig.stride_product
    enc_channels = SpotConfig.enc_channels
    dec_channels = SpotConfig.dec_channels
    bottleneck_channels = SpotConfig.bottleneck_channels
    padded_length = SpotConfig.padded_length
    latent_frames = SpotConfig.latent_frames


SEP_FULL = SepConfig()
# Smallest configuration the MFMA tiles accept: two levels, bottleneck width 128 (8 heads of 16).
SEP_SMALL = SepConfig(stride_list=(2, 4), channels=64, encoder_channels=256, ffw_dim=128, bottleneck_layers=2,
                      bottleneck_ksize=7)


def sep_param_shapes(cfg: SepConfig):
    """Ordered (name, shape) list of the joint separation ``Network`` state dict.  The U-Net and
    mask-path names are the reference's own (SpeakerSeparation/network.py:343-404).  The
    bottleneck (:270-321) is built from speechbrain modules that are absent here; their
    parameter names below follow the published speechbrain definitions (ConformerEncoder /
    ConformerEncoderLayer / ConvolutionModule / RelPosMHAXL / PositionalwiseFeedForward and the
    speechbrain LayerNorm wrapper, whose inner module is ``.norm``) and torch's
    nn.TransformerEncoderLayer -- "parity unpinned" for the speechbrain names, SURVEY.md §8c."""
    K = cfg.kernel_size
    shapes = [("preproc.weight", (cfg.channels, cfg.n_mics, 1)), ("preproc.bias", (cfg.channels,))]

    def res(prefix, c):
        for j in range(cfg.residual_layers):
            shapes.extend([
                (f"{prefix}.res.seq.{j}.conv.weight", (c, c, K)), (f"{prefix}.res.seq.{j}.conv.bias", (c,)),
                (f"{prefix}.res.seq.{j}.norm.weight", (c,)), (f"{prefix}.res.seq.{j}.norm.bias", (c,)),
            ])

    for i, (cin, cout) in enumerate(cfg.enc_channels()):
        p = f"encoder.module_list.{i}"
        res(p, cin)
        shapes.extend([(f"{p}.conv1.weight", (2 * cout, cin, K)), (f"{p}.conv1.bias", (2 * cout,)),
                       (f"{p}.norm1.weight", (2 * cout,)), (f"{p}.norm1.bias", (2 * cout,))])
    d, f, H, BK = cfg.bottleneck_channels, cfg.ffw_dim, cfg.num_head, cfg.bottleneck_ksize
    shapes.append(("bottleneck.pe_single.inv_freq", (d // 2,)))          # registered buffer of RelPosEncXL
    for l in range(cfg.bottleneck_layers):
        c = f"bottleneck.module_list.{l}.intra.layers.0"
        shapes.extend([
            (f"{c}.mha_layer.in_proj_weight", (3 * d, d)),
            (f"{c}.mha_layer.pos_bias_u", (d // H, H)), (f"{c}.mha_layer.pos_bias_v", (d // H, H)),
            (f"{c}.mha_layer.out_proj.weight", (d, d)), (f"{c}.mha_layer.out_proj.bias", (d,)),
            (f"{c}.mha_layer.linear_pos.weight", (d, d)),
            (f"{c}.convolution_module.layer_norm.weight", (d,)), (f"{c}.convolution_module.layer_norm.bias", (d,)),
            (f"{c}.convolution_module.bottleneck.0.weight", (2 * d, d, 1)),
            (f"{c}.convolution_module.bottleneck.0.bias", (2 * d,)),
            (f"{c}.convolution_module.conv.weight", (d, 1, BK)), (f"{c}.convolution_module.conv.bias", (d,)),
            (f"{c}.convolution_module.after_conv.0.weight", (d,)), (f"{c}.convolution_module.after_conv.0.bias", (d,)),
            (f"{c}.convolution_module.after_conv.2.weight", (d, d)), (f"{c}.convolution_module.after_conv.2.bias", (d,)),
        ])
        for m in ("ffn_module1", "ffn_module2"):
            shapes.extend([
                (f"{c}.{m}.0.weight", (d,)), (f"{c}.{m}.0.bias", (d,)),
                (f"{c}.{m}.1.ffn.0.weight", (f, d)), (f"{c}.{m}.1.ffn.0.bias", (f,)),
                (f"{c}.{m}.1.ffn.3.weight", (d, f)), (f"{c}.{m}.1.ffn.3.bias", (d,)),
            ])
        shapes.extend([
            (f"{c}.norm1.norm.weight", (d,)), (f"{c}.norm1.norm.bias", (d,)),
            (f"{c}.norm2.norm.weight", (d,)), (f"{c}.norm2.norm.bias", (d,)),
            (f"bottleneck.module_list.{l}.intra.norm.norm.weight", (d,)),
            (f"bottleneck.module_list.{l}.intra.norm.norm.bias", (d,)),
        ])
        t = f"bottleneck.module_list.{l}.inter.layers.0"
        shapes.extend([
            (f"{t}.self_attn.in_proj_weight", (3 * d, d)), (f"{t}.self_attn.in_proj_bias", (3 * d,)),
            (f"{t}.self_attn.out_proj.weight", (d, d)), (f"{t}.self_attn.out_proj.bias", (d,)),
            (f"{t}.linear1.weight", (f, d)), (f"{t}.linear1.bias", (f,)),
            (f"{t}.linear2.weight", (d, f)), (f"{t}.linear2.bias", (d,)),
            (f"{t}.norm1.weight", (d,)), (f"{t}.norm1.bias", (d,)),
            (f"{t}.norm2.weight", (d,)), (f"{t}.norm2.bias", (d,)),
        ])
    for i, (cin, cout, s) in enumerate(cfg.dec_channels()):
        p = f"decoder.module_list.{i}"
        shapes.extend([(f"{p}.upsample.conv.weight", (cin, 2 * cout, s)), (f"{p}.upsample.conv.bias", (2 * cout,)),
                       (f"{p}.norm1.weight", (2 * cout,)), (f"{p}.norm1.bias", (2 * cout,))])
        res(p, cout)
    E, EK = cfg.encoder_channels, cfg.encoder_kernel_size
    shapes.extend([
        ("reference_bypass.weight", (E, 1, EK)), ("reference_bypass.bias", (E,)),
        ("mask_encoder.weight", (E, cfg.channels, EK)), ("mask_encoder.bias", (E,)),
        ("output_decoder.weight", (E, 1, EK)), ("output_decoder.bias", (1,)),
    ])
    return shapes
