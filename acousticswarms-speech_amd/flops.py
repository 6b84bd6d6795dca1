"""Algorithmic work of one TDoA candidate: 2 x MACs of every conv,
transposed conv, linear and attention matmul of the spot forward.  Reproduces the
survey's hook count: 126.52 GFLOP at (M=7, T=48 000), 380.62 GFLOP at T=144 000."""
from .config import SpotConfig


def flops_per_candidate(cfg: SpotConfig, T: int) -> dict:
    Tp = cfg.padded_length(T)
    K = cfg.kernel_size
    Tl = [Tp]
    for s in cfg.stride_list:
        Tl.append(Tl[-1] // s)
    out = {"preproc": 2.0 * cfg.n_mics * cfg.channels * Tp, "res": 0.0, "down": 0.0, "up": 0.0}
    for i, (cin, cout) in enumerate(cfg.enc_channels()):
        out["res"] += 2 * cfg.residual_layers * 2.0 * K * cin * cin * Tl[i]      # encoder + decoder stacks
        out["down"] += 2.0 * K * cin * 2 * cout * Tl[i + 1]
    for j, (cin, cout, s) in enumerate(cfg.dec_channels()):
        out["up"] += 2.0 * cin * 2 * cout * s * Tl[cfg.depth - j]
    L, d, f = Tl[-1], cfg.bottleneck_channels, cfg.ffw_dim
    out["transformer"] = cfg.num_transformer_layers * (2.0 * L * d * 3 * d + 2.0 * L * d * d + 4.0 * L * L * d
                                                       + 4.0 * L * d * f)
    F, E, EK = cfg.latent_frames(Tp), cfg.encoder_channels, cfg.encoder_kernel_size
    out["mask"] = 2.0 * EK * cfg.channels * E * F + 2.0 * EK * E * F + 2.0 * E * EK * F
    out["total"] = sum(out.values())
    return out
