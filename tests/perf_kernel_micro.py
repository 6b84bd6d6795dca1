"""Micro-benchmark of ONE layer through the C ABI (for rocprofv3 --pmc runs and A/B timing).
Not a pytest module.  Usage: python3 tests/perf_kernel_micro.py <layer>  (under the profiler:
`rocprofv3 --pmc ... -- python3 tests/perf_kernel_micro.py <layer>`, the interpreter itself after `--`)
 [--reps N] [--batch B] [--precision f16x3|f32]
layers: res64d1 res64d7 res64d49 res128d7 res256d7 res512d7 mask down0 qkv"""
import argparse
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("layer")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--precision", default="f16x3")
    ap.add_argument("--no-frag", action="store_true")
    a = ap.parse_args()
    from acousticswarms_speech_amd import ops
    g = torch.Generator().manual_seed(0)
    B = a.batch

    def rnd(*s, scale=1.0):
        return (torch.randn(*s, generator=g) * scale).cuda()
    L = a.layer
    kw = dict(precision=a.precision, use_fragments=not a.no_frag)
    if L.startswith("res"):
        C, d = [int(v) for v in L[3:].split("d")]
        T = {64: 48128, 128: 12032, 256: 3008, 512: 752}[C]
        x = rnd(B, T, C)
        w = ops.pack_conv_weight(rnd(C, C, 7, scale=1 / math.sqrt(7 * C)))
        b, gm, be = rnd(C, scale=0.1), 1 + rnd(C, scale=0.1), rnd(C, scale=0.1)
        flops = 2.0 * B * T * C * C * 7
        out = torch.empty_like(x)

        def run():
            ops.convgemm(x, w, T, C, C, taps=7, dil=d, pad=3 * d, bias=b, relu=True, resid=x, ln=(gm, be), out=out, **kw)
    elif L == "mask":
        T, C, E = 48128, 64, 2048
        F = 3008
        x = rnd(B, T, C)
        w = ops.pack_conv_weight(rnd(E, C, 33, scale=1 / math.sqrt(33 * C)))
        b = rnd(E, scale=0.1)
        y = rnd(B, F, E)
        flops = 2.0 * B * F * E * C * 33

        def run():
            ops.convgemm(x, w, F, E, C, taps=33, stride=16, pad=16, bias=b, relu=True, mul=y, out=y, **kw)
    elif L == "down0":
        T, C, N = 48128, 64, 128
        x = rnd(B, T, C)
        w = ops.pack_conv_weight(rnd(N, C, 7, scale=1 / math.sqrt(7 * C)))
        b = rnd(N, scale=0.1)
        flops = 2.0 * B * (T // 2) * N * C * 7
        out = torch.empty((B, T // 2, N), device="cuda")

        def run():
            ops.convgemm(x, w, T // 2, N, C, taps=7, stride=2, pad=3, bias=b, stats_chan_mod=N, out=out, **kw)
    elif L == "qkv":
        rows, d = B * 188, 1024
        x = rnd(1, rows, d)
        w = rnd(3 * d, d, scale=1 / math.sqrt(d))
        flops = 2.0 * rows * 3 * d * d
        out = torch.empty((1, rows, 3 * d), device="cuda")

        def run():
            ops.convgemm(x, w, rows, 3 * d, d, B=1, out=out, **kw)
    else:
        raise SystemExit("unknown layer")
    # ops.convgemm re-splits the weights on the host per call; time only the kernel via events
    run()
    torch.cuda.synchronize()
    from acousticswarms_speech_amd import native
    import ctypes
    import json
    Lb = native.lib()
    Lb.asw_profile_enable(1)
    for _ in range(a.reps):
        run()
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 14)
    Lb.asw_profile_report(buf, len(buf))
    prof = json.loads(buf.value.decode())
    for k, v in prof.items():
        print(f"{L} {k}: {v['ms'] / v['launches']:.4f} ms/launch, {flops / (v['ms'] / v['launches'] * 1e-3) / 1e12:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
