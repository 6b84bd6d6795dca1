"""Diagnostic only: phase shares of the generic f16x3 GEMM main loop from an instrumented
build (ASW_LIB_PATH=.abl/libasw_stamp.so).  Shares, not durations, are meaningful."""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from acousticswarms_speech_amd import native, ops
import math

L = native.lib()
g = torch.Generator().manual_seed(0)
layer = sys.argv[1] if len(sys.argv) > 1 else "qkv"
B = 32
if layer == "qkv":
    rows, d = B * 188, 1024
    x = (torch.randn(1, rows, d, generator=g)).cuda()
    w = (torch.randn(3 * d, d, generator=g) / math.sqrt(d)).cuda()
    run = lambda: ops.convgemm(x, w, rows, 3 * d, d, B=1, precision="f16x3")
else:
    T, C, E, F = 48128, 64, 2048, 3008
    x = torch.randn(B, T, C, generator=g).cuda()
    w = ops.pack_conv_weight(torch.randn(E, C, 33, generator=g) / math.sqrt(33 * C)).cuda()
    y = torch.randn(B, F, E, generator=g).cuda()
    run = lambda: ops.convgemm(x, w, F, E, C, taps=33, stride=16, pad=16, relu=True, mul=y, out=y, precision="f16x3")
buf = (ctypes.c_ulonglong * 16)()
run(); torch.cuda.synchronize()
L.asw_dbg_read(buf, 1)
run(); torch.cuda.synchronize()
L.asw_dbg_read(buf, 0)
names = ["wait barrier1 (prev compute done)", "lstore (wait loads + split + LDS writes)", "wait barrier2", "gload issue",
         "compute (LDS reads + MFMA)", "prologue loads", "epilogue"]
tot = sum(buf[i] for i in range(7))
print(layer, "waves", buf[8])
for i, n in enumerate(names):
    print(f"  {n:45s} {100.0 * buf[i] / tot:5.1f} %   {buf[i] / max(buf[8], 1):10.0f} cycles/wave")
