"""Diagnostic: where does a workgroup of the fused residual-stack kernel (csrc/resstack.hip) spend its cycles?
Needs a diagnostic build of the library with the cycle stamps compiled in, selected through ASW_LIB_PATH (the
.so is not kept in the tree):

    cd acousticswarms-speech_amd && python -c "import native; native.build()" &&
    hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -DASW_PHASE_TIMING -c csrc/resstack.hip -o /tmp/resstack_dbg.o &&
    hipcc --offload-arch=gfx950 -fPIC -shared -o ../tests/micro/libasw_hip_rsphase.so /tmp/resstack_dbg.o \
        $(ls build/*.o | grep -v resstack.o)

Prints, per stack shape, the mean cycles wave 0 of a workgroup spends staging, in each layer's k-loop and in each
layer's epilogue (+ hand-over barriers / stores)."""
import ctypes
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["ASW_LIB_PATH"] = os.path.join(ROOT, "tests", "micro", "libasw_hip_rsphase.so")
import torch  # noqa: E402
from acousticswarms_speech_amd import native, ops  # noqa: E402

L = native.lib()
L.asw_debug_resstack_cycles.argtypes = [ctypes.c_void_p, ctypes.c_int]
g = torch.Generator().manual_seed(0)


def rnd(*s, scale=1.0):
    return (torch.randn(*s, generator=g) * scale).cuda()


def run(dils, T, B=32, reps=3):
    x = rnd(B, T, 64)
    layers = [(ops.pack_conv_weight(rnd(64, 64, 7, scale=1 / math.sqrt(448))), rnd(64, scale=0.1), 1 + rnd(64, scale=0.1),
               rnd(64, scale=0.1), d) for d in dils]
    out = torch.empty_like(x)
    buf = (ctypes.c_ulonglong * 6)()
    ops.resstack(x, layers, out=out)
    torch.cuda.synchronize()
    L.asw_debug_resstack_cycles(buf, 1)
    for _ in range(reps):
        ops.resstack(x, layers, out=out)
    torch.cuda.synchronize()
    L.asw_debug_resstack_cycles(buf, 1)
    n = max(1, buf[5])
    v = [buf[i] / n for i in range(5)]
    tot = sum(v)
    names = ["staging", "k-loop 0", "epilogue 0", "k-loop 1", "epilogue 1 / stores"]
    print(f"dils={dils} T={T} B={B}: per workgroup {tot:8.0f} cycles: " +
          ", ".join(f"{nm} {c:7.0f} ({100 * c / tot:4.1f} %)" for nm, c in zip(names, v) if c > 0) +
          f"; workgroups {n // reps}", flush=True)


for dils, T in (((1, 7), 48128), ((1,), 48128), ((49,), 48128), ((1, 7), 24064)):
    run(dils, T)
