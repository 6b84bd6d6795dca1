"""Diagnostic: where does a residual-layer workgroup spend its cycles?  Needs a diagnostic build of the library with the cycle counters compiled in, selected through
ASW_LIB_PATH (the .so is not kept in the tree):

    cd acousticswarms-speech_amd && python -c "import native; native.build()" &&
    hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -DASW_PHASE_TIMING -c csrc/convgemm.hip -o /tmp/convgemm_dbg.o &&
    hipcc --offload-arch=gfx950 -fPIC -shared -o ../tests/micro/libasw_hip_phase.so /tmp/convgemm_dbg.o \
        $(ls build/*.o | grep -v convgemm.o)

Round-2 result (T = 48 128, batch 32, before the k-loop was software-pipelined): C = 64: staging 16 %,
k-loop 40 %, epilogue 44 % of a workgroup's cycles; C = 128: 12 / 51 / 37; C = 256: 8 / 66 / 26;
C = 512: 6 / 83 / 11.  Prints, per layer shape, the mean cycles wave 0 of a workgroup spends staging its
image, in the taps x k-steps loop and in the epilogue."""
import ctypes
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["ASW_LIB_PATH"] = os.path.join(ROOT, "tests", "micro", "libasw_hip_phase.so")
import torch  # noqa: E402
from acousticswarms_speech_amd import native, ops  # noqa: E402

L = native.lib()
L.asw_debug_phase_cycles.argtypes = [ctypes.c_void_p, ctypes.c_int]
g = torch.Generator().manual_seed(0)


def rnd(*s, scale=1.0):
    return (torch.randn(*s, generator=g) * scale).cuda()


def run(C, d, T, B=32, reps=3):
    x = rnd(B, T, C)
    w = ops.pack_conv_weight(rnd(C, C, 7, scale=1 / math.sqrt(7 * C)))
    b, gm, be = rnd(C, scale=0.1), 1 + rnd(C, scale=0.1), rnd(C, scale=0.1)
    out = torch.empty_like(x)
    buf = (ctypes.c_ulonglong * 4)()
    ops.convgemm(x, w, T, C, C, taps=7, dil=d, pad=3 * d, bias=b, relu=True, resid=x, ln=(gm, be), out=out, precision="f16x3")
    torch.cuda.synchronize()
    L.asw_debug_phase_cycles(buf, 1)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for _ in range(reps):
        # ops.convgemm splits the weights on the host per call: time only the launch with events is not possible
        # from here, so the wall figure below includes that; the cycle counters do not.
        ops.convgemm(x, w, T, C, C, taps=7, dil=d, pad=3 * d, bias=b, relu=True, resid=x, ln=(gm, be), out=out, precision="f16x3")
    torch.cuda.synchronize()
    L.asw_debug_phase_cycles(buf, 1)
    n = max(1, buf[3])
    st, lp, ep = buf[0] / n, buf[1] / n, buf[2] / n
    tot = st + lp + ep
    print(f"C={C:3d} dil={d:2d} T={T}: per workgroup {tot:8.0f} cycles = staging {st:7.0f} ({100 * st / tot:4.1f} %) + "
          f"k-loop {lp:7.0f} ({100 * lp / tot:4.1f} %) + epilogue {ep:7.0f} ({100 * ep / tot:4.1f} %); workgroups {n // reps}", flush=True)


for C, d, T in ((64, 1, 48128), (64, 7, 48128), (64, 49, 48128), (128, 1, 12032), (256, 1, 3008), (512, 1, 752)):
    run(C, d, T)


def run_mask(B=32, reps=3):
    """Generic 256x256 GEMM on the mask-encoder shape."""
    T, C, E, F = 48128, 64, 2048, 3008
    L.asw_debug_gemm_cycles.argtypes = [ctypes.c_void_p, ctypes.c_int]
    x = rnd(B, T, C)
    w = ops.pack_conv_weight(rnd(E, C, 33, scale=1 / math.sqrt(33 * C)))
    b = rnd(E, scale=0.1)
    y = rnd(B, F, E)
    buf = (ctypes.c_ulonglong * 5)()
    ops.convgemm(x, w, F, E, C, taps=33, stride=16, pad=16, bias=b, relu=True, mul=y, out=y, precision="f16x3")
    torch.cuda.synchronize()
    L.asw_debug_gemm_cycles(buf, 1)
    for _ in range(reps):
        ops.convgemm(x, w, F, E, C, taps=33, stride=16, pad=16, bias=b, relu=True, mul=y, out=y, precision="f16x3")
    torch.cuda.synchronize()
    L.asw_debug_gemm_cycles(buf, 1)
    n = max(1, buf[4])
    wt, st, cp, ep = (buf[i] / n for i in range(4))
    tot = wt + st + cp + ep
    print(f"mask encoder 256x256 tile, 66 chunks: per workgroup {tot:8.0f} cycles = barrier wait {wt:7.0f} ({100 * wt / tot:4.1f} %) + "
          f"deposit {st:7.0f} ({100 * st / tot:4.1f} %) + loads/MFMA {cp:7.0f} ({100 * cp / tot:4.1f} %) + epilogue {ep:7.0f} "
          f"({100 * ep / tot:4.1f} %); per chunk: wait {wt / 66:.0f} deposit {st / 66:.0f} compute {cp / 66:.0f}", flush=True)


run_mask()
