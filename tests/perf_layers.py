"""Per-layer GEMM table of one spot-network batch: every launch of the conv-as-GEMM kernels
tagged with its shape (asw_profile_enable(2)), time and achieved TFLOP/s.  Diagnostic."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acousticswarms_speech_amd import native  # noqa: E402
from acousticswarms_speech_amd.config import FULL  # noqa: E402
from acousticswarms_speech_amd.scenes import make_scene, random_offsets  # noqa: E402
from acousticswarms_speech_amd.spot import SpotModel  # noqa: E402
from acousticswarms_speech_amd.weights import make_spot_state_dict  # noqa: E402


def main(T=48000, batch=64, precision="f16x3", reps=3):
    dev = torch.device("cuda", 0)
    m = SpotModel(FULL, make_spot_state_dict(FULL, 5), batch_size=batch, precision=precision).to(dev)
    mix = torch.from_numpy(make_scene(1001, 3, 7, T).mix).to(dev)
    offs = torch.from_numpy(random_offsets(7, batch, 6, 140)).to(dev)
    L = native.lib()
    for _ in range(2):
        m.shift_and_sep_device(mix, offs, strict=1, want_wave=False, want_energy=True, window=12000)
    torch.cuda.synchronize()
    L.asw_profile_enable(2)
    for _ in range(reps):
        m.shift_and_sep_device(mix, offs, strict=1, want_wave=False, want_energy=True, window=12000)
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 18)
    native.check(L.asw_profile_report(buf, len(buf)))
    L.asw_profile_enable(0)
    prof = json.loads(buf.value.decode())
    rows = sorted(prof.items(), key=lambda kv: -kv[1]["ms"])
    tot = sum(v["ms"] for v in prof.values()) / reps
    print(f"T={T} batch={batch} {precision}: {tot:.2f} ms of launches per batch")
    print(f"{'kernel[shape]':78s} {'n':>3s} {'ms/batch':>9s} {'TFLOP/s':>8s} {'cum%':>6s}")
    cum = 0.0
    for k, v in rows:
        ms = v["ms"] / reps
        cum += ms
        tf = v["work"] / max(v["ms"], 1e-9) / 1e9
        print(f"{k:78s} {v['launches'] // reps:3d} {ms:9.3f} {tf:8.1f} {100 * cum / tot:6.1f}")


if __name__ == "__main__":
    main(T=int(sys.argv[1]) if len(sys.argv) > 1 else 48000)
