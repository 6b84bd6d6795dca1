"""Diagnostic (not a pytest module): wall time of SepModel.infer_device for S speakers at T
samples, f16x3 / f32.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel table."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acousticswarms_speech_amd.config import SEP_FULL  # noqa: E402
from acousticswarms_speech_amd.scenes import make_scene, random_offsets  # noqa: E402
from acousticswarms_speech_amd.sep import SepModel  # noqa: E402
from acousticswarms_speech_amd.weights import make_sep_state_dict  # noqa: E402


def main(S=5, T=48000, precision="f16x3", reps=5):
    dev = torch.device("cuda", 0)
    m = SepModel(SEP_FULL, make_sep_state_dict(SEP_FULL, 9), precision=precision).to(dev)
    mix = torch.from_numpy(make_scene(1010, 5, 7, T, reverb=True).mix).to(dev)
    offs = torch.from_numpy(random_offsets(7, S, 6, 140)).to(dev)
    for _ in range(2):
        m.infer_device(mix, offs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        m.infer_device(mix, offs)
    torch.cuda.synchronize()
    print(f"sep infer S={S} T={T} {precision}: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms per call")


if __name__ == "__main__":
    a = sys.argv[1:]
    main(S=int(a[0]) if a else 5, T=int(a[1]) if len(a) > 1 else 48000, precision=a[2] if len(a) > 2 else "f16x3")
