"""Diagnostic (not a pytest module): candidates/s of the hot call with one lane vs two lanes
(asw_spot_set_lanes) at several internal batch sizes; checks the results are identical."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acousticswarms_speech_amd.config import FULL  # noqa: E402
from acousticswarms_speech_amd.scenes import make_scene, random_offsets  # noqa: E402
from acousticswarms_speech_amd.spot import SpotModel  # noqa: E402
from acousticswarms_speech_amd.weights import make_spot_state_dict  # noqa: E402


def main(T=48000, N=256, steps=4):
    dev = torch.device("cuda", 0)
    m = SpotModel(FULL, make_spot_state_dict(FULL, 5), batch_size=256, precision="f16x3").to(dev)
    mix = torch.from_numpy(make_scene(1010, 5, 7, T, reverb=True).mix).to(dev)
    offs = torch.from_numpy(random_offsets(7, N, 6, 140)).to(dev)
    ref = None
    for lanes, batch in ((1, 256), (1, 128), (2, 128), (2, 64), (2, 32), (1, 64)):
        m.set_lanes(lanes)
        m.set_batch_size(batch)
        for _ in range(2):
            _, en = m.shift_and_sep_device(mix, offs, strict=1, want_wave=False, want_energy=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            _, en = m.shift_and_sep_device(mix, offs, strict=1, want_wave=False, want_energy=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        if ref is None:
            ref = en.clone()
        same = bool(torch.equal(en, ref))
        print(f"lanes={lanes} batch={batch}: {N / dt:8.1f} cand/s  ({dt * 1e3:.1f} ms/step)  identical={same}", flush=True)


if __name__ == "__main__":
    main()
