"""Stress workload of BASELINE config 5 on one GPU: 16-mic array, dense width-2 TDoA lattice
(SRP-PHAT bypassed), FULL spot net with n_mics = 16 and seeded random weights (no 16-mic
checkpoint exists).  Prints the lattice size and the candidates/s of the energies-only scorer
over a bounded slice of it.  Diagnostic, not the headline bench."""
import dataclasses
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from acousticswarms_speech_amd.config import FULL  # noqa: E402
from acousticswarms_speech_amd.dense_grid import dense_tdoa_candidates  # noqa: E402
from acousticswarms_speech_amd.scenes import make_scene  # noqa: E402
from acousticswarms_speech_amd.spot import SpotModel  # noqa: E402
from acousticswarms_speech_amd.weights import make_spot_state_dict  # noqa: E402


def main(n_eval=2048, T=48000):
    cfg = dataclasses.replace(FULL, n_mics=16)
    sc = make_scene(1010, 5, 16, T)
    t0 = time.time()
    offs, counts, _ = dense_tdoa_candidates(sc.mic_positions, sc.speaker_range, width=2, step=0.05, with_points=False)
    t_enum = time.time() - t0
    m = SpotModel(cfg, make_spot_state_dict(cfg, 1), batch_size=64, precision="f16x3").to("cuda")
    mix = torch.from_numpy(sc.mix).cuda()

    class P:
        def __init__(self, o):
            self.sample_offset = o
    step = max(1, len(offs) // n_eval)
    pick = [P(o) for o in offs[::step][:n_eval]]
    m.shift_and_score(mix, pick[:128], Strict=1)              # warm-up
    torch.cuda.synchronize()
    t0 = time.time()
    en = m.shift_and_score(mix, pick, Strict=1)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(json.dumps({"workload": "16 mics, 5 talkers, dense width-2 TDoA lattice at 5 cm, T=%d" % T,
                      "lattice_candidates": int(len(offs)), "enumeration_s": round(t_enum, 2),
                      "evaluated": len(pick), "candidates_per_s": round(len(pick) / dt, 1),
                      "full_lattice_s_at_this_rate": round(len(offs) * dt / len(pick), 1),
                      "max_power2": float(en[:, 1].max())}))


if __name__ == "__main__":
    main()
