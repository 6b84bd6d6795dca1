"""Diagnostic: distribution of the five stage times of JointModel.forward over 12 forwards."""
import io, os, sys
from contextlib import redirect_stdout
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from acousticswarms_speech_amd.config import FULL, SEP_FULL
from acousticswarms_speech_amd.joint import JointModel
from acousticswarms_speech_amd.scenes import make_scene
from acousticswarms_speech_amd.sep import SepModel
from acousticswarms_speech_amd.spot import SpotModel
from acousticswarms_speech_amd.weights import make_sep_state_dict, make_spot_state_dict
scene = make_scene(1010, 5, 7, 48000, reverb=True)
spot = SpotModel(FULL, make_spot_state_dict(FULL, 5), batch_size=256, precision="f16x3").to("cuda")
sep = SepModel(SEP_FULL, make_sep_state_dict(SEP_FULL, 9), precision="f16x3").to("cuda")
jm = JointModel(spot, sep, device="cuda")
mix = torch.from_numpy(scene.mix)
rows = []
with redirect_stdout(io.StringIO()):
    jm.setup(scene.mic_positions, scene.speaker_range)
    for k in range(15):
        out = jm.forward(mix)
        if k >= 3:
            rows.append([t * 1e3 for t in jm.times])
r = np.array(rows)
print("pageable copies" if os.environ.get("ASW_PAGEABLE_COPIES") else "pinned copies")
for i, n in enumerate(["srp", "coarse", "fine", "clustering", "joint_sep"]):
    print(f"{n:11s} median {np.median(r[:, i]):7.1f}  min {r[:, i].min():7.1f}  max {r[:, i].max():7.1f}   " + " ".join(f"{v:5.0f}" for v in r[:, i]))
print(f"total       median {np.median(r.sum(1)):7.1f}  min {r.sum(1).min():7.1f}  max {r.sum(1).max():7.1f}")
