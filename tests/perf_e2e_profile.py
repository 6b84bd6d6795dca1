"""Diagnostic: cProfile of one JointModel.forward (host-side hot spots of the search)."""
import cProfile
import io
import os
import pstats
import sys
from contextlib import redirect_stdout
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from acousticswarms_speech_amd.config import FULL, SEP_FULL
from acousticswarms_speech_amd.joint import JointModel
from acousticswarms_speech_amd.scenes import make_scene
from acousticswarms_speech_amd.sep import SepModel
from acousticswarms_speech_amd.spot import SpotModel
from acousticswarms_speech_amd.weights import make_sep_state_dict, make_spot_state_dict

scene = make_scene(1010, 5, 7, 48000, reverb=True)                      # the bench's configs[2] scene
spot = SpotModel(FULL, make_spot_state_dict(FULL, 5), batch_size=256, precision="f16x3").to("cuda")
sep = SepModel(SEP_FULL, make_sep_state_dict(SEP_FULL, 9), precision="f16x3").to("cuda")
jm = JointModel(spot, sep, device="cuda")
mix = torch.from_numpy(scene.mix)
with redirect_stdout(io.StringIO()):
    jm.setup(scene.mic_positions, scene.speaker_range)
    jm.forward(mix)
pr = cProfile.Profile()
with redirect_stdout(io.StringIO()):
    pr.enable()
    jm.forward(mix)
    pr.disable()
print("times", [round(t * 1e3, 1) for t in jm.times])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:12000])
