"""Diagnostic: run a few forwards under `rocprofv3 --hip-trace --kernel-trace` and mark the stage boundaries
with hipDeviceSynchronize pairs, to see which HIP call of the separation stage absorbs the 20-25 ms."""
import io, os, sys
from contextlib import redirect_stdout
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
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
with redirect_stdout(io.StringIO()):
    jm.setup(scene.mic_positions, scene.speaker_range)
    for k in range(8):
        jm.forward(mix)
        print("times", [round(t * 1e3, 1) for t in jm.times], file=sys.stderr)
