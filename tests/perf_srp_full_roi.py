import io, sys, time
from contextlib import redirect_stdout
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from acousticswarms_speech_amd.scenes import make_scene
from acousticswarms_speech_amd.mic_array import MicArray
sc = make_scene(1010, 5, 7, 144000)
roi = [-2.2, 2.25, 0.0, 6.2, 0.0, 0.9]
t0 = time.time()
with redirect_stdout(io.StringIO()):
    ma = MicArray(sc.mic_positions, Spk_Range=roi, device="cuda")
t_setup = time.time() - t0
mix = torch.from_numpy(sc.mix)
with redirect_stdout(io.StringIO()):
    ma.Apply_SRP_PHAT(mix)
torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.time()
    with redirect_stdout(io.StringIO()):
        p, _ = ma.Apply_SRP_PHAT(mix)
    torch.cuda.synchronize()
    ts.append(time.time() - t0)
node = ma.SRP_node
t0 = time.time(); node.SRP_Map_WINDOW_new(sc.mix, window=36000); torch.cuda.synchronize(); t_map = time.time() - t0
print({"G_clusters": len(node.clusters), "setup_s": round(t_setup, 2), "apply_srp_phat_ms": [round(t * 1e3, 1) for t in ts],
       "map_only_ms": round(t_map * 1e3, 1), "coarse_patches": len(p), "T": 144000})
