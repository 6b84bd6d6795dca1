"""Information only: how far apart two fp32 executions of the same network already are.
Runs the oracle's torch.nn.functional statement of shift_and_sep with stock PyTorch-ROCm fp32
ops on the MI355X and compares it with the reference's own CPU output (fixture g4b), next to the
HIP path in both arithmetic modes.  Puts the 104-105 dB of the f16x3 mode in context."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def snr_db(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return float(10 * np.log10(np.sum(ref ** 2) / max(np.sum((got - ref) ** 2), 1e-300)))


def main():
    from acousticswarms_speech_amd.config import FULL
    from acousticswarms_speech_amd.scenes import make_scene
    from acousticswarms_speech_amd.spot import SpotModel
    from acousticswarms_speech_amd.weights import make_spot_state_dict
    from oracle import spot_ref
    g = np.load(os.path.join(ROOT, "tests", "golden", "g4b_shift_and_sep_full.npz"))
    sd_np = make_spot_state_dict(FULL, 5)
    mix = torch.from_numpy(make_scene(2, 3, 7, 6000).mix)
    offs = [o for o in g["offsets"]]
    ref = g["y_strict1"]
    dev = torch.device("cuda:0")
    sd_gpu = {k: torch.from_numpy(v).to(dev) for k, v in sd_np.items()}
    torch.backends.cudnn.allow_tf32 = False
    w = torch.tensor([1.0, 0.0], device=dev)
    rows = []
    for i in range(0, len(offs), 2):                     # the reference's loop, forward on the GPU
        data = torch.stack([spot_ref.roll_channels(mix.to(torch.float32), o) for o in offs[i:i + 2]])
        dn, mu, sg = spot_ref.normalize_input(data)
        y = spot_ref.spot_forward(sd_gpu, FULL, dn.to(dev), w.expand(dn.shape[0], 2))
        rows.append(spot_ref.unnormalize_input(y.cpu(), mu, sg)[:, 0].numpy())
    y_stock = np.concatenate(rows)
    out = {"stock_pytorch_rocm_fp32_vs_reference_cpu_db": round(snr_db(y_stock, ref), 1)}
    for prec in ("f32", "f16x3"):
        m = SpotModel(FULL, sd_np, batch_size=2, precision=prec).to(dev)

        class P:
            def __init__(self, o):
                self.sample_offset = o
        y = m.shift_and_sep(mix, [P(o) for o in offs], Strict=1)
        out[f"hip_{prec}_vs_reference_cpu_db"] = round(snr_db(y, ref), 1)
        out[f"hip_{prec}_vs_stock_gpu_db"] = round(snr_db(y, y_stock), 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
