"""Diagnostic: the single-pass f16 mode (precision "f16": one MFMA per product) against f16x3 and f32 --
throughput on the bench workload and SNR of the waveforms / energies."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from acousticswarms_speech_amd.config import FULL
from acousticswarms_speech_amd.scenes import make_scene, random_offsets
from acousticswarms_speech_amd.spot import SpotModel
from acousticswarms_speech_amd.weights import make_spot_state_dict
dev = torch.device("cuda", 0)
m = SpotModel(FULL, make_spot_state_dict(FULL, 5), batch_size=256, precision="f32").to(dev)
mix = torch.from_numpy(make_scene(1010, 5, 7, 48000, reverb=True).mix).to(dev)
offs = torch.from_numpy(random_offsets(7, 256, 6, 140)).to(dev)
def snr(a, b): return 10 * np.log10((b.astype(np.float64) ** 2).sum() / max(((a.astype(np.float64) - b) ** 2).sum(), 1e-300))
ref = None
for prec in ("f32", "f16x3", "f16"):
    m.set_precision(prec)
    w, e = m.shift_and_sep_device(mix, offs[:32], strict=1, want_wave=True, want_energy=True, window=12000)
    w, e = w.cpu().numpy(), e.cpu().numpy()
    if ref is None: ref = (w, e)
    m.shift_and_sep_device(mix, offs, strict=1, want_wave=False, want_energy=True, window=12000)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): m.shift_and_sep_device(mix, offs, strict=1, want_wave=False, want_energy=True, window=12000)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    per = [snr(w[i], ref[0][i]) for i in range(32)]
    print(f"{prec:6s}: {256 / dt:7.1f} cand/s; waveform SNR vs f32 min {min(per):6.1f} dB median {np.median(per):6.1f} dB; "
          f"energy rel err max {np.abs(e / ref[1] - 1).max():.2e}")
