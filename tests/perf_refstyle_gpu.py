"""Information only (not a pytest module, not the product): times the REFERENCE-STYLE GPU
path on the MI355X -- the oracle's torch.nn.functional statement of the spot network
executed by stock PyTorch-ROCm ops (MIOpen / rocBLAS) under the reference's loop
structure (per-candidate roll loop, normalise, batched forward of `--batch` candidates,
sep/training/JointModel/network.py:75-96).  This is the number the north-star's ">= 8x the
reference single-GPU throughput" target is measured against.  Writes one JSON line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--candidates", type=int, default=64)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--samples", type=int, default=48000)
    ap.add_argument("--repeats", type=int, default=2)
    ap.add_argument("--host-energies", action="store_true",
                    help="include the reference's host-side energy loop in the timed region")
    args = ap.parse_args()
    from acousticswarms_speech_amd.config import FULL
    from acousticswarms_speech_amd.scenes import make_scene, random_offsets
    from acousticswarms_speech_amd.weights import make_spot_state_dict
    from oracle import spot_ref
    dev = torch.device("cuda:0")
    sd = {k: torch.from_numpy(v).to(dev) for k, v in make_spot_state_dict(FULL, 5).items()}
    mix = torch.from_numpy(make_scene(1001, 3, 7, args.samples).mix).to(dev)
    offs = random_offsets(7, args.candidates, 6, 140)
    w = torch.tensor([1.0, 0.0], device=dev)

    def roll(mix, off):
        M, T = mix.shape
        o = torch.tensor([0, *[int(v) for v in off]], device=dev).view(M, 1)
        idx = (torch.arange(T, device=dev).view(1, T) + o) % T
        return torch.gather(mix, 1, idx)

    def run():
        outs = []
        for i in range(0, args.candidates, args.batch):
            chunk = offs[i:i + args.batch]
            data = torch.stack([roll(mix, o) for o in chunk])
            dn, mu, sg = spot_ref.normalize_input(data)
            y = spot_ref.spot_forward(sd, FULL, dn, w.expand(len(chunk), 2))
            outs.append(spot_ref.unnormalize_input(y, mu, sg)[:, 0])
        r = torch.cat(outs).cpu().numpy()
        if args.host_energies:
            # the reference then scores every candidate on the host (mean removal, sum of squares,
            # windowed RMS through scipy's uniform_filter1d: local_utils_3d.py:13-17,349-354)
            spot_ref.candidate_energies(r)
        return r

    run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.repeats):
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    dt = min(ts)
    print(json.dumps({"what": "reference-style GPU path (stock PyTorch-ROCm ops, oracle statement)",
                      "candidates_per_s": args.candidates / dt, "candidates": args.candidates,
                      "batch": args.batch, "T": args.samples, "seconds": dt, "dtype": "f32",
                      "host_energies_included": bool(args.host_energies)}))


if __name__ == "__main__":
    main()
