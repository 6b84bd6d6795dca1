#!/usr/bin/env python3
"""Benchmark of the Spotforming candidate hot path on MI355X.

One "step" = one fine-stage call of the hot loop (the shape of
Mic_Array.Spotform_Small_Patch_Parallel -> spot_model.shift_and_sep(..., Strict=1),
sep/Mic_Array.py:263): `--candidates` TDoA candidates of one synthetic 7-mic mixture are
shifted, normalised, run through the FULL 47.27 M-parameter spot network and reduced to
(power, power2) energies, with mixture, offsets and weights already resident in HBM.
With N GPUs the candidate list is N times longer and sharded (weak scaling); each step
ends with the stage's one exchange, an all-gather of the energies over RCCL.

Prints ONE JSON line (rank 0):  metric = TDoA candidates/s (whole job).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

# MI355X_MICROARCH.md: dense f32 matrix peak (exact fp32 MFMA) and dense f16 MFMA peak.  In the
# f16x3 mode every algorithmic product costs three f16 MFMAs, so the ceiling for ALGORITHMIC
# flops is 2500 / 3.
PEAK_TFLOPS = {"f32": 157.3, "f16x3": 2500.0 / 3.0}
# What the matrix pipe sustains on this chip with random (non-zero) fp16 operands and nothing else
# running: 1.63 PFLOP/s of v_mfma_f32_32x32x16_f16 (2.2 with all-zero operands), i.e. the clock is
# power-limited under dense MFMA load (tests/micro/cu_probe.hip, profiles/r1/cu_probe_random_operands.txt).
# Reported beside the nominal peak; `frac` stays on the nominal figure.
SUSTAINED_TFLOPS = {"f16x3": 1630.0 / 3.0}
DTYPE_NAME = {"f32": "f32", "f16x3": "f32 via f16x3 split-operand MFMA (fp32 accumulate)"}


def cpu_baseline(cfg, sd, mix, offsets, n_sample):
    """The oracle (CPU restatement of the reference, fixture-pinned) on a bounded sample of the
    same workload, on this box's host cores."""
    from oracle import spot_ref
    # the GPU box gives one GPU a 16-core share of the host; more threads only oversubscribe
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(16, avail)))
    offs = [o for o in offsets[:n_sample]]
    spot_ref.shift_and_sep(sd, cfg, mix, offs[:1], strict=1)            # warm-up (thread pools, caches)
    t0 = time.perf_counter()
    y = spot_ref.shift_and_sep(sd, cfg, mix, offs, strict=1, batch_size=1)
    spot_ref.candidate_energies(y)
    dt = time.perf_counter() - t0
    return {"value": len(offs) / dt, "unit": "candidates/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{len(offs)} candidates of the same mixture (M=7, T={mix.shape[1]}), batch 1, "
                      f"oracle.spot_ref.shift_and_sep + energies, {dt:.1f} s"}


def e2e_latency(model, scene, dev):
    """Second half of BASELINE.json's metric: end-to-end localise(+separate) latency of ONE
    mixture through the whole pipeline (JointModel.forward: SRP-PHAT -> coarse -> fine ->
    clustering), per stage as JointModel.times (sep/training/JointModel/network.py:143-194),
    with device synchronisation at every stage boundary; geometry setup() excluded, as the
    reference's README says.  Seeded random weights: the candidate counts of the search are
    those a random network produces (close to the worst case of 30 coarse survivors)."""
    import io
    from contextlib import redirect_stdout
    from acousticswarms_speech_amd.joint import JointModel
    jm = JointModel(model, None, device=dev)
    mix = torch.from_numpy(scene.mix)
    with redirect_stdout(io.StringIO()):
        t0 = time.perf_counter()
        jm.setup(scene.mic_positions, scene.speaker_range)
        setup_s = time.perf_counter() - t0
        jm.forward(mix)                                   # warm-up (workspace growth, gate cache)
        out = jm.forward(mix)
    mp = jm.Mic_processor
    return {"unit": "ms", "total": round(sum(jm.times) * 1e3, 2),
            "stages": {k: round(v * 1e3, 2) for k, v in zip(["srp_phat", "coarse", "fine", "clustering", "joint_sep"],
                                                            jm.times)},
            "spot_calls": {"coarse": int(mp.big_spotforming_times), "fine": int(mp.spotforming_times)},
            "talkers_found": len(out[0]), "setup_excluded_s": round(setup_s, 2),
            "note": "joint separation network is a next-row component (stage 5 = 0)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--candidates", type=int, default=256, help="candidates per GPU per step")
    ap.add_argument("--samples", type=int, default=48000, help="T: 48000 = 3 s @ 16 kHz (BASELINE literal)")
    ap.add_argument("--batch", type=int, default=256, help="internal candidate batch (spot_batch_size)")
    ap.add_argument("--cpu-sample", type=int, default=96, help="candidates timed on the CPU baseline (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="skip the in-library per-kernel event timing")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end pipeline latency measurement")
    ap.add_argument("--precision", choices=["f32", "f16x3"], default="f16x3",
                    help="GEMM arithmetic: exact f32 MFMA, or f16x3 split-operand MFMA (105 dB SNR vs the reference)")
    args = ap.parse_args()

    from acousticswarms_speech_amd import native
    from acousticswarms_speech_amd.config import FULL
    from acousticswarms_speech_amd.flops import flops_per_candidate
    from acousticswarms_speech_amd.scenes import make_scene, random_offsets
    from acousticswarms_speech_amd.spot import SpotModel
    from acousticswarms_speech_amd.weights import make_spot_state_dict

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    cfg, T = FULL, args.samples
    sd = make_spot_state_dict(cfg, seed=5)
    model = SpotModel(cfg, sd, batch_size=args.batch, precision=args.precision).to(dev)
    scene = make_scene(1001, n_speakers=3, n_mics=7, T=T)
    mix_d = torch.from_numpy(scene.mix).to(dev)
    n_total = args.candidates * world
    offsets = random_offsets(7, n_total, 6, 140)
    off_d = torch.from_numpy(offsets).to(dev)

    def local_score(_mix, offs_local_dev):
        _, en = model.shift_and_sep_device(mix_d, offs_local_dev, strict=1, want_wave=False, want_energy=True,
                                           window=12000)
        return en

    from acousticswarms_speech_amd.shard import shard_bounds
    b = shard_bounds(n_total, world)
    my_off = off_d[b[rank]:b[rank + 1]].contiguous()

    def step():
        en = local_score(None, my_off)
        if world > 1:
            width = max(b[r + 1] - b[r] for r in range(world))
            buf = torch.zeros((width, 2), dtype=torch.float64, device=dev)
            buf[:en.shape[0]] = en
            out = torch.empty((world * width, 2), dtype=torch.float64, device=dev)
            dist.all_gather_into_tensor(out, buf)
            return out
        return en

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    L = native.lib()
    profile = not args.no_profile
    barrier()
    if profile:
        L.asw_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    barrier()
    dt = time.perf_counter() - t0
    prof = {}
    if profile:
        buf = ctypes.create_string_buffer(1 << 16)
        native.check(L.asw_profile_report(buf, len(buf)))
        L.asw_profile_enable(0)
        prof = json.loads(buf.value.decode())
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(last).all()

    if rank == 0:
        fl = flops_per_candidate(cfg, T)
        value = n_total * args.steps / dt
        # dominant kernel = the convgemm instantiation with the largest measured time
        roof = None
        peak = PEAK_TFLOPS[args.precision]
        if prof:
            name, rec = max(prof.items(), key=lambda kv: kv[1]["ms"])
            ach = rec["work"] / (rec["ms"] * 1e-3) / 1e12
            tot_ms = sum(r["ms"] for r in prof.values())
            tot_work = sum(r["work"] for r in prof.values())
            traffic = traffic_detail = None
            try:      # measured offline by rocprofv3 --pmc (separate passes); only valid for the same kernel
                tj = json.load(open(os.path.join(ROOT, "profiles", "r1", "traffic.json")))
                if tj["kernel"] == name:
                    # per launch of the dominant kernel = one internal batch; counters were taken at batch tj["batch"]
                    scale = min(args.batch, args.candidates) / float(tj["batch"])
                    traffic = int(tj["corrected_bytes_per_launch"] * scale)
                    traffic_detail = {"fetch_size_bytes_raw": int(tj["read_bytes_raw"] * scale),
                                      "write_size_bytes": int(tj["write_bytes"] * scale),
                                      "algorithmic_bytes": int(tj["algorithmic_bytes"] * scale),
                                      "correction": "FETCH_SIZE x2 (gfx950 wide streaming reads) + WRITE_SIZE",
                                      "counters_measured_at_batch": tj["batch"], "source": "profiles/r1/traffic.json"}
            except (OSError, KeyError, ValueError):
                pass
            roof = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": round(peak, 1),
                    "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                    "traffic_detail": traffic_detail,
                    "sustained_peak": ({"tflops": round(SUSTAINED_TFLOPS[args.precision], 1),
                                        "frac": round(ach / SUSTAINED_TFLOPS[args.precision], 4),
                                        "note": "MFMA-only loop on random operands, power-limited clock; "
                                                "profiles/r1/cu_probe_random_operands.txt"}
                                       if args.precision in SUSTAINED_TFLOPS else None),
                    "avg_launch_ms": round(rec["ms"] / rec["launches"], 4), "launches": rec["launches"],
                    "all_gemm_kernels": {"achieved": round(tot_work / (tot_ms * 1e-3) / 1e12, 2),
                                         "frac": round(tot_work / (tot_ms * 1e-3) / 1e12 / peak, 4),
                                         "share_of_step_time": round(tot_ms * 1e-3 / dt, 3)},
                    "per_kernel": {k: {"ms_per_step": round(v["ms"] / args.steps, 3),
                                       "tflops": round(v["work"] / (v["ms"] * 1e-3) / 1e12, 2)}
                                   for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}}
        e2e = None
        if not args.no_e2e:
            e2e = e2e_latency(model, scene, dev)
        cpu = None
        if args.cpu_sample > 0 and world == 1:           # reported on rank 0 at N = 1 only
            cpu = cpu_baseline(cfg, sd, torch.from_numpy(scene.mix), offsets, args.cpu_sample)
        line = {
            "metric": "TDoA candidates/sec (shift+normalise+spot forward+energies)", "value": round(value, 2),
            "unit": "candidates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE_NAME[args.precision], "data": "synthetic",
            "config": {"workload": "configs[1]: 3-speaker free-field scene, 7 mics, T=%d samples (3 s), fine-stage "
                                   "(Strict=1) candidate batch of one mixture, FULL spot net 47.27 M params, seeded "
                                   "random weights" % T,
                       "candidates_per_gpu_per_step": args.candidates, "internal_batch": args.batch,
                       "gflop_per_candidate": round(fl["total"] / 1e9, 2), "parallelism": f"candidate-shard x{world}"},
            "effective_tflops": round(value * fl["total"] / 1e12, 2),
            "roofline": roof, "cpu_baseline": cpu, "e2e_latency": e2e,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
