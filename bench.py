"""Benchmark of the Spotforming candidate hot path on MI355X.

One "step" = one fine-stage call of the hot loop (the shape of
Mic_Array.Spotform_Small_Patch_Parallel -> spot_model.shift_and_sep(..., Strict=1),
sep/Mic_Array.py:263): `--candidates` TDoA candidates of one synthetic 7-mic, 5-speaker
mixture are shifted, normalised, run through the FULL 47.27 M-parameter spot network and
reduced to (power, power2) energies, with mixture, offsets and weights already resident in
HBM.  With N GPUs the candidate list is N times longer and sharded (weak scaling); each
step ends with the stage's one exchange, an all-gather of the energies over RCCL.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no torchrun environment the script starts the N ranks itself
(`python -m torch.distributed.run ...`, one process per GPU) BEFORE anything touches the
GPU, and exits with their status; under torchrun it is one of the ranks.

Prints ONE JSON line (rank 0):  metric = TDoA candidates/s (whole job).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

# MI355X_MICROARCH.md: dense f32 matrix peak (exact fp32 MFMA) and dense f16 MFMA peak.  In the
# f16x3 mode every algorithmic product costs three f16 MFMAs, so the ceiling for ALGORITHMIC
# flops is 2500 / 3.
PEAK_TFLOPS = {"f32": 157.3, "f16x3": 2500.0 / 3.0}
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured for a float4 copy)
# What the matrix pipe sustains on this chip with random (non-zero) fp16 operands and nothing else
# running: 1.63 PFLOP/s of v_mfma_f32_32x32x16_f16 (2.2 with all-zero operands), i.e. the clock is
# power-limited under dense MFMA load (tests/micro/cu_probe.hip, profiles/r1/cu_probe_random_operands.txt).
# Reported beside the nominal peak; `frac` stays on the nominal figure.
SUSTAINED_TFLOPS = {"f16x3": 1630.0 / 3.0}
DTYPE_NAME = {"f32": "f32", "f16x3": "f32 via f16x3 split-operand MFMA (fp32 accumulate)"}
WORKLOAD_SEED, WORKLOAD_SPEAKERS = 1010, 5      # configs[2] scene: 5 talkers, image-source reverb


def host_threads():
    # the GPU box gives one GPU a 16-core share of the host; more threads only oversubscribe
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(16, avail))


def cpu_baseline(cfg, sd, mix, offsets, n_sample):
    """Baseline leg: the oracle (CPU restatement of the reference, fixture-pinned) on a bounded
    sample of the same workload, on this box's host cores."""
    from oracle import spot_ref
    torch.set_num_threads(host_threads())
    offs = [o for o in offsets[:n_sample]]
    spot_ref.shift_and_sep(sd, cfg, mix, offs[:1], strict=1)            # warm-up (thread pools, caches)
    t0 = time.perf_counter()
    y = spot_ref.shift_and_sep(sd, cfg, mix, offs, strict=1, batch_size=1)
    spot_ref.candidate_energies(y)
    dt = time.perf_counter() - t0
    return {"value": len(offs) / dt, "unit": "candidates/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{len(offs)} candidates of the same mixture (M=7, T={mix.shape[1]}), batch 1, "
                      f"oracle.spot_ref.shift_and_sep + energies, {dt:.1f} s"}


def refstyle_gpu_baseline(cfg, sd, mix_dev, offsets, n_cand, batch=128):
    """Baseline leg: the REFERENCE-STYLE single-GPU path on this MI355X -- the oracle's
    torch.nn.functional statement of the spot network run by stock PyTorch-ROCm ops (MIOpen /
    rocBLAS) under the reference's loop structure (per-candidate roll loop, normalise, batched
    forward of 128 candidates, D2H of every waveform, host energy loop:
    sep/training/JointModel/network.py:75-99, sep/helpers/local_utils_3d.py:349-354).  This is
    what the north star's ">= 8x the reference single-GPU throughput" is measured against."""
    from oracle import spot_ref
    dev = mix_dev.device
    sdd = {k: torch.from_numpy(v).to(dev) for k, v in sd.items()}
    w = torch.tensor([1.0, 0.0], device=dev)
    offs = offsets[:n_cand]
    M, T = mix_dev.shape
    ar = torch.arange(T, device=dev).view(1, T)

    def roll(off):
        o = torch.tensor([0, *[int(v) for v in off]], device=dev).view(M, 1)      # one H2D per candidate, as :82
        return torch.gather(mix_dev, 1, (ar + o) % T)

    def run():
        outs = []
        for i in range(0, len(offs), batch):
            chunk = offs[i:i + batch]
            data = torch.stack([roll(o) for o in chunk])
            dn, mu, sg = spot_ref.normalize_input(data)
            y = spot_ref.spot_forward(sdd, cfg, dn, w.expand(len(chunk), 2))
            outs.append(spot_ref.unnormalize_input(y, mu, sg)[:, 0])
        r = torch.cat(outs).cpu().numpy()
        spot_ref.candidate_energies(r)
        return r

    run()                                                # warm-up: MIOpen kernel selection
    torch.cuda.synchronize()
    ts = []
    for _ in range(2):
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    del sdd
    torch.cuda.empty_cache()
    dt = min(ts)
    return {"value": round(len(offs) / dt, 2), "unit": "candidates/s", "dtype": "f32", "batch": batch,
            "candidates": len(offs), "seconds": round(dt, 3),
            "what": "reference-style single-GPU path: stock PyTorch-ROCm ops (MIOpen/rocBLAS) under the reference's "
                    "per-candidate roll loop, batch 128, D2H of all waveforms + host energy loop"}


def e2e_latency(model, sep_model, scene, dev):
    """Second half of BASELINE.json's metric: end-to-end localise+separate latency of ONE
    mixture through the whole pipeline (JointModel.forward: SRP-PHAT -> coarse -> fine ->
    clustering -> joint separation), per stage as JointModel.times
    (sep/training/JointModel/network.py:143-194), with device synchronisation at every stage
    boundary; geometry setup() excluded, as the reference's README says.  Seeded random weights:
    the candidate counts of the search are those a random network produces (close to the worst
    case of 30 coarse survivors), and the "talkers" it reports are not real talkers."""
    import io
    from contextlib import redirect_stdout
    from acousticswarms_speech_amd.joint import JointModel
    jm = JointModel(model, sep_model, device=dev)
    mix = torch.from_numpy(scene.mix)
    with redirect_stdout(io.StringIO()):
        t0 = time.perf_counter()
        jm.setup(scene.mic_positions, scene.speaker_range)
        setup_s = time.perf_counter() - t0
        for _ in range(3):                                # warm-up: workspace growth, gate cache, torch's
            jm.forward(mix)                               # caching allocator reaching its steady state
        runs = []
        for _ in range(3):
            out = jm.forward(mix)
            runs.append(list(jm.times))
        # one more forward with the in-library launch timer on: the SRP map kernel (K10) against its rooflines
        srp_rec = None
        try:
            from acousticswarms_speech_amd import native
            L = native.lib()
            L.asw_profile_enable(1)
            jm.forward(mix)
            torch.cuda.synchronize()
            buf = ctypes.create_string_buffer(1 << 16)
            native.check(L.asw_profile_report(buf, len(buf)))
            L.asw_profile_enable(0)
            pr = json.loads(buf.value.decode()).get("srp_map")
            if pr:
                srp_rec = {"ms": round(pr["ms"] / pr["launches"], 4),
                           "achieved_gbs": round(pr["bytes"] / (pr["ms"] * 1e-3) / 1e9, 1),
                           "frac_of_8tbs": round(pr["bytes"] / (pr["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 5),
                           "gop_per_s": round(pr["work"] / (pr["ms"] * 1e-3) / 1e9, 1),
                           "note": "steered-response accumulation: one sincos + complex MAC per (grid point, bin, pair, "
                                   "window), VALU-bound; its bytes (cross spectra + delays + map) are far from the HBM roof"}
        except Exception as exc:                          # diagnostic only
            srp_rec = {"error": f"{type(exc).__name__}: {exc}"}
    runs.sort(key=sum)
    times = runs[1]                                       # the median forward of three
    mp = jm.Mic_processor
    names = ["srp_phat", "coarse", "fine", "clustering", "joint_sep"]
    stages = {k: round(v * 1e3, 2) for k, v in zip(names, times)}
    rec = {"unit": "ms", "stages": stages, "protocol": "3 warm-up forwards, median (by total) of 3 measured forwards",
           "totals_of_the_three_runs": [round(sum(r) * 1e3, 2) for r in runs],
           "spot_calls": {"coarse": int(mp.big_spotforming_times), "fine": int(mp.spotforming_times)},
           "talkers_found": len(out[0]), "setup_excluded_s": round(setup_s, 2), "srp_map": srp_rec}
    if sep_model is None:
        # no separation network behind sep_model: the figure is localisation only, stage 5 is not part of it
        del stages["joint_sep"]
        rec["total_localize_only"] = round(sum(times[:4]) * 1e3, 2)
        rec["note"] = "localize-only latency (no joint separation network attached)"
    else:
        rec["total"] = round(sum(times) * 1e3, 2)
        rec["separated_rows"] = 0 if out[2] is None else int(out[2].shape[0])
    return rec


def launch_ranks(n: int) -> int:
    """`--gpus N` outside torchrun: start N ranks (one process per GPU) and wait.  This process
    has not touched the GPU (counting devices does not initialise it), so nothing is re-exec'ed
    after GPU init; the ranks are children and their exit status is ours."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


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
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra legs (reference-style GPU baseline, f32 mode, T=144000)")
    ap.add_argument("--precision", choices=["f32", "f16x3"], default="f16x3",
                    help="GEMM arithmetic: exact f32 MFMA, or f16x3 split-operand MFMA (105 dB SNR vs the reference)")
    ap.add_argument("--stub", action="store_true",
                    help="plumbing self-test without a GPU: gloo backend and a stand-in scorer (tests/test_bench_ranks.py); "
                         "the printed value is meaningless")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if not args.stub and torch.cuda.device_count() < args.gpus and os.environ.get("ASW_BENCH_BACKEND", "nccl") == "nccl":
            raise SystemExit(f"--gpus {args.gpus}: only {torch.cuda.device_count()} HIP devices visible")
        raise SystemExit(launch_ranks(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    if args.stub:
        return stub_main(args, world, rank)

    from acousticswarms_speech_amd import native
    from acousticswarms_speech_amd.config import FULL
    from acousticswarms_speech_amd.flops import flops_per_candidate
    from acousticswarms_speech_amd.scenes import make_scene, random_offsets
    from acousticswarms_speech_amd.shard import shard_bounds
    from acousticswarms_speech_amd.spot import SpotModel
    from acousticswarms_speech_amd.weights import make_spot_state_dict

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # ASW_BENCH_BACKEND=gloo is a REHEARSAL switch (tests / a one-GPU box): the ranks then share the visible
    # devices and the collectives run on host tensors.  The measured multi-GPU path is RCCL ("nccl").
    backend = os.environ.get("ASW_BENCH_BACKEND", "nccl")
    dev = torch.device("cuda", local_rank if backend == "nccl" else local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")       # where the exchanged tensors live
    close_ranks.device = coll_dev
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"{dist.get_world_size()} ranks joined, --gpus {args.gpus} asked")

    cfg, T = FULL, args.samples
    sd = make_spot_state_dict(cfg, seed=5)
    model = SpotModel(cfg, sd, batch_size=args.batch, precision=args.precision).to(dev)
    scene = make_scene(WORKLOAD_SEED, n_speakers=WORKLOAD_SPEAKERS, n_mics=7, T=T, reverb=True)
    mix_d = torch.from_numpy(scene.mix).to(dev)
    n_total = args.candidates * world
    offsets = random_offsets(7, n_total, 6, 140)
    off_d = torch.from_numpy(offsets).to(dev)
    b = shard_bounds(n_total, world)
    my_off = off_d[b[rank]:b[rank + 1]].contiguous()
    width = max(b[r + 1] - b[r] for r in range(world))

    def make_step(mdl, mix_dev):
        def step():
            _, en = mdl.shift_and_sep_device(mix_dev, my_off, strict=1, want_wave=False, want_energy=True, window=12000)
            if world > 1:
                buf = torch.zeros((width, 2), dtype=torch.float64, device=coll_dev)
                buf[:en.shape[0]] = en.to(coll_dev)
                out = torch.empty((world * width, 2), dtype=torch.float64, device=coll_dev)
                dist.all_gather_into_tensor(out, buf)          # the stage's one exchange
                return out
            return en
        return step

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        # + one event round trip: after a bare device synchronize the runtime intermittently starts the next
        # submission 20-25 ms late; the stage boundaries of the pipeline do the same
        ev = torch.cuda.Event()
        ev.record()
        ev.synchronize()

    def timed(step, steps, warmup):
        for _ in range(warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            last = step()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        assert torch.isfinite(last).all()
        return dt

    step = make_step(model, mix_d)
    L = native.lib()
    profile = not args.no_profile
    for _ in range(args.warmup):
        step()
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
        tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(last).all()
    from acousticswarms_speech_amd.ops import f16x3_overflow_count
    overflow = f16x3_overflow_count(reset=True)         # f16x3 range guard over warm-up + timed steps
    if overflow:
        raise SystemExit(f"f16x3 range guard: {overflow} activations beyond the fp16 range; rerun with --precision f32")

    # ---- extra legs (every rank takes part in the collectives; rank 0 reports) ---------------
    extras = {}
    if not args.no_extras:
        other = "f32" if args.precision == "f16x3" else "f16x3"
        model.set_precision(other)
        dt2 = timed(step, 2, 1)
        extras[f"precision_{other}"] = {"value": round(n_total * 2 / dt2, 2), "unit": "candidates/s", "steps": 2,
                                        "note": "same workload, other arithmetic mode of the GEMM-class layers"}
        model.set_precision(args.precision)
        if args.precision == "f16x3":
            # the OPTIONAL single-pass f16 mode (one MFMA per product): reduced precision, reported beside the
            # fp32-class headline, never as it
            model.set_precision("f16")
            dt4 = timed(step, 2, 1)
            extras["precision_f16_single_pass"] = {
                "value": round(n_total * 2 / dt4, 2), "unit": "candidates/s", "steps": 2,
                "note": "optional mode, one f16 MFMA per product: 47-48 dB from the reference's outputs (f16x3: 100+), "
                        "about 2 % of the search's hard decisions change on the seeded random weights "
                        "(profiles/r2/flip_rate_f16_single_pass.json); not comparable with the fp32-class headline"}
            model.set_precision(args.precision)
        if T != 144000:
            sc3 = make_scene(WORKLOAD_SEED, n_speakers=WORKLOAD_SPEAKERS, n_mics=7, T=144000, reverb=True)
            mix3 = torch.from_numpy(sc3.mix).to(dev)
            dt3 = timed(make_step(model, mix3), 2, 1)       # same internal batch (256: 140 GB of workspace; 686 cand/s at 64, 704 at 256)
            extras["T144000"] = {"value": round(n_total * 2 / dt3, 2), "unit": "candidates/s", "steps": 2,
                                 "internal_batch": args.batch,
                                 "gflop_per_candidate": round(flops_per_candidate(cfg, 144000)["total"] / 1e9, 2),
                                 "effective_tflops": round(n_total * 2 / dt3 * flops_per_candidate(cfg, 144000)["total"] / 1e12, 2),
                                 "note": "3 s at the reference's native 48 kHz"}
            del mix3

    line = None
    if rank == 0:
        fl = flops_per_candidate(cfg, T)
        value = n_total * args.steps / dt
        # dominant kernel = the GEMM-class instantiation with the largest measured time
        roof = None
        peak = PEAK_TFLOPS[args.precision]
        if prof:
            # the memory-bound passes (K1-K3 shift / normalise / preproc, GroupNorm + GLU, overlap-add,
            # K9 energies) against the HBM roofline: algorithmic bytes of the launch / its event time
            hbm = {k: {"ms_per_step": round(v["ms"] / args.steps, 4),
                       "gbytes_per_launch": round(v["bytes"] / v["launches"] / 1e9, 4),
                       "achieved_gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1),
                       "frac_of_8tbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
                   for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]) if v.get("bytes", 0) > 0 and v["work"] == 0}
            prof = {k: v for k, v in prof.items() if v["work"] > 0}
            name, rec = max(prof.items(), key=lambda kv: kv[1]["ms"])
            ach = rec["work"] / (rec["ms"] * 1e-3) / 1e12
            tot_ms = sum(r["ms"] for r in prof.values())
            tot_work = sum(r["work"] for r in prof.values())
            # HBM / fabric bytes per launch of the dominant kernel need separate rocprofv3 --pmc passes (the guide's
            # recipe): taken OFFLINE over this very command and kept in profiles/rN/traffic.json.  Only quoted when
            # the kernel and its per-launch batch are the ones of this run; otherwise null.
            traffic = traffic_src = None
            for rnd in ("r3", "r2"):
                try:
                    tj = json.load(open(os.path.join(ROOT, "profiles", rnd, "traffic.json")))
                    if tj["kernel"] == name and tj["batch"] == min(args.batch, args.candidates) and T == 48000:
                        traffic = int(tj["corrected_bytes_per_launch"])
                        traffic_src = {"source": f"offline: profiles/{rnd}/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                                 "passes of `python bench.py`, same batch)",
                                       "algorithmic_bytes": int(tj["algorithmic_bytes"]), "correction": tj["_correction"]}
                        break
                except (OSError, KeyError, ValueError):
                    pass
            roof = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": round(peak, 1),
                    "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic, "traffic_detail": traffic_src,
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
                                   for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
                    "hbm_bound_kernels": {"peak_gbs": PEAK_HBM_GBS, "unit": "GB/s", "kernels": hbm,
                                          "note": "algorithmic bytes (inputs read once + outputs written once) / HIP-event "
                                                  "time of the launch; srp_map is reported under e2e_latency.srp_map"}}
        e2e = None
        if not args.no_e2e and world == 1:
            e2e = e2e_latency(model, build_sep_model(dev, args.precision), scene, dev)
        if e2e is not None and not args.no_extras and T != 144000:
            # the same pipeline on the reference-native clip: 3 s at 48 kHz (sep/helpers/constants.py:8)
            try:
                sc3 = make_scene(WORKLOAD_SEED, n_speakers=WORKLOAD_SPEAKERS, n_mics=7, T=144000, reverb=True)
                extras["e2e_latency_T144000"] = e2e_latency(model, build_sep_model(dev, args.precision), sc3, dev)
                extras["e2e_latency_T144000"]["note"] = "3 s at the reference's native 48 kHz, same scene seed, same protocol"
            except Exception as exc:                      # an extra: reported, never fatal to the line
                extras["e2e_latency_T144000"] = {"error": f"{type(exc).__name__}: {exc}"}
            torch.cuda.empty_cache()
        cpu = refstyle = None
        if world == 1:                                   # baselines: reported on rank 0 at N = 1 only
            if args.cpu_sample > 0:
                cpu = cpu_baseline(cfg, sd, torch.from_numpy(scene.mix), offsets, args.cpu_sample)
            if not args.no_extras:
                refstyle = refstyle_gpu_baseline(cfg, sd, mix_d, offsets, min(256, n_total))
        line = {
            "metric": "TDoA candidates/sec (shift+normalise+spot forward+energies)", "value": round(value, 2),
            "unit": "candidates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            # BASELINE.md holds no published number for this metric; the measured reference-style GPU
            # figure of the same run is in `refstyle_gpu` (ratio = vs_refstyle_gpu)
            "vs_baseline": None, "dtype": DTYPE_NAME[args.precision], "data": "synthetic",
            "config": {"workload": "configs[2] scene: 5-speaker reverberant mixture (image sources), 7 mics, T=%d samples "
                                   "(BASELINE's '3 s @ 16 kHz' sample count; the reference pipeline and this scene "
                                   "generator run at 48 kHz, where it is a 1 s clip -- the reference-native 3 s is "
                                   "extras.T144000), fine-stage (Strict=1) candidate batch of one mixture, FULL spot net "
                                   "47.27 M params, seeded random weights" % T,
                       "candidates_per_gpu_per_step": args.candidates, "internal_batch": args.batch,
                       "gflop_per_candidate": round(fl["total"] / 1e9, 2), "parallelism": f"candidate-shard x{world}"},
            "effective_tflops": round(value * fl["total"] / 1e12, 2), "f16x3_overflow_count": overflow,
            "roofline": roof, "cpu_baseline": cpu, "refstyle_gpu": refstyle,
            "vs_refstyle_gpu": (round(value / refstyle["value"], 2) if refstyle else None),
            "extras": extras or None, "e2e_latency": e2e,
        }
    # ---- end-to-end latency on N ranks: the same pipeline with the candidates of the coarse and the fine stage
    # sharded over the ranks (shard.ShardedSpotModel: one energy all-gather per stage, one object gather of the
    # cluster heads); SRP-PHAT, the global clustering and the separation call are replica work.  Every rank takes
    # part.  The throughput line is complete before this starts: a watchdog turns a hung collective into a
    # reported error (rank 0 still prints the line) instead of a hung or failed run.
    rc = 0
    if world > 1 and not args.no_e2e:
        import threading
        done = threading.Event()

        def watchdog():
            if not done.wait(float(os.environ.get("ASW_E2E_TIMEOUT_S", "240"))):
                if rank == 0:
                    line["e2e_latency"] = {"error": "multi-rank end-to-end measurement timed out"}
                    print(json.dumps(line), flush=True)
                os._exit(EXIT_HUNG)                       # the line is out, the failure is in the status
        threading.Thread(target=watchdog, daemon=True).start()
        try:
            from acousticswarms_speech_amd.shard import ShardedSpotModel
            e2e_sharded = e2e_latency(ShardedSpotModel(model, device=coll_dev), build_sep_model(dev, args.precision), scene, dev)
            e2e_sharded["ranks"] = world
            e2e_sharded["note"] = ("candidates of the coarse and the fine stage sharded over the ranks; SRP-PHAT, global "
                                   "clustering and the separation call replicated")
        except Exception as exc:                          # the throughput line stands, the run still fails
            e2e_sharded = {"error": f"{type(exc).__name__}: {exc}"}
            rc = EXIT_E2E_FAILED
        done.set()
        if rank == 0:
            line["e2e_latency"] = e2e_sharded
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        rc = close_ranks(dist, rc)
    if rc:
        raise SystemExit(rc)


# exit statuses of a multi-rank run whose JSON line was still printed (the driver must see the failure)
EXIT_HUNG, EXIT_E2E_FAILED, EXIT_CLOSE_HUNG = 2, 3, 4


def close_ranks(dist, rc):
    """Closing barrier of a multi-rank run.  The line is out; a rank that failed or left early must not turn
    the barrier into a ten-minute hang, and must not look like success either: if the barrier does not
    complete within a minute every waiting rank exits with EXIT_CLOSE_HUNG."""
    import threading
    guard = threading.Timer(float(os.environ.get("ASW_CLOSE_TIMEOUT_S", "60")), lambda: os._exit(EXIT_CLOSE_HUNG))
    guard.daemon = True
    guard.start()
    dist.barrier()
    # every rank learns whether any rank failed: the launcher's status is then the same whichever rank it reads
    flag = torch.tensor([float(rc)], dtype=torch.float64, device=getattr(close_ranks, "device", "cpu"))
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    guard.cancel()
    dist.destroy_process_group()
    return int(flag.item())


def build_sep_model(dev, precision):
    """The joint separation network behind JointModel.sep_model (seeded random weights, like the
    spot model: checkpoints are not available offline)."""
    try:
        from acousticswarms_speech_amd.sep import SepModel
    except ImportError:
        return None
    from acousticswarms_speech_amd.config import SEP_FULL
    from acousticswarms_speech_amd.weights import make_sep_state_dict
    return SepModel(SEP_FULL, make_sep_state_dict(SEP_FULL, seed=9), precision=precision).to(dev)


def stub_main(args, world, rank):
    """--stub: the rank plumbing of this script (launch, shard, all-gather, max-over-ranks
    timing, one JSON line) with the gloo backend and a deterministic stand-in for the scorer.
    Used by the CPU test of `--gpus 2`; measures nothing."""
    import numpy as np
    from acousticswarms_speech_amd.scenes import random_offsets
    from acousticswarms_speech_amd.shard import ShardedScorer
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")

    def fake(_mix, offs):
        o = np.asarray(offs, dtype=np.float64)
        return np.stack([np.abs(o).sum(1) + 1.0, np.sqrt((o ** 2).sum(1) + 1.0)], axis=1)

    n_total = args.candidates * world
    offsets = random_offsets(7, n_total, 6, 140)
    scorer = ShardedScorer(fake)
    for _ in range(args.warmup):
        scorer.score(None, offsets, device="cpu")
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full = scorer.score(None, offsets, device="cpu")
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ok = bool(np.array_equal(full, fake(None, offsets)))
    line = None
    if rank == 0:
        line = {"metric": "stub", "value": round(n_total * args.steps / dt, 2), "unit": "candidates/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "stub (no GPU, gloo)",
                "config": {"workload": "rank plumbing self-test", "world": scorer.world},
                "all_ranks_hold_all_energies": ok}
    rc = 0 if ok else EXIT_E2E_FAILED
    if dist:
        # second phase, watched exactly like the sharded end-to-end measurement of the real run: the line is
        # complete before it starts; if a rank never joins the collective (ASW_STUB_FAULT_RANK: that rank skips
        # it) rank 0 still prints the line and the run ends with a non-zero status
        import threading
        done = threading.Event()

        def watchdog():
            if not done.wait(float(os.environ.get("ASW_E2E_TIMEOUT_S", "240"))):
                if rank == 0:
                    line["e2e_latency"] = {"error": "multi-rank end-to-end measurement timed out"}
                    print(json.dumps(line), flush=True)
                os._exit(EXIT_HUNG)
        threading.Thread(target=watchdog, daemon=True).start()
        fault = os.environ.get("ASW_STUB_FAULT_RANK")
        if fault is None or int(fault) != rank:
            scorer.score(None, offsets, device="cpu")
        done.set()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist:
        close_ranks.device = "cpu"
        rc = close_ranks(dist, rc)
    if rc:
        raise SystemExit(rc)


if __name__ == "__main__":
    main()
