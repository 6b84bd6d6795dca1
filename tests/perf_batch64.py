"""BASELINE config 4 on ONE GPU (the driver owns the 8-GPU node): 64 five-speaker mixtures
(seeds 2000-2063), the complete search of each through shard.localize_batch -- which on N ranks
deals whole mixtures to ranks and all-gathers the per-mixture results.  Prints mixtures/s,
candidates/s inside the search and the stage split.  Diagnostic, not the headline bench."""
import io
import json
import os
import sys
import time
from contextlib import redirect_stdout

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acousticswarms_speech_amd.config import FULL  # noqa: E402
from acousticswarms_speech_amd.joint import JointModel  # noqa: E402
from acousticswarms_speech_amd.scenes import make_scene  # noqa: E402
from acousticswarms_speech_amd.shard import localize_batch  # noqa: E402
from acousticswarms_speech_amd.spot import SpotModel  # noqa: E402
from acousticswarms_speech_amd.weights import make_spot_state_dict  # noqa: E402


def main(n_mix=64, T=48000, concurrent=2, batch=256):
    model = SpotModel(FULL, make_spot_state_dict(FULL, 5), batch_size=batch, precision="f16x3").to("cuda")
    jm = JointModel(model, None, device="cuda")
    sc0 = make_scene(2000, 5, 7, T)
    # one array geometry for the whole batch (a recording session): setup() once, excluded like the reference's
    mixes = [torch.from_numpy(make_scene(2000 + i, 5, 7, T, mic_positions=sc0.mic_positions).mix) for i in range(n_mix)]
    with redirect_stdout(io.StringIO()):
        jm.setup(sc0.mic_positions, sc0.speaker_range)
        jm.forward(mixes[0])                                       # warm-up
    torch.cuda.synchronize()
    sampler = None
    if os.environ.get("ASW_SAMPLE_THREADS"):
        # poor man's profiler: where the search threads are, every 5 ms (innermost frame inside this package)
        import collections
        import threading
        hist, stop = collections.Counter(), threading.Event()

        def watch():
            me = threading.get_ident()
            while not stop.wait(0.005):
                for tid, fr in sys._current_frames().items():
                    if tid == me:
                        continue
                    inner, f = None, fr
                    while f is not None:
                        fn = f.f_code.co_filename
                        if "acousticswarms" in fn:
                            inner = (os.path.basename(fn), f.f_lineno, f.f_code.co_name)
                            break
                        f = f.f_back
                    if inner:
                        hist[inner] += 1
        sampler = threading.Thread(target=watch, daemon=True)
        sampler.start()
    prof = None
    if os.environ.get("ASW_CPROFILE"):                          # host functions of the search by own time
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    with redirect_stdout(io.StringIO()):
        out = localize_batch(jm, mixes, concurrent=concurrent)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if prof is not None:
        import pstats
        prof.disable()
        pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(28)
    if sampler is not None:
        stop.set()
        sampler.join()
        tot = sum(hist.values())
        for (fn, ln, name), n in hist.most_common(25):
            print(f"  {100.0 * n / tot:5.1f} %  {fn}:{ln} {name}", file=sys.stderr)
    cands = sum(r["spot_times"] for r in out)
    stages = np.array([r["times"] for r in out]).sum(0)
    print(json.dumps({"workload": "64 five-speaker mixtures, 7 mics, T=%d, full search each, 1 GPU" % T,
                      "concurrent_searches": concurrent, "internal_batch": batch,
                      "mixtures_per_s": round(n_mix / dt, 2), "s_total": round(dt, 2),
                      "spot_candidates": int(cands), "candidates_per_s_in_search": round(cands / dt, 1),
                      "batcher": (lambda st: None if st is None else {
                          "launches": st["launches"], "requests": st["requests"], "launch_rule": st["launch_rule"], "spot_gpu_s": round(st["spot_gpu_s"], 2), "enqueue_host_s": round(st["enqueue_host_s"], 2),
                          "wait_pending_s": round(st["wait_pending_s"], 2), "wait_device_s": round(st["wait_device_s"], 2),
                          "worker_s": round(st["worker_s"], 2),
                          "median_launch": int(np.median(st["launch_sizes"])), "max_launch": int(max(st["launch_sizes"]))})(
                          getattr(localize_batch, "last_stats", None) if concurrent > 1 else None),
                      "talkers_found_mean": round(float(np.mean([len(r["names"]) for r in out])), 2),
                      "stage_seconds": {k: round(float(v), 2) for k, v in
                                        zip(["srp_phat", "coarse", "fine", "clustering", "joint_sep"], stages)}}))


if __name__ == "__main__":
    main(concurrent=int(sys.argv[1]) if len(sys.argv) > 1 else 2, n_mix=int(sys.argv[2]) if len(sys.argv) > 2 else 64)
