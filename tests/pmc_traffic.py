"""Build profiles/<round>/traffic.json from the rocprofv3 --pmc passes over the bench command
(one pass per counter: FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES) and the kernel-stats CSV of
the same command.  Per-launch means of the dominant kernel at the bench's own internal batch,
corrected as MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE is in KB and counts 64 B per
fabric read request, wide coalesced streams issue 128-B requests -> x2; WRITE_SIZE (KB) is exact.
Usage: python3 tests/pmc_traffic.py <gpurun_out prefix> <out.json>"""
import csv
import json
import os
import sys
from collections import defaultdict

prefix, out_path = sys.argv[1], sys.argv[2]
DOM = "maskpath16p_kernel<256, 3>"      # the mask path in one launch (bypass + mask encoder + decoder taps)


def short(name):
    return name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(asw_")[0].split("(float")[0]


def counter_means(counter):
    acc = defaultdict(list)
    with open(os.path.join(f"{prefix}_pmc_{counter}", "pmc_counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


fetch, write, busy = counter_means("FETCH_SIZE"), counter_means("WRITE_SIZE"), counter_means("SQ_VALU_MFMA_BUSY_CYCLES")
stats = {}
with open(os.path.join(prefix, "bench_kernel_stats.csv")) as f:
    for r in csv.DictReader(f):
        stats[short(r["Name"])] = (float(r["AverageNs"]), int(r["Calls"]), float(r["Percentage"]))
B, F, K, E, C = 256, 3008, 2112, 2048, 64
# the latents never leave the chip: decoder-block output read once, weights (fp16 hi + lo of the mask encoder, the
# 48-tap bypass and the 64-row decoder matrix), the padded reference channel, 8 partial tap tensors of 33 columns
alg = {"activation_read": B * 48128 * C * 4, "weights_read": E * (K + 48 + 64) * 4, "reference_read": B * (48128 + 144) * 4,
       "partial_taps_write": (E // 256) * B * F * 33 * 4}
alg_total = sum(alg.values())
fr, fn = fetch[DOM]
wr, _ = write[DOM]
rec = {
    "_comment": "HBM / fabric traffic of the dominant kernel of `python bench.py` (fused mask path, internal batch 256, T=48000), "
                "from separate rocprofv3 --pmc passes over that very command; per-launch means.",
    "kernel": "maskpath16p<256,256,32>", "rocprof_name": DOM, "batch": B, "dispatches_averaged": fn,
    "FETCH_SIZE_KB_raw": fr, "WRITE_SIZE_KB": wr,
    "read_bytes_corrected_2x": int(fr * 1024 * 2), "write_bytes": int(wr * 1024),
    "corrected_bytes_per_launch": int(fr * 1024 * 2 + wr * 1024),
    "algorithmic_bytes": alg_total, "algorithmic_detail": alg,
    "ratio_to_algorithmic": round((fr * 1024 * 2 + wr * 1024) / alg_total, 3),
    "avg_launch_ms_kernel_stats": round(stats[DOM][0] / 1e6, 4),
    "fabric_TB_per_s": round((fr * 1024 * 2 + wr * 1024) / (stats[DOM][0] * 1e-9) / 1e12, 3),
    "_correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE x2 for wide coalesced streaming reads on gfx950, WRITE_SIZE "
                   "exact; the counters are fabric-side, Infinity-Cache hits are included",
}
mf = {}
for k, (v, n) in busy.items():
    if k in stats and stats[k][0] > 0:
        # busy SIMD-cycles / (kernel time x 2.4 GHz x 1024 SIMDs)
        mf[k] = {"mfma_busy_frac_nominal_clock": round(v / (stats[k][0] * 1e-9 * 2.4e9 * 1024), 4), "share_of_gpu_time_pct": stats[k][2]}
rec["mfma_busy"] = dict(sorted(mf.items(), key=lambda kv: -kv[1]["share_of_gpu_time_pct"])[:16])
with open(out_path, "w") as f:
    json.dump(rec, f, indent=1)
print(json.dumps({k: rec[k] for k in ("FETCH_SIZE_KB_raw", "WRITE_SIZE_KB", "corrected_bytes_per_launch", "algorithmic_bytes",
                                      "ratio_to_algorithmic", "avg_launch_ms_kernel_stats", "fabric_TB_per_s")}))
for k, v in rec["mfma_busy"].items():
    print(f"{k[:70]:70s} {v}")
