"""Build profiles/<round>/ summaries from the rocprofv3 passes of tests/collect_profiles.sh (ROCm 7.2 writes rocpd
SQLite databases): the kernel-stats table of the bench command, and per kernel (and launch grid) the per-launch
means of FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES from the separate --pmc passes, corrected as
MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE is in KB and counts 64 B per fabric read request while
wide coalesced streams issue 128-B requests -> x2; WRITE_SIZE (KB) is exact.
Usage: python3 tests/pmc_traffic.py <gpurun_out/r3prof> <profiles/r3>"""
import csv
import json
import os
import re
import sqlite3
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)


def short(name):
    n = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", n)


def stats_table(db):
    con = sqlite3.connect(db)
    rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    mn = dict(con.execute("select name, min(duration) from kernels group by name").fetchall())
    mx = dict(con.execute("select name, max(duration) from kernels group by name").fetchall())
    return [(n, c, t * 1e3, a * 1e3, p, mn.get(n, 0), mx.get(n, 0)) for n, c, t, a, p in rows]


def counter_groups(db, counter):
    """{(kernel, grid_size): (mean value, mean duration ns, n)}"""
    con = sqlite3.connect(db)
    acc = defaultdict(list)
    for name, grid, val, dur in con.execute(
            "select kernel_name, grid_size, value, duration from counters_collection where counter_name = ?", (counter,)):
        acc[(short(name), int(grid))].append((float(val), float(dur)))
    return {k: (sum(v for v, _ in vs) / len(vs), sum(d for _, d in vs) / len(vs), len(vs)) for k, vs in acc.items()}


st = stats_table(os.path.join(src, "stats", "bench_results.db"))
with open(os.path.join(dst, "bench_kernel_stats_f16x3_batch256.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in sorted(st, key=lambda r: -r[2]):
        w.writerow([r[0], r[1], int(r[2]), round(r[3], 1), round(r[4], 4), r[5], r[6]])

out = {}
for tag, sub in (("", "pmc_"), ("per_layer_residual_kernels_", "pmc_old_")):
    fetch = counter_groups(os.path.join(src, sub + "FETCH_SIZE", "pmc_results.db"), "FETCH_SIZE")
    write = counter_groups(os.path.join(src, sub + "WRITE_SIZE", "pmc_results.db"), "WRITE_SIZE")
    busy = counter_groups(os.path.join(src, "pmc_SQ_VALU_MFMA_BUSY_CYCLES", "pmc_results.db"), "SQ_VALU_MFMA_BUSY_CYCLES") \
        if not tag else {}
    rec = {}
    for key in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, (0, 0, 0))[0] * fetch.get(k, (0, 0, 0))[2])):
        name, grid = key
        if not any(t in name for t in ("convgemm", "resconv", "resstack", "downconv", "maskpath", "attention", "gn_glu", "preproc",
                                       "energy", "overlap_add", "add_layernorm", "shift_stats")):
            continue
        fr, fdur, fn = fetch.get(key, (0.0, 0.0, 0))
        wr, wdur, wn = write.get(key, (0.0, 0.0, 0))
        e = {"launches_averaged": fn, "FETCH_SIZE_KB_raw": round(fr, 1), "WRITE_SIZE_KB": round(wr, 1),
             "read_bytes_corrected_2x": int(fr * 2048), "write_bytes": int(wr * 1024),
             "corrected_bytes_per_launch": int(fr * 2048 + wr * 1024),
             "avg_launch_ms_in_pmc_pass": round(fdur / 1e6, 4),
             "fabric_TB_per_s": round((fr * 2048 + wr * 1024) / max(fdur, 1.0) / 1e3, 3)}
        if key in busy:
            e["mfma_busy_frac_nominal_clock"] = round(busy[key][0] / (busy[key][1] * 1e-9 * 2.4e9 * 1024), 4)
        rec[f"{name} grid={grid}"] = e
    out[tag + "per_kernel"] = rec
out["_correction"] = ("MI355X_MICROARCH.md HBM section: FETCH_SIZE x2 for wide coalesced streaming reads on gfx950, WRITE_SIZE "
                      "exact; the counters are fabric-side, Infinity-Cache hits are included")
out["_command"] = "python3 bench.py --no-e2e --no-extras --cpu-sample 0 --steps 3 --warmup 1 --no-profile (tests/collect_profiles.sh)"
with open(os.path.join(dst, "pmc_per_kernel.json"), "w") as f:
    json.dump(out, f, indent=1)
for tag in ("per_kernel", "per_layer_residual_kernels_per_kernel"):
    print("==", tag)
    for k, v in list(out[tag].items())[:40]:
        print(f"{k[:92]:92s} n={v['launches_averaged']:3d} rd {v['read_bytes_corrected_2x'] / 1e9:7.3f} GB wr {v['write_bytes'] / 1e9:7.3f} GB "
              f"{v['avg_launch_ms_in_pmc_pass']:8.3f} ms {v['fabric_TB_per_s']:6.2f} TB/s busy {v.get('mfma_busy_frac_nominal_clock', '')}")

# ---- traffic.json: the dominant kernel of the bench line (read by bench.py) + the fused residual pair against
# the two per-layer launches it replaces
B, F, K, E, C = 256, 3008, 2112, 2048, 64
pk = out["per_kernel"]
dom_key = next(k for k in pk if k.startswith("maskpath16p_kernel"))
dom = pk[dom_key]
alg = {"activation_read": B * 48128 * C * 4, "weights_read": E * (K + 48 + 64) * 4, "reference_read": B * (48128 + 144) * 4,
       "partial_taps_write": (E // 256) * B * F * 33 * 4}
old = out["per_layer_residual_kernels_per_kernel"]


def total(rec, pred):
    return sum(v["corrected_bytes_per_launch"] for k, v in rec.items() if pred(k))


# full-rate level (M = 48 128): the plain pair of the encoder and the GroupNorm + GLU pair of the decoder
pair_new = pk[next(k for k in pk if k.startswith("resstack64_kernel<2, 4, 2, 1, false, false") and "grid=14745600" in k)]
pair_glu = pk[next(k for k in pk if k.startswith("resstack64_kernel<2, 4, 2, 1, false, true") and "grid=14745600" in k)]
o_d1 = old[next(k for k in old if k.startswith("resconv16_kernel<128, 64, 2, 2, 1, 4, false, false") and "grid=24641536" in k)]
o_d1g = old[next(k for k in old if k.startswith("resconv16_kernel<128, 64, 2, 2, 1, 4, false, true") and "grid=24641536" in k)]
o_d7 = old[next(k for k in old if k.startswith("resconv16_kernel<128, 64, 2, 2, 1, 4, true, false") and "grid=24772608" in k)]
rec = {
    "_comment": "HBM / fabric traffic of the dominant kernel of `python bench.py` (fused mask path, internal batch 256, T=48000), "
                "from separate rocprofv3 --pmc passes over that very command; per-launch means.  Round 3.",
    "kernel": "maskpath16p<256,256,32>", "rocprof_name": dom_key, "batch": B, "dispatches_averaged": dom["launches_averaged"],
    "FETCH_SIZE_KB_raw": dom["FETCH_SIZE_KB_raw"], "WRITE_SIZE_KB": dom["WRITE_SIZE_KB"],
    "read_bytes_corrected_2x": dom["read_bytes_corrected_2x"], "write_bytes": dom["write_bytes"],
    "corrected_bytes_per_launch": dom["corrected_bytes_per_launch"],
    "algorithmic_bytes": sum(alg.values()), "algorithmic_detail": alg,
    "ratio_to_algorithmic": round(dom["corrected_bytes_per_launch"] / sum(alg.values()), 3),
    "fabric_TB_per_s": dom["fabric_TB_per_s"], "mfma_busy_frac_nominal_clock": dom.get("mfma_busy_frac_nominal_clock"),
    "_correction": out["_correction"],
    "fused_residual_pair_full_rate_level": {
        "what": "layers 0 + 1 (dilation 1 and 7) of a 64-channel DilatedResidualSequence at M = 48 128, 256 candidates: "
                "ONE resstack64 launch against the two per-layer resconv16 launches (ASW_NO_RESSTACK=1), corrected bytes",
        "algorithmic_bytes_pair": 2 * B * 48128 * C * 4,
        "encoder_pair": {"fused": pair_new["corrected_bytes_per_launch"],
                         "two_launches": o_d1["corrected_bytes_per_launch"] + o_d7["corrected_bytes_per_launch"],
                         "ratio": round(pair_new["corrected_bytes_per_launch"] /
                                        (o_d1["corrected_bytes_per_launch"] + o_d7["corrected_bytes_per_launch"]), 3)},
        "decoder_pair_groupnorm_glu_on_load": {"fused": pair_glu["corrected_bytes_per_launch"],
                                               "two_launches": o_d1g["corrected_bytes_per_launch"] + o_d7["corrected_bytes_per_launch"],
                                               "ratio": round(pair_glu["corrected_bytes_per_launch"] /
                                                              (o_d1g["corrected_bytes_per_launch"] + o_d7["corrected_bytes_per_launch"]), 3)}},
}
with open(os.path.join(dst, "traffic.json"), "w") as f:
    json.dump(rec, f, indent=1)
print(json.dumps({k: rec[k] for k in ("corrected_bytes_per_launch", "algorithmic_bytes", "ratio_to_algorithmic")}),
      json.dumps(rec["fused_residual_pair_full_rate_level"]["encoder_pair"]),
      json.dumps(rec["fused_residual_pair_full_rate_level"]["decoder_pair_groupnorm_glu_on_load"]))
