"""Search-stage fixture generators (g6-g8, g10): drive the REFERENCE's Mic_Array /
SRP_PHAT / search_area in this container and record what they produce.  Called from
make_golden.py (which installs the import stubs first).

The reference gets its STFT framing and librosa RMS/split from OUR restatements
(tests/golden/thirdparty_restated.py) because those packages are absent: these fixtures
pin everything except that third-party framing.
"""
import io
import os
from contextlib import redirect_stdout

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))

# reduced region of interest so the reference's pure-Python geometry init takes seconds
ROI = [-1.6, 1.6, 0.35, 2.75, 0.1, 0.7]
SCENE_SEED, N_SPK, T = 41, 3, 48000


def _save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def scene():
    """Three talkers inside the reduced ROI, free field, 1 s at 48 kHz."""
    from acousticswarms_speech_amd.scenes import make_scene
    sc = make_scene(SCENE_SEED, N_SPK, 7, T)
    return sc


def scene_in_roi():
    """Deterministic scene with talkers re-drawn inside ROI (make_scene's ROI is larger)."""
    from acousticswarms_speech_amd import scenes
    rng = np.random.default_rng(SCENE_SEED)
    mics, _ = scenes.desk_mics(rng, 7)
    spk = np.array([[-0.9, 1.4, 0.45], [0.7, 2.1, 0.30], [1.1, 0.9, 0.55]])
    mix = np.zeros((7, T))
    for s in range(N_SPK):
        x = scenes._speech_like(rng, T, 48000) * (0.5 - 0.1 * s)
        d = np.linalg.norm(spk[s] - mics, axis=1)
        for m in range(7):
            mix[m] += scenes._frac_delay(x, (d[m] - d[0]) / scenes.SPEED_OF_SOUND * 48000) * min(1.0, 1.0 / d[m])
    mix += 1e-3 * rng.standard_normal(mix.shape)
    return mics, spk, mix.astype(np.float32)


def _patch_arrays(patches):
    return dict(offsets=np.stack([np.asarray(p.sample_offset, dtype=np.float64) for p in patches]),
                widths=np.stack([np.asarray(p.width_list, dtype=np.float64) for p in patches]),
                peaks=np.stack([np.asarray(p.peak_pos, dtype=np.float64) if p.peak_pos is not None
                                else np.full(3, np.nan) for p in patches]),
                npoints=np.array([p.area_size() for p in patches]),
                centroid=np.stack([p.area_points.mean(1) if p.area_size() else np.full(3, np.nan)
                                   for p in patches]))


def _mic_array(mics):
    from sep.Mic_Array import Mic_Array
    with redirect_stdout(io.StringIO()):
        return Mic_Array(mics, Spk_Range=ROI)


def g7_g8_g6():
    """g7: geometry tables + SRP map; g8: peak list + width-8 patches; g6: search_area children."""
    from sep.helpers.local_utils_3d import search_area
    mics, spk, mix = scene_in_roi()
    ma = _mic_array(mics)
    node = ma.SRP_node
    with redirect_stdout(io.StringIO()):
        patches, _ = ma.Apply_SRP_PHAT(torch.from_numpy(mix))
    srp = node.SRP_map.numpy().astype(np.float64)
    _save("g7_srp_map", roi=np.array(ROI), mics=mics, speakers=spk, grids=node.grids,
          cluster_offsets=np.stack([c.sample_offset for c in node.clusters]),
          cluster_sizes=np.array([c.cluster_size() for c in node.clusters]),
          power_index=node.POWER_INDEX.astype(np.int32), srp_map=srp,
          max_power=np.float64(node.MAX_POWER), min_power=np.float64(node.Min_POWER))
    with redirect_stdout(io.StringIO()):
        peak_index = node.find_valid_peak_new()
    _save("g8_srp_patches", peak_index=np.array(peak_index), **_patch_arrays(patches))
    out = {}
    for k in range(min(4, len(patches))):
        with redirect_stdout(io.StringIO()):
            kids = search_area([patches[k]], mics, ma.upper_bound_pairwise)
        for key, v in _patch_arrays(kids).items():
            if key != "peaks":
                out[f"p{k}_{key}"] = v
        out[f"p{k}_parent_offset_after"] = np.asarray(patches[k].sample_offset, dtype=np.float64)
        out[f"p{k}_parent_width_after"] = np.asarray(patches[k].width_list, dtype=np.float64)
    _save("g6_search_area", n=np.int64(min(4, len(patches))), **out)


def g10():
    """End-to-end stage trace of the reference's Mic_Array driven by the surrogate scorer."""
    from tests.golden.surrogate import SurrogateSpot
    mics, spk, mix = scene_in_roi()
    ma = _mic_array(mics)
    spot = SurrogateSpot()
    mix_t = torch.from_numpy(mix)
    log = io.StringIO()
    with redirect_stdout(log):
        p1, _ = ma.Apply_SRP_PHAT(mix_t)
        srp_offsets = np.stack([np.asarray(p.sample_offset, dtype=np.float64) for p in p1])
        p2 = ma.Spotform_Big_Patch(mix_t, p1, spot)
        kept = [int(np.flatnonzero([q is p for q in p1])[0]) for p in p2]
        pairs = ma.Spotform_Small_Patch_Parallel(mix_t, p2, spot)
        audio, final, spot_times, _ = ma.Clustering_new(pairs)
    _save("g10_stage_trace", n_srp=np.int64(len(p1)), srp_offsets=srp_offsets, kept=np.array(kept),
          calls=np.array(spot.calls), n_pairs=np.int64(len(pairs)),
          pair_names=np.array([p[3] for p in pairs]), pair_power=np.array([p[2] for p in pairs]),
          pair_audio_offset=np.stack([np.asarray(p[4]["audio_offset"], dtype=np.float64) for p in pairs]),
          pair_loc_offset=np.stack([np.asarray(p[4]["localization_offset"], dtype=np.float64) for p in pairs]),
          pair_center=np.stack([p[0].center_pos() for p in pairs]),
          final_names=np.array([p[3] for p in final]),
          final_center=np.stack([p[0].center_pos() for p in final]) if final else np.zeros((0, 3)),
          spot_times=np.int64(spot_times), speakers=spk, mics=mics,
          final_audio_l2=np.array([float(np.linalg.norm(a)) for a in audio]))
    print(f"g10: SRP {len(p1)} -> coarse {len(p2)} -> pairs {len(pairs)} -> final {len(final)}; calls {spot.calls}")
    for p in final:
        c = p[0].center_pos()
        print("   final", p[3], np.round(c, 3), "nearest talker", np.round(np.linalg.norm(spk - c, axis=1).min(), 3))


GENERATORS = {"g7": g7_g8_g6, "g10": g10}
