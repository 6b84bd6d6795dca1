"""Host-side search logic (Patch, geometry tables, peak picking, subdivision, stage
orchestration, clustering) against fixtures recorded from the reference's own code
(tests/golden/make_golden_search.py).  CPU only: the SRP map is injected from fixture g7
and the spot network is replaced by the model-free surrogate, so what is pinned here is
exactly the host logic.  Integer / index results are compared bit-exact."""
import io
from contextlib import redirect_stdout

import numpy as np
import pytest

from acousticswarms_speech_amd.mic_array import (MicArray, check_sisnr_win, find_merge_center, weight_mean_pos)
from acousticswarms_speech_amd.patch import Patch
from acousticswarms_speech_amd.search import search_area
from tests.golden.make_golden_search import ROI, scene_in_roi
from tests.golden.surrogate import SurrogateSpot


@pytest.fixture(scope="module")
def world(golden):
    mics, spk, mix = scene_in_roi()
    g7 = golden("g7_srp_map")
    np.testing.assert_array_equal(mics, g7["mics"])
    with redirect_stdout(io.StringIO()):
        ma = MicArray(mics, Spk_Range=ROI)
    return ma, mics, spk, mix, g7


def test_patch_semantics():
    p = Patch(np.array([10.0, -3.0]), [8, 8], None)
    offs = np.array([[6.0, 14.0, 14.002, 10.0], [1.0, -7.0, 0.0, -8.0]])
    assert p.hyperbola_sample(offs).tolist() == [1, 1, 0, 0]
    assert p.center_pos() is None and p.area_size() == 0
    q = Patch(np.array([40.0, -50.0, 3.0]), np.array([8.0, 8.0, 8.0]), None)
    q.check_out(np.array([36.0, 36.0, 36.0]))
    # pulled inside by quarter widths while halving, stops at width 4 (Patch_3D.py:69-87)
    assert q.sample_offset.tolist() == [38.0, -48.0, 3.0] and q.width_list.tolist() == [4.0, 4.0, 8.0]
    assert q.check_ready_Spotforming(4) == (False, 2)
    assert Patch(np.array([1.0, 2.0]), [4, 4], None).check_gt(np.array([[3.5], [-0.9]]))
    assert not Patch(np.array([1.0, 2.0]), [4, 4], None).check_gt(np.array([[4.5], [2.0]]))


def test_geometry_tables_match_reference(world):
    """Map_3D_TDoA + BFS clustering (SRP_Prunning.py:277-344): same clusters, same order."""
    ma, _, _, _, g7 = world
    node = ma.SRP_node
    assert node.grids.shape == g7["grids"].shape
    np.testing.assert_array_equal(np.stack([c.sample_offset for c in node.clusters]), g7["cluster_offsets"])
    np.testing.assert_array_equal(np.array([c.cluster_size() for c in node.clusters]), g7["cluster_sizes"])
    np.testing.assert_array_equal(node.POWER_INDEX, g7["power_index"])
    np.testing.assert_allclose(node.grids, g7["grids"], rtol=0, atol=1e-12)


def test_peaks_and_patches_match_reference(world, golden):
    """find_valid_peak_new + local_source_adaptive (SRP_Prunning.py:500-643) on the
    reference's own SRP map."""
    ma, _, _, _, g7 = world
    g8 = golden("g8_srp_patches")
    node = ma.SRP_node
    node.set_map(g7["srp_map"])
    assert abs(node.MAX_POWER - float(g7["max_power"])) < 1e-6
    with redirect_stdout(io.StringIO()):
        peaks = node.find_valid_peak_new()
        patches = node.local_source_adaptive()
    assert peaks == g8["peak_index"].tolist()
    assert len(patches) == g8["offsets"].shape[0]
    np.testing.assert_array_equal(np.stack([p.sample_offset for p in patches]), g8["offsets"])
    np.testing.assert_array_equal(np.stack([p.width_list for p in patches]), g8["widths"])
    np.testing.assert_array_equal(np.array([p.area_size() for p in patches]), g8["npoints"])
    np.testing.assert_allclose(np.stack([p.peak_pos for p in patches]), g8["peaks"], atol=1e-12)
    np.testing.assert_allclose(np.stack([p.area_points.mean(1) for p in patches]), g8["centroid"], atol=1e-9)


def test_search_area_children_match_reference(world, golden):
    ma, mics, _, _, g7 = world
    g6 = golden("g6_search_area")
    node = ma.SRP_node
    node.set_map(g7["srp_map"])
    with redirect_stdout(io.StringIO()):
        patches = node.local_source_adaptive()
    for k in range(int(g6["n"])):
        kids = search_area([patches[k]], mics, ma.upper_bound_pairwise)
        np.testing.assert_array_equal(np.stack([c.sample_offset for c in kids]), g6[f"p{k}_offsets"])
        np.testing.assert_array_equal(np.stack([c.width_list for c in kids]), g6[f"p{k}_widths"])
        np.testing.assert_array_equal(np.array([c.area_size() for c in kids]), g6[f"p{k}_npoints"])
        np.testing.assert_allclose(np.stack([c.area_points.mean(1) for c in kids]), g6[f"p{k}_centroid"], atol=1e-9)
        # check_out mutates the caller's patch (local_utils_3d.py:250)
        np.testing.assert_array_equal(patches[k].sample_offset, g6[f"p{k}_parent_offset_after"])
        np.testing.assert_array_equal(patches[k].width_list, g6[f"p{k}_parent_width_after"])


def test_cluster_helpers(golden):
    g = golden("g9_sisdr")
    wins = [[-1.0, -3.0], [-3.0, -8.0], [-1.5, -6.9], [5.0], [-2.0, -2.0]]
    assert [check_sisnr_win(w) for w in wins] == g["check_win"].tolist()
    assert [check_sisnr_win(w, SISNR_THRESHOLD=-1, SISNR_THRESHOLD2=-5) for w in wins] == g["check_win2"].tolist()
    patches = [Patch(o, [4] * 6, None, pk) for o, pk in zip(g["wm_offsets"], g["wm_peaks"])]
    pos, off = weight_mean_pos(patches, list(g["wm_powers"]), [0, 1, 2, 3, 4])
    np.testing.assert_allclose(pos, g["wm_pos"], rtol=1e-12)
    np.testing.assert_allclose(off, g["wm_off"], rtol=1e-12)


def test_stage_trace_matches_reference(world, golden):
    """All four stages end to end with the surrogate scorer: candidate counts and order,
    coarse survivors, fine-stage clusters, merged offsets, final talkers (fixture g10)."""
    import torch
    ma, mics, spk, mix, g7 = world
    g = golden("g10_stage_trace")
    node = ma.SRP_node
    node.SRP_Map_WINDOW_new = lambda signal, window=36000: node.set_map(g7["srp_map"])   # no GPU here
    spot = SurrogateSpot()
    mix_t = torch.from_numpy(mix)
    with redirect_stdout(io.StringIO()):
        p1, _ = ma.Apply_SRP_PHAT(mix_t)
        assert len(p1) == int(g["n_srp"])
        np.testing.assert_array_equal(np.stack([p.sample_offset for p in p1]), g["srp_offsets"])
        p2 = ma.Spotform_Big_Patch(mix_t, p1, spot)
        kept = [int(np.flatnonzero([q is p for q in p1])[0]) for p in p2]
        assert kept == g["kept"].tolist()
        pairs = ma.Spotform_Small_Patch_Parallel(mix_t, p2, spot)
        audio, final, spot_times, _ = ma.Clustering_new(pairs)
    assert spot.calls == [tuple(c) for c in g["calls"].tolist()]
    assert [p[3] for p in pairs] == g["pair_names"].tolist()
    np.testing.assert_allclose(np.array([p[2] for p in pairs]), g["pair_power"], rtol=1e-5)
    np.testing.assert_array_equal(np.stack([p[4]["audio_offset"] for p in pairs]), g["pair_audio_offset"])
    np.testing.assert_allclose(np.stack([p[4]["localization_offset"] for p in pairs]), g["pair_loc_offset"],
                               rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(np.stack([p[0].center_pos() for p in pairs]), g["pair_center"], atol=1e-6)
    assert [p[3] for p in final] == g["final_names"].tolist()
    np.testing.assert_allclose(np.stack([p[0].center_pos() for p in final]), g["final_center"], atol=1e-6)
    assert spot_times == int(g["spot_times"])
    np.testing.assert_allclose([float(np.linalg.norm(a)) for a in audio], g["final_audio_l2"], rtol=1e-5)


def test_find_merge_center_fallback():
    mics = np.array([[0, 0, 0.02], [0.3, 0.1, 0.02], [-0.3, 0.1, 0.02]], dtype=float)
    area = np.array([[1.0, 1.01], [1.0, 1.0], [0.3, 0.3]])
    far = np.array([200.0, -200.0])
    p = find_merge_center(far, area, mics, np.array([1.0, 1.0, 0.3]))
    assert p.area_points is None and p.center_pos().tolist() == [1.0, 1.0, 0.3]


def test_dense_tdoa_lattice_invariants():
    """16-mic dense lattice (BASELINE config 5, SRP bypassed): every grid point belongs to exactly
    one cube, cube centres are multiples of the width, members lie within +-width/2."""
    from acousticswarms_speech_amd.dense_grid import dense_tdoa_candidates, roi_grid
    from acousticswarms_speech_amd.patch import pair_offsets
    from acousticswarms_speech_amd.scenes import desk_mics
    mics = np.asarray(desk_mics(np.random.default_rng(5), 16)[0])
    roi = [-1.0, 1.0, 0.4, 2.0, 0.1, 0.5]
    for width in (2, 4):
        offs, counts, patches = dense_tdoa_candidates(mics, roi, width=width, step=0.05)
        n_pts = roi_grid(roi, 0.05).shape[1]
        assert offs.shape[1] == 15 and counts.sum() == n_pts and len(patches) == len(offs)
        assert np.all(offs % width == 0)
        assert len(np.unique(offs, axis=0)) == len(offs)
        for p, c in list(zip(patches, counts))[::97]:
            t = pair_offsets(p.area_points, mics)
            assert p.area_points.shape[1] == c
            assert np.all(np.abs(t - p.sample_offset[:, None]) <= width / 2 + 1e-9)
            assert np.all(p.hyperbola_sample(t) == 1)
    assert len(offs) > 1000          # thousands of candidates even in this small ROI
