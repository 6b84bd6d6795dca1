"""Host-side candidate bookkeeping of the coarse and fine Spotforming stages
(SURVEY.md §8 a-K, a-L): hypercube subdivision and coarse-stage thresholding.

Follows sep/helpers/local_utils_3d.py:212-388 and the constants of
sep/helpers/constants.py:23-47.  All tensor arithmetic of the candidates (shift,
network, energies) happens on the GPU behind ``spot_model``; this module only builds
candidate lists and applies the reference's thresholds to the returned energies.
"""
import numpy as np

from .hostdsp import max_avg_power
from .patch import FS as FS_F, SPEED_OF_SOUND as SPEED_OF_SOUND_F, Patch, pair_offsets

# sep/helpers/constants.py:31-41
MIN_AREA = 400
MIN_WIDTH = 3
MIN_TOLERANCE = 4
MAX_BIG_PATCH = 30
MIN_WIDTH_REQUIRED = 2
USE_RELATIVE_SPOT_POWER = False
SPOT_POWER_THRESHOLD1 = 0.008
SPOT_POWER_THRESHOLD2 = 0.01
SI_SNR_POWER_THRESHOLD = 4e-3
INIT_WIDTH = 8


def _halve(patch, samples, dim):
    """The two children of ``patch`` along pair ``dim`` (offset -/+ w/4, width w/2) with the
    points / offsets that fall inside each (local_utils_3d.py:273-307).  Empty children
    are dropped from the returned lists; sizes are returned for both."""
    kids, kid_samples, sizes = [], [], []
    for sign in (-1.0, 1.0):
        off = np.copy(patch.sample_offset)
        off[dim] += sign * patch.width_list[dim] / 4
        w = np.copy(patch.width_list)
        w[dim] /= 2
        child = Patch(off, w, None)
        inside = child.hyperbola_sample(samples) == 1
        n = int(np.sum(inside))
        sizes.append(n)
        if n > 0:
            child.area_points = patch.area_points[:, inside]
            kids.append(child)
            kid_samples.append(samples[:, inside])
    return kids, kid_samples, sizes, w[dim]


def binary_area_divide_width(patch, samples0, mic_positions, upper_bound_pairwise):
    """One subdivision step (local_utils_3d.py:248-335).  Returns
    (True, [children], [their offsets]) or (False, patch, samples0) when the patch is
    final.  Splits along the pair whose halves hold the most balanced point counts."""
    if upper_bound_pairwise is not None:
        patch.check_out(upper_bound_pairwise)          # mutates the caller's patch (a-L)
    widths = patch.width_list
    if (np.amax(widths) / 2 <= MIN_WIDTH_REQUIRED) and patch.area_size() <= MIN_AREA:
        return False, patch, samples0
    best, best_samples, best_diff = None, None, 2500000
    wide_seen = False
    last_kids = None
    for i in range(patch.sample_offset.shape[0]):
        if widths[i] / 2 < MIN_WIDTH:
            continue
        kids, kid_samples, sizes, half_w = _halve(patch, samples0, i)
        last_kids = kids
        diff = abs(sizes[0] - sizes[1])
        if half_w > MIN_WIDTH_REQUIRED:
            if not wide_seen or diff < best_diff:
                best, best_samples, best_diff = kids, kid_samples, diff
            wide_seen = True
        elif not wide_seen and diff < best_diff:
            best, best_samples, best_diff = kids, kid_samples, diff
    if best is None or len(last_kids) == 0:
        return False, patch, samples0
    return True, best, best_samples


def search_area(patch_list, mic_positions, upper_bound_pairwise):
    """Breadth-first subdivision of ONE coarse patch into fine hypercubes
    (local_utils_3d.py:212-246): patch_list = [coarse_patch].  Runs in the native library
    (csrc/search_host.cpp, same float64 arithmetic, ~8x faster); ``search_area_py`` below is
    the same algorithm in numpy, kept as the readable statement the tests compare against.
    Like every other entry of the path this one needs the built library (``native.lib()``
    raises otherwise)."""
    from . import native
    L = native.lib()
    import ctypes
    from ctypes import byref, c_int, c_void_p
    root = patch_list[0]
    P = root.sample_offset.shape[0]
    pts = np.ascontiguousarray(root.area_points, dtype=np.float64)
    mic = np.ascontiguousarray(mic_positions, dtype=np.float64)
    off = np.ascontiguousarray(root.sample_offset, dtype=np.float64).copy()
    wid = np.ascontiguousarray(root.width_list, dtype=np.float64).copy()
    ub = None if upper_bound_pairwise is None else np.ascontiguousarray(upper_bound_pairwise, dtype=np.float64)
    nc = c_int()
    po, pw, pc, pi = c_void_p(), c_void_p(), c_void_p(), c_void_p()
    native.check(L.asw_search_area(c_void_p(pts.ctypes.data), pts.shape[1], c_void_p(mic.ctypes.data), mic.shape[0],
                                   c_void_p(off.ctypes.data), c_void_p(wid.ctypes.data),
                                   None if ub is None else c_void_p(ub.ctypes.data), SPEED_OF_SOUND_F, FS_F,
                                   byref(nc), byref(po), byref(pw), byref(pc), byref(pi)))
    try:
        n = nc.value
        offs = np.ctypeslib.as_array(ctypes.cast(po, ctypes.POINTER(ctypes.c_double)), shape=(max(n, 1), P))[:n].copy()
        wids = np.ctypeslib.as_array(ctypes.cast(pw, ctypes.POINTER(ctypes.c_double)), shape=(max(n, 1), P))[:n].copy()
        cnt = np.ctypeslib.as_array(ctypes.cast(pc, ctypes.POINTER(ctypes.c_int)), shape=(max(n, 1),))[:n].copy()
        tot = int(cnt.sum())
        idx = np.ctypeslib.as_array(ctypes.cast(pi, ctypes.POINTER(ctypes.c_int)), shape=(max(tot, 1),))[:tot].copy()
    finally:
        for q in (po, pw, pc, pi):
            L.asw_free(q)
    # check_out mutates the caller's patch in place (same dtype handling as the numpy path)
    root.sample_offset[...] = off.astype(root.sample_offset.dtype, copy=False)
    root.width_list[...] = wid.astype(root.width_list.dtype, copy=False)
    kids, pos = [], 0
    for k in range(n):
        sel = idx[pos:pos + cnt[k]]
        pos += cnt[k]
        if n == 1 and cnt[k] == pts.shape[1] and np.array_equal(offs[k], off) and np.array_equal(wids[k], wid):
            kids.append(root)                          # not subdivided: the reference returns the patch itself
        else:
            kids.append(Patch(offs[k].astype(root.sample_offset.dtype), wids[k].astype(root.width_list.dtype),
                              root.area_points[:, sel]))
    return kids


def search_area_py(patch_list, mic_positions, upper_bound_pairwise):
    """numpy statement of search_area (kept as the readable reference of the native version)."""
    root = patch_list[0]
    frontier = [root]
    frontier_samples = [pair_offsets(root.area_points, mic_positions)]
    done = []
    while frontier:
        nxt, nxt_samples = [], []
        for p, smp in zip(frontier, frontier_samples):
            more, kids, kid_samples = binary_area_divide_width(p, smp, mic_positions, upper_bound_pairwise)
            if more:
                nxt.extend(kids)
                nxt_samples.extend(kid_samples)
            else:
                done.append(kids)
        frontier, frontier_samples = nxt, nxt_samples
    return done


def stage_energies(spot_model, mix_data, patch_list, strict):
    """(power, power2) of every mean-removed candidate output.  Uses the device
    reduction when the model offers it (``shift_and_score``); otherwise reproduces the
    reference's host loop on the returned waveforms (local_utils_3d.py:349-354)."""
    if hasattr(spot_model, "shift_and_score"):
        en = spot_model.shift_and_score(mix_data, patch_list, Strict=strict, keep_waveforms=False)
        return en[:, 0], en[:, 1]
    sep = spot_model.shift_and_sep(mix_data, patch_list, Strict=strict)
    p, pw = [], []
    for i in range(sep.shape[0]):
        x = sep[i, :] - np.mean(sep[i, :])
        p.append(np.sum(x ** 2))
        pw.append(max_avg_power(x))
    return np.array(p), np.array(pw)


def binary_search_baseline(mix_data, spot_model, patch_list, mic_positions):
    """Coarse stage scoring (local_utils_3d.py:339-388): run the spot model with the relaxed
    window on every SRP patch, weight the windowed RMS by (1 + distance to mic 0), keep
    those above SPOT_POWER_THRESHOLD1 in descending windowed-RMS order, at most
    MAX_BIG_PATCH.  Returns (kept patches, powers_with_dis, relative_threshold*1.2)."""
    _, powers_win = stage_energies(spot_model, mix_data, patch_list, 0)
    with_dis = []
    for i, p in enumerate(patch_list):
        c = p.center_pos()
        d = np.linalg.norm(c - mic_positions[0]) if c.shape[0] == 3 else 4
        with_dis.append(powers_win[i] * (d + 1))
    order = np.argsort(-1 * np.array(powers_win))
    if USE_RELATIVE_SPOT_POWER:
        thr = min([0.4 * max(with_dis), SPOT_POWER_THRESHOLD1])
    else:
        thr = SPOT_POWER_THRESHOLD1
    kept = []
    for i in order:
        if with_dis[i] < thr:
            continue
        if len(kept) >= MAX_BIG_PATCH:
            print("warning too many patch remaining, only keep the best 30")
            break
        kept.append(patch_list[i])
    return kept, with_dis, thr * 1.2
