"""Dense TDoA-lattice candidate enumeration (BASELINE.json config 5: "16-mic array, dense
TDoA grid, SRP-PHAT bypassed").

With the pruning stage bypassed every hypercube of the integer TDoA lattice that some point
of the region of interest falls into is a candidate.  The cubes follow the conventions of the
search stages: offsets are relative to microphone 0 in samples (``pair_offsets``,
sep/helpers/local_utils_3d.py:221-225), a cube of width w is centred on a multiple of w in
every pair dimension and owns the grid points whose TDoA rounds to it (the membership test
of ``Patch.hyperbola_sample``, sep/Traditional_SP/Patch_3D.py:40-47, without its 1e-3 slack,
so every point belongs to exactly one cube).  The reference has no function for this
configuration; it only defines the stress workload, so there is no fixture to pin and the
enumeration is checked by its own invariants (tests/test_search_host.py).
"""
import numpy as np

from .patch import Patch, pair_offsets


def roi_grid(roi, step):
    """[3,n] float64 grid points of the ROI [x0,x1,y0,y1,z0,z1] at ``step`` metres."""
    ax = [np.arange(roi[2 * k], roi[2 * k + 1] + 1e-9, step) for k in range(3)]
    X, Y, Z = np.meshgrid(*ax, indexing="ij")
    return np.stack([X.ravel(), Y.ravel(), Z.ravel()])


def dense_tdoa_candidates(mic_positions, roi, width=2, step=0.02, chunk=1 << 20, with_points=True):
    """All non-empty width-``width`` TDoA cubes of the ROI.

    Returns (offsets int64 [N, M-1] sorted lexicographically, counts [N], patches) where
    ``patches`` is a list of ``Patch`` (sample_offset, width_list = width, area_points = the
    grid points inside) or None when ``with_points`` is False (tens of thousands of candidates:
    the scorer only needs the offsets)."""
    mic = np.asarray(mic_positions, dtype=np.float64)
    P = mic.shape[0] - 1
    pts = roi_grid(roi, step)
    n = pts.shape[1]
    cells = np.empty((n, P), dtype=np.int64)
    for lo in range(0, n, chunk):                            # bounded temporaries: chunk x P float64
        hi = min(n, lo + chunk)
        cells[lo:hi] = np.rint(pair_offsets(pts[:, lo:hi], mic) / width).astype(np.int64).T
    key = np.ascontiguousarray(cells).view([("", np.int64)] * P).ravel()
    order = np.argsort(key, kind="stable")
    sk = key[order]
    first = np.flatnonzero(np.concatenate([[True], sk[1:] != sk[:-1]]))
    counts = np.diff(np.concatenate([first, [n]]))
    offsets = cells[order[first]] * width
    patches = None
    if with_points:
        patches = []
        for k, (a, c) in enumerate(zip(first, counts)):
            patches.append(Patch(offsets[k].astype(np.float64), np.full(P, float(width)), pts[:, order[a:a + c]]))
    return offsets, counts, patches
