"""TDoA hypercube value type at the boundary of the search (SURVEY.md §8 a-A).

Same attribute and method names as the reference ``Patch``
(sep/Traditional_SP/Patch_3D.py:3-93) so objects are interchangeable with anything
duck-typed on it (``spot_model.shift_and_sep`` reads ``.sample_offset``).  Written
vectorised; behaviour, including the in-place mutation done by ``check_out``, follows
the reference line by line.
"""
import numpy as np

SPEED_OF_SOUND = 343.0   # sep/helpers/constants.py:7
FS = 48000               # sep/helpers/constants.py:8


def pair_offsets(points: np.ndarray, mic_positions: np.ndarray, sound_speed: float = SPEED_OF_SOUND,
                 fs: float = FS) -> np.ndarray:
    """TDoA in samples of 3-D points [3,n] for every pair (m, 0), m = 1..M-1 -> [M-1, n].
    Same expression order as Patch_3D.py:31-33 / local_utils_3d.py:222-224."""
    X, Y, Z = points[0], points[1], points[2]
    d0 = (((X - mic_positions[0, 0]) ** 2 + (Y - mic_positions[0, 1]) ** 2
           + (Z - mic_positions[0, 2]) ** 2) ** 0.5) / sound_speed * fs
    rows = []
    for m in range(1, mic_positions.shape[0]):
        dm = (((X - mic_positions[m, 0]) ** 2 + (Y - mic_positions[m, 1]) ** 2
               + (Z - mic_positions[m, 2]) ** 2) ** 0.5) / sound_speed * fs
        rows.append(dm - d0)
    return np.array(rows)


class Patch(object):
    def __init__(self, sample_offset, width_list, area_points, peak_pos=None):
        self.sample_offset = sample_offset                 # ndarray [P] (not copied: Patch_3D.py:5)
        self.width_list = np.copy(width_list)              # copied: Patch_3D.py:6
        self.area_points = area_points                     # float64 [3,n] or None
        self.num_pair = sample_offset.shape[0]
        self.peak_pos = peak_pos
        self._centroid = None                              # (points array it was computed from, value)

    # ---- geometry ---------------------------------------------------------------
    def area_size(self):
        if self.area_points is None:
            return 0
        return self.area_points.shape[1]

    def center_pos(self):
        """peak position if known, else the centroid of the contained points (:19-26)."""
        if self.peak_pos is not None:
            return self.peak_pos
        ap = self.area_points
        if ap is None or ap.shape[1] == 0:
            return None
        c = self._centroid
        if c is None or c[0] is not ap:                    # the stage loops ask for the same centroid many times
            c = self._centroid = (ap, np.mean(ap, axis=1))
        return c[1].copy()

    def _inside(self, offsets: np.ndarray) -> np.ndarray:
        """Box test +-width/2 (+-1e-3) on offsets [P, n] (:36-37,44-45)."""
        off = np.asarray(self.sample_offset, dtype=np.float64).reshape(-1, 1)
        half = np.asarray(self.width_list, dtype=np.float64).reshape(-1, 1) / 2 + 1e-3
        return np.all((offsets >= off - half) & (offsets <= off + half), axis=0)

    def hyperbola_general_area(self, X, Y, Z, mic_position, sound_speed, fs):
        """1 where the 3-D point lies inside the hypercube (:28-38)."""
        offs = pair_offsets(np.stack([X, Y, Z]), mic_position, sound_speed, fs)
        return self._inside(offs).astype(int)

    def hyperbola_sample(self, offset):
        """1 where the pre-computed offsets [P,n] lie inside the hypercube (:40-47)."""
        return self._inside(np.asarray(offset)).astype(int)

    def check_gt(self, sample_offsets_gt):
        """True if a ground-truth TDoA column lies within width/2+1 on every pair (:50-66)."""
        gt = np.asarray(sample_offsets_gt)
        off = np.asarray(self.sample_offset).reshape(-1, 1)
        tol = np.asarray(self.width_list).reshape(-1, 1) / 2 + 1
        return bool(np.any(np.all(np.abs(gt[:self.num_pair] - off) <= tol, axis=0)))

    # ---- mutation -----------------------------------------------------------------
    def check_out(self, upper_bound_pairwise):
        """Pull offsets beyond the physical bound back inside by halving the width (:69-87).
        Mutates sample_offset and width_list in place, as the reference does."""
        for i in range(self.num_pair):
            ub = upper_bound_pairwise[i]
            while abs(self.sample_offset[i]) > ub and self.width_list[i] > 4:
                res = self.width_list[i]
                if self.sample_offset[i] > ub:
                    self.sample_offset[i] = self.sample_offset[i] - res / 4
                elif self.sample_offset[i] < -ub:
                    self.sample_offset[i] = self.sample_offset[i] + res / 4
                self.width_list[i] = res / 2

    def check_ready_Spotforming(self, MIN_TOLERANCE):
        """(:89-93)"""
        for i in range(self.num_pair):
            if self.width_list[i] > MIN_TOLERANCE:
                return False, i
        return True, -1
