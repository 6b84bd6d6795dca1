"""SRP-PHAT pruning stage: geometry tables on the host, steered-response map on the GPU.

Mirrors ``SRP_PHAT`` (sep/Traditional_SP/SRP_Prunning.py:101-643; SURVEY.md §8 a-O, a-P,
a-Q) with the same method names used by the search (``reset``,
``SRP_Map_WINDOW_new``, ``local_source_adaptive``).  Differences by design:

* the one-off 3-D -> TDoA map (``Map_3D_TDoA`` + BFS ``search_cluster``, :277-344; 44 s
  of pure-Python loops in the reference) is computed with vectorised numpy and a sparse
  connected-components pass -- same clusters, same order;
* the [G,198,21] complex128 steering table (:368-381,230-246; 1.16 GB) is never built:
  the HIP map kernel regenerates exp(j w dtau) from the [G,M] propagation delays;
* the map itself (``SRP_Map_WINDOW_torch``, :387-434) runs in libasw_hip.so
  (csrc/srp_kernels.hip) -- there is no host fallback.
"""
import numpy as np
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components

from .patch import Patch

ERR_TOLERANCE = 0.2        # SRP_Prunning.py:17


class GridCluster(object):
    """A set of voxels sharing one quantised TDoA vector (SRP_Prunning.py:68-96)."""
    __slots__ = ("sample_offset", "grids", "index")

    def __init__(self, sample_offset, grids, index):
        self.sample_offset = sample_offset      # int [M-1]
        self.grids = grids                      # float [n,3]
        self.index = index                      # int [n,3] voxel indices

    def cluster_size(self):
        return len(self.index)

    def center_pos(self):
        return np.mean(self.grids, axis=0)


def _offsets_within(offsets, center, width):
    """All pairs within +-width/2 of ``center`` (hyperbola_offset / hyperbola_area_sample,
    SRP_Prunning.py:19-39); offsets [..., P].  The same comparisons as the all-pairs mask,
    evaluated pair by pair on the survivors only (the cube is a tiny part of the table)."""
    c = np.asarray(center, dtype=np.float64)
    lo, hi = c - width / 2, c + width / 2
    P = offsets.shape[-1]
    flat = offsets.reshape(-1, P)
    v = flat[:, 0]
    idx = np.flatnonzero((v >= lo[0]) & (v <= hi[0]))
    for p in range(1, P):
        if idx.shape[0] == 0:
            break
        v = flat[idx, p]
        idx = idx[(v >= lo[p]) & (v <= hi[p])]
    mask = np.zeros(flat.shape[0], dtype=bool)
    mask[idx] = True
    return mask.reshape(offsets.shape[:-1])


class SRPPhat(object):
    def __init__(self, mic_pos, freq_bins, Range_spk, C=343, FS=16000, n_fft=1024, grid_size=0.06,
                 grid_size_z=0.1, sample_resolution=4, threshold=0.03, WIDTH=8, device=None):
        self.device = device
        self.C, self.FS, self.n_fft = C, FS, n_fft
        self.freq_bins = np.asarray(freq_bins)
        self.mic_pos = np.asarray(mic_pos, dtype=np.float64)
        self.num_mic = self.mic_pos.shape[0]
        self.mic_center = self.mic_pos.mean(0)
        self.sample_resolution = sample_resolution
        self.WIDTH = WIDTH
        self.threshold = threshold
        self.Range_spk = Range_spk
        r = Range_spk
        self.x_grids = np.arange(r[0], r[1], grid_size)
        self.y_grids = np.arange(r[2], r[3], grid_size)
        self.z_grids = np.arange(r[4], r[5], grid_size_z)
        self.Lx, self.Ly, self.Lz = len(self.x_grids), len(self.y_grids), len(self.z_grids)
        gx, gy = np.meshgrid(self.x_grids, self.y_grids, indexing="ij")
        self.dis_matrix = np.sqrt((gx - self.mic_center[0]) ** 2 + (gy - self.mic_center[1]) ** 2) + 1e-8
        self.Axis_range = [[r[0], r[1]], [r[2], r[3]], [r[4], r[5]]]

        # 5 cm and 1 cm lookup grids with their TDoA vectors (:149-170)
        self.Pos_5, self.Offset_5 = self._lookup_grid(0.05)
        self.Pos_1, self.Offset_1 = self._lookup_grid(0.01)
        # pair-major copies for the cube scans (asw_cube_select_planes streams one pair's plane)
        self._planes_5 = np.ascontiguousarray(np.moveaxis(self.Offset_5, 3, 0))
        self._planes_1 = np.ascontiguousarray(np.moveaxis(self.Offset_1, 3, 0))

        keepout = 0.2                                       # :174-180
        self.array_border = [self.mic_pos[:, 0].min() - keepout, self.mic_pos[:, 1].min() - keepout,
                             self.mic_pos[:, 0].max() + keepout, self.mic_pos[:, 1].max() + keepout]
        self._map_3d_tdoa()

        # propagation delays used by the steering term; mic z is ignored and the point's z is
        # taken absolute, exactly as generate_mod_vector does (:368-381)
        dx = self.grids[:, None, 0] - self.mic_pos[None, :, 0]
        dy = self.grids[:, None, 1] - self.mic_pos[None, :, 1]
        self.tau = np.sqrt(dx ** 2 + dy ** 2 + self.grids[:, None, 2] ** 2) / self.C      # [G,M] seconds
        self.omega = 2 * np.pi * FS * self.freq_bins / n_fft
        ii, jj = np.triu_indices(self.num_mic, k=1)          # row-major upper triangle == mask_triu order
        self.pair_i, self.pair_j = ii.astype(np.int32), jj.astype(np.int32)
        self.SRP_map = np.zeros(self.grids.shape[0], dtype=np.float32)
        self.MAX_POWER, self.Min_POWER = -100, 0.0
        self._dev = None

    # ---- one-off geometry ---------------------------------------------------------
    def _lookup_grid(self, step):
        r = self.Range_spk
        xx, yy, zz = np.arange(r[0], r[1], step), np.arange(r[2], r[3], step), np.arange(r[4], r[5], 0.1)
        X, Y, Z = np.meshgrid(xx, yy, zz)
        pos = np.stack((X, Y, Z), axis=3)
        d0 = np.linalg.norm(pos - self.mic_pos[0, :], axis=3) / self.C * self.FS
        offs = [np.linalg.norm(pos - self.mic_pos[i, :], axis=3) / self.C * self.FS - d0
                for i in range(1, self.num_mic)]
        return pos, np.stack(offs, axis=3)

    def _valid_mask(self):
        b = self.array_border
        inside = ((self.x_grids[:, None] > b[0]) & (self.x_grids[:, None] < b[2])
                  & (self.y_grids[None, :] > b[1]) & (self.y_grids[None, :] < b[3]))
        return np.broadcast_to(~inside[:, :, None], (self.Lx, self.Ly, self.Lz))

    def _map_3d_tdoa(self):
        """Quantised TDoA per voxel, then merge 26-connected voxels with identical TDoA
        (Map_3D_TDoA + search_cluster, :277-344).  Cluster order = order in which the
        reference's (ix,iy,iz) scan first meets each cluster."""
        Lx, Ly, Lz = self.Lx, self.Ly, self.Lz
        X, Y, Z = np.meshgrid(self.x_grids, self.y_grids, self.z_grids, indexing="ij")
        pos = np.stack([X, Y, Z], axis=-1)
        d = np.linalg.norm(pos[..., None, :] - self.mic_pos[None, None, None], axis=-1)
        off = (d[..., 1:] - d[..., :1]) / self.C * self.FS
        q = np.round(off / self.sample_resolution).astype(int) * self.sample_resolution
        valid = self._valid_mask()
        n = Lx * Ly * Lz
        flat_valid = valid.reshape(n)
        qf = q.reshape(n, -1)
        idx3 = np.arange(n).reshape(Lx, Ly, Lz)
        rows, cols = [], []
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dz in (-1, 0, 1):
                    if (dx, dy, dz) <= (0, 0, 0):
                        continue                        # each undirected neighbour pair once
                    sa = (slice(max(0, -dx), Lx - max(0, dx)), slice(max(0, -dy), Ly - max(0, dy)),
                          slice(max(0, -dz), Lz - max(0, dz)))
                    sb = (slice(max(0, dx), Lx - max(0, -dx)), slice(max(0, dy), Ly - max(0, -dy)),
                          slice(max(0, dz), Lz - max(0, -dz)))
                    a, b = idx3[sa].ravel(), idx3[sb].ravel()
                    same = flat_valid[a] & flat_valid[b] & np.all(qf[a] == qf[b], axis=1)
                    rows.append(a[same])
                    cols.append(b[same])
        rows, cols = np.concatenate(rows), np.concatenate(cols)
        graph = coo_matrix((np.ones(len(rows), dtype=np.int8), (rows, cols)), shape=(n, n))
        _, lab = connected_components(graph, directed=False)
        vid = np.flatnonzero(flat_valid)
        vlab = lab[vid]
        _, first = np.unique(vlab, return_index=True)        # first voxel (scan order) of each label
        order = np.argsort(first)
        rank_of = np.empty(vlab.max() + 1, dtype=np.int64)
        rank_of[np.unique(vlab)[order]] = np.arange(len(order))
        cid = rank_of[vlab]
        G = len(order)
        self.POWER_MAP = np.zeros((Lx, Ly, Lz))
        self.POWER_INDEX = np.zeros((Lx, Ly, Lz), dtype=int)
        self.POWER_INDEX.reshape(n)[vid] = cid
        self._valid_flat = vid
        self._valid_cid = cid
        posf = pos.reshape(n, 3)
        srt = np.argsort(cid, kind="stable")
        bounds = np.searchsorted(cid[srt], np.arange(G + 1))
        ix3 = np.stack(np.unravel_index(vid, (Lx, Ly, Lz)), axis=1)
        self.clusters = []
        centers = np.zeros((G, 3))
        for g in range(G):
            mem = srt[bounds[g]:bounds[g + 1]]
            pts = posf[vid[mem]]
            self.clusters.append(GridCluster(qf[vid[mem[0]]], pts, ix3[mem]))
            centers[g] = pts.mean(axis=0)
        self.grids = centers
        self.SRP_times = G

    # ---- per-mixture -----------------------------------------------------------------
    def reset(self):
        self.SRP_map = np.zeros(self.grids.shape[0], dtype=np.float32)

    def _device_tables(self, dev):
        import torch
        if self._dev is None or self._dev["dev"] != dev:
            nb = len(self.freq_bins)
            nb_pad = ((nb + 63) // 64) * 64
            k = self.freq_bins.astype(np.float64)[:, None]
            nn = np.arange(self.n_fft, dtype=np.float64)[None, :]
            ang = 2 * np.pi * k * nn / self.n_fft
            tw = np.zeros((2 * nb_pad, self.n_fft), dtype=np.float32)
            tw[:nb] = np.cos(ang)
            tw[nb_pad:nb_pad + nb] = -np.sin(ang)
            self._dev = {"dev": dev, "nb_pad": nb_pad,
                         "tw": torch.from_numpy(tw).to(dev),
                         "tau": torch.from_numpy(np.ascontiguousarray(self.tau)).to(dev),
                         "omega": torch.from_numpy(np.ascontiguousarray(self.omega, dtype=np.float64)).to(dev),
                         "pi": torch.from_numpy(self.pair_i).to(dev), "pj": torch.from_numpy(self.pair_j).to(dev)}
        return self._dev

    def SRP_Map_WINDOW_new(self, signal, window=36000, tol=1e-8):
        """Steered-response map, maximum over half-overlapped windows (:383-434)."""
        import torch
        from . import native
        dev = torch.device(self.device if self.device is not None else "cuda")
        if dev.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("SRP map runs only on the MI355X (no host fallback)")
        M, T = signal.shape
        assert self.mic_pos.shape[0] == M
        step = window // 2
        n_win = 0
        for j in range(0, T // step - 1):                  # :398-402
            if j * step + window > T:
                break
            n_win += 1
        t = self._device_tables(dev)
        sig = torch.as_tensor(signal, dtype=torch.float32).to(dev)
        Tp = (T + 3) // 4 * 4
        if Tp != T:
            sig = torch.nn.functional.pad(sig, (0, Tp - T))
        sig = sig.contiguous()
        hop = self.n_fft // 4
        # torch.ops.asw.srp_phat_map: DFT-GEMM + PHAT + cross-spectra, then the steered map
        out = native.torch_ops().srp_phat_map(sig, t["tw"], t["pi"], t["pj"], t["tau"], t["omega"], int(window), int(step),
                                              int(n_win), int(self.n_fft), int(hop), float(tol))
        self.SRP_map = out.cpu().numpy()
        self._finish_map()

    def set_map(self, srp_map):
        """Install a map computed elsewhere (tests / oracle) and refresh the derived state."""
        self.SRP_map = np.asarray(srp_map, dtype=np.float32)
        self._finish_map()

    def _finish_map(self):
        self.MAX_POWER = float(np.amax(self.SRP_map))
        self.Min_POWER = float(np.amin(self.SRP_map))
        # fill_powermap_torch (:347-357): every voxel takes its cluster's value
        self.POWER_MAP.reshape(-1)[self._valid_flat] = self.SRP_map[self._valid_cid]

    # ---- peak picking -------------------------------------------------------------------
    def find_valid_peak_new(self, rato=4):
        """Cluster ids of local maxima (5x5 in x,y; dz in {-1,0}) above the distance-weighted
        adaptive threshold, or of any voxel above `rato` times it (:500-544)."""
        thr = self.threshold[0] * self.MAX_POWER
        thr = min(max(thr, self.threshold[1]), self.threshold[2])
        thr2 = thr * rato
        print("Adaptive threshold: ", self.MAX_POWER, thr, thr2)
        pm = self.POWER_MAP
        NX, NY, NZ = pm.shape
        core = pm[2:-2, 2:-2, 1:-1]
        t1 = (thr * (0.9 + 1 / self.dis_matrix))[2:-2, 2:-2, None]
        t2 = (thr2 * (1 + 1 / self.dis_matrix))[2:-2, 2:-2, None]
        is_max = np.ones(core.shape, dtype=bool)
        for dx in range(-2, 3):
            for dy in range(-2, 3):
                for dz in range(-1, 1):
                    if dx == 0 and dy == 0 and dz == 0:
                        continue
                    is_max &= core >= pm[2 + dx:NX - 2 + dx, 2 + dy:NY - 2 + dy, 1 + dz:NZ - 1 + dz]
        sel = (is_max & (core > t1) & (core <= t2)) | (core > t2)
        peaks, seen = [], set()
        for ix, iy, iz in np.transpose(np.nonzero(sel)):
            g = int(self.POWER_INDEX[ix + 2, iy + 2, iz + 1])
            if g not in seen:
                seen.add(g)
                peaks.append(g)
        return peaks

    def hyperbola_area_init(self, sample_offsets, width):
        """3-D points (1 cm grid) whose TDoA lies inside the cube, found through the 5 cm
        grid's bounding box (:41-61).  Returns [3,n] or None."""
        pts = self._cube_points(self.Pos_5, self._planes_5, sample_offsets, width)
        if pts.shape[0] == 0:
            return None
        ax = self.Axis_range
        x0, x1 = max(ax[0][0], pts[:, 0].min() - 0.05), min(ax[0][1], pts[:, 0].max() + 0.05)
        y0, y1 = max(ax[1][0], pts[:, 1].min() - 0.05), min(ax[1][1], pts[:, 1].max() + 0.05)
        xi0, xi1 = int(np.floor((x0 - ax[0][0]) / 0.01)), int(np.ceil((x1 - ax[0][0]) / 0.01))
        yi0, yi1 = int(np.floor((y0 - ax[1][0]) / 0.01)), int(np.ceil((y1 - ax[1][0]) / 0.01))
        return self._cube_points(self.Pos_1, self._planes_1, sample_offsets, width, yi0, yi1, xi0, xi1).T

    @staticmethod
    def _cube_points(pos, planes, center, width, y0=0, y1=None, x0=0, x1=None):
        """Points [n,3] of the lookup grid (sub-box [y0,y1) x [x0,x1)) whose TDoA lies within
        +-width/2 of ``center`` on every pair: the native box scan (csrc/search_host.cpp,
        same comparisons and order as the boolean mask of ``_offsets_within``) over the
        pair-major table ``planes`` [P, ny, nx, nz]."""
        from ctypes import byref, c_int64, c_void_p
        from . import native
        P, ny, nx, nz = planes.shape
        y1 = ny if y1 is None else min(y1, ny)
        x1 = nx if x1 is None else min(x1, nx)
        y0, x0 = max(y0, 0), max(x0, 0)
        if y1 <= y0 or x1 <= x0:
            return np.zeros((0, 3))
        c = np.asarray(center, dtype=np.float64)
        lo, hi = np.ascontiguousarray(c - width / 2), np.ascontiguousarray(c + width / 2)
        cap = (y1 - y0) * (x1 - x0) * nz
        idx = np.empty(cap, dtype=np.int32)
        n = c_int64()
        native.check(native.lib().asw_cube_select_planes(c_void_p(planes.ctypes.data), ny, nx, nz, P, y0, y1, x0, x1,
                                                  c_void_p(lo.ctypes.data), c_void_p(hi.ctypes.data),
                                                  c_void_p(idx.ctypes.data), cap, byref(n)))
        return pos.reshape(-1, 3)[idx[:n.value]]

    def local_source_adaptive(self):
        """Greedy peak -> width-8 hypercube list (:547-643): strongest peak first, each new
        cube trimmed on its high side against the cubes already accepted, peaks covered by
        a cube are not revisited."""
        peak_index = self.find_valid_peak_new()
        print("peak_index: ", len(peak_index))
        peaks = self.SRP_map[peak_index]
        peaks_pos = self.grids[peak_index]
        self.peaks, self.peaks_pos = peaks, peaks_pos
        peaks_sample = np.array([self.clusters[i].sample_offset for i in peak_index])
        visited = np.zeros_like(peaks)
        P = self.num_mic - 1
        W = self.WIDTH
        accepted, peak_candidate = [], []
        for k in np.argsort(-1 * peaks):
            if visited[k] >= 1:
                continue
            center = peaks_sample[k]
            peak_candidate.append(peaks_pos[k, :])
            occupy = np.ones((P, W))
            for p in accepted:
                delta = p.sample_offset - center
                lo1 = delta - p.width_list / 2
                hi1 = delta + p.width_list / 2
                d1 = int(round((lo1 - W / 2).max()))
                d2 = int(round((hi1 + W / 2).min()))
                if d1 >= 0 or d2 <= 0:
                    continue                                   # disjoint in some pair
                # d1 < 0 always holds here, so only the high side is ever trimmed (a-Q)
                if W + d1 < 0:
                    occupy[:, :] = 0
                else:
                    occupy[:, W + d1:] = 0
            widths, offs, dead = [], [], False
            for i in range(P):
                on = np.where(occupy[i])[0]
                if on.shape[0] == 0:
                    dead = True
                    break
                widths.append(on.shape[0])
                offs.append(int(round(center[i] + (on[0] + on[-1] - W + 1) / 2)))
            if dead:
                continue
            visited += _offsets_within(peaks_sample, center, W + ERR_TOLERANCE).astype(int)
            widths, offs = np.array(widths), np.array(offs)
            # the reference uses the FIRST pair's trimmed width for every pair here (:629)
            area = self.hyperbola_area_init(offs, widths[0] + ERR_TOLERANCE)
            if area is None or area.shape[-1] == 0:
                continue
            accepted.append(Patch(offs, widths, area, peaks_pos[k, :]))
        print("SRP-PHAT candidate number: ", len(accepted))
        self.peak_candidate = np.array(peak_candidate)
        return accepted
