// search_host.cpp -- host-side hypercube subdivision of the fine Spotforming stage, native.
//
// Replaces search_area / binary_area_divide_width (sep/helpers/local_utils_3d.py:212-335) and
// Patch.check_out / hyperbola_sample (sep/Traditional_SP/Patch_3D.py:40-47,69-87): one coarse
// +-4-sample hypercube is split breadth-first, one pair dimension at a time (the split whose
// halves hold the most balanced point counts), until every width is <= 4 and the cell holds
// <= 400 points.  Pure float64 host arithmetic in the reference's expression order, so the
// children are bit-identical to the Python statement (tests/test_search_host.py, fixture g6);
// ~30x faster than the numpy version (8 ms -> 0.3 ms per coarse patch), which matters
// because this runs between the coarse and the fine GPU calls of every mixture.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "asw_common.h"

namespace {

constexpr int MIN_AREA = 400;              // sep/helpers/constants.py:31-35
constexpr double MIN_WIDTH = 3;
constexpr double MIN_WIDTH_REQUIRED = 2;

struct Cell {
  std::vector<double> off, width;
  std::vector<int> idx;                    // indices into the coarse patch's point list
  bool filtered = false;                   // every point of idx passed the box test of THIS box
};

bool check_out(Cell& c, const double* ub, int P) {   // Patch_3D.py:69-87; true if the box moved
  bool moved = false;
  for (int i = 0; i < P; ++i) {
    while (std::fabs(c.off[i]) > ub[i] && c.width[i] > 4) {
      const double res = c.width[i];
      if (c.off[i] > ub[i]) c.off[i] = c.off[i] - res / 4;
      else if (c.off[i] < -ub[i]) c.off[i] = c.off[i] + res / 4;
      c.width[i] = res / 2;
      moved = true;
    }
  }
  return moved;
}

// the same test on one pair only: enough for a child whose parent's points already passed the
// parent's box (the halves differ from it in that pair alone)
inline bool inside_dim(const double* s, long n_pts, int p_idx, int i, const double* off, const double* width) {
  const double half = width[i] / 2 + 1e-3;
  const double v = s[(long)i * n_pts + p_idx];
  return v >= off[i] - half && v <= off[i] + half;
}

// box test +-width/2 (+-1e-3) on every pair (Patch_3D.py:40-47)
inline bool inside(const double* s, long n_pts, int p_idx, const double* off, const double* width, int P) {
  for (int i = 0; i < P; ++i) {
    const double half = width[i] / 2 + 1e-3;
    const double v = s[(long)i * n_pts + p_idx];
    if (!(v >= off[i] - half && v <= off[i] + half)) return false;
  }
  return true;
}

}  // namespace

extern "C" int asw_search_area(const double* points, int n_pts, const double* mic, int M, double* offset,
                               double* width, const double* ub, double sound_speed, double fs, int* n_children,
                               double** child_offset, double** child_width, int** child_count, int** child_index) {
  ASW_CHECK_ARG(points && mic && offset && width && n_children && child_offset && child_width && child_count &&
                    child_index,
                "search_area: null pointer");
  ASW_CHECK_ARG(n_pts >= 0 && M >= 2, "search_area: bad shape");
  const int P = M - 1;
  // TDoA of every point for every pair (local_utils_3d.py:221-225, expression order kept)
  std::vector<double> samples((size_t)P * n_pts);
  for (int j = 0; j < n_pts; ++j) {
    const double X = points[j], Y = points[(long)n_pts + j], Z = points[2L * n_pts + j];
    const double d0 = std::sqrt((X - mic[0]) * (X - mic[0]) + (Y - mic[1]) * (Y - mic[1]) + (Z - mic[2]) * (Z - mic[2])) /
                      sound_speed * fs;
    for (int i = 0; i < P; ++i) {
      const double* m = mic + 3 * (i + 1);
      const double di = std::sqrt((X - m[0]) * (X - m[0]) + (Y - m[1]) * (Y - m[1]) + (Z - m[2]) * (Z - m[2])) /
                        sound_speed * fs;
      samples[(size_t)i * n_pts + j] = di - d0;
    }
  }
  std::vector<Cell> frontier(1), done;
  frontier[0].off.assign(offset, offset + P);
  frontier[0].width.assign(width, width + P);
  frontier[0].idx.resize(n_pts);
  for (int j = 0; j < n_pts; ++j) frontier[0].idx[j] = j;
  bool first = true;
  while (!frontier.empty()) {
    std::vector<Cell> next;
    for (Cell& c : frontier) {
      if (ub && check_out(c, ub, P)) c.filtered = false;
      if (first) {                                     // the caller's patch is mutated (a-L)
        std::memcpy(offset, c.off.data(), sizeof(double) * P);
        std::memcpy(width, c.width.data(), sizeof(double) * P);
        first = false;
      }
      double wmax = c.width[0];
      for (int i = 1; i < P; ++i) wmax = c.width[i] > wmax ? c.width[i] : wmax;
      if (wmax / 2 <= MIN_WIDTH_REQUIRED && (int)c.idx.size() <= MIN_AREA) { done.push_back(std::move(c)); continue; }
      std::vector<Cell> best, last;
      bool have_best = false, wide_seen = false, any = false;
      long best_diff = 2500000;
      for (int i = 0; i < P; ++i) {
        if (c.width[i] / 2 < MIN_WIDTH) continue;
        any = true;
        std::vector<Cell> kids;
        long sizes[2] = {0, 0};
        double half_w = 0;
        for (int sgn = 0; sgn < 2; ++sgn) {
          Cell k;
          k.off = c.off;
          k.off[i] += (sgn == 0 ? -1.0 : 1.0) * c.width[i] / 4;
          k.width = c.width;
          k.width[i] /= 2;
          half_w = k.width[i];
          if (c.filtered) {
            for (int pj : c.idx)
              if (inside_dim(samples.data(), n_pts, pj, i, k.off.data(), k.width.data())) k.idx.push_back(pj);
          } else {
            for (int pj : c.idx)
              if (inside(samples.data(), n_pts, pj, k.off.data(), k.width.data(), P)) k.idx.push_back(pj);
          }
          k.filtered = true;
          sizes[sgn] = (long)k.idx.size();
          if (!k.idx.empty()) kids.push_back(std::move(k));
        }
        const long diff = std::labs(sizes[0] - sizes[1]);
        if (half_w > MIN_WIDTH_REQUIRED) {
          if (!wide_seen || diff < best_diff) { best = kids; best_diff = diff; have_best = true; }
          wide_seen = true;
        } else if (!wide_seen && diff < best_diff) {
          best = kids; best_diff = diff; have_best = true;
        }
        last = std::move(kids);
      }
      if (!any || !have_best || last.empty()) { done.push_back(std::move(c)); continue; }
      for (Cell& k : best) next.push_back(std::move(k));
    }
    frontier = std::move(next);
  }
  // ---- hand the result to the caller (freed with asw_free)
  const int nc = (int)done.size();
  size_t total = 0;
  for (const Cell& c : done) total += c.idx.size();
  double* co = (double*)std::malloc(sizeof(double) * (size_t)(nc > 0 ? nc : 1) * P);
  double* cw = (double*)std::malloc(sizeof(double) * (size_t)(nc > 0 ? nc : 1) * P);
  int* cc = (int*)std::malloc(sizeof(int) * (size_t)(nc > 0 ? nc : 1));
  int* ci = (int*)std::malloc(sizeof(int) * (total > 0 ? total : 1));
  if (!co || !cw || !cc || !ci) return asw::set_error(ASW_ERR_NOMEM, "search_area: host allocation failed");
  size_t pos = 0;
  for (int k = 0; k < nc; ++k) {
    std::memcpy(co + (size_t)k * P, done[k].off.data(), sizeof(double) * P);
    std::memcpy(cw + (size_t)k * P, done[k].width.data(), sizeof(double) * P);
    cc[k] = (int)done[k].idx.size();
    std::memcpy(ci + pos, done[k].idx.data(), sizeof(int) * done[k].idx.size());
    pos += done[k].idx.size();
  }
  *n_children = nc;
  *child_offset = co; *child_width = cw; *child_count = cc; *child_index = ci;
  return ASW_OK;
}

extern "C" void asw_free(void* p) { std::free(p); }

// Grid points whose TDoA vector lies inside a cube: the box scan of hyperbola_offset /
// hyperbola_area_sample (sep/Traditional_SP/SRP_Prunning.py:19-61) over a sub-box
// [y0,y1) x [x0,x1) x all z of a lookup table offsets[ny][nx][nz][P].  Writes the flat indices
// ((y*nx + x)*nz + z) of the hits in scan order (the order of the reference's boolean mask).
// The same scan over a table stored pair-major, planes[P][ny][nx][nz]: almost every point fails on the first
// pair, so the scan streams 8 bytes per point instead of the 8 P of the interleaved layout (the 30 cubes of one
// SRP-PHAT stage visit 3.2 M points: 25 -> 4 ms on the build host).  Same comparisons, same order.
extern "C" int asw_cube_select_planes(const double* planes, int ny, int nx, int nz, int P, int y0, int y1, int x0, int x1,
                                      const double* lo, const double* hi, int32_t* out_idx, int64_t cap, int64_t* count) {
  ASW_CHECK_ARG(planes && lo && hi && out_idx && count, "cube_select: null pointer");
  ASW_CHECK_ARG(ny > 0 && nx > 0 && nz > 0 && P > 0 && 0 <= y0 && y0 <= y1 && y1 <= ny && 0 <= x0 && x0 <= x1 && x1 <= nx,
                "cube_select: bad box");
  const size_t plane = (size_t)ny * nx * nz;
  const double l0 = lo[0], h0 = hi[0];
  int64_t n = 0;
  for (int y = y0; y < y1; ++y) {
    const size_t i0 = ((size_t)y * nx + x0) * nz, i1 = ((size_t)y * nx + x1) * nz;   // one contiguous run per row
    size_t i = i0;
    while (i < i1) {
      // first pair, eight points at a time without a branch (the compiler vectorises the block); the rare
      // survivors are then checked against the other pairs in scan order
      const size_t blk = i1 - i < 8 ? i1 - i : 8;
      unsigned m = 0;
      if (blk == 8) {
        for (int k = 0; k < 8; ++k) m |= (unsigned)((planes[i + k] >= l0) & (planes[i + k] <= h0)) << k;
      } else {
        for (size_t k = 0; k < blk; ++k) m |= (unsigned)((planes[i + k] >= l0) & (planes[i + k] <= h0)) << k;
      }
      for (size_t k = 0; m; ++k, m >>= 1) {
        if (!(m & 1u)) continue;
        bool in = true;
        for (int p = 1; p < P; ++p) {
          const double v = planes[(size_t)p * plane + i + k];
          if (!(v >= lo[p] && v <= hi[p])) { in = false; break; }
        }
        if (in) {
          if (n >= cap) return asw::set_error(ASW_ERR_ARG, "cube_select: output capacity %lld too small", (long long)cap);
          out_idx[n++] = (int32_t)(i + k);
        }
      }
      i += blk;
    }
  }
  *count = n;
  return ASW_OK;
}

extern "C" int asw_cube_select(const double* offsets, int ny, int nx, int nz, int P, int y0, int y1, int x0, int x1,
                               const double* lo, const double* hi, int32_t* out_idx, int64_t cap, int64_t* count) {
  ASW_CHECK_ARG(offsets && lo && hi && out_idx && count, "cube_select: null pointer");
  ASW_CHECK_ARG(ny > 0 && nx > 0 && nz > 0 && P > 0 && 0 <= y0 && y0 <= y1 && y1 <= ny && 0 <= x0 && x0 <= x1 && x1 <= nx,
                "cube_select: bad box");
  int64_t n = 0;
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x) {
      const double* row = offsets + ((size_t)y * nx + x) * nz * P;
      for (int z = 0; z < nz; ++z) {
        const double* v = row + (size_t)z * P;
        bool in = true;
        for (int p = 0; p < P; ++p)
          if (!(v[p] >= lo[p] && v[p] <= hi[p])) { in = false; break; }
        if (in) {
          if (n >= cap) return asw::set_error(ASW_ERR_ARG, "cube_select: output capacity %lld too small", (long long)cap);
          out_idx[n++] = (int32_t)(((size_t)y * nx + x) * nz + z);
        }
      }
    }
  *count = n;
  return ASW_OK;
}
