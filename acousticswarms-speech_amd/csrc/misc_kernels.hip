// misc_kernels.hip -- the memory-bound / small kernels around the MFMA GEMMs.
#include "asw_common.h"

namespace {

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---------------------------------------------------------------------------
// GroupNorm(2) + GLU.  EncoderBlock / DecoderBlock tail
// (sep/training/SpeakerLocalization/network.py:107-113,194-197).  With 2 groups over
// 2C channels, group 0 is exactly the GLU value half and group 1 the gate half.
// Statistics arrive as per-tile partial sums from the producing GEMM's epilogue and are
// reduced here in double (deterministic, no atomics).
// ---------------------------------------------------------------------------
// (mean0, rstd0, mean1, rstd1) of batch item b from the partial sums, by a 256-thread workgroup; ends on a barrier
__device__ __forceinline__ void gn_group_stats(const float* __restrict__ stats, int n_partials, int b, int T, int C, float eps,
                                               double (&red)[4][4], float (&mr)[4]) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  double s[4] = {0, 0, 0, 0};
  const float* sp = stats + (long)b * n_partials * 4;
  for (int i = threadIdx.x; i < n_partials; i += blockDim.x) {
    const float4 v = *reinterpret_cast<const float4*>(sp + (long)i * 4);
    s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    s[k] = wave_sum_d(s[k]);
    if (lane == 0) red[wid][k] = s[k];
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    const int g = threadIdx.x;
    const double cnt = (double)T * (double)C;
    const double sum = red[0][2 * g] + red[1][2 * g] + red[2][2 * g] + red[3][2 * g];
    const double sq = red[0][2 * g + 1] + red[1][2 * g + 1] + red[2][2 * g + 1] + red[3][2 * g + 1];
    const double mean = sum / cnt;
    double var = sq / cnt - mean * mean;   // biased, as nn.GroupNorm
    if (var < 0) var = 0;
    mr[2 * g] = (float)mean;
    mr[2 * g + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
}

// the statistics alone, for a consumer that normalises while it loads (asw_convgemm_args.glu_raw)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ stats, int n_partials, int T, int C,
                                                          float eps, float* __restrict__ out) {
  __shared__ double red[4][4];
  __shared__ float mr[4];
  gn_group_stats(stats, n_partials, blockIdx.x, T, C, eps, red, mr);
  if (threadIdx.x < 4) out[blockIdx.x * 4 + threadIdx.x] = mr[threadIdx.x];
}

__global__ __launch_bounds__(256) void gn_glu_kernel(const float* __restrict__ raw, const float* __restrict__ stats,
                                                     int n_partials, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, int T, int C, float eps,
                                                     float* __restrict__ out, int items_per_block) {
  __shared__ double red[4][4];
  __shared__ float mr[4];                 // mean0, rstd0, mean1, rstd1
  const int b = blockIdx.y;
  gn_group_stats(stats, n_partials, b, T, C, eps, red, mr);
  const float m0 = mr[0], r0 = mr[1], m1 = mr[2], r1 = mr[3];
  const int c4n = C >> 2;
  const long total = (long)T * c4n;
  const long i0 = (long)blockIdx.x * items_per_block;
  const long i1 = i0 + items_per_block < total ? i0 + items_per_block : total;
  const float* rb = raw + (long)b * T * 2 * C;
  float* ob = out + (long)b * T * C;
  for (long i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
    const long t = i / c4n;
    const int c = (int)(i - t * c4n) * 4;
    const float4 a = *reinterpret_cast<const float4*>(rb + t * 2 * C + c);
    const float4 g = *reinterpret_cast<const float4*>(rb + t * 2 * C + C + c);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + c), ba = *reinterpret_cast<const float4*>(beta + c);
    const float4 gg = *reinterpret_cast<const float4*>(gamma + C + c), bg = *reinterpret_cast<const float4*>(beta + C + c);
    float4 o;
#define ASW_GLU(f)                                              \
  {                                                             \
    o.f = asw::gn_glu_value(a.f, g.f, m0, r0, m1, r1, ga.f, ba.f, gg.f, bg.f); \
  }
    ASW_GLU(x) ASW_GLU(y) ASW_GLU(z) ASW_GLU(w)
#undef ASW_GLU
    *reinterpret_cast<float4*>(ob + t * C + c) = o;
  }
}

// ---------------------------------------------------------------------------
// Self-attention core (nn.MultiheadAttention inside nn.TransformerEncoderLayer,
// network.py:254): per (batch, head, 16-query tile) a flash-style sweep over 128-key
// tiles held in LDS, online softmax, fp32 VALU.  The bottleneck is 1-4 % of the
// forward's FLOPs (sequence T/256 = 188 or 563), so this kernel is written for
// exactness and bounded LDS rather than MFMA rate.
// thread = (query q = tid/16, sub-lane sl = tid%16): scores for keys sl+16*jj,
// output dims sl+16*i.
// ---------------------------------------------------------------------------
constexpr int ATT_BQ = 16, ATT_KT = 128, ATT_MAXD = 8;   // head_dim <= 128

__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ qkv, int L, int d, int nhead,
                                                        float* __restrict__ ctx) {
  extern __shared__ __align__(16) float smem[];
  const int hd = d / nhead;
  const int ldk = hd + 1;
  float* Qs = smem;                       // [BQ][hd+1]
  float* Ks = Qs + ATT_BQ * ldk;          // [KT][hd+1]
  float* Vs = Ks + ATT_KT * ldk;          // [KT][hd]
  float* Ps = Vs + ATT_KT * hd;           // [BQ][KT]
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * ATT_BQ;
  const int tid = threadIdx.x, q = tid >> 4, sl = tid & 15;
  const float* base = qkv + (long)b * L * 3 * d;
  const float scale = 1.0f / sqrtf((float)hd);
  const int nd = hd >> 4;                 // output dims per thread

  for (int i = tid; i < ATT_BQ * hd; i += 256) {
    const int r = i / hd, c = i - r * hd;
    Qs[r * ldk + c] = (q0 + r < L) ? base[(long)(q0 + r) * 3 * d + h * hd + c] * scale : 0.f;
  }
  float acc[ATT_MAXD];
#pragma unroll
  for (int i = 0; i < ATT_MAXD; ++i) acc[i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  for (int k0 = 0; k0 < L; k0 += ATT_KT) {
    const int kn = L - k0 < ATT_KT ? L - k0 : ATT_KT;
    __syncthreads();                      // previous tile fully consumed (also covers Qs)
    for (int i = tid; i < ATT_KT * hd; i += 256) {
      const int r = i / hd, c = i - r * hd;
      float kv = 0.f, vv = 0.f;
      if (r < kn) {
        const float* row = base + (long)(k0 + r) * 3 * d + h * hd + c;
        kv = row[d];
        vv = row[2 * d];
      }
      Ks[r * ldk + c] = kv;
      Vs[r * hd + c] = vv;
    }
    __syncthreads();
    // scores for this thread's keys
    float sc[ATT_KT / 16];
    float tmax = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < ATT_KT / 16; ++jj) {
      const int j = sl + 16 * jj;
      float s = 0.f;
      for (int c = 0; c < hd; ++c) s = fmaf(Qs[q * ldk + c], Ks[j * ldk + c], s);
      sc[jj] = (j < kn) ? s : -INFINITY;
      tmax = fmaxf(tmax, sc[jj]);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = (m_run == -INFINITY) ? 0.f : expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int jj = 0; jj < ATT_KT / 16; ++jj) {
      const float pv = (sc[jj] == -INFINITY) ? 0.f : expf(sc[jj] - m_new);
      Ps[q * ATT_KT + sl + 16 * jj] = pv;
      psum += pv;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) psum += __shfl_xor(psum, o, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ATT_MAXD; ++i) acc[i] *= alpha;
    for (int j = 0; j < kn; ++j) {
      const float pv = Ps[q * ATT_KT + j];
#pragma unroll
      for (int i = 0; i < ATT_MAXD; ++i)
        if (i < nd) acc[i] = fmaf(pv, Vs[j * hd + sl + 16 * i], acc[i]);
    }
  }
  if (q0 + q < L) {
    const float inv = 1.0f / l_run;
    float* o = ctx + ((long)b * L + q0 + q) * d + h * hd;
#pragma unroll
    for (int i = 0; i < ATT_MAXD; ++i)
      if (i < nd) o[sl + 16 * i] = acc[i] * inv;
  }
}

// ---------------------------------------------------------------------------
// output_decoder ConvTranspose1d(E->1, k=taps, stride=hop) as overlap-add of the
// per-frame tap products D[b][f][j] (computed by a GEMM), trim [trim_left : ...], keep
// the last t samples (network.py:400-405), then y*std+mean (JointModel/network.py:96).
// ---------------------------------------------------------------------------
// D may arrive as `nparts` partial tap tensors (the fused mask path writes one per 256-channel column
// tile of the latent), part_stride floats apart; they are added in a fixed order.
__global__ __launch_bounds__(256) void overlap_add_kernel(const float* __restrict__ D, int nparts, long part_stride,
                                                          int F, int ldd, int taps,
                                                          int hop, int t, int lead, float bias,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ stdv, float* __restrict__ out) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= t) return;
  const int u = i + lead;                       // index into the untrimmed ConvTranspose output
  float s = bias;
  const float* Db = D + (long)b * F * ldd;
  for (int j = u % hop; j < taps; j += hop) {
    const int f = (u - j) / hop;
    if (f >= 0 && f < F) {
      float v = Db[(long)f * ldd + j];
      for (int q = 1; q < nparts; ++q) v += Db[q * part_stride + (long)f * ldd + j];
      s += v;
    }
  }
  if (mean) s = s * stdv[b] + mean[b];
  out[(long)b * t + i] = s;
}

// ---------------------------------------------------------------------------
// Candidate energies (sep/helpers/local_utils_3d.py:13-17,349-354; Mic_Array.py:290-295):
// x = y - mean(y); power = sum x^2; power2 = max_i sqrt(|mean(x^2[i:i+W])|) with zeros
// past the end, i.e. window sums c[min(i+W,T)] - c[i] of the prefix sum c of x^2 (double).
//
// One workgroup of 16 waves per candidate, every read coalesced (VEC consecutive samples per lane,
// 64 * VEC per wave instruction), no prefix array in memory:
//   1. mean (double sum, rounded to float32 like np.mean of a float32 row);
//   2. the sum of x^2 of every 64*VEC-sample block -> LDS, exclusive scan by wave 0: P[j] = c[j * 64 * VEC];
//   3. every wave walks a contiguous run of blocks with TWO cursors, one over x^2[i] and one over
//      x^2[i + W]: in-lane prefix + a wave scan (shuffles) per block give c[i] and c[i + W] for the
//      lane's samples, the running carries start from P[] (+ the partial block in front of i0 + W).
// Bytes: the row is read four times (L2 hits after the first); round 2's version wrote and re-read a
// (T+1)-double prefix row per candidate with per-thread contiguous (uncoalesced) chunks: 0.44 ms per
// 256 candidates at T = 48 000.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_incl_scan_d(double v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double u = __shfl_up(v, o, 64);
    if (lane >= o) v += u;
  }
  return v;
}

constexpr int ENERGY_MAX_BLOCKS = 4096;

template <int VEC>
__global__ __launch_bounds__(1024) void energy_kernel(const float* __restrict__ y, int T, int window, double* __restrict__ out) {
  constexpr int BLK = 64 * VEC;
  __shared__ double red[16];
  __shared__ double P[ENERGY_MAX_BLOCKS + 1];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float* yb = y + (long)b * T;
  const int nblk = (T + BLK - 1) / BLK;
  // VEC samples of block j owned by this lane, zeros past the end of the row
  auto load = [&](long i0, float (&v)[VEC]) {
    if (VEC == 4 && i0 + 3 < T) {
      const float4 f = *reinterpret_cast<const float4*>(yb + i0);
      v[0] = f.x; v[1 % VEC] = f.y; v[2 % VEC] = f.z; v[3 % VEC] = f.w;
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] = i0 + e < T ? yb[i0 + e] : 0.f;
    }
  };
  // ---- 1. mean
  double acc = 0.0;
  for (int j = wid; j < nblk; j += 16) {
    float v[VEC];
    load((long)j * BLK + lane * VEC, v);
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc += (double)v[e];
  }
  acc = wave_sum_d(acc);
  if (lane == 0) red[wid] = acc;
  __syncthreads();
  double tot = 0.0;
  for (int i = 0; i < 16; ++i) tot += red[i];
  const float mu = (float)(tot / (double)T);     // np.mean of a float32 row is float32
  // squared, mean-removed samples of the lane (0 past the end)
  auto squares = [&](long i0, float (&q)[VEC]) {
    float v[VEC];
    load(i0, v);
#pragma unroll
    for (int e = 0; e < VEC; ++e) { const float x = v[e] - mu; q[e] = i0 + e < T ? x * x : 0.f; }
  };
  // ---- 2. block sums and their exclusive scan
  for (int j = wid; j < nblk; j += 16) {
    float q[VEC];
    squares((long)j * BLK + lane * VEC, q);
    double s = 0.0;
#pragma unroll
    for (int e = 0; e < VEC; ++e) s += (double)q[e];
    s = wave_sum_d(s);
    if (lane == 0) P[j + 1] = s;
  }
  __syncthreads();
  if (wid == 0) {
    double carry = 0.0;
    if (lane == 0) P[0] = 0.0;
    for (int j0 = 0; j0 < nblk; j0 += 64) {
      const double v = j0 + lane < nblk ? P[j0 + lane + 1] : 0.0;
      const double inc = wave_incl_scan_d(v, lane) + carry;
      if (j0 + lane < nblk) P[j0 + lane + 1] = inc;
      carry = __shfl(inc, 63, 64);
    }
  }
  __syncthreads();
  const double total = P[nblk];
  // ---- 3. windowed maximum: blocks [jb, je) of this wave
  const int per = (nblk + 15) / 16;
  const int jb = wid * per, je = jb + per < nblk ? jb + per : nblk;
  double best = 0.0;
  if (jb < je) {
    double carryA = P[jb];
    // c[min(jb*BLK + W, T)]: whole blocks from P, the partial block in front of it summed here
    double carryB;
    {
      const long s = (long)jb * BLK + window;
      if (s >= T) carryB = total;
      else {
        const int js = (int)(s / BLK);
        float q[VEC];
        squares((long)js * BLK + lane * VEC, q);
        double part = 0.0;
#pragma unroll
        for (int e = 0; e < VEC; ++e) part += ((long)js * BLK + lane * VEC + e < s) ? (double)q[e] : 0.0;
        carryB = P[js] + wave_sum_d(part);
      }
    }
    for (int j = jb; j < je; ++j) {
      const long i0 = (long)j * BLK + lane * VEC;
      float qa[VEC], qb[VEC];
      squares(i0, qa);
      squares(i0 + window, qb);
      double ta = 0.0, tb = 0.0;
#pragma unroll
      for (int e = 0; e < VEC; ++e) { ta += (double)qa[e]; tb += (double)qb[e]; }
      const double ia = wave_incl_scan_d(ta, lane), ib = wave_incl_scan_d(tb, lane);
      double ca = carryA + (ia - ta), cb = carryB + (ib - tb);       // c[i0], c[i0 + W]
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        if (i0 + e < T) {
          const double w = fabs((cb - ca) / (double)window);
          best = w > best ? w : best;
        }
        ca += (double)qa[e];
        cb += (double)qb[e];
      }
      carryA += __shfl(ia, 63, 64);
      carryB += __shfl(ib, 63, 64);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const double v = __shfl_xor(best, o, 64); best = v > best ? v : best; }
  __syncthreads();
  if (lane == 0) red[wid] = best;
  __syncthreads();
  if (tid == 0) {
    double m = 0.0;
    for (int i = 0; i < 16; ++i) m = red[i] > m ? red[i] : m;
    out[b * 2 + 0] = total;
    out[b * 2 + 1] = sqrt(m);
  }
}

// ---------------------------------------------------------------------------
// SI-SDR of every ordered pair (sep/helpers/eval_utils.py:11-39), (i = estimate, j = reference):
//   a = <s,e>/<s,s>;  10 log10(|a s|^2 / (|e - a s|^2 + 1e-8)),  |e - a s|^2 = <e,e> - <s,e>^2/<s,s>.
// The three inner products are exact float32 products accumulated in double (a float32 x float32
// product is exact in double), so the Gram form loses nothing against the reference's element-wise
// float32 residual up to > 100 dB; the residual is clamped at 0 before MIN_ERR is added.
// A workgroup owns a 16 x 16 block of pairs and streams T in 64-sample chunks through LDS: every
// sample of the 32 rows is fetched once per block instead of once per pair (n = 301 waveforms of the
// bench scene: 6.2 -> 1.6 ms).
// ---------------------------------------------------------------------------
__device__ __forceinline__ double sisdr_from_products(double ee, double ss, double es) {
  const double sss = es * es / ss;
  double snn = ee - sss;
  snn = (snn > 0.0 ? snn : 0.0) + 1e-8;
  return 10.0 * log10(sss / snn);
}

__global__ __launch_bounds__(256) void pair_sisdr_kernel(const float* __restrict__ y, int n, int T,
                                                         double* __restrict__ out) {
  constexpr int KC = 64;
  __shared__ float E[16][KC + 1], S[16][KC + 1];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int i0 = blockIdx.x * 16, j0 = blockIdx.y * 16;
  double ee = 0, ss = 0, es = 0;
  float pe[4], ps[4];
  auto fetch = [&](int t0) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int idx = tid + 256 * m, r = idx >> 6, c = idx & 63;
      const bool okt = t0 + c < T;
      pe[m] = (okt && i0 + r < n) ? y[(long)(i0 + r) * T + t0 + c] : 0.f;
      ps[m] = (okt && j0 + r < n) ? y[(long)(j0 + r) * T + t0 + c] : 0.f;
    }
  };
  fetch(0);
  for (int t0 = 0; t0 < T; t0 += KC) {
    __syncthreads();                                   // previous chunk consumed
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int idx = tid + 256 * m, r = idx >> 6, c = idx & 63;
      E[r][c] = pe[m];
      S[r][c] = ps[m];
    }
    __syncthreads();
    if (t0 + KC < T) fetch(t0 + KC);                   // in flight under the products of this chunk
#pragma unroll 16
    for (int k = 0; k < KC; ++k) {
      const double e = (double)E[ty][k], sv = (double)S[tx][k];
      ee += e * e;
      ss += sv * sv;
      es += e * sv;
    }
  }
  const int i = i0 + ty, j = j0 + tx;
  if (i < n && j < n) out[(long)i * n + j] = sisdr_from_products(ee, ss, es);
}

// The same for a FEW waveforms (the per-coarse-patch calls of the fine stage: n ~ 12-38): 16 x 16-pair tiles
// would leave a handful of workgroups streaming T one chunk after the other (measured 2.0 ms per call), so
// here a workgroup owns 2 x 2 pairs and its 256 threads split T; the partial products meet in LDS.
__global__ __launch_bounds__(256) void pair_sisdr_small_kernel(const float* __restrict__ y, int n, int T,
                                                               double* __restrict__ out) {
  __shared__ double red[4][8];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int i0 = blockIdx.x * 2, j0 = blockIdx.y * 2;
  const int i1 = i0 + 1 < n ? i0 + 1 : i0, j1 = j0 + 1 < n ? j0 + 1 : j0;      // a ragged edge repeats its row
  const float* e0 = y + (long)i0 * T;
  const float* e1 = y + (long)i1 * T;
  const float* s0 = y + (long)j0 * T;
  const float* s1 = y + (long)j1 * T;
  double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};       // ee0 ee1 ss0 ss1 e0s0 e0s1 e1s0 e1s1
  for (int t = tid; t < T; t += 256) {
    const double x0 = (double)e0[t], x1 = (double)e1[t], z0 = (double)s0[t], z1 = (double)s1[t];
    a[0] += x0 * x0; a[1] += x1 * x1; a[2] += z0 * z0; a[3] += z1 * z1;
    a[4] += x0 * z0; a[5] += x0 * z1; a[6] += x1 * z0; a[7] += x1 * z1;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    a[k] = wave_sum_d(a[k]);
    if (lane == 0) red[wid][k] = a[k];
  }
  __syncthreads();
  if (tid < 4) {
    const int di = tid >> 1, dj = tid & 1;
    const int i = i0 + di, j = j0 + dj;
    if (i < n && j < n) {
      double v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
      out[(long)i * n + j] = sisdr_from_products(v[di], v[2 + dj], v[4 + di * 2 + dj]);
    }
  }
}

// Segment-wise SI-SDR (split_wise_sisdr, sep/helpers/eval_utils.py:73-82; call site
// Mic_Array.py:432-458): for every ordered pair (i = estimate, j = reference) the SI-SDR over
// each voiced segment [a,b) of waveform i, same Gram form.  A workgroup owns estimate i and 64
// references: thread (j, q) accumulates the products of samples q*16 .. q*16+15 of every 64-sample
// chunk (references through an LDS tile, one coalesced fetch per chunk, the next chunk prefetched
// into registers), the four partial sums meet in LDS at the end of a segment.
__global__ __launch_bounds__(256) void seg_sisdr_kernel(const float* __restrict__ y, int n, int T,
                                                        const int* __restrict__ seg, const int* __restrict__ cnt, int kmax,
                                                        double* __restrict__ out) {
  constexpr int KC = 64;
  __shared__ float Ev[KC], S[64][KC + 1];
  __shared__ double red[3][4][64];
  const int i = blockIdx.x, j0 = blockIdx.y * 64;
  const int tid = threadIdx.x, jl = tid & 63, q = tid >> 6;
  const int nseg = cnt[i];
  const float* __restrict__ ei = y + (long)i * T;
  float ps[16], pe = 0.f;
  for (int k = 0; k < nseg; ++k) {
    const int a = seg[((long)i * kmax + k) * 2], b = seg[((long)i * kmax + k) * 2 + 1];
    double ee = 0, ss = 0, es = 0;
    auto fetch = [&](int t0) {
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const int idx = tid + 256 * m, r = idx >> 6, c = idx & 63;
        ps[m] = (t0 + c < b && j0 + r < n) ? y[(long)(j0 + r) * T + t0 + c] : 0.f;
      }
      if (tid < KC) pe = (t0 + tid < b) ? ei[t0 + tid] : 0.f;
    };
    fetch(a);
    for (int t0 = a; t0 < b; t0 += KC) {
      __syncthreads();                                 // previous chunk (and the previous segment's red[]) consumed
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const int idx = tid + 256 * m, r = idx >> 6, c = idx & 63;
        S[r][c] = ps[m];
      }
      if (tid < KC) Ev[tid] = pe;
      __syncthreads();
      if (t0 + KC < b) fetch(t0 + KC);
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const double e = (double)Ev[q * 16 + c], sv = (double)S[jl][q * 16 + c];
        ee += e * e;
        ss += sv * sv;
        es += e * sv;
      }
    }
    __syncthreads();
    red[0][q][jl] = ee; red[1][q][jl] = ss; red[2][q][jl] = es;
    __syncthreads();
    if (q == 0 && j0 + jl < n) {
      const double fe = (red[0][0][jl] + red[0][1][jl]) + (red[0][2][jl] + red[0][3][jl]);
      const double fs = (red[1][0][jl] + red[1][1][jl]) + (red[1][2][jl] + red[1][3][jl]);
      const double fx = (red[2][0][jl] + red[2][1][jl]) + (red[2][2][jl] + red[2][3][jl]);
      out[((long)i * n + j0 + jl) * kmax + k] = sisdr_from_products(fe, fs, fx);
    }
  }
}

// x[b][:] -= mean(x[b][:])   (sep/Mic_Array.py:291: the stage loops centre every candidate
// output before measuring / comparing it); mean accumulated in double, applied as float32
__global__ __launch_bounds__(1024) void center_rows_kernel(float* __restrict__ y, int T) {
  __shared__ double red[16];
  float* yb = y + (long)blockIdx.x * T;
  double acc = 0.0;
  for (int i = threadIdx.x; i < T; i += 1024) acc += (double)yb[i];
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  double tot = 0.0;
  for (int i = 0; i < 16; ++i) tot += red[i];
  const float mu = (float)(tot / (double)T);
  for (int i = threadIdx.x; i < T; i += 1024) yb[i] -= mu;
}

// out[r][:] = LayerNorm(x[r][:] + resid[r][:]) * gamma + beta   (post-norm transformer block,
// nn.TransformerEncoderLayer norm1/norm2, sep/training/SpeakerLocalization/network.py:254).
// One wave per row, the row in registers (NV float4 per lane), two-pass statistics by wave
// shuffles -- the same arithmetic as the GEMM epilogue's fused LayerNorm.  Used for rows too
// wide (d = 1024) for a LayerNorm-fused GEMM tile of useful height: the GEMM then runs on the
// wide 256x256 tile and this pass costs one read + one write of the activations.
template <int NV>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const float* __restrict__ x, const float* __restrict__ resid,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            int rows, float eps, float* __restrict__ out) {
  constexpr int N = NV * 256;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float4* xr = reinterpret_cast<const float4*>(x + (long)row * N);
  const float4* rr = reinterpret_cast<const float4*>(resid + (long)row * N);
  float4 v[NV];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float4 a = xr[i * 64 + lane], r = rr[i * 64 + lane];
    v[i] = make_float4(a.x + r.x, a.y + r.y, a.z + r.z, a.w + r.w);
    sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float mu = sum / (float)N;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float dx = v[i].x - mu, dy = v[i].y - mu, dz = v[i].z - mu, dw = v[i].w - mu;
    sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
  const float rstd = 1.0f / sqrtf(sq / (float)N + eps);
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
  float4* orow = reinterpret_cast<float4*>(out + (long)row * N);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float4 g = g4[i * 64 + lane], b = b4[i * 64 + lane];
    orow[i * 64 + lane] = make_float4((v[i].x - mu) * rstd * g.x + b.x, (v[i].y - mu) * rstd * g.y + b.y,
                                      (v[i].z - mu) * rstd * g.z + b.z, (v[i].w - mu) * rstd * g.w + b.w);
  }
}

}  // namespace

extern "C" int asw_add_layernorm(const float* x, const float* resid, const float* gamma, const float* beta, int rows,
                                 int N, float eps, float* out, void* stream) {
  ASW_CHECK_ARG(x && resid && gamma && beta && out, "add_layernorm: null pointer");
  ASW_CHECK_ARG(rows >= 0 && N > 0 && N % 256 == 0 && N <= 2048, "add_layernorm: N=%d must be a multiple of 256 <= 2048", N);
  if (rows == 0) return ASW_OK;
  hipStream_t s = asw::as_stream(stream);
  dim3 grid(asw::cdiv(rows, 4));
  asw::ProfScope prof(s, "add_layernorm", 0.0, (double)rows * N * 12);     // x, residual read; result written
  switch (N / 256) {
#define ASW_ALN(NV) case NV: hipLaunchKernelGGL(add_layernorm_kernel<NV>, grid, dim3(256), 0, s, x, resid, gamma, beta, rows, eps, out); break;
    ASW_ALN(1) ASW_ALN(2) ASW_ALN(3) ASW_ALN(4) ASW_ALN(5) ASW_ALN(6) ASW_ALN(7) ASW_ALN(8)
#undef ASW_ALN
  }
  ASW_LAUNCH_CHECK();
  return ASW_OK;
}

extern "C" int asw_center_rows(float* y, int B, int T, void* stream) {
  ASW_CHECK_ARG(y && T > 0, "center_rows: bad argument");
  if (B <= 0) return ASW_OK;
  hipLaunchKernelGGL(center_rows_kernel, dim3(B), dim3(1024), 0, asw::as_stream(stream), y, T);
  ASW_LAUNCH_CHECK();
  return ASW_OK;
}

extern "C" int asw_gn_glu(const float* raw, const float* stats, int n_partials, const float* gamma,
                          const float* beta, int B, int T, int C, float eps, float* out, void* stream) {
  ASW_CHECK_ARG(raw && stats && gamma && beta && out, "gn_glu: null pointer");
  ASW_CHECK_ARG(B > 0 && T > 0 && C > 0 && C % 4 == 0 && n_partials > 0, "gn_glu: bad shape");
  const int items = 256 * 8;
  dim3 grid(asw::cdiv((long)T * (C / 4), items), B);
  // algorithmic bytes: the 2C-channel pre-GLU tensor read, the C-channel result written
  asw::ProfScope prof(asw::as_stream(stream), "gn_glu", 0.0, (double)B * T * C * 12);
  hipLaunchKernelGGL(gn_glu_kernel, grid, dim3(256), 0, asw::as_stream(stream), raw, stats, n_partials, gamma, beta,
                     T, C, eps, out, items);
  ASW_LAUNCH_CHECK();
  return ASW_OK;
}

extern "C" int asw_gn_finalize(const float* stats, int n_partials, int B, int T, int C, float eps, float* mr, void* stream) {
  ASW_CHECK_ARG(stats && mr, "gn_finalize: null pointer");
  ASW_CHECK_ARG(B > 0 && T > 0 && C > 0 && n_partials > 0, "gn_finalize: bad shape");
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, asw::as_stream(stream), stats, n_partials, T, C, eps, mr);
  ASW_LAUNCH_CHECK();
  return ASW_OK;
}

namespace asw { int attention_mfma(const float* qkv, int B, int L, int d, int nhead, float* ctx, hipStream_t s, int precision); }

extern "C" int asw_attention(const float* qkv, int B, int L, int d, int nhead, float* ctx, void* stream) {
  return asw_attention_prec(qkv, B, L, d, nhead, 0, ctx, stream);
}

extern "C" int asw_attention_prec(const float* qkv, int B, int L, int d, int nhead, int precision, float* ctx, void* stream) {
  ASW_CHECK_ARG(qkv && ctx, "attention: null pointer");
  ASW_CHECK_ARG(precision >= 0 && precision <= 2, "attention: precision %d", precision);
  ASW_CHECK_ARG(B > 0 && L > 0 && nhead > 0 && d % nhead == 0, "attention: bad shape");
  ASW_CHECK_ARG(B <= 65535 && nhead <= 65535, "attention: grid too large");
  {
    const int rc = asw::attention_mfma(qkv, B, L, d, nhead, ctx, asw::as_stream(stream), precision);   // head_dim 128: MFMA kernels
    if (rc != 1) return rc;
  }
  const int hd = d / nhead;
  ASW_CHECK_ARG(hd % 16 == 0 && hd <= 16 * ATT_MAXD, "attention: head_dim %d must be a multiple of 16, <= 128", hd);
  ASW_CHECK_ARG(B <= 65535 && nhead <= 65535, "attention: grid too large");
  const size_t smem = sizeof(float) * ((size_t)ATT_BQ * (hd + 1) + (size_t)ATT_KT * (hd + 1) + (size_t)ATT_KT * hd +
                                       (size_t)ATT_BQ * ATT_KT);
  static asw::SmemAttr smem_attr;                      // per device
  if (int rc = smem_attr.ensure(reinterpret_cast<const void*>(attention_kernel), smem)) return rc;
  dim3 grid(asw::cdiv(L, ATT_BQ), nhead, B);
  hipLaunchKernelGGL(attention_kernel, grid, dim3(256), smem, asw::as_stream(stream), qkv, L, d, nhead, ctx);
  ASW_LAUNCH_CHECK();
  return ASW_OK;
}

extern "C" int asw_overlap_add_unnorm(const float* D, int B, int F, int ldd, int taps, int hop, int t,
                                      int trim_left, int trim_right, float bias, const float* mean, const float* std, float* out,
                                      void* stream) {
  return asw_overlap_add_parts(D, 1, B, F, ldd, taps, hop, t, trim_left, trim_right, bias, mean, std, out, stream);
}

extern "C" int asw_overlap_add_parts(const float* D, int nparts, int B, int F, int ldd, int taps, int hop, int t,
                                     int trim_left, int trim_right, float bias, const float* mean, const float* std, float* out,
                                     void* stream) {
  ASW_CHECK_ARG(D && out, "overlap_add: null pointer");
  ASW_CHECK_ARG(nparts >= 1 && nparts <= 64, "overlap_add: 1..64 partial tap tensors");
  ASW_CHECK_ARG(B > 0 && F > 0 && taps > 0 && taps <= ldd && hop > 0 && t > 0, "overlap_add: bad shape");
  const int kept = (F - 1) * hop + taps - trim_left - trim_right;
  ASW_CHECK_ARG(trim_left >= 0 && trim_right >= 0 && kept >= t, "overlap_add: %d samples after the trim, %d requested", kept, t);
  const int lead = kept - t + trim_left;
  ASW_CHECK_ARG((mean == nullptr) == (std == nullptr), "overlap_add: mean/std must both be given or both NULL");
  ASW_CHECK_ARG(B <= 65535, "overlap_add: batch too large");
  dim3 grid(asw::cdiv(t, 256), B);
  // algorithmic bytes: the tap tensors read, the waveform written
  asw::ProfScope prof(asw::as_stream(stream), "overlap_add", 0.0, (double)nparts * B * F * taps * 4 + (double)B * t * 4);
  hipLaunchKernelGGL(overlap_add_kernel, grid, dim3(256), 0, asw::as_stream(stream), D, nparts, (long)B * F * ldd, F, ldd,
                     taps, hop, t, lead, bias, mean, std, out);
  ASW_LAUNCH_CHECK();
  return ASW_OK;
}

extern "C" int asw_energies(const float* y, int B, int T, int window, double* scratch, double* out, void* stream) {
  ASW_CHECK_ARG(y && out, "energies: null pointer");
  ASW_CHECK_ARG(T > 0 && window > 0, "energies: bad shape");
  if (B == 0) return ASW_OK;
  (void)scratch;                                       // kept in the ABI; the prefix sums no longer go through memory
  hipStream_t s = asw::as_stream(stream);
  asw::ProfScope prof(s, "energy", 0.0, (double)B * T * 4);
  // four samples per lane when every row is 16-byte aligned, one otherwise
  if (T % 4 == 0 && window % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0) {
    ASW_CHECK_ARG(T <= ENERGY_MAX_BLOCKS * 256, "energies: T=%d too long", T);
    hipLaunchKernelGGL(energy_kernel<4>, dim3(B), dim3(1024), 0, s, y, T, window, out);
  } else {
    ASW_CHECK_ARG(T <= ENERGY_MAX_BLOCKS * 64, "energies: T=%d too long for unaligned rows", T);
    hipLaunchKernelGGL(energy_kernel<1>, dim3(B), dim3(1024), 0, s, y, T, window, out);
  }
  ASW_LAUNCH_CHECK();
  return ASW_OK;
}

extern "C" int asw_segment_sisdr(const float* y, int n, int T, const int32_t* segments, const int32_t* seg_count,
                                 int kmax, double* out, void* stream) {
  ASW_CHECK_ARG(y && segments && seg_count && out, "segment_sisdr: null pointer");
  ASW_CHECK_ARG(n > 0 && n <= 65535 && T > 0 && kmax > 0, "segment_sisdr: bad shape");
  hipLaunchKernelGGL(seg_sisdr_kernel, dim3(n, asw::cdiv(n, 64)), dim3(256), 0, asw::as_stream(stream), y, n, T, segments,
                     seg_count, kmax, out);
  ASW_LAUNCH_CHECK();
  return ASW_OK;
}

extern "C" int asw_pair_sisdr(const float* y, int n, int T, double* out, void* stream) {
  ASW_CHECK_ARG(y && out, "pair_sisdr: null pointer");
  ASW_CHECK_ARG(n > 0 && n <= 65535 && T > 0, "pair_sisdr: bad shape");
  if (n <= 64)
    hipLaunchKernelGGL(pair_sisdr_small_kernel, dim3(asw::cdiv(n, 2), asw::cdiv(n, 2)), dim3(256), 0, asw::as_stream(stream),
                       y, n, T, out);
  else
    hipLaunchKernelGGL(pair_sisdr_kernel, dim3(asw::cdiv(n, 16), asw::cdiv(n, 16)), dim3(256), 0, asw::as_stream(stream), y, n, T,
                       out);
  ASW_LAUNCH_CHECK();
  return ASW_OK;
}
