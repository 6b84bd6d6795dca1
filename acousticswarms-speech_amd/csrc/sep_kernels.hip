// sep_kernels.hip -- the small (memory / latency bound, VALU) kernels of the joint separation
// network: joint normalisation statistics of the S x M zero-fill-shifted channels
// (sep/training/SpeakerSeparation/network.py:510-534 with :28-40) and the pieces of its
// bottleneck (:270-321) that are not GEMMs -- row LayerNorms with fused residual adds, GLU,
// depthwise convolution + LayerNorm + Swish, relative-position self-attention over time
// (speechbrain RelPosMHAXL as published; the library itself is absent, see oracle/sep_ref.py)
// and the inter-speaker attention over the S speakers of one time step
// (nn.TransformerEncoderLayer, :290-291,311-316).
//
// The bottleneck tensors are [S][L][d] fp32 channels-last with L = T/64 (750 at T = 48 000),
// d = 512: a few MB, so every kernel here is one short pass and the stage is launch-latency
// bound; the GEMMs between them run on the MFMA kernels of convgemm.hip.
#include <cstdlib>

#include "asw_common.h"

namespace {

__device__ __forceinline__ float quant16(float x) { return rintf(x * 32768.0f) * (1.0f / 32768.0f); }
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wsum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float swishf(float v) { return v / (1.0f + expf(-v)); }

// ---------------------------------------------------------------------------------------
// Joint statistics: ref[t] = mean over the S*M channels of the int16-quantised, zero-fill
// shifted mixture (channel (s,m) = mix[m] advanced by offsets[s][m-1], mic 0 unshifted);
// mean = mean_t ref, std = unbiased std_t ref.  Pass 1: per-block partial (sum, sum of
// squares) in double over a strided set of samples; pass 2: one wave reduces the partials and
// writes S copies (the preproc / un-normalise kernels take per-sequence arrays).
// ---------------------------------------------------------------------------------------
constexpr int JS_MAXCH = 2048;
constexpr int JS_BLOCKS = 128;

__global__ __launch_bounds__(256) void joint_stats_partial_kernel(const float* __restrict__ mix, int M, int T,
                                                                  const int32_t* __restrict__ offsets, int S,
                                                                  double* __restrict__ part) {
  __shared__ int off[JS_MAXCH];
  __shared__ double red[4][2];
  const int SM = S * M;
  for (int i = threadIdx.x; i < SM; i += 256) {
    const int s = i / M, m = i - s * M;
    off[i] = m == 0 ? 0 : offsets[(long)s * (M - 1) + m - 1];
  }
  __syncthreads();
  double a0 = 0.0, a1 = 0.0;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < T; t += gridDim.x * 256) {
    float acc = 0.f;
    for (int s = 0; s < S; ++s)
      for (int m = 0; m < M; ++m) {
        const int i = t + off[s * M + m];
        const float x = (i >= 0 && i < T) ? mix[(long)m * T + i] : 0.f;
        acc += quant16(x);
      }
    const float r = acc / (float)SM;
    a0 += (double)r;
    a1 += (double)r * (double)r;
  }
  a0 = wsum_d(a0); a1 = wsum_d(a1);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { red[wid][0] = a0; red[wid][1] = a1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x * 2 + 0] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    part[blockIdx.x * 2 + 1] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
  }
}

__global__ __launch_bounds__(64) void joint_stats_final_kernel(const double* __restrict__ part, int nblocks, int T, int S,
                                                               float* __restrict__ mean_out, float* __restrict__ std_out) {
  double a0 = 0.0, a1 = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 64) { a0 += part[i * 2]; a1 += part[i * 2 + 1]; }
  a0 = wsum_d(a0); a1 = wsum_d(a1);
  const double mean = a0 / (double)T;
  double var = (a1 - a0 * a0 / (double)T) / (double)(T - 1);      // Bessel, torch.std default
  if (var < 0) var = 0;
  const float mu = (float)mean, sg = (float)sqrt(var);
  for (int s = threadIdx.x; s < S; s += 64) { mean_out[s] = mu; std_out[s] = sg; }
}

// ---------------------------------------------------------------------------------------
// Row kernel: s = x + alpha * y (y optional); sum_out = s (optional);
// ln_out = act(LayerNorm(s) * gamma + beta) (optional; act 0 = none, 2 = Swish).
// One wave per row, the row in registers (NV float4 per lane, N <= 256 * NV, N % 4 == 0),
// two-pass statistics by wave shuffles -- the arithmetic of the GEMM epilogue's LayerNorm.
// ---------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void add_ln2_kernel(const float* __restrict__ x, const float* __restrict__ y, float alpha,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      int rows, int N, float eps, int act, float* __restrict__ sum_out,
                                                      float* __restrict__ ln_out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int n4 = N >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + (long)row * N);
  const float4* yr = y ? reinterpret_cast<const float4*>(y + (long)row * N) : nullptr;
  float4 v[NV];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = i * 64 + lane;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < n4) {
      a = xr[c4];
      if (yr) { const float4 b = yr[c4]; a.x += alpha * b.x; a.y += alpha * b.y; a.z += alpha * b.z; a.w += alpha * b.w; }
      if (sum_out) reinterpret_cast<float4*>(sum_out + (long)row * N)[c4] = a;
    }
    v[i] = a;
    sum += (a.x + a.y) + (a.z + a.w);
  }
  if (!ln_out) return;
  const float mu = wsum(sum) / (float)N;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (i * 64 + lane < n4) {
      const float dx = v[i].x - mu, dy = v[i].y - mu, dz = v[i].z - mu, dw = v[i].w - mu;
      sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  }
  const float rstd = 1.0f / sqrtf(wsum(sq) / (float)N + eps);
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
  float4* orow = reinterpret_cast<float4*>(ln_out + (long)row * N);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = i * 64 + lane;
    if (c4 < n4) {
      const float4 g = g4[c4], b = b4[c4];
      float4 o = make_float4((v[i].x - mu) * rstd * g.x + b.x, (v[i].y - mu) * rstd * g.y + b.y,
                             (v[i].z - mu) * rstd * g.z + b.z, (v[i].w - mu) * rstd * g.w + b.w);
      if (act == 2) { o.x = swishf(o.x); o.y = swishf(o.y); o.z = swishf(o.z); o.w = swishf(o.w); }
      orow[c4] = o;
    }
  }
}

// out[r][c] = raw[r][c] * sigmoid(raw[r][C + c])   (nn.GLU over channels-last rows of 2C)
__global__ __launch_bounds__(256) void glu_rows_kernel(const float* __restrict__ raw, long rows, int C, float* __restrict__ out) {
  const int c4n = C >> 2;
  const long total = rows * c4n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / c4n;
    const int c = (int)(i - r * c4n) * 4;
    const float4 a = *reinterpret_cast<const float4*>(raw + r * 2 * C + c);
    const float4 g = *reinterpret_cast<const float4*>(raw + r * 2 * C + C + c);
    float4 o;
    o.x = a.x / (1.0f + expf(-g.x)); o.y = a.y / (1.0f + expf(-g.y));
    o.z = a.z / (1.0f + expf(-g.z)); o.w = a.w / (1.0f + expf(-g.w));
    *reinterpret_cast<float4*>(out + r * C + c) = o;
  }
}

// Depthwise Conv1d(d, d, k, padding (k-1)/2, groups = d) over time within each sequence,
// + bias, then LayerNorm over channels and Swish (ConvolutionModule: conv -> after_conv[0..1]).
// u, out: [BS][L][d]; wT: [K][d] (tap-major so a wave reads contiguous channel rows).
// One wave per (sequence, time) row.
template <int NV>
__global__ __launch_bounds__(256) void dwconv_ln_swish_kernel(const float* __restrict__ u, const float* __restrict__ wT,
                                                              const float* __restrict__ bias, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int BS, int L, int d, int K,
                                                              float eps, float* __restrict__ out) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= (long)BS * L) return;
  const int seq = (int)(row / L), t = (int)(row - (long)seq * L);
  const int n4 = d >> 2, pad = (K - 1) / 2;
  float4 acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = i * 64 + lane;
    acc[i] = c4 < n4 ? reinterpret_cast<const float4*>(bias)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int k = 0; k < K; ++k) {
    const int tt = t + k - pad;
    if (tt < 0 || tt >= L) continue;                     // wave-uniform: zero padding
    const float4* ur = reinterpret_cast<const float4*>(u + ((long)seq * L + tt) * d);
    const float4* wr = reinterpret_cast<const float4*>(wT + (long)k * d);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c4 = i * 64 + lane;
      if (c4 < n4) {
        const float4 a = ur[c4], w = wr[c4];
        acc[i].x = fmaf(a.x, w.x, acc[i].x); acc[i].y = fmaf(a.y, w.y, acc[i].y);
        acc[i].z = fmaf(a.z, w.z, acc[i].z); acc[i].w = fmaf(a.w, w.w, acc[i].w);
      }
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) sum += (acc[i].x + acc[i].y) + (acc[i].z + acc[i].w);     // lanes past the row hold zeros
  const float mu = wsum(sum) / (float)d;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (i * 64 + lane < n4) {
      const float dx = acc[i].x - mu, dy = acc[i].y - mu, dz = acc[i].z - mu, dw = acc[i].w - mu;
      sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  const float rstd = 1.0f / sqrtf(wsum(sq) / (float)d + eps);
  float4* orow = reinterpret_cast<float4*>(out + row * d);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = i * 64 + lane;
    if (c4 < n4) {
      const float4 g = reinterpret_cast<const float4*>(gamma)[c4], b = reinterpret_cast<const float4*>(beta)[c4];
      orow[c4] = make_float4(swishf((acc[i].x - mu) * rstd * g.x + b.x), swishf((acc[i].y - mu) * rstd * g.y + b.y),
                             swishf((acc[i].z - mu) * rstd * g.z + b.z), swishf((acc[i].w - mu) * rstd * g.w + b.w));
    }
  }
}

// ---------------------------------------------------------------------------------------
// Relative-position self-attention over time (RelPosMHAXL as published by speechbrain):
//   score[i][j] = scale * ( (q_i + u_h) . k_j  +  (q_i + v_h) . P_h[(L-1) + j - i] ),
//   ctx = softmax_j(score) V,   scale = 1/sqrt(embed_dim), P = linear_pos(sinusoid table).
// qkv [BS][L][3d] in the standard layout (Q | K | V, head h at columns h*HD; the launcher's
// caller permutes speechbrain's per-head (q,k,v) interleave when packing in_proj_weight),
// P [2L-1][d], bu / bv [d] (flat, head-major as pos_bias_*.view(1,1,H,hd) reads them).
// Per (sequence, head, 16-query tile) a flash-style sweep over 64-key tiles held in LDS with
// an online softmax, exact fp32 VALU arithmetic; the P rows a tile needs are the 79
// consecutive rows (L-1) + k0 - q0 - 15 .. + 78.  thread = (query q = tid/16, sub-lane
// sl = tid%16): scores for keys sl + 16 jj, output dims sl + 16 i.
// ---------------------------------------------------------------------------------------
constexpr int RA_BQ = 16, RA_KT = 64, RA_PR = RA_KT + RA_BQ - 1;

template <int HD>
__global__ __launch_bounds__(256) void relpos_attention_kernel(const float* __restrict__ qkv, const float* __restrict__ P,
                                                               const float* __restrict__ bu, const float* __restrict__ bv,
                                                               int L, int d, float scale, float* __restrict__ ctx) {
  constexpr int LDK = HD + 1, ND = HD / 16;
  extern __shared__ __align__(16) float smem[];
  float* Qu = smem;                       // [BQ][HD]   (q + u) * scale
  float* Qv = Qu + RA_BQ * HD;            // [BQ][HD]   (q + v) * scale
  float* Ks = Qv + RA_BQ * HD;            // [KT][HD+1]
  float* Vs = Ks + RA_KT * LDK;           // [KT][HD]
  float* Pw = Vs + RA_KT * HD;            // [PR][HD+1] window of P rows for this (query tile, key tile)
  float* Pr = Pw + RA_PR * LDK;           // [BQ][KT] probabilities
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * RA_BQ;
  const int tid = threadIdx.x, q = tid >> 4, sl = tid & 15;
  const float* base = qkv + (long)b * L * 3 * d;

  for (int i = tid; i < RA_BQ * HD; i += 256) {
    const int r = i / HD, c = i - r * HD;
    const float qq = (q0 + r < L) ? base[(long)(q0 + r) * 3 * d + h * HD + c] : 0.f;
    Qu[i] = (qq + bu[h * HD + c]) * scale;
    Qv[i] = (qq + bv[h * HD + c]) * scale;
  }
  float acc[ND];
#pragma unroll
  for (int i = 0; i < ND; ++i) acc[i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  for (int k0 = 0; k0 < L; k0 += RA_KT) {
    const int kn = L - k0 < RA_KT ? L - k0 : RA_KT;
    __syncthreads();                      // previous tile fully consumed (also covers Qu / Qv)
    for (int i = tid; i < RA_KT * HD; i += 256) {
      const int r = i / HD, c = i - r * HD;
      float kv = 0.f, vv = 0.f;
      if (r < kn) {
        const float* row = base + (long)(k0 + r) * 3 * d + h * HD + c;
        kv = row[d];
        vv = row[2 * d];
      }
      Ks[r * LDK + c] = kv;
      Vs[r * HD + c] = vv;
    }
    const int pbase = (L - 1) + k0 - q0 - (RA_BQ - 1);
    for (int i = tid; i < RA_PR * HD; i += 256) {
      const int r = i / HD, c = i - r * HD;
      const int m = pbase + r;
      Pw[r * LDK + c] = (m >= 0 && m <= 2 * L - 2) ? P[(long)m * d + h * HD + c] : 0.f;
    }
    __syncthreads();
    float sc[RA_KT / 16];
    float tmax = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < RA_KT / 16; ++jj) {
      const int j = sl + 16 * jj;
      const float* kr = Ks + j * LDK;
      const float* pr = Pw + (j - q + RA_BQ - 1) * LDK;
      float s = 0.f;
#pragma unroll 8
      for (int c = 0; c < HD; ++c) s = fmaf(Qu[q * HD + c], kr[c], fmaf(Qv[q * HD + c], pr[c], s));
      sc[jj] = (j < kn) ? s : -INFINITY;
      tmax = fmaxf(tmax, sc[jj]);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = (m_run == -INFINITY) ? 0.f : expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int jj = 0; jj < RA_KT / 16; ++jj) {
      const float pv = (sc[jj] == -INFINITY) ? 0.f : expf(sc[jj] - m_new);
      Pr[q * RA_KT + sl + 16 * jj] = pv;
      psum += pv;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) psum += __shfl_xor(psum, o, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ND; ++i) acc[i] *= alpha;
    for (int j = 0; j < kn; ++j) {
      const float pv = Pr[q * RA_KT + j];
#pragma unroll
      for (int i = 0; i < ND; ++i) acc[i] = fmaf(pv, Vs[j * HD + sl + 16 * i], acc[i]);
    }
  }
  if (q0 + q < L) {
    const float inv = 1.0f / l_run;
    float* o = ctx + ((long)b * L + q0 + q) * d + h * HD;
#pragma unroll
    for (int i = 0; i < ND; ++i) o[sl + 16 * i] = acc[i] * inv;
  }
}

// ---------------------------------------------------------------------------------------
// The same attention on the f32 matrix cores (head_dim 64, the full network's shape): exact fp32
// arithmetic (v_mfma_f32_32x32x2_f32 is an fmaf chain), one workgroup (4 waves) per (sequence,
// head, 64-query tile), online softmax over 64-key tiles.
//   AC  = (Q+u) K^T            4 MFMA tiles (2 query x 2 key halves), one per wave
//   BD: the relative-position term needs, for query i and key j of the tile, row (L-1)+j-i of P --
//       127 consecutive rows per (query tile, key tile).  Wave (qi, kj) multiplies its 32 queries
//       with the 64 P rows its block can touch (2 MFMA tiles), parks the 32 x 64 product in LDS
//       and reads it back skewed: bd[i][j] = G[i][(j - i) + 31].
//   O   = O * alpha + P V      wave (qi, dj) owns a 32 x 32 block of the 64 x 64 output tile.
// ---------------------------------------------------------------------------------------
typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int RM_BQ = 64, RM_BK = 64, RM_HD = 64, RM_LD = RM_HD + 4;     // row stride 68 floats: conflict-free b128 reads

__device__ __forceinline__ floatx16 rm_mma(const float* a_row, const float* b_row, floatx16 acc) {
  // a_row / b_row include this lane's (row, 4*(lane>>5)) offset; 8 k per iteration, hd = 64
#pragma unroll
  for (int kk = 0; kk < RM_HD / 8; ++kk) {
    const float4 a = *reinterpret_cast<const float4*>(a_row + kk * 8);
    const float4 b = *reinterpret_cast<const float4*>(b_row + kk * 8);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
  }
  return acc;
}

__global__ __launch_bounds__(256) void relpos_attention_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ P,
                                                                    const float* __restrict__ bu, const float* __restrict__ bv,
                                                                    int L, int d, float scale, float* __restrict__ ctx) {
  extern __shared__ __align__(16) float smem[];
  float* Qu = smem;                          // [64][68]
  float* Qv = Qu + RM_BQ * RM_LD;            // [64][68]
  float* KV = Qv + RM_BQ * RM_LD;            // [64][68] K tile, then V tile
  float* Pw = KV + RM_BK * RM_LD;            // [128][68] window of P rows; later 4 x [32][68] skew buffers
  float* Sc = Pw + 2 * RM_BK * RM_LD;        // [64][68] scores, then probabilities
  float* rm = Sc + RM_BQ * RM_LD;            // [64] running max
  float* rl = rm + RM_BQ;                    // [64] running sum
  float* ra = rl + RM_BQ;                    // [64] rescale factor of this step
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * RM_BQ;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int qi = wid >> 1, kj = wid & 1;
  const float* base = qkv + (long)b * L * 3 * d;

  for (int i = tid; i < RM_BQ * (RM_HD / 4); i += 256) {
    const int r = i / (RM_HD / 4), c4 = (i - r * (RM_HD / 4)) * 4;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q0 + r < L) q = *reinterpret_cast<const float4*>(base + (long)(q0 + r) * 3 * d + h * RM_HD + c4);
    const float4 u = *reinterpret_cast<const float4*>(bu + h * RM_HD + c4);
    const float4 v = *reinterpret_cast<const float4*>(bv + h * RM_HD + c4);
    *reinterpret_cast<float4*>(Qu + r * RM_LD + c4) = make_float4((q.x + u.x) * scale, (q.y + u.y) * scale, (q.z + u.z) * scale, (q.w + u.w) * scale);
    *reinterpret_cast<float4*>(Qv + r * RM_LD + c4) = make_float4((q.x + v.x) * scale, (q.y + v.y) * scale, (q.z + v.z) * scale, (q.w + v.w) * scale);
  }
  if (tid < RM_BQ) { rm[tid] = -INFINITY; rl[tid] = 0.f; }
  floatx16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  float* G = Pw + wid * 32 * RM_LD;          // this wave's skew buffer (aliases the P window once it is consumed)
  const int rt0 = kj - qi + 1;               // first of the two 32-row blocks of the P window this wave needs

  for (int k0 = 0; k0 < L; k0 += RM_BK) {
    __syncthreads();                         // previous step fully consumed (also covers Qu / Qv / rm / rl)
    for (int i = tid; i < RM_BK * (RM_HD / 4); i += 256) {
      const int r = i / (RM_HD / 4), c4 = (i - r * (RM_HD / 4)) * 4;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k0 + r < L) kv = *reinterpret_cast<const float4*>(base + (long)(k0 + r) * 3 * d + d + h * RM_HD + c4);
      *reinterpret_cast<float4*>(KV + r * RM_LD + c4) = kv;
    }
    const int pbase = (L - 1) + k0 - q0 - (RM_BQ - 1);
    for (int i = tid; i < 2 * RM_BK * (RM_HD / 4); i += 256) {
      const int r = i / (RM_HD / 4), c4 = (i - r * (RM_HD / 4)) * 4;
      const int m = pbase + r;
      float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m >= 0 && m <= 2 * L - 2) pv = *reinterpret_cast<const float4*>(P + (long)m * d + h * RM_HD + c4);
      *reinterpret_cast<float4*>(Pw + r * RM_LD + c4) = pv;
    }
    __syncthreads();
    floatx16 ac, g0, g1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { ac[r] = 0.f; g0[r] = 0.f; g1[r] = 0.f; }
    const float* qa = Qu + (qi * 32 + lr) * RM_LD + lh * 4;
    const float* qb = Qv + (qi * 32 + lr) * RM_LD + lh * 4;
    ac = rm_mma(qa, KV + (kj * 32 + lr) * RM_LD + lh * 4, ac);
    g0 = rm_mma(qb, Pw + (rt0 * 32 + lr) * RM_LD + lh * 4, g0);
    g1 = rm_mma(qb, Pw + ((rt0 + 1) * 32 + lr) * RM_LD + lh * 4, g1);
    __syncthreads();                         // every wave is done with the P window and the K tile
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
      G[row * RM_LD + lr] = g0[r];
      G[row * RM_LD + 32 + lr] = g1[r];
    }
    // V tile (the K tile is consumed) while the skew buffers settle
    for (int i = tid; i < RM_BK * (RM_HD / 4); i += 256) {
      const int r = i / (RM_HD / 4), c4 = (i - r * (RM_HD / 4)) * 4;
      float4 vv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k0 + r < L) vv = *reinterpret_cast<const float4*>(base + (long)(k0 + r) * 3 * d + 2 * d + h * RM_HD + c4);
      *reinterpret_cast<float4*>(KV + r * RM_LD + c4) = vv;
    }
    __syncthreads();
    {
      const bool valid = k0 + kj * 32 + lr < L;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float s = ac[r] + G[row * RM_LD + (lr - row + 31)];            // bd[i][j] = G[i][(j - i) + 31]
        Sc[(qi * 32 + row) * RM_LD + kj * 32 + lr] = valid ? s : -INFINITY;
      }
    }
    __syncthreads();
    {                                        // online softmax: 4 lanes per query row
      const int row = tid >> 2, sub = tid & 3;
      float* pr = Sc + row * RM_LD;
      float m = -INFINITY;
      for (int j = sub; j < RM_BK; j += 4) m = fmaxf(m, pr[j]);
      m = fmaxf(m, __shfl_xor(m, 1, 64));
      m = fmaxf(m, __shfl_xor(m, 2, 64));
      const float m_old = rm[row], m_new = fmaxf(m_old, m);
      float sum = 0.f;
      for (int j = sub; j < RM_BK; j += 4) {
        const float e = expf(pr[j] - m_new);                                 // exp(-inf) = 0 for padded keys
        pr[j] = e;
        sum += e;
      }
      sum += __shfl_xor(sum, 1, 64);
      sum += __shfl_xor(sum, 2, 64);
      if (sub == 0) {
        const float alpha = expf(m_old - m_new);                             // 0 on the first step (m_old = -inf)
        ra[row] = alpha;
        rl[row] = rl[row] * alpha + sum;
        rm[row] = m_new;
      }
    }
    __syncthreads();
    {                                        // O = O * alpha + P V : wave (qi, dj = kj) owns a 32 x 32 block
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] *= ra[qi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
      const float* pa = Sc + (qi * 32 + lr) * RM_LD + lh * 4;
      const float* vb = KV + (lh * 4) * RM_LD + kj * 32 + lr;
#pragma unroll 4
      for (int kk = 0; kk < RM_BK / 8; ++kk) {
        const float4 a = *reinterpret_cast<const float4*>(pa + kk * 8);
        const float b0 = vb[(kk * 8 + 0) * RM_LD], b1 = vb[(kk * 8 + 1) * RM_LD], b2 = vb[(kk * 8 + 2) * RM_LD],
                    b3 = vb[(kk * 8 + 3) * RM_LD];
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b2, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b3, o, 0, 0, 0);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = qi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    const int q = q0 + row;
    if (q < L) ctx[((long)b * L + q) * d + h * RM_HD + kj * 32 + lr] = o[r] / rl[row];
  }
}

// ---------------------------------------------------------------------------------------
// Inter-speaker attention: at every (item n, time step t, head h) the S speakers are the
// sequence (nn.MultiheadAttention inside nn.TransformerEncoderLayer(batch_first), applied to
// x.reshape(N*T, S, F), SpeakerSeparation/network.py:311-316).  qkv [N][S][L][3d] standard
// layout with in_proj bias already added; ctx [N][S][L][d].
// One wave per (n, t, head): the S x hd blocks of q, k, v are staged once into LDS (rows padded
// to hd+1 floats so the per-lane row walks are conflict free); lane s2 computes the scores
// (s1, s2) for every s1 with an in-lane dot product, the softmax runs across lanes, and for the
// output lane = head dimension.  S <= 64, hd <= 64.  IA_WAVES waves (units) per workgroup: 4 while their
// regions fit the CU's LDS (S <= 38 at hd = 64), else 2 (any S <= 64: 2 x 66 KB at S = 64, hd = 64) or 1.
// ---------------------------------------------------------------------------------------
template <int IA_WAVES>
__global__ __launch_bounds__(64 * IA_WAVES) void inter_attention_kernel(const float* __restrict__ qkv, int NB, int S, int L,
                                                                        int d, int nhead, float* __restrict__ ctx) {
  extern __shared__ __align__(16) float smem[];
  const int hd = d / nhead, ldk = hd + 1;
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* Qs = smem + (size_t)wid * (3 * S * ldk + S * 64);     // per-wave region: Q, K, V [S][hd+1], P [S][64]
  float* Ks = Qs + S * ldk;
  float* Vs = Ks + S * ldk;
  float* Ps = Vs + S * ldk;
  long unit = (long)blockIdx.x * IA_WAVES + wid;                  // (n, t, h)
  const bool active = unit < (long)NB * L * nhead;                // wave-uniform
  if (!active) unit = 0;
  const int h = (int)(unit % nhead);
  const long nt = unit / nhead;
  const int n = (int)(nt / L), t = (int)(nt - (long)n * L);
  const float scale = 1.0f / sqrtf((float)hd);
  for (int i = lane; active && i < S * hd; i += 64) {
    const int s = i / hd, c = i - s * hd;
    const float* row = qkv + ((long)(n * S + s) * L + t) * 3 * d + h * hd + c;
    Qs[s * ldk + c] = row[0] * scale;
    Ks[s * ldk + c] = row[d];
    Vs[s * ldk + c] = row[2 * d];
  }
  __syncthreads();
  for (int s1 = 0; active && s1 < S; ++s1) {
    float sc = -INFINITY;
    if (lane < S) {
      float acc = 0.f;
      const float* qr = Qs + s1 * ldk;
      const float* kr = Ks + lane * ldk;
      for (int c = 0; c < hd; ++c) acc = fmaf(qr[c], kr[c], acc);
      sc = acc;
    }
    const float mx = wmax(sc);
    const float e = lane < S ? expf(sc - mx) : 0.f;
    const float p = e / wsum(e);
    Ps[s1 * 64 + lane] = p;
  }
  __syncthreads();
  if (active && lane < hd) {
    for (int s1 = 0; s1 < S; ++s1) {
      float acc = 0.f;
      for (int s2 = 0; s2 < S; ++s2) acc = fmaf(Ps[s1 * 64 + s2], Vs[s2 * ldk + lane], acc);
      ctx[((long)(n * S + s1) * L + t) * d + h * hd + lane] = acc;
    }
  }
}

}  // namespace

extern "C" int asw_joint_shift_stats(const float* mix, int M, int T, const int32_t* offsets, int S, double* scratch,
                                     float* mean, float* std, void* stream) {
  ASW_CHECK_ARG(mix && offsets && scratch && mean && std, "joint_shift_stats: null pointer");
  ASW_CHECK_ARG(M >= 1 && S >= 1 && S * M <= JS_MAXCH && T >= 2, "joint_shift_stats: S=%d M=%d T=%d unsupported (S*M <= %d)", S,
                M, T, JS_MAXCH);
  hipStream_t s = asw::as_stream(stream);
  const int nb = asw::cdiv(T, 256) < JS_BLOCKS ? asw::cdiv(T, 256) : JS_BLOCKS;
  hipLaunchKernelGGL(joint_stats_partial_kernel, dim3(nb), dim3(256), 0, s, mix, M, T, offsets, S, scratch);
  ASW_LAUNCH_CHECK();
  hipLaunchKernelGGL(joint_stats_final_kernel, dim3(1), dim3(64), 0, s, scratch, nb, T, S, mean, std);
  ASW_LAUNCH_CHECK();
  return ASW_OK;
}

extern "C" int asw_joint_shift_stats_scratch_doubles(void) { return 2 * JS_BLOCKS; }

extern "C" int asw_add_layernorm2(const float* x, const float* y, float alpha, const float* gamma, const float* beta,
                                  int rows, int N, float eps, int act, float* sum_out, float* ln_out, void* stream) {
  ASW_CHECK_ARG(x && (sum_out || ln_out), "add_layernorm2: null pointer");
  ASW_CHECK_ARG(!ln_out || (gamma && beta), "add_layernorm2: LayerNorm needs gamma and beta");
  ASW_CHECK_ARG(rows >= 0 && N > 0 && N % 4 == 0 && N <= 2048, "add_layernorm2: N=%d must be a multiple of 4, <= 2048", N);
  ASW_CHECK_ARG(act == 0 || act == 2, "add_layernorm2: act must be 0 (none) or 2 (Swish)");
  if (rows == 0) return ASW_OK;
  hipStream_t s = asw::as_stream(stream);
  dim3 grid(asw::cdiv(rows, 4));
  switch ((N + 255) / 256) {
#define ASW_ALN2(NV) case NV: hipLaunchKernelGGL(add_ln2_kernel<NV>, grid, dim3(256), 0, s, x, y, alpha, gamma, beta, rows, N, eps, act, sum_out, ln_out); break;
    ASW_ALN2(1) ASW_ALN2(2) ASW_ALN2(3) ASW_ALN2(4) ASW_ALN2(5) ASW_ALN2(6) ASW_ALN2(7) ASW_ALN2(8)
#undef ASW_ALN2
  }
  ASW_LAUNCH_CHECK();
  return ASW_OK;
}

extern "C" int asw_glu_rows(const float* raw, long rows, int C, float* out, void* stream) {
  ASW_CHECK_ARG(raw && out, "glu_rows: null pointer");
  ASW_CHECK_ARG(rows >= 0 && C > 0 && C % 4 == 0, "glu_rows: C=%d must be a multiple of 4", C);
  if (rows == 0) return ASW_OK;
  const long total = rows * (C / 4);
  const int blocks = (int)(total / 256 + 1 < 4096 ? total / 256 + 1 : 4096);
  hipLaunchKernelGGL(glu_rows_kernel, dim3(blocks), dim3(256), 0, asw::as_stream(stream), raw, rows, C, out);
  ASW_LAUNCH_CHECK();
  return ASW_OK;
}

extern "C" int asw_dwconv_ln_swish(const float* u, const float* wT, const float* bias, const float* gamma, const float* beta,
                                   int BS, int L, int d, int K, float eps, float* out, void* stream) {
  ASW_CHECK_ARG(u && wT && bias && gamma && beta && out, "dwconv_ln_swish: null pointer");
  ASW_CHECK_ARG(BS > 0 && L > 0 && d > 0 && d % 4 == 0 && d <= 1024 && K >= 1 && K % 2 == 1,
                "dwconv_ln_swish: bad shape (d %% 4, d <= 1024, odd K)");
  hipStream_t s = asw::as_stream(stream);
  dim3 grid(asw::cdiv((long)BS * L, 4));
  switch ((d + 255) / 256) {
#define ASW_DW(NV) case NV: hipLaunchKernelGGL(dwconv_ln_swish_kernel<NV>, grid, dim3(256), 0, s, u, wT, bias, gamma, beta, BS, L, d, K, eps, out); break;
    ASW_DW(1) ASW_DW(2) ASW_DW(3) ASW_DW(4)
#undef ASW_DW
  }
  ASW_LAUNCH_CHECK();
  return ASW_OK;
}

namespace {
template <int HD>
int launch_relpos(const float* qkv, const float* P, const float* bu, const float* bv, int BS, int L, int d, int nhead,
                  float scale, float* ctx, hipStream_t s) {
  constexpr size_t smem = sizeof(float) * ((size_t)2 * RA_BQ * HD + (size_t)RA_KT * (HD + 1) + (size_t)RA_KT * HD +
                                           (size_t)RA_PR * (HD + 1) + (size_t)RA_BQ * RA_KT);
  static asw::SmemAttr attr;                                   // per device
  if (int rc = attr.ensure(reinterpret_cast<const void*>(relpos_attention_kernel<HD>), smem)) return rc;
  dim3 grid(asw::cdiv(L, RA_BQ), nhead, BS);
  asw::ProfScope prof(s, "relpos_attention", 6.0 * BS * nhead * (double)L * L * HD);
  hipLaunchKernelGGL(relpos_attention_kernel<HD>, grid, dim3(256), smem, s, qkv, P, bu, bv, L, d, scale, ctx);
  ASW_LAUNCH_CHECK();
  return ASW_OK;
}
}  // namespace

extern "C" int asw_relpos_attention(const float* qkv, const float* P, const float* bias_u, const float* bias_v, int BS, int L,
                                    int d, int nhead, float scale, float* ctx, void* stream) {
  ASW_CHECK_ARG(qkv && P && bias_u && bias_v && ctx, "relpos_attention: null pointer");
  ASW_CHECK_ARG(BS > 0 && BS <= 65535 && L > 0 && nhead > 0 && nhead <= 65535 && d % nhead == 0, "relpos_attention: bad shape");
  hipStream_t s = asw::as_stream(stream);
  static const bool valu_only = getenv("ASW_RELPOS_VALU") != nullptr;          // A/B switch for measurements
  if (d / nhead == RM_HD && d % 4 == 0 && !valu_only) {
    constexpr size_t smem = sizeof(float) * ((size_t)(3 * RM_BQ + 2 * RM_BK + RM_BQ) * RM_LD + 3 * RM_BQ);
    static asw::SmemAttr attr;                                 // per device
    if (int rc = attr.ensure(reinterpret_cast<const void*>(relpos_attention_mfma_kernel), smem)) return rc;
    dim3 grid(asw::cdiv(L, RM_BQ), nhead, BS);
    asw::ProfScope prof(s, "relpos_attention_mfma", 6.0 * BS * nhead * (double)L * L * RM_HD);
    hipLaunchKernelGGL(relpos_attention_mfma_kernel, grid, dim3(256), smem, s, qkv, P, bias_u, bias_v, L, d, scale, ctx);
    ASW_LAUNCH_CHECK();
    return ASW_OK;
  }
  switch (d / nhead) {
    case 16: return launch_relpos<16>(qkv, P, bias_u, bias_v, BS, L, d, nhead, scale, ctx, s);
    case 32: return launch_relpos<32>(qkv, P, bias_u, bias_v, BS, L, d, nhead, scale, ctx, s);
    case 64: return launch_relpos<64>(qkv, P, bias_u, bias_v, BS, L, d, nhead, scale, ctx, s);
    default: return asw::set_error(ASW_ERR_ARG, "relpos_attention: head_dim %d unsupported (16, 32, 64)", d / nhead);
  }
}

extern "C" int asw_inter_attention(const float* qkv, int NB, int S, int L, int d, int nhead, float* ctx, void* stream) {
  ASW_CHECK_ARG(qkv && ctx, "inter_attention: null pointer");
  ASW_CHECK_ARG(NB > 0 && S > 0 && S <= 64 && L > 0 && nhead > 0 && d % nhead == 0 && d / nhead <= 64,
                "inter_attention: bad shape (S <= 64, head_dim <= 64)");
  const int hd = d / nhead;
  const size_t per_wave = sizeof(float) * (3 * (size_t)S * (hd + 1) + (size_t)S * 64);
  const size_t lds_max = 160 * 1024;
  const int waves = 4 * per_wave <= lds_max ? 4 : 2 * per_wave <= lds_max ? 2 : 1;
  ASW_CHECK_ARG(per_wave <= lds_max, "inter_attention: S=%d x head_dim %d needs %zu bytes of LDS per wave", S, hd, per_wave);
  const size_t smem = per_wave * waves;
  const void* kern = waves == 4 ? reinterpret_cast<const void*>(inter_attention_kernel<4>)
                   : waves == 2 ? reinterpret_cast<const void*>(inter_attention_kernel<2>)
                                : reinterpret_cast<const void*>(inter_attention_kernel<1>);
  static asw::SmemAttr attr[3];                                // per device and instantiation
  if (int rc = attr[waves == 4 ? 0 : waves == 2 ? 1 : 2].ensure(kern, smem)) return rc;
  dim3 grid(asw::cdiv((long)NB * L * nhead, waves));
  hipStream_t st = asw::as_stream(stream);
  if (waves == 4) hipLaunchKernelGGL(inter_attention_kernel<4>, grid, dim3(256), smem, st, qkv, NB, S, L, d, nhead, ctx);
  else if (waves == 2) hipLaunchKernelGGL(inter_attention_kernel<2>, grid, dim3(128), smem, st, qkv, NB, S, L, d, nhead, ctx);
  else hipLaunchKernelGGL(inter_attention_kernel<1>, grid, dim3(64), smem, st, qkv, NB, S, L, d, nhead, ctx);
  ASW_LAUNCH_CHECK();
  return ASW_OK;
}
