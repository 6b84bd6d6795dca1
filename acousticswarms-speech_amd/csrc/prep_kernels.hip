// prep_kernels.hip -- candidate alignment front end (HBM/L2-bound, VALU only).
//
// K1+K2+K3 of SURVEY.md §2.2 fused: integer circular shift of the M channels
// (sep/training/JointModel/network.py:12-25,80-83), int16 quantise + normalise by the
// mean / unbiased std of the mic-average (sep/training/SpeakerLocalization/network.py:
// 28-40), left zero-pad to a multiple of the stride product (:377-378) and the 1x1
// preproc convolution (:305-307,385).  The shifted [N][M][T] tensor the reference
// materialises is never written: every candidate re-reads the (L2-resident) [M][T]
// mixture at shifted addresses and writes only the channels-last [N][T_pad][C]
// network input plus the normalised reference channel.
#include "asw_common.h"

namespace {

constexpr int MAX_MICS = 32;

__device__ __forceinline__ float quant16(float x) {
  // (x * 2**15).round() / 2**15, round-half-even like torch.round (network.py:34)
  return rintf(x * 32768.0f) * (1.0f / 32768.0f);
}

__device__ __forceinline__ double block_sum(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  double s = 0.0;
  for (int i = 0; i < nw; ++i) s += red[i];
  return s;
}

// One workgroup per candidate: two passes over T (sum, then centred sum of squares),
// double accumulation; the mixture stays in L2 across candidates.
__global__ __launch_bounds__(1024) void shift_stats_kernel(const float* __restrict__ mixes, int M, int T,
                                                           const int32_t* __restrict__ offsets,
                                                           const int32_t* __restrict__ mix_index, int circular,
                                                           float* __restrict__ mean_out,
                                                           float* __restrict__ std_out) {
  __shared__ int off[MAX_MICS];
  __shared__ double red[16];
  const int n = blockIdx.x;
  // candidates of several mixtures in one launch: candidate n reads mixture mix_index[n] of [K][M][T]
  const float* __restrict__ mix = mixes + (mix_index ? (long)mix_index[n] * M * T : 0);
  if (threadIdx.x < M) {
    int o = threadIdx.x == 0 ? 0 : offsets[(long)n * (M - 1) + threadIdx.x - 1];
    if (circular) { o %= T; if (o < 0) o += T; }
    off[threadIdx.x] = o;
  }
  __syncthreads();
  const float inv_m = 1.0f / (float)M;
  auto ref_at = [&](int t) -> float {
    float s = 0.f;
    for (int m = 0; m < M; ++m) {
      int i = t + off[m];
      float x;
      if (circular) { if (i >= T) i -= T; x = mix[(long)m * T + i]; }
      else x = (i >= 0 && i < T) ? mix[(long)m * T + i] : 0.f;
      s += quant16(x);
    }
    return s * inv_m;
  };
  double acc = 0.0;
  for (int t = threadIdx.x; t < T; t += blockDim.x) acc += (double)ref_at(t);
  const double mean = block_sum(acc, red) / (double)T;
  acc = 0.0;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    const double d = (double)ref_at(t) - mean;
    acc += d * d;
  }
  const double var = block_sum(acc, red) / (double)(T - 1);   // Bessel, torch.std default
  if (threadIdx.x == 0) {
    mean_out[n] = (float)mean;
    std_out[n] = (float)sqrt(var);
  }
}

// grid (row tiles, N).  Two phases per 64-row tile: (1) the M aligned, quantised, normalised
// input samples of every row are computed once into LDS (one (row, mic) pair per thread);
// (2) thread = (row, float4 of output channels) does only the M fmas per channel from LDS
// and a wave writes whole 256-byte channel rows back to back.
template <bool SHIFTED>
__global__ __launch_bounds__(256) void preproc_kernel(const float* __restrict__ srcs, int M, int T, int T_pad,
                                                      const int32_t* __restrict__ offsets,
                                                      const int32_t* __restrict__ mix_index, int circular,
                                                      const float* __restrict__ mean, const float* __restrict__ stdv,
                                                      const float* __restrict__ w, const float* __restrict__ bias,
                                                      int C, float* __restrict__ x0, float* __restrict__ refn,
                                                      long refn_stride, int rows_per_block) {
  __shared__ int off[MAX_MICS];
  __shared__ float vs[64 * MAX_MICS];       // [row][m], rows_per_block <= 64
  __shared__ float ws[128 * MAX_MICS];      // preproc weights [C][M], C <= 128
  const int n = blockIdx.y;
  const float* __restrict__ src = srcs + ((SHIFTED && mix_index) ? (long)mix_index[n] * M * T : 0);
  for (int i = threadIdx.x; i < C * M; i += blockDim.x) ws[i] = w[i];
  const int c4n = C >> 2;                   // float4 groups per row
  if (threadIdx.x < M) {
    int o = 0;
    if (SHIFTED && threadIdx.x > 0) {
      o = offsets[(long)n * (M - 1) + threadIdx.x - 1];
      if (circular) { o %= T; if (o < 0) o += T; }
    }
    off[threadIdx.x] = o;
  }
  __syncthreads();
  const int pad = T_pad - T;
  const float mu = SHIFTED ? mean[n] : 0.f;
  const float sg = SHIFTED ? stdv[n] : 1.f;
  const int row0 = blockIdx.x * rows_per_block;
  for (int it = threadIdx.x; it < rows_per_block * M; it += blockDim.x) {
    const int m = it / rows_per_block, r = it - m * rows_per_block;   // consecutive lanes -> consecutive samples
    const int tp = row0 + r;
    float x = 0.f;
    if (tp >= pad && tp < T_pad) {
      const int t = tp - pad;
      if (SHIFTED) {
        int i = t + off[m];
        if (circular) { if (i >= T) i -= T; x = src[(long)m * T + i]; }
        else x = (i >= 0 && i < T) ? src[(long)m * T + i] : 0.f;
        x = (quant16(x) - mu) / sg;
      } else {
        x = src[((long)n * M + m) * T + t];
      }
    }
    vs[r * M + m] = x;
  }
  __syncthreads();
  const int items = rows_per_block * c4n;
  for (int it = threadIdx.x; it < items; it += blockDim.x) {
    const int r = it / c4n, c4 = it - r * c4n;
    const int tp = row0 + r;
    if (tp >= T_pad) break;
    const int c = c4 * 4;
    float4 o = *reinterpret_cast<const float4*>(bias + c);
    if (tp >= pad) {
      for (int m = 0; m < M; ++m) {
        const float x = vs[r * M + m];
        o.x = fmaf(ws[(c + 0) * M + m], x, o.x);
        o.y = fmaf(ws[(c + 1) * M + m], x, o.y);
        o.z = fmaf(ws[(c + 2) * M + m], x, o.z);
        o.w = fmaf(ws[(c + 3) * M + m], x, o.w);
      }
    }
    *reinterpret_cast<float4*>(x0 + ((long)n * T_pad + tp) * C + c) = o;
    if (c4 == 0) refn[(long)n * refn_stride + tp] = vs[r * M];
  }
}

}  // namespace

extern "C" int asw_shift_stats(const float* mix, int M, int T, const int32_t* offsets, int N, int circular,
                               float* mean, float* std, void* stream) {
  return asw_shift_stats_multi(mix, M, T, offsets, nullptr, N, circular, mean, std, stream);
}

extern "C" int asw_shift_stats_multi(const float* mix, int M, int T, const int32_t* offsets, const int32_t* mix_index, int N,
                                     int circular, float* mean, float* std, void* stream) {
  ASW_CHECK_ARG(mix && offsets && mean && std, "shift_stats: null pointer");
  ASW_CHECK_ARG(M >= 1 && M <= MAX_MICS && T >= 2, "shift_stats: M=%d T=%d unsupported", M, T);
  if (N == 0) return ASW_OK;
  // algorithmic bytes: the M x T mixture once (every candidate re-reads it from L2) + the statistics
  asw::ProfScope prof(asw::as_stream(stream), "shift_stats", 0.0, (double)M * T * 4 + (double)N * 8);
  hipLaunchKernelGGL(shift_stats_kernel, dim3(N), dim3(1024), 0, asw::as_stream(stream), mix, M, T, offsets, mix_index,
                     circular, mean, std);
  ASW_LAUNCH_CHECK();
  return ASW_OK;
}

extern "C" int asw_shift_norm_preproc(const float* mix, int M, int T, int T_pad, const int32_t* offsets, int N,
                                      int circular, const float* mean, const float* std, const float* w,
                                      const float* b, int C, float* x0, float* refn, long refn_stride,
                                      void* stream) {
  return asw_shift_norm_preproc_multi(mix, M, T, T_pad, offsets, nullptr, N, circular, mean, std, w, b, C, x0, refn,
                                      refn_stride, stream);
}

extern "C" int asw_shift_norm_preproc_multi(const float* mix, int M, int T, int T_pad, const int32_t* offsets,
                                            const int32_t* mix_index, int N, int circular, const float* mean,
                                            const float* std, const float* w, const float* b, int C, float* x0, float* refn,
                                            long refn_stride, void* stream) {
  ASW_CHECK_ARG(refn_stride >= T_pad, "shift_norm_preproc: refn_stride < T_pad");
  ASW_CHECK_ARG(mix && offsets && mean && std && w && b && x0 && refn, "shift_norm_preproc: null pointer");
  ASW_CHECK_ARG(M >= 1 && M <= MAX_MICS && T >= 1 && T_pad >= T && C % 4 == 0 && C > 0 && C <= 128,
                "shift_norm_preproc: bad shape M=%d T=%d T_pad=%d C=%d (C <= 128)", M, T, T_pad, C);
  if (N == 0) return ASW_OK;
  const int rows = 64;
  dim3 grid(asw::cdiv(T_pad, rows), N);
  // algorithmic bytes: the mixture once + the [N][T_pad][C] activation and the reference channel written
  asw::ProfScope prof(asw::as_stream(stream), "preproc", 0.0, (double)M * T * 4 + (double)N * T_pad * (C + 1) * 4);
  hipLaunchKernelGGL(preproc_kernel<true>, grid, dim3(256), 0, asw::as_stream(stream), mix, M, T, T_pad, offsets, mix_index,
                     circular, mean, std, w, b, C, x0, refn, refn_stride, rows);
  ASW_LAUNCH_CHECK();
  return ASW_OK;
}

extern "C" int asw_pad_preproc(const float* x, int B, int M, int t, int T_pad, const float* w, const float* b,
                               int C, float* x0, float* refn, long refn_stride, void* stream) {
  ASW_CHECK_ARG(refn_stride >= T_pad, "pad_preproc: refn_stride < T_pad");
  ASW_CHECK_ARG(x && w && b && x0 && refn, "pad_preproc: null pointer");
  ASW_CHECK_ARG(M >= 1 && M <= MAX_MICS && t >= 1 && T_pad >= t && C % 4 == 0 && C > 0 && C <= 128,
                "pad_preproc: bad shape M=%d t=%d T_pad=%d C=%d (C <= 128)", M, t, T_pad, C);
  if (B == 0) return ASW_OK;
  const int rows = 64;
  dim3 grid(asw::cdiv(T_pad, rows), B);
  hipLaunchKernelGGL(preproc_kernel<false>, grid, dim3(256), 0, asw::as_stream(stream), x, M, t, T_pad,
                     (const int32_t*)nullptr, (const int32_t*)nullptr, 1, (const float*)nullptr, (const float*)nullptr, w, b, C, x0, refn,
                     refn_stride, rows);
  ASW_LAUNCH_CHECK();
  return ASW_OK;
}
