// attention_mfma.hip -- bottleneck self-attention on the f32 MFMA pipe for the shapes the
// spot network produces at T <= 49 152 samples: sequence L <= 192 (= T/256), head_dim 128.
// nn.MultiheadAttention core inside nn.TransformerEncoderLayer
// (sep/training/SpeakerLocalization/network.py:254): ctx = softmax(Q K^T / sqrt(hd)) V.
//
// One workgroup (4 waves) per (batch item, head, 32-query tile).  Everything a tile needs
// fits in LDS, so there is no online-softmax rescaling and the arithmetic is an exact fp32
// fmaf chain (v_mfma_f32_32x32x2_f32), like the fp32 GEMMs:
//   phase 1  S[32][192] = (Q/sqrt(hd)) K^T   Q tile + all keys in LDS; 6 column tiles over 4 waves
//   softmax  row-wise over the L valid keys (8 lanes per row), padded keys -> 0
//   phase 2  O[32][128] = P V                 V staged TRANSPOSED (Vt[n][key]) so the MFMA B
//                                              operand is one ds_read_b128; 1 column tile per wave
// LDS rows are padded by 4 floats: the per-lane 16-byte operand reads are conflict free.
// Longer sequences (T = 144 000 -> L = 563) use the tiled flash-style kernel in
// misc_kernels.hip.
#include "asw_common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int AQ = 32;            // queries per workgroup
constexpr int AL = 192;           // padded key count
constexpr int AD = 128;           // head_dim
constexpr int LDQ = AD + 4;       // Q / K row stride (floats)
constexpr int LDP = AL + 4;       // P / Vt row stride (floats)

__device__ __forceinline__ floatx16 mma_row(const float* a_row, const float* b_row, int ksteps, floatx16 acc) {
  // a_row / b_row already include this lane's (row, 4*(lane>>5)) offset; 8 k per iteration
  for (int kk = 0; kk < ksteps; ++kk) {
    const float4 a = *reinterpret_cast<const float4*>(a_row + kk * 8);
    const float4 b = *reinterpret_cast<const float4*>(b_row + kk * 8);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
  }
  return acc;
}

__global__ __launch_bounds__(256) void attention_mfma_kernel(const float* __restrict__ qkv, int L, int d,
                                                             float* __restrict__ ctx) {
  extern __shared__ __align__(16) float smem[];
  float* Qs = smem;                          // [AQ][LDQ]
  float* Ps = Qs + AQ * LDQ;                 // [AQ][LDP]
  float* KV = Ps + AQ * LDP;                 // K: [AL][LDQ]   then   Vt: [AD][LDP]
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * AQ;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float* base = qkv + (long)b * L * 3 * d + h * AD;
  const float scale = 1.0f / sqrtf((float)AD);

  // ---- stage Q (scaled) and K; rows past L are zero
  for (int i = tid; i < AQ * (AD / 4); i += 256) {
    const int r = i / (AD / 4), c4 = i - r * (AD / 4);
    const int q = q0 + r;
    float4 v = *reinterpret_cast<const float4*>(base + (long)(q < L ? q : 0) * 3 * d + c4 * 4);
    if (q >= L) v = make_float4(0.f, 0.f, 0.f, 0.f);
    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
    *reinterpret_cast<float4*>(Qs + r * LDQ + c4 * 4) = v;
  }
  for (int i = tid; i < AL * (AD / 4); i += 256) {
    const int r = i / (AD / 4), c4 = i - r * (AD / 4);
    float4 v = *reinterpret_cast<const float4*>(base + (long)(r < L ? r : 0) * 3 * d + d + c4 * 4);
    if (r >= L) v = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(KV + r * LDQ + c4 * 4) = v;
  }
  __syncthreads();

  // ---- phase 1: S tiles.  wave w owns column tiles w and (w < 2) w + 4
  const int lr = lane & 31, lh = lane >> 5;
  for (int ct = wid; ct < AL / 32; ct += 4) {
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = mma_row(Qs + lr * LDQ + lh * 4, KV + (ct * 32 + lr) * LDQ + lh * 4, AD / 8, acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
      Ps[row * LDP + ct * 32 + lr] = acc[r];
    }
  }
  __syncthreads();

  // ---- stage V transposed over the K buffer (all waves are past phase 1) ...
  for (int i = tid; i < AL * (AD / 4); i += 256) {
    const int j = i / (AD / 4), c4 = i - j * (AD / 4);
    float4 v = *reinterpret_cast<const float4*>(base + (long)(j < L ? j : 0) * 3 * d + 2 * d + c4 * 4);
    if (j >= L) v = make_float4(0.f, 0.f, 0.f, 0.f);
    KV[(c4 * 4 + 0) * LDP + j] = v.x;
    KV[(c4 * 4 + 1) * LDP + j] = v.y;
    KV[(c4 * 4 + 2) * LDP + j] = v.z;
    KV[(c4 * 4 + 3) * LDP + j] = v.w;
  }
  // ---- ... and the row softmax: 8 lanes per query row
  {
    const int row = tid >> 3, sub = tid & 7;
    float* pr = Ps + row * LDP;
    float m = -INFINITY;
    for (int j = sub; j < L; j += 8) m = fmaxf(m, pr[j]);
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float s = 0.f;
    for (int j = sub; j < AL; j += 8) {
      const float e = j < L ? expf(pr[j] - m) : 0.f;
      pr[j] = e;
      s += e;
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float inv = 1.0f / s;
    for (int j = sub; j < AL; j += 8) pr[j] *= inv;
  }
  __syncthreads();

  // ---- phase 2: O column tile `wid` = P @ V
  {
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = mma_row(Ps + lr * LDP + lh * 4, KV + (wid * 32 + lr) * LDP + lh * 4, AL / 8, acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int q = q0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (q < L) ctx[((long)b * L + q) * d + h * AD + wid * 32 + lr] = acc[r];
    }
  }
}

}  // namespace

namespace asw {
// returns 1 when the shape is not an MFMA-kernel case
int attention_mfma(const float* qkv, int B, int L, int d, int nhead, float* ctx, hipStream_t s) {
  if (d / nhead != AD || L > AL) return 1;
  const size_t smem = sizeof(float) * ((size_t)AQ * LDQ + (size_t)AQ * LDP + (size_t)AL * LDQ);
  static_assert((size_t)AL * LDQ >= (size_t)AD * LDP, "Vt must fit in the K buffer");
  static bool attr = false;
  if (!attr) {
    ASW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_mfma_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr = true;
  }
  dim3 grid(cdiv(L, AQ), nhead, B);
  ProfScope prof(s, "attention_mfma", 4.0 * B * nhead * (double)L * L * AD);
  hipLaunchKernelGGL(attention_mfma_kernel, grid, dim3(256), smem, s, qkv, L, d, ctx);
  ASW_LAUNCH_CHECK();
  return ASW_OK;
}
}  // namespace asw
