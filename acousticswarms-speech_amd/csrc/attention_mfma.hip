// attention_mfma.hip -- bottleneck self-attention on the f32 MFMA pipe for the shapes the
// spot network produces: sequence L = T/256 (188 at T = 48 000, 563 at T = 144 000; any
// L <= 672 fits), head_dim 128.
// nn.MultiheadAttention core inside nn.TransformerEncoderLayer
// (sep/training/SpeakerLocalization/network.py:254): ctx = softmax(Q K^T / sqrt(hd)) V.
//
// One workgroup (4 waves) per (batch item, head, 32-query tile).  The whole score row of the
// tile stays in LDS, so there is no online-softmax rescaling and the arithmetic is an exact
// fp32 fmaf chain (v_mfma_f32_32x32x2_f32), like the fp32 GEMMs:
//   phase 1  S[32][L] = (Q/sqrt(hd)) K^T   keys staged 96 at a time (3 column tiles, waves 0-2)
//   softmax  row-wise over the L valid keys (8 lanes per row), padded keys -> 0
//   phase 2  O[32][128] = P V              V staged TRANSPOSED (Vt[n][key]) 96 keys at a time so
//                                          the MFMA B operand is one ds_read_b128; wave w owns
//                                          output columns [32w, 32w+32) across all key tiles
// LDS rows are padded by 4 floats: the per-lane 16-byte operand reads are conflict free.
// Sequences beyond 672 fall back to the flash-style VALU kernel in misc_kernels.hip.
#include "asw_common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int AQ = 32;            // queries per workgroup
constexpr int KT = 96;            // keys per staged tile (3 MFMA column tiles)
constexpr int AD = 128;           // head_dim
constexpr int LDQ = AD + 4;       // Q / K row stride (floats)
constexpr int LDV = KT + 4;       // Vt row stride (floats)
constexpr int KVF = (KT * LDQ > AD * LDV) ? KT * LDQ : AD * LDV;   // floats of the shared K / Vt buffer

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

// LP = L rounded up to a multiple of KT; the whole score row of a query tile lives in LDS
// (Ps[AQ][LP+4]), so the softmax is exact and needs no running rescale.  LDS:
// 17 KB (Q) + 50 KB (K / Vt tile) + 32*(LP+4)*4 B (scores): L <= 672 fits 160 KB.
__global__ __launch_bounds__(256) void attention_mfma_kernel(const float* __restrict__ qkv, int L, int LP, int d,
                                                             float* __restrict__ ctx) {
  extern __shared__ __align__(16) float smem[];
  const int LDP = LP + 4;
  float* Qs = smem;                          // [AQ][LDQ]
  float* KV = Qs + AQ * LDQ;                 // K tile [KT][LDQ]   or   Vt tile [AD][LDV]
  float* Ps = KV + KVF;                      // [AQ][LDP]
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * AQ;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float* base = qkv + (long)b * L * 3 * d + h * AD;
  const float scale = 1.0f / sqrtf((float)AD);
  const int lr = lane & 31, lh = lane >> 5;

  for (int i = tid; i < AQ * (AD / 4); i += 256) {
    const int r = i / (AD / 4), c4 = i - r * (AD / 4);
    const int q = q0 + r;
    float4 v = *reinterpret_cast<const float4*>(base + (long)(q < L ? q : 0) * 3 * d + c4 * 4);
    if (q >= L) v = make_float4(0.f, 0.f, 0.f, 0.f);
    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
    *reinterpret_cast<float4*>(Qs + r * LDQ + c4 * 4) = v;
  }

  // ---- phase 1: S = (Q/sqrt(hd)) K^T, one 96-key tile at a time; waves 0..2 own a column tile
  for (int k0 = 0; k0 < LP; k0 += KT) {
    __syncthreads();                                   // previous tile consumed (and Q staged)
    for (int i = tid; i < KT * (AD / 4); i += 256) {
      const int r = i / (AD / 4), c4 = i - r * (AD / 4);
      const int j = k0 + r;
      float4 v = *reinterpret_cast<const float4*>(base + (long)(j < L ? j : 0) * 3 * d + d + c4 * 4);
      if (j >= L) v = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(KV + r * LDQ + c4 * 4) = v;
    }
    __syncthreads();
    if (wid < KT / 32) {
      floatx16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      acc = mma_row(Qs + lr * LDQ + lh * 4, KV + (wid * 32 + lr) * LDQ + lh * 4, AD / 8, acc);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        Ps[row * LDP + k0 + wid * 32 + lr] = acc[r];
      }
    }
  }
  __syncthreads();

  // ---- row softmax over the L valid keys: 8 lanes per query row, padded keys -> 0
  {
    const int row = tid >> 3, sub = tid & 7;
    float* pr = Ps + row * LDP;
    float m = -INFINITY;
    for (int j = sub; j < L; j += 8) m = fmaxf(m, pr[j]);
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float s = 0.f;
    for (int j = sub; j < LP; j += 8) {
      const float e = j < L ? expf(pr[j] - m) : 0.f;
      pr[j] = e;
      s += e;
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float inv = 1.0f / s;
    for (int j = sub; j < LP; j += 8) pr[j] *= inv;
  }

  // ---- phase 2: O = P V, V staged transposed (Vt[n][key]) per 96-key tile; wave w owns O
  //      columns [32w, 32w+32) and keeps its accumulator across the tiles
  floatx16 oacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
  for (int k0 = 0; k0 < LP; k0 += KT) {
    __syncthreads();                                   // softmax done / previous Vt tile consumed
    for (int i = tid; i < KT * (AD / 4); i += 256) {
      const int r = i / (AD / 4), c4 = i - r * (AD / 4);
      const int j = k0 + r;
      float4 v = *reinterpret_cast<const float4*>(base + (long)(j < L ? j : 0) * 3 * d + 2 * d + c4 * 4);
      if (j >= L) v = make_float4(0.f, 0.f, 0.f, 0.f);
      KV[(c4 * 4 + 0) * LDV + r] = v.x;
      KV[(c4 * 4 + 1) * LDV + r] = v.y;
      KV[(c4 * 4 + 2) * LDV + r] = v.z;
      KV[(c4 * 4 + 3) * LDV + r] = v.w;
    }
    __syncthreads();
    oacc = mma_row(Ps + lr * LDP + k0 + lh * 4, KV + (wid * 32 + lr) * LDV + lh * 4, KT / 8, oacc);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int q = q0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (q < L) ctx[((long)b * L + q) * d + h * AD + wid * 32 + lr] = oacc[r];
  }
}

}  // namespace

namespace asw {
// returns 1 when the shape is not an MFMA-kernel case
int attention_mfma(const float* qkv, int B, int L, int d, int nhead, float* ctx, hipStream_t s) {
  if (d / nhead != AD) return 1;
  const int LP = cdiv(L, KT) * KT;
  const size_t smem = sizeof(float) * ((size_t)AQ * LDQ + (size_t)KVF + (size_t)AQ * (LP + 4));
  if (smem > 160 * 1024) return 1;                     // very long sequences: flash-style VALU kernel
  static size_t attr = 0;
  if (smem > attr) {
    ASW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_mfma_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr = smem;
  }
  dim3 grid(cdiv(L, AQ), nhead, B);
  ProfScope prof(s, "attention_mfma", 4.0 * B * nhead * (double)L * L * AD);
  hipLaunchKernelGGL(attention_mfma_kernel, grid, dim3(256), smem, s, qkv, L, LP, d, ctx);
  ASW_LAUNCH_CHECK();
  return ASW_OK;
}
}  // namespace asw
