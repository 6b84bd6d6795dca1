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


// ---- 64-query variant for short sequences (L <= 352: T = 48 000 gives L = 188) -------------
// Two query tiles per workgroup halve the K / V re-staging, the 64 x 64 score tile of a key
// block is four MFMA tiles (one per wave, none idle), and V is staged row-major with float4
// stores (the B operand of O = P V is then read as four ds_read_b32 per four MFMAs instead of
// being transposed through scalar LDS stores).  Same exact fp32 arithmetic as above.
constexpr int BQ = 64;            // queries per workgroup
constexpr int BK = 64;            // keys per staged tile
typedef int intx4a __attribute__((ext_vector_type(4)));
typedef float floatx4a __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 row_load4(__amdgpu_buffer_rsrc_t r, long elem, bool ok) {
  // rows past the sequence end read as zeros through the descriptor's range check (no branch)
  const intx4a v = __builtin_amdgcn_raw_buffer_load_b128(r, ok ? (int)(elem * 4) : (int)0x80000000, 0, 0);
  const floatx4a f = __builtin_bit_cast(floatx4a, v);
  return make_float4(f[0], f[1], f[2], f[3]);
}

__global__ __launch_bounds__(256) void attention_mfma64_kernel(const float* __restrict__ qkv, int L, int LP, int d,
                                                               float* __restrict__ ctx) {
  extern __shared__ __align__(16) float smem[];
  const int LDP = LP + 4;
  float* Qs = smem;                          // [BQ][LDQ]
  float* KV = Qs + BQ * LDQ;                 // K tile [BK][LDQ], then V tile [BK][LDQ] (row-major)
  float* Ps = KV + BK * LDQ;                 // [BQ][LDP]
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * BQ;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const float scale = 1.0f / sqrtf((float)AD);
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(qkv + (long)b * L * 3 * d), 0, L * 3 * d * 4, 0x00020000);

  for (int i = tid; i < BQ * (AD / 4); i += 256) {
    const int r = i / (AD / 4), c4 = i - r * (AD / 4);
    float4 v = row_load4(rs, (long)(q0 + r) * 3 * d + h * AD + c4 * 4, q0 + r < L);
    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
    *reinterpret_cast<float4*>(Qs + r * LDQ + c4 * 4) = v;
  }
  // ---- phase 1: S = (Q/sqrt(hd)) K^T; wave w owns score tile (query tile w>>1, key tile w&1)
  const int qi = wid >> 1, kj = wid & 1;
  for (int k0 = 0; k0 < LP; k0 += BK) {
    __syncthreads();
    for (int i = tid; i < BK * (AD / 4); i += 256) {
      const int r = i / (AD / 4), c4 = i - r * (AD / 4);
      *reinterpret_cast<float4*>(KV + r * LDQ + c4 * 4) =
          row_load4(rs, (long)(k0 + r) * 3 * d + d + h * AD + c4 * 4, k0 + r < L);
    }
    __syncthreads();
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = mma_row(Qs + (qi * 32 + lr) * LDQ + lh * 4, KV + (kj * 32 + lr) * LDQ + lh * 4, AD / 8, acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = qi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      Ps[row * LDP + k0 + kj * 32 + lr] = acc[r];
    }
  }
  __syncthreads();
  // ---- row softmax over the L valid keys: 4 lanes per query row, padded keys -> 0
  {
    const int row = tid >> 2, sub = tid & 3;
    float* pr = Ps + row * LDP;
    float m = -INFINITY;
    for (int j = sub; j < L; j += 4) m = fmaxf(m, pr[j]);
    m = fmaxf(m, __shfl_xor(m, 1, 64));
    m = fmaxf(m, __shfl_xor(m, 2, 64));
    float s = 0.f;
    for (int j = sub; j < LP; j += 4) {
      const float e = j < L ? expf(pr[j] - m) : 0.f;
      pr[j] = e;
      s += e;
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    const float inv = 1.0f / s;
    for (int j = sub; j < LP; j += 4) pr[j] *= inv;
  }
  // ---- phase 2: O = P V; wave w owns output columns [32w, 32w+32) of both query tiles
  floatx16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  for (int k0 = 0; k0 < LP; k0 += BK) {
    __syncthreads();                                   // softmax done / previous V tile consumed
    for (int i = tid; i < BK * (AD / 4); i += 256) {
      const int r = i / (AD / 4), c4 = i - r * (AD / 4);
      *reinterpret_cast<float4*>(KV + r * LDQ + c4 * 4) =
          row_load4(rs, (long)(k0 + r) * 3 * d + 2 * d + h * AD + c4 * 4, k0 + r < L);
    }
    __syncthreads();
    const float* p0 = Ps + lr * LDP + k0 + lh * 4;
    const float* p1 = p0 + 32 * LDP;
    const float* vb = KV + (lh * 4) * LDQ + wid * 32 + lr;     // V[key 4*lh + i][column]
#pragma unroll 4
    for (int kk = 0; kk < BK / 8; ++kk) {
      const float4 a0 = *reinterpret_cast<const float4*>(p0 + kk * 8);
      const float4 a1 = *reinterpret_cast<const float4*>(p1 + kk * 8);
      const float b0 = vb[(kk * 8 + 0) * LDQ], b1 = vb[(kk * 8 + 1) * LDQ], b2 = vb[(kk * 8 + 2) * LDQ],
                  b3 = vb[(kk * 8 + 3) * LDQ];
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b0, o1, 0, 0, 0);
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b1, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1, o1, 0, 0, 0);
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b2, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b2, o1, 0, 0, 0);
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b3, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b3, o1, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int q = q0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (q < L) ctx[((long)b * L + q) * d + h * AD + wid * 32 + lr] = o0[r];
    if (q + 32 < L) ctx[((long)b * L + q + 32) * d + h * AD + wid * 32 + lr] = o1[r];
  }
}

}  // namespace

namespace asw {
// returns 1 when the shape is not an MFMA-kernel case
int attention_mfma(const float* qkv, int B, int L, int d, int nhead, float* ctx, hipStream_t s) {
  if (d / nhead != AD) return 1;
  {
    // short sequences: 64 queries per workgroup (scores of both tiles must fit beside Q and K/V)
    const int LP64 = cdiv(L, BK) * BK;
    const size_t smem64 = sizeof(float) * ((size_t)BQ * LDQ + (size_t)BK * LDQ + (size_t)BQ * (LP64 + 4));
    if (smem64 <= 160 * 1024 && (long)L * 3 * d * 4 < (1L << 31)) {
      static size_t attr64 = 0;
      if (smem64 > attr64) {
        ASW_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_mfma64_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem64));
        attr64 = smem64;
      }
      dim3 grid(cdiv(L, BQ), nhead, B);
      ProfScope prof(s, "attention_mfma64", 4.0 * B * nhead * (double)L * L * AD);
      hipLaunchKernelGGL(attention_mfma64_kernel, grid, dim3(256), smem64, s, qkv, L, LP64, d, ctx);
      ASW_LAUNCH_CHECK();
      return ASW_OK;
    }
  }
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
