// attention_mfma.hip -- bottleneck self-attention on the f32 MFMA pipe for the shapes the
// spot network produces: sequence L = T/256 (188 at T = 48 000, 563 at T = 144 000), head_dim 128.
// nn.MultiheadAttention core inside nn.TransformerEncoderLayer
// (sep/training/SpeakerLocalization/network.py:254): ctx = softmax(Q K^T / sqrt(hd)) V.
//
// One workgroup (4 waves) per (batch item, head, 64-query tile); exact fp32 arithmetic
// (v_mfma_f32_32x32x2_f32 is an fmaf chain), K and V tiles of 64 keys staged row-major through
// a buffer descriptor (rows past the sequence read as zeros, no branch):
//   L <= 352: the whole score row of the tile stays in LDS -> exact softmax, one Q K^T;
//   longer:   key-tiled two-pass kernel (row statistics first, scores recomputed).
// Other head sizes fall back to the flash-style VALU kernel in misc_kernels.hip.
#include "asw_common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int AD = 128;           // head_dim
constexpr int LDQ = AD + 4;       // Q / K row stride (floats)

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

// ---- short sequences (L <= 352: T = 48 000 gives L = 188): whole score row in LDS ----------
// 64 queries per workgroup; the 64 x 64 score tile of a key block is four MFMA tiles (one per
// wave); V is staged row-major with float4 stores, so the B operand of O = P V is read as four
// ds_read_b32 per four MFMAs (no transposition through scalar LDS stores).
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

// ---- long sequences (L > 352: T = 144 000 gives L = 563): key-tiled, two passes ------------
// The score row of 64 queries no longer fits LDS beside Q, K and V, so the row statistics are
// taken first and the scores recomputed: pass A walks the key tiles keeping each query's
// running maximum m and sum l = sum exp(s - m) (rescaled when m grows); pass B recomputes the
// same score tiles, turns them into probabilities exp(s - m) / l and accumulates O = P V.
// Costs one extra Q K^T (a third more MFMAs) but stages K and V row-major with float4 stores,
// keeps all four waves busy and reuses every K / V tile for 64 queries instead of 32.
__global__ __launch_bounds__(256) void attention_mfma_flash_kernel(const float* __restrict__ qkv, int L, int d,
                                                                   float* __restrict__ ctx) {
  extern __shared__ __align__(16) float smem[];
  constexpr int LDS_P = BK + 4;
  float* Qs = smem;                          // [BQ][LDQ]
  float* KV = Qs + BQ * LDQ;                 // K tile / V tile [BK][LDQ]
  float* Pt = KV + BK * LDQ;                 // score / probability tile [BQ][BK + 4]
  float* rm = Pt + BQ * LDS_P;               // [BQ] running max
  float* rl = rm + BQ;                       // [BQ] running sum
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * BQ;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const float scale = 1.0f / sqrtf((float)AD);
  const int LPk = (L + BK - 1) / BK * BK;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(qkv + (long)b * L * 3 * d), 0, L * 3 * d * 4, 0x00020000);
  for (int i = tid; i < BQ * (AD / 4); i += 256) {
    const int r = i / (AD / 4), c4 = i - r * (AD / 4);
    float4 v = row_load4(rs, (long)(q0 + r) * 3 * d + h * AD + c4 * 4, q0 + r < L);
    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
    *reinterpret_cast<float4*>(Qs + r * LDQ + c4 * 4) = v;
  }
  if (tid < BQ) { rm[tid] = -INFINITY; rl[tid] = 0.f; }
  const int qi = wid >> 1, kj = wid & 1;
  auto stage = [&](int k0, int which) {      // which: 1 = K, 2 = V
    for (int i = tid; i < BK * (AD / 4); i += 256) {
      const int r = i / (AD / 4), c4 = i - r * (AD / 4);
      *reinterpret_cast<float4*>(KV + r * LDQ + c4 * 4) =
          row_load4(rs, (long)(k0 + r) * 3 * d + which * d + h * AD + c4 * 4, k0 + r < L);
    }
  };
  auto score_tile = [&](int k0) {            // Pt = (Q/sqrt(hd)) K^T for this key tile; padded keys -> -inf
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = mma_row(Qs + (qi * 32 + lr) * LDQ + lh * 4, KV + (kj * 32 + lr) * LDQ + lh * 4, AD / 8, acc);
    const bool valid = k0 + kj * 32 + lr < L;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = qi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      Pt[row * LDS_P + kj * 32 + lr] = valid ? acc[r] : -INFINITY;
    }
  };
  // ---- pass A: running max / sum per query row (4 lanes per row)
  for (int k0 = 0; k0 < LPk; k0 += BK) {
    __syncthreads();
    stage(k0, 1);
    __syncthreads();
    score_tile(k0);
    __syncthreads();
    {
      const int row = tid >> 2, sub = tid & 3;
      const float* pr = Pt + row * LDS_P;
      float m = -INFINITY;
      for (int j = sub; j < BK; j += 4) m = fmaxf(m, pr[j]);
      m = fmaxf(m, __shfl_xor(m, 1, 64));
      m = fmaxf(m, __shfl_xor(m, 2, 64));
      const float m_old = rm[row], m_new = fmaxf(m_old, m);
      float sum = 0.f;
      for (int j = sub; j < BK; j += 4) sum += expf(pr[j] - m_new);       // exp(-inf) = 0 for padded keys
      sum += __shfl_xor(sum, 1, 64);
      sum += __shfl_xor(sum, 2, 64);
      if (sub == 0) {
        rl[row] = rl[row] * expf(m_old - m_new) + sum;
        rm[row] = m_new;
      }
    }
  }
  // ---- pass B: probabilities and O = P V; wave w owns output columns [32w, 32w+32)
  floatx16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  for (int k0 = 0; k0 < LPk; k0 += BK) {
    __syncthreads();
    stage(k0, 1);
    __syncthreads();
    score_tile(k0);
    __syncthreads();
    {
      const int row = tid >> 2, sub = tid & 3;
      float* pr = Pt + row * LDS_P;
      const float m = rm[row], inv = 1.0f / rl[row];
      for (int j = sub; j < BK; j += 4) pr[j] = expf(pr[j] - m) * inv;
    }
    stage(k0, 2);                            // K tile is consumed (score_tile done before the barrier above)
    __syncthreads();
    const float* p0 = Pt + lr * LDS_P + lh * 4;
    const float* p1 = p0 + 32 * LDS_P;
    const float* vb = KV + (lh * 4) * LDQ + wid * 32 + lr;
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
      static SmemAttr attr64;                           // per device
      if (int rc = attr64.ensure(reinterpret_cast<const void*>(attention_mfma64_kernel), smem64)) return rc;
      dim3 grid(cdiv(L, BQ), nhead, B);
      ProfScope prof(s, "attention_mfma64", 4.0 * B * nhead * (double)L * L * AD);
      hipLaunchKernelGGL(attention_mfma64_kernel, grid, dim3(256), smem64, s, qkv, L, LP64, d, ctx);
      ASW_LAUNCH_CHECK();
      return ASW_OK;
    }
  }
  if ((long)L * 3 * d * 4 < (1L << 31)) {
    // long sequences: key-tiled two-pass kernel (any L)
    constexpr size_t smemf = sizeof(float) * ((size_t)BQ * LDQ + (size_t)BK * LDQ + (size_t)BQ * (BK + 4) + 2 * BQ);
    static SmemAttr attrf;                              // per device
    if (int rc = attrf.ensure(reinterpret_cast<const void*>(attention_mfma_flash_kernel), smemf)) return rc;
    dim3 grid(cdiv(L, BQ), nhead, B);
    ProfScope prof(s, "attention_mfma_flash", 4.0 * B * nhead * (double)L * L * AD);
    hipLaunchKernelGGL(attention_mfma_flash_kernel, grid, dim3(256), smemf, s, qkv, L, d, ctx);
    ASW_LAUNCH_CHECK();
    return ASW_OK;
  }
  return 1;                                            // row descriptor too large: VALU fallback
}
}  // namespace asw
