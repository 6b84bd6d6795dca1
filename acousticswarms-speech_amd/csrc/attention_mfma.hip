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
#include <cstdlib>

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


// ---- short sequences on the f16 matrix pipe (f16x3 split operands) --------------------------
// Same structure as attention_mfma64_kernel (64 queries per workgroup, exact fp32 softmax over a score
// row that stays in LDS), with both products on v_mfma_f32_32x32x16_f16: every fp32 operand x is
// hi = fp16(x) + lo = fp16(x - hi) and a product is lo*hi + hi*lo + hi*hi with fp32 accumulation
// (operands good to 2^-21; the f32 MFMA runs at 1/16 of this pipe's rate, i.e. 3/16 of it per product).
//   * Q and the K tile are staged as fp16 hi / lo images, row = 128 hi + 128 lo halves + 16 B pad
//     (528 B: the 16-byte fragment reads of 32 consecutive rows are bank-conflict free);
//   * S = (Q / sqrt(hd)) K^T goes to LDS in fp32, the softmax runs there as before, and each row is
//     then REPLACED by its probabilities as fp16 hi | lo (same bytes: 2 + 2 per element), read back as
//     the A operand of O = P V;
//   * V stays row-major [key][hd] (coalesced staging): the B operand of O = P V needs, per hd column,
//     8 consecutive KEYS, which ds_read_b64_tr_b16 delivers from the row-major image (per 16-lane
//     group a 4-row x 16-column block, column-major); V rows are 256 B of hi (or lo) + 64 B pad
//     = 320 B, which keeps those reads conflict-free (bank of row q, 8-byte chunk p: 16 q + 2 p + 8 g).
typedef _Float16 ahalf8 __attribute__((ext_vector_type(8)));
typedef _Float16 ahalf4 __attribute__((ext_vector_type(4)));
typedef __fp16 afp16x2 __attribute__((ext_vector_type(2)));
typedef __fp16 afp16x4 __attribute__((ext_vector_type(4)));

constexpr int QRS = 528;          // bytes per Q / K image row (hi 256 | lo 256 | pad 16)
constexpr int VRS = 320;          // bytes per V image row (256 + 64 pad); hi image, then lo image

// hi by packed round-toward-zero conversion, lo = the remainder ROUNDED TO NEAREST (|x - hi - lo| <= 2^-23 |x|, no
// one-sided bias): the softmax exponentiates the score error, so the attention products keep the extra bit that
// the truncating split of the convolution kernels (convgemm.hip split4) gives up
__device__ __forceinline__ void split4h(const float4 x, ahalf4& hi, ahalf4& lo) {
  const afp16x2 h01 = __builtin_amdgcn_cvt_pkrtz(x.x, x.y), h23 = __builtin_amdgcn_cvt_pkrtz(x.z, x.w);
  union { afp16x2 v[2]; ahalf4 h; } uh;
  uh.v[0] = h01; uh.v[1] = h23;
  hi = uh.h;
  lo = ahalf4{(_Float16)(x.x - (float)h01[0]), (_Float16)(x.y - (float)h01[1]), (_Float16)(x.z - (float)h23[0]),
              (_Float16)(x.w - (float)h23[1])};
}

// transposed fragment: 8 consecutive k (rows of the row-major image) of this lane's column
__device__ __forceinline__ ahalf8 tr_frag(const char* blk_lo4, const char* blk_hi4) {
  typedef __attribute__((address_space(3))) afp16x4 lds_h4;
  const afp16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4*)(blk_lo4));
  const afp16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4*)(blk_hi4));
  union { afp16x4 v[2]; ahalf8 h; } u;
  u.v[0] = a; u.v[1] = b;
  return u.h;
}

__global__ __launch_bounds__(256) void attention_mfma16_kernel(const float* __restrict__ qkv, int L, int LP, int d,
                                                               float* __restrict__ ctx, int pf) {
  extern __shared__ __align__(16) float smem[];
  const int RSP = LP * 4 + 16;               // bytes per score / probability row
  char* Qi = reinterpret_cast<char*>(smem);  // [BQ] x QRS
  char* KV = Qi + BQ * QRS;                  // K tile [BK] x QRS, then V tile: hi [BK] x VRS | lo [BK] x VRS
  char* Pb = KV + 2 * BK * VRS;              // [BQ] x RSP: fp32 scores, then fp16 hi | lo probabilities
  static_assert(2 * BK * VRS >= BK * QRS, "the V images cover the K tile");
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * BQ;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const float scale = 1.0f / sqrtf((float)AD);
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(qkv + (long)b * L * 3 * d), 0, L * 3 * d * 4, 0x00020000);
  // Tiles of 64 rows x 128 floats travel global -> registers -> (split) -> LDS; the loads of the NEXT tile are issued
  // before the MFMAs of the current one (8 float4 per thread in flight), so with one workgroup per CU the memory
  // latency of a tile hides behind the products of the previous one (and the first V tile behind the softmax).
  constexpr int TV = BK * (AD / 4) / 256;               // float4 per thread and tile
  auto issue = [&](float4 (&r)[TV], int row0, int which) {
#pragma unroll
    for (int u = 0; u < TV; ++u) {
      const int i = tid + u * 256;
      const int rr = i / (AD / 4), c4 = i - rr * (AD / 4);
      r[u] = row_load4(rs, (long)(row0 + rr) * 3 * d + which * d + h * AD + c4 * 4, row0 + rr < L);
    }
  };
  auto deposit = [&](const float4 (&r)[TV], char* hi_img, char* lo_img, int rstride, float mul) {
#pragma unroll
    for (int u = 0; u < TV; ++u) {
      const int i = tid + u * 256;
      const int rr = i / (AD / 4), c4 = i - rr * (AD / 4);
      float4 v = r[u];
      v.x *= mul; v.y *= mul; v.z *= mul; v.w *= mul;
      ahalf4 hi, lo;
      split4h(v, hi, lo);
      *reinterpret_cast<ahalf4*>(hi_img + rr * rstride + c4 * 8) = hi;
      *reinterpret_cast<ahalf4*>(lo_img + rr * rstride + c4 * 8) = lo;
    }
  };
  float4 stg[TV];
  issue(stg, q0, 0);
  deposit(stg, Qi, Qi + 256, QRS, scale);
  // pf: bit 0 = K tiles, bit 1 = V tiles requested one tile ahead (3 in production; the launcher's ASW_ATTN_PF
  // selects other combinations for measurements).  The flags are RUNTIME values on purpose: with the two prefetches
  // as straight-line code hipcc (ROCm 7.2) produced a kernel that returned wrong values for every shape, while each
  // of the four flag combinations of this form is correct (tests/micro/att_dbg.py; test_attention covers pf = 3).
  if (pf & 1) issue(stg, 0, 1);
  // ---- phase 1: S = (Q/sqrt(hd)) K^T; wave w owns score tile (query tile w>>1, key tile w&1)
  const int qi = wid >> 1, kj = wid & 1;
  for (int k0 = 0; k0 < LP; k0 += BK) {
    __syncthreads();                                   // previous K tile consumed
    if (!(pf & 1)) issue(stg, k0, 1);
    deposit(stg, KV, KV + 256, QRS, 1.0f);
    __syncthreads();
    if (k0 + BK < LP) { if (pf & 1) issue(stg, k0 + BK, 1); }      // next K tile
    else if (pf & 2) issue(stg, 0, 2);                 // first V tile: in flight under the softmax
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const char* qa = Qi + (qi * 32 + lr) * QRS + lh * 16;
    const char* kb = KV + (kj * 32 + lr) * QRS + lh * 16;
#pragma unroll
    for (int ks = 0; ks < AD / 16; ++ks) {
      const ahalf8 qh = *reinterpret_cast<const ahalf8*>(qa + ks * 32), ql = *reinterpret_cast<const ahalf8*>(qa + 256 + ks * 32);
      const ahalf8 kh = *reinterpret_cast<const ahalf8*>(kb + ks * 32), kl = *reinterpret_cast<const ahalf8*>(kb + 256 + ks * 32);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ql, kh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh, kl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh, kh, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = qi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      *reinterpret_cast<float*>(Pb + (size_t)row * RSP + (k0 + kj * 32 + lr) * 4) = acc[r];
    }
  }
  __syncthreads();
  // ---- row softmax over the L valid keys (fp32, 4 lanes per query row), then the row is rewritten in place as
  //      fp16 hi | lo probabilities: every lane first pulls its share of the row into registers
  {
    const int row = tid >> 2, sub = tid & 3;
    char* pr = Pb + (size_t)row * RSP;
    constexpr int MAXV = 352 / 16;                       // float4 groups per lane (LP <= 352)
    float4 v[MAXV];
    const int nv = LP / 16;                               // LP is a multiple of 64
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      if (i < nv) {
        v[i] = *reinterpret_cast<const float4*>(pr + (i * 4 + sub) * 16);
        const int j = (i * 4 + sub) * 4;
        if (j + 0 < L) m = fmaxf(m, v[i].x);
        if (j + 1 < L) m = fmaxf(m, v[i].y);
        if (j + 2 < L) m = fmaxf(m, v[i].z);
        if (j + 3 < L) m = fmaxf(m, v[i].w);
      }
    }
    m = fmaxf(m, __shfl_xor(m, 1, 64));
    m = fmaxf(m, __shfl_xor(m, 2, 64));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      if (i < nv) {
        const int j = (i * 4 + sub) * 4;
        v[i].x = j + 0 < L ? expf(v[i].x - m) : 0.f;
        v[i].y = j + 1 < L ? expf(v[i].y - m) : 0.f;
        v[i].z = j + 2 < L ? expf(v[i].z - m) : 0.f;
        v[i].w = j + 3 < L ? expf(v[i].w - m) : 0.f;
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      }
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    // probabilities are stored times 2^12 (undone in the output): the lo half of a small probability would otherwise
    // fall into the fp16 subnormals (p = 1/300 has lo ~ 1e-6) and lose the bits it is there to carry
    const float inv = 4096.0f / s;
    // the four lanes of a row have read all of it (same wave: the shuffles above order the reads before these writes)
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      if (i < nv) {
        const float4 pv = make_float4(v[i].x * inv, v[i].y * inv, v[i].z * inv, v[i].w * inv);
        ahalf4 hi, lo;
        split4h(pv, hi, lo);
        const int j = (i * 4 + sub) * 4;
        *reinterpret_cast<ahalf4*>(pr + j * 2) = hi;
        *reinterpret_cast<ahalf4*>(pr + LP * 2 + j * 2) = lo;
      }
    }
  }
  // ---- phase 2: O = P V; wave w owns output columns [32w, 32w+32) of both query tiles
  floatx16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  char* Vh = KV;
  char* Vl = KV + BK * VRS;
  // transposed reads: 16-lane group g = lane >> 4 covers columns 16 (g & 1) .. +15 of this wave's 32, k half (g >> 1);
  // lane 4q + p of a group addresses row q, 8-byte chunk p of the 4-row x 16-column block
  const int g = lane >> 4, gq = (lane & 15) >> 2, gp = lane & 3;
  const int tr_off = (8 * (g >> 1) + gq) * VRS + (wid * 32 + 16 * (g & 1) + 4 * gp) * 2;
  for (int k0 = 0; k0 < LP; k0 += BK) {
    __syncthreads();                                   // probabilities written / previous V tile consumed
    if (!(pf & 2)) issue(stg, k0, 2);
    deposit(stg, Vh, Vl, VRS, 1.0f);
    __syncthreads();
    // The two image bases go through an opaque asm placed after the barrier: the transposed reads cannot be scheduled
    // before it, and every read keeps a small immediate offset from its own base register.
    const char* vh_base = Vh + tr_off;
    const char* vl_base = Vl + tr_off;
    asm volatile("" : "+v"(vh_base), "+v"(vl_base));
    if ((pf & 2) && k0 + BK < LP) issue(stg, k0 + BK, 2);          // next V tile
    const char* p0 = Pb + (size_t)lr * RSP + (k0 + lh * 8) * 2;
    const char* p1 = p0 + (size_t)32 * RSP;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const ahalf8 a0h = *reinterpret_cast<const ahalf8*>(p0 + ks * 32), a0l = *reinterpret_cast<const ahalf8*>(p0 + LP * 2 + ks * 32);
      const ahalf8 a1h = *reinterpret_cast<const ahalf8*>(p1 + ks * 32), a1l = *reinterpret_cast<const ahalf8*>(p1 + LP * 2 + ks * 32);
      const int ro = ks * 16 * VRS;                    // rows 16 ks + 8 (g >> 1) + {0..3}, then + 4
      const ahalf8 vh = tr_frag(vh_base + ro, vh_base + ro + 4 * VRS);
      const ahalf8 vl = tr_frag(vl_base + ro, vl_base + ro + 4 * VRS);
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, vh, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, vh, o1, 0, 0, 0);
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, vl, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, vl, o1, 0, 0, 0);
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, vh, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, vh, o1, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int q = q0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (q < L) ctx[((long)b * L + q) * d + h * AD + wid * 32 + lr] = o0[r] * (1.0f / 4096.0f);
    if (q + 32 < L) ctx[((long)b * L + q + 32) * d + h * AD + wid * 32 + lr] = o1[r] * (1.0f / 4096.0f);
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
int attention_mfma(const float* qkv, int B, int L, int d, int nhead, float* ctx, hipStream_t s, int precision) {
  if (d / nhead != AD) return 1;
  static const bool no16 = getenv("ASW_NO_ATTN16") != nullptr;          // A/B switch for measurements
  if (precision >= 1 && !no16) {
    // f16x3 arithmetic: short sequences only (the probability rows replace the score rows in LDS)
    const int LP = cdiv(L, BK) * BK;
    const size_t smem16 = (size_t)BQ * QRS + (size_t)2 * BK * VRS + (size_t)BQ * (LP * 4 + 16);
    if (LP <= 352 && smem16 <= 160 * 1024 && (long)L * 3 * d * 4 < (1L << 31)) {
      static SmemAttr attr16;                           // per device
      if (int rc = attr16.ensure(reinterpret_cast<const void*>(attention_mfma16_kernel), smem16)) return rc;
      dim3 grid(cdiv(L, BQ), nhead, B);
      ProfScope prof(s, "attention_mfma16", 4.0 * B * nhead * (double)L * L * AD);
      static const int pf = getenv("ASW_ATTN_PF") ? atoi(getenv("ASW_ATTN_PF")) & 3 : 3;
      hipLaunchKernelGGL(attention_mfma16_kernel, grid, dim3(256), smem16, s, qkv, L, LP, d, ctx, pf);
      ASW_LAUNCH_CHECK();
      return ASW_OK;
    }
  }
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
