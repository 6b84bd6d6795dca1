// mfma_util.h -- device helpers shared by the MFMA kernels (convgemm.hip, resstack.hip): vector types,
// the buffer-descriptor activation load and the fp32 -> fp16 hi / lo operand split of the f16x3 mode.
#pragma once
#include "asw_common.h"

namespace asw_mfma {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Activation loads go through a raw buffer descriptor: an offset outside [0, bytes) -- the
// zero padding of the convolution, rows past the end of the sequence, or a disabled lane --
// makes the hardware return zeros, so the staging loads carry no branch and no
// select.  (With plain pointers hipcc turns "load, then select zero" back into a branch around
// the load and waits vmcnt(0) between loads: a full memory latency per K chunk, exposed.)
typedef int intx4 __attribute__((ext_vector_type(4)));
typedef float floatx4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t act_rsrc(const float* base, long elems) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)(elems * 4), 0x00020000);
}
__device__ __forceinline__ float4 act_load4(__amdgpu_buffer_rsrc_t r, long elem, bool ok) {
  // disabled / negative -> 0x80000000: beyond any descriptor (bytes < 2^31) without wrapping
  const intx4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (ok && elem >= 0) ? (int)(elem * 4) : (int)0x80000000, 0, 0);
  const floatx4v f = __builtin_bit_cast(floatx4v, v);
  return make_float4(f[0], f[1], f[2], f[3]);
}

// x -> hi + lo with two packed round-toward-zero conversions per pair (v_cvt_pkrtz_f16_f32:
// finite overflow saturates at +-65504 instead of becoming inf).  hi carries 11 bits, the
// remainder x - hi is exact in fp32, lo carries its next 11 bits truncated: |x - hi - lo| < 2^-20 |x|.
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split4(const float4 x, half4& hi, half4& lo) {
  const fp16x2 h01 = __builtin_amdgcn_cvt_pkrtz(x.x, x.y);
  const fp16x2 h23 = __builtin_amdgcn_cvt_pkrtz(x.z, x.w);
  const fp16x2 l01 = __builtin_amdgcn_cvt_pkrtz(x.x - (float)h01[0], x.y - (float)h01[1]);
  const fp16x2 l23 = __builtin_amdgcn_cvt_pkrtz(x.z - (float)h23[0], x.w - (float)h23[1]);
  union { fp16x2 v[2]; half4 h; } uh, ul;
  uh.v[0] = h01; uh.v[1] = h23;
  ul.v[0] = l01; ul.v[1] = l23;
  hi = uh.h;
  lo = ul.h;
}

// Single-pass f16 mode (precision 2: one MFMA per product, the hi halves only): the hi half is rounded to
// nearest instead of truncated -- with no lo half to catch the remainder, truncation would bias every
// product low by ~2^-12.  lo is still produced (the C = 64 residual layer reads its residual from the image).
__device__ __forceinline__ void split4_rn(const float4 x, half4& hi, half4& lo) {
  const _Float16 h0 = (_Float16)x.x, h1 = (_Float16)x.y, h2 = (_Float16)x.z, h3 = (_Float16)x.w;
  const fp16x2 l01 = __builtin_amdgcn_cvt_pkrtz(x.x - (float)h0, x.y - (float)h1);
  const fp16x2 l23 = __builtin_amdgcn_cvt_pkrtz(x.z - (float)h2, x.w - (float)h3);
  hi = half4{h0, h1, h2, h3};
  union { fp16x2 v[2]; half4 h; } ul;
  ul.v[0] = l01; ul.v[1] = l23;
  lo = ul.h;
}
template <int NTERM>
__device__ __forceinline__ void split4t(const float4 x, half4& hi, half4& lo) {
  if constexpr (NTERM == 1) split4_rn(x, hi, lo);
  else split4(x, hi, lo);
}

}  // namespace asw_mfma
