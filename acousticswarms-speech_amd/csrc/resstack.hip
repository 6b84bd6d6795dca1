// resstack.hip -- a stack of 64-channel DilatedResidualLayers in ONE launch on the gfx950 matrix cores
// (f16x3 split-operand arithmetic), the layers' intermediate tensors never leaving LDS.
//
// Replaces (reference file:line): DilatedResidualLayer / DilatedResidualSequence
// (sep/training/SpeakerLocalization/network.py:50-82, sep/training/SpeakerSeparation/network.py, same
// classes): out_i = LayerNorm(ReLU(conv_{d_i}(x_i) + b_i) + x_i), x_{i+1} = out_i.
//
// Why a second residual kernel.  At C = 64 the halo-staged layer of convgemm.hip (resconv16_kernel)
// moves 112 FLOP per byte of HBM traffic -- exactly the ridge of this chip -- and spends 44 % of a
// workgroup's time in its epilogue (accumulators -> LDS slab -> row-wise LayerNorm -> store).  Two changes:
//
//  * The accumulators are TRANSPOSED: the weight fragment is the A operand of v_mfma_f32_32x32x16_f16
//    and the activation rows are the B operand, so a lane holds, for ONE time row (lane & 31), 16
//    channels of each 32-channel block: (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  A wave owns all 64
//    channels of its rows, hence LayerNorm is 32 in-lane adds + ONE cross-lane shuffle, bias / ReLU /
//    residual are applied in registers, four consecutive channels of a row are one float4 (global
//    store) or one 8-byte hi / lo pair (LDS image): no slab, no barriers inside the epilogue.
//  * Consecutive layers with small dilation are FUSED: the workgroup stages the rows the whole stack
//    needs (output rows + the summed halos) once, layer i writes its output -- already split to fp16
//    hi / lo -- over the LDS image it has just consumed, layer i + 1 reads it from there.  With
//    R0 = 256 rows computed by layer 0 the pair (dilation 1, dilation 7; halo 3 + 21) produces
//    256 - 42 = 214 finished rows per workgroup: 12 % more MFMA work, half the HBM round trips.
//
// Large dilations (49) cannot be fused (the halo would be the tile); they run through the same
// kernel as single layers on polyphase row sets (x[phase + dil * j] is a dilation-1 convolution in j),
// which in the transposed layout costs nothing extra: every lane stores its own row anyway.
//
// Image layout (as resconv16_kernel): one row = 64 channels as 128 B of fp16 hi + 128 B of fp16 lo +
// 16 B pad (272 B: the 16-byte fragment reads of 32 consecutive rows are bank-conflict free).  Weights
// never touch LDS: MFMA-fragment order (asw_pack_fragments_f16), one coalesced 1 KiB load per fragment,
// QD k-steps ahead.
#include <cstdlib>
#include <type_traits>

#include "asw_common.h"
#include "mfma_util.h"

namespace {
using namespace asw_mfma;

constexpr int RS = 272;                  // bytes per image row
constexpr int C = 64;
constexpr int MAXL = 3;

struct LayerDev {
  const half8* Wh;
  const half8* Wl;
  const float* bias;
  const float* gamma;
  const float* beta;
  int dil, pad;
  int halo;                              // rows of halo still needed AFTER this layer (sum of later pads)
  int nfrag;                             // 32-row fragments this layer computes
  float scale;                           // 2^-w_shift
};

#ifdef ASW_PHASE_TIMING
// Diagnostic build only (tests/micro/resstack_phases.py): cycles wave 0 of every workgroup spends per phase, summed
// over workgroups: [0] staging, [1] layer-0 k-loop, [2] layer-0 epilogue + hand-over, [3] layer-1 k-loop,
// [4] layer-1 epilogue + stores, [5] workgroups.
__device__ unsigned long long g_rs_cycles[6] = {0, 0, 0, 0, 0, 0};
#define RS_MARK(var) const unsigned long long var = __builtin_readcyclecounter()
#else
#define RS_MARK(var)
#endif

struct KArgs {
  const float* x;
  float* out;
  const float* glu_raw;
  const float* glu_mr;
  const float* glu_gamma;
  const float* glu_beta;
  float* glu_out;                        // optional: the GroupNorm + GLU rows of this tile's own output range, fp32
  int B, T, taps, ntile, BM, img_rows;
  float eps;
  LayerDev L[MAXL];
};

// taps x 4 k-steps of one layer for NF row fragments of this wave: acc[i][cb] += W(cb) . X(i)^T
template <int NF, int QD, int NTERM>
__device__ __forceinline__ void kloop(floatx16 (&acc)[NF][2], const char* img, const int (&xb)[NF], int taps, int tapstep,
                                      const half8* __restrict__ Wh, const half8* __restrict__ Wl, int lane) {
  auto wload = [&](int kg, half8 (&h)[2], half8 (&l)[2]) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const long o = ((long)kg * 2 + cb) * 64 + lane;
      h[cb] = Wh[o];
      if (NTERM == 3) l[cb] = Wl[o];
    }
  };
  auto xload = [&](int tap, int ks, half8 (&h)[NF], half8 (&l)[NF]) {
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const char* q = img + xb[i] + tap * tapstep + ks * 32;
      h[i] = *reinterpret_cast<const half8*>(q);
      if (NTERM == 3) l[i] = *reinterpret_cast<const half8*>(q + 128);
    }
  };
  half8 wh[QD][2], wl[QD][2];
  half8 xh[2][NF], xl[2][NF];
#pragma unroll
  for (int q = 0; q < QD; ++q) wload(q, wh[q], wl[q]);
  xload(0, 0, xh[0], xl[0]);
  // Every load in the loop body is UNCONDITIONAL (past-the-end indices are clamped to the last fragment, which
  // is simply fetched again): a load inside an `if` sits in its own basic block, and at the join hipcc waits
  // vmcnt(0) -- i.e. for the weight fragments requested one k-step earlier -- once per tap, instead of counting.
  const int nks = taps * 4;
  for (int tap = 0; tap < taps; ++tap) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int nk = (ks + 1) & 3;
      int nt = tap + (ks == 3 ? 1 : 0);
      nt = nt < taps ? nt : taps - 1;
      xload(nt, nk, xh[(ks + 1) & 1], xl[(ks + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);       // (the next k-step's LDS reads go out BEFORE this k-step's MFMAs, not after)
      const int s = ks % QD, xbuf = ks & 1;
#pragma unroll
      for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          if (NTERM == 3) {
            // same term order as the row-major kernels: x_lo * w_hi, x_hi * w_lo, x_hi * w_hi
            acc[i][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s][cb], xl[xbuf][i], acc[i][cb], 0, 0, 0);
            acc[i][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[s][cb], xh[xbuf][i], acc[i][cb], 0, 0, 0);
          }
          acc[i][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s][cb], xh[xbuf][i], acc[i][cb], 0, 0, 0);
        }
      int kg = tap * 4 + ks + QD;
      kg = kg < nks ? kg : nks - 1;
      wload(kg, wh[s], wl[s]);
      // pin the k-step order: left alone, hipcc sinks the four k-steps' weight loads to the end of the tap body and
      // waits for them at the top of the next one -- the L2 latency of the weight stream exposed once per tap
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// NL fused layers (contiguous row tiles) or one layer on polyphase row sets (POLY, NL == 1).
// NW waves, each owning TM 32-row fragments x all 64 channels.
template <int NL, int NW, int TM, int PH, bool POLY, bool GLU, int NTERM, int QD>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(TM >= 4 ? 1 : 2)))
void resstack64_kernel(const KArgs p) {
  static_assert(!POLY || NL == 1, "polyphase row sets: single layers only");
  static_assert(QD == 2 || QD == 4, "weight prefetch depth in k-steps");
  constexpr int NTHR = 64 * NW, SROWS = NTHR / 16;
  constexpr int R0 = 32 * NW * TM;                      // rows layer 0 computes
  constexpr int BMJ = R0 / PH;                          // polyphase: rows of j per phase
  static_assert(BMJ % 32 == 0, "a row fragment must stay inside one phase");

  extern __shared__ __align__(16) char smem[];
  char* img = smem;
  float* tab = reinterpret_cast<float*>(smem + (size_t)p.img_rows * RS);       // [NL][bias | gamma | beta][64]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // XCD-aware tile order: workgroup L runs on XCD L % 8 (round-robin dispatch); XCD x walks a CONTIGUOUS
  // run of (item, row tile) pairs, so the halo rows two neighbouring tiles share meet in one L2.
  const int total = p.B * p.ntile;
  const int per = (total + 7) >> 3;
  const int idx = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (idx >= total) return;
  const int b = idx / p.ntile, tile = idx - b * p.ntile;
  const int T = p.T, taps = p.taps;
  const int dil0 = p.L[0].dil, pad0 = p.L[0].pad;
  const int n_pb = POLY ? (dil0 + PH - 1) / PH : 1;
  const int jb = POLY ? tile / n_pb : 0, pb = POLY ? tile - jb * n_pb : 0;
  const int m0 = tile * p.BM;
  const int RJ = BMJ + taps - 1;                        // polyphase: image rows per phase
  const int R_img = POLY ? PH * RJ : R0 + 2 * pad0;

  RS_MARK(t_start);
  // ---- per-layer vectors -> LDS
  for (int i = tid; i < NL * 3 * 16; i += NTHR) {
    const int li = i / 48, w = (i - li * 48) / 16, c4 = i & 15;
    const float* src = w == 0 ? p.L[li].bias : w == 1 ? p.L[li].gamma : p.L[li].beta;
    reinterpret_cast<float4*>(tab)[i] = *reinterpret_cast<const float4*>(src + c4 * 4);
  }
  // ---- stage + split the input rows of layer 0 (16 threads per row, 8 rows per thread in flight)
  {
    const __amdgpu_buffer_rsrc_t rX = GLU ? act_rsrc(p.glu_raw + (long)b * T * 2 * C, (long)T * 2 * C)
                                          : act_rsrc(p.x + (long)b * T * C, (long)T * C);
    const int srow = tid >> 4, sc4 = tid & 15;
    float gm0 = 0.f, gr0 = 0.f, gm1 = 0.f, gr1 = 0.f;
    float4 gga, gba, ggg, gbg;
    if (GLU) {
      gm0 = p.glu_mr[b * 4 + 0]; gr0 = p.glu_mr[b * 4 + 1]; gm1 = p.glu_mr[b * 4 + 2]; gr1 = p.glu_mr[b * 4 + 3];
      gga = *reinterpret_cast<const float4*>(p.glu_gamma + sc4 * 4);
      gba = *reinterpret_cast<const float4*>(p.glu_beta + sc4 * 4);
      ggg = *reinterpret_cast<const float4*>(p.glu_gamma + C + sc4 * 4);
      gbg = *reinterpret_cast<const float4*>(p.glu_beta + C + sc4 * 4);
    }
    // the whole image is requested before the first row is converted: one exposed memory latency per tile (with 8
    // rows per thread in flight the three rounds of a 262-row image took 17 % of a workgroup's time)
    constexpr int SU = GLU ? 9 : 18;
    for (int r0 = 0; r0 < R_img; r0 += SROWS * SU) {
      float4 buf[SU];
      float4 gate[GLU ? SU : 1];
      bool okr[GLU ? SU : 1];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int row = r0 + u * SROWS + srow;
        int g;
        bool ok = row < R_img;
        if (!POLY) {
          g = m0 - p.L[0].halo - pad0 + row;
        } else {
          const int ph = pb * PH + row / RJ;
          const int jj = jb * BMJ + row % RJ - (taps - 1) / 2;
          g = dil0 * jj + ph;
          ok = ok && ph < dil0 && jj >= 0;
        }
        ok = ok && g >= 0 && g < T;
        if (GLU) {
          buf[u] = act_load4(rX, (long)g * 2 * C + sc4 * 4, ok);
          gate[u] = act_load4(rX, (long)g * 2 * C + C + sc4 * 4, ok);
          okr[u] = ok;
        } else {
          buf[u] = act_load4(rX, (long)g * C + sc4 * 4, ok);
        }
      }
      if (GLU) {
        // the arithmetic of gn_glu_kernel, expression for expression; rows outside the sequence stay zero
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          float4 o;
#define ASW_GLU(f)                                                       \
  {                                                                      \
    const float gl = asw::gn_glu_value(buf[u].f, gate[u].f, gm0, gr0, gm1, gr1, gga.f, gba.f, ggg.f, gbg.f); \
    o.f = okr[u] ? gl : 0.f;                                             \
  }
          ASW_GLU(x) ASW_GLU(y) ASW_GLU(z) ASW_GLU(w)
#undef ASW_GLU
          buf[u] = o;
          // the normalised tensor is also a skip connection: each tile writes the rows of its own output range once
          if (p.glu_out && okr[u]) {
            const int g = m0 - p.L[0].halo - pad0 + r0 + u * SROWS + srow;
            if (g >= m0 && g < m0 + p.BM)
              *reinterpret_cast<float4*>(p.glu_out + ((long)b * T + g) * C + sc4 * 4) = o;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int row = r0 + u * SROWS + srow;
        if (row < R_img) {
          half4 hi, lo;
          split4t<NTERM>(buf[u], hi, lo);
          *reinterpret_cast<half4*>(img + row * RS + sc4 * 8) = hi;
          *reinterpret_cast<half4*>(img + row * RS + 128 + sc4 * 8) = lo;      // (the residual reads it in every mode)
        }
      }
    }
  }
  __syncthreads();
  RS_MARK(t_staged);
#ifdef ASW_PHASE_TIMING
  unsigned long long t_marks[2 * MAXL + 1];
  t_marks[0] = t_staged;
#endif

  const int h = lane >> 5;
  floatx16 acc[TM][2];
  auto layer = [&](auto lic) {
    constexpr int li = decltype(lic)::value;
    constexpr bool last = li == NL - 1;
    const LayerDev& Ld = p.L[li];
    const int tapstep = (POLY ? 1 : Ld.dil) * RS;
    const int nf = Ld.nfrag - wid * TM;                 // fragments of this wave that exist in this layer (<= 0: none)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][cb][r] = 0.f;        // (all of acc defined on every path: it stays in registers)
    // image row of output row `trow` at tap 0
    auto img_row = [&](int trow) { return POLY ? (trow / BMJ) * RJ + trow % BMJ : trow; };
    // this wave's first nf (<= TM) fragments: one k-loop instantiation per count (wave-uniform choice)
    auto run = [&](auto nfc) {
      constexpr int NF = decltype(nfc)::value;
      int xb[NF];
      floatx16 a[NF][2];
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        xb[i] = (img_row((wid * TM + i) * 32) + (lane & 31)) * RS + h * 16;
        a[i][0] = acc[i][0];
        a[i][1] = acc[i][1];
      }
      kloop<NF, QD, NTERM>(a, img, xb, taps, tapstep, Ld.Wh, Ld.Wl, lane);
#pragma unroll
      for (int i = 0; i < NF; ++i) { acc[i][0] = a[i][0]; acc[i][1] = a[i][1]; }
    };
    if (nf >= TM) run(std::integral_constant<int, TM>{});
    else if constexpr (TM > 1) {
      if (nf == 1) run(std::integral_constant<int, 1>{});
      if constexpr (TM > 2) {
        if (nf == 2) run(std::integral_constant<int, 2>{});
        if (nf == 3) run(std::integral_constant<int, 3>{});
      }
    }
#ifdef ASW_PHASE_TIMING
    {
      float sink = acc[0][0][0];
      asm volatile("" ::"v"(sink));                     // the stamp waits for the MFMAs
      t_marks[2 * li + 1] = __builtin_readcyclecounter();
    }
#endif
    // ---- epilogue in registers: + bias, ReLU, + residual (the layer's own input, from the image), LayerNorm
    const float* tb = tab + li * 192;
    const int cpad = POLY ? (taps - 1) / 2 : Ld.pad;    // image row of the residual = tap-0 row + centre tap
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (i < nf) {
        const int trow = (wid * TM + i) * 32 + (lane & 31);
        const char* rrow = img + (img_row(trow) + cpad) * RS;
        float s = 0.f;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int c0 = cb * 32 + q * 8 + h * 4;
            const float4 bi = *reinterpret_cast<const float4*>(tb + c0);
            const half4 rh = *reinterpret_cast<const half4*>(rrow + c0 * 2);
            const half4 rl = *reinterpret_cast<const half4*>(rrow + 128 + c0 * 2);
            const float bv[4] = {bi.x, bi.y, bi.z, bi.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float v = fmaxf(acc[i][cb][q * 4 + j] * Ld.scale + bv[j], 0.f);
              v += (float)rh[j] + (float)rl[j];
              acc[i][cb][q * 4 + j] = v;
              s += v;
            }
          }
        s += __shfl_xor(s, 32, 64);
        const float mean = s * (1.0f / C);
        float d = 0.f;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int r = 0; r < 16; ++r) { const float cx = acc[i][cb][r] - mean; d += cx * cx; }
        d += __shfl_xor(d, 32, 64);
        const float rstd = 1.0f / sqrtf(d * (1.0f / C) + p.eps);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int c0 = cb * 32 + q * 8 + h * 4;
            const float4 g4 = *reinterpret_cast<const float4*>(tb + 64 + c0);
            const float4 b4 = *reinterpret_cast<const float4*>(tb + 128 + c0);
            const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, be[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][cb][q * 4 + j] = (acc[i][cb][q * 4 + j] - mean) * rstd * gv[j] + be[j];
          }
      }
    }
    if constexpr (!last) {
      // the next layer's input replaces this layer's in the image: rows outside the sequence are its zero padding
      __syncthreads();                                  // every wave is done reading the old image
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (i < nf) {
          const int trow = (wid * TM + i) * 32 + (lane & 31);
          const int t = m0 - Ld.halo + trow;
          const bool inside = t >= 0 && t < T;
          char* wrow = img + trow * RS;
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int c0 = cb * 32 + q * 8 + h * 4;
              float4 v = make_float4(acc[i][cb][q * 4 + 0], acc[i][cb][q * 4 + 1], acc[i][cb][q * 4 + 2], acc[i][cb][q * 4 + 3]);
              if (!inside) v = make_float4(0.f, 0.f, 0.f, 0.f);
              half4 hi, lo;
              split4t<NTERM>(v, hi, lo);
              *reinterpret_cast<half4*>(wrow + c0 * 2) = hi;
              *reinterpret_cast<half4*>(wrow + 128 + c0 * 2) = lo;
            }
        }
      }
      __syncthreads();
    } else {
      float* __restrict__ ob = p.out + (long)b * T * C;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (i < nf) {
          const int trow = (wid * TM + i) * 32 + (lane & 31);
          int t;
          bool ok;
          if (!POLY) {
            t = m0 + trow;
            ok = trow < p.BM && t < T;
          } else {
            const int ph = pb * PH + trow / BMJ;
            t = dil0 * (jb * BMJ + trow % BMJ) + ph;
            ok = ph < dil0 && t < T;
          }
          if (ok) {
            float* orow = ob + (long)t * C;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int c0 = cb * 32 + q * 8 + h * 4;
                *reinterpret_cast<float4*>(orow + c0) =
                    make_float4(acc[i][cb][q * 4 + 0], acc[i][cb][q * 4 + 1], acc[i][cb][q * 4 + 2], acc[i][cb][q * 4 + 3]);
              }
          }
        }
      }
    }
  };
  layer(std::integral_constant<int, 0>{});
#ifdef ASW_PHASE_TIMING
  t_marks[2] = __builtin_readcyclecounter();
#endif
  if constexpr (NL > 1) layer(std::integral_constant<int, 1>{});
  if constexpr (NL > 2) layer(std::integral_constant<int, 2>{});
#ifdef ASW_PHASE_TIMING
  if (threadIdx.x == 0 && NL <= 2) {
    const unsigned long long t_end = __builtin_readcyclecounter();
    atomicAdd(&g_rs_cycles[0], t_staged - t_start);
    atomicAdd(&g_rs_cycles[1], t_marks[1] - t_marks[0]);
    if (NL == 1) {
      atomicAdd(&g_rs_cycles[4], t_end - t_marks[1]);
    } else {
      atomicAdd(&g_rs_cycles[2], t_marks[2] - t_marks[1]);
      atomicAdd(&g_rs_cycles[3], t_marks[3] - t_marks[2]);
      atomicAdd(&g_rs_cycles[4], t_end - t_marks[3]);
    }
    atomicAdd(&g_rs_cycles[5], 1ull);
  }
#endif
}

template <int NL, int NW, int TM, int PH, bool POLY, bool GLU, int QD>
int launch_stack(const KArgs& k, int precision, size_t smem, double flops, const char* tag, hipStream_t s) {
  const bool x1 = precision == 2;
  auto kern = x1 ? resstack64_kernel<NL, NW, TM, PH, POLY, GLU, 1, QD> : resstack64_kernel<NL, NW, TM, PH, POLY, GLU, 3, QD>;
  static asw::SmemAttr attr[2];                         // per device and instantiation
  if (int rc = attr[x1].ensure(reinterpret_cast<const void*>(kern), 160 * 1024)) return rc;
  const int total = k.B * k.ntile;
  dim3 grid(((total + 7) / 8) * 8);
  char nm[128];
  int nl = snprintf(nm, sizeof nm, "resstack64<%s,%dx%d%s%s>", tag, NW, TM, POLY ? (PH == 1 ? ",poly1" : PH == 2 ? ",poly2" : ",poly4") : "",
                    GLU ? ",glu" : "");
  if (asw::prof_detail()) snprintf(nm + nl, sizeof nm - nl, "[B%d M%d d%d]", k.B, k.T, k.L[0].dil);
  // algorithmic bytes: the stack's input read once, its output written once
  asw::ProfScope prof(s, nm, flops, (double)k.B * k.T * C * 4 * (GLU ? 3 : 2));
  hipLaunchKernelGGL(kern, grid, dim3(64 * NW), smem, s, k);
  ASW_LAUNCH_CHECK();
  return ASW_OK;
}

template <int NW, int TM>
int dispatch(const asw_resstack_args& a, KArgs& k, bool glu, double flops, hipStream_t s) {
  constexpr int R0 = 32 * NW * TM;
  const int dil0 = k.L[0].dil;
  // one layer of large dilation: polyphase row sets while every phase still fills its share of the tile
  const int rpp = a.T / dil0;                            // rows per phase
  if (a.n_layers == 1 && dil0 >= 7 && rpp >= 48 && !glu) {
    const int PH = rpp >= R0 - 32 ? 1 : rpp >= R0 / 2 - 32 ? 2 : 4;
    const int BMJ = R0 / PH;
    k.BM = R0;
    k.L[0].nfrag = NW * TM;
    k.ntile = asw::cdiv(asw::cdiv(a.T, dil0), BMJ) * asw::cdiv(dil0, PH);
    k.img_rows = PH * (BMJ + a.taps - 1);
    const size_t smem = (size_t)k.img_rows * RS + 3 * 64 * sizeof(float);
    switch (PH) {
      case 1: return launch_stack<1, NW, TM, 1, true, false, 4>(k, a.precision, smem, flops, "1", s);
      case 2: return launch_stack<1, NW, TM, 2, true, false, 4>(k, a.precision, smem, flops, "1", s);
      default: return launch_stack<1, NW, TM, 4, true, false, 4>(k, a.precision, smem, flops, "1", s);
    }
  }
  // contiguous tiles: layer 0 computes R0 rows, the stack delivers BM = R0 - 2 * (halo after layer 0)
  k.BM = R0 - 2 * k.L[0].halo;
  ASW_CHECK_ARG(k.BM >= R0 / 2, "resstack: the later layers' halo (%d rows per side) leaves %d of %d rows; fuse fewer layers",
                k.L[0].halo, k.BM, R0);
  k.ntile = asw::cdiv(a.T, k.BM);
  int rows = 0;
  for (int i = 0; i < a.n_layers; ++i) {
    k.L[i].nfrag = asw::cdiv(k.BM + 2 * k.L[i].halo, 32);
    const int r = 32 * k.L[i].nfrag + 2 * k.L[i].pad;
    rows = r > rows ? r : rows;
  }
  k.img_rows = rows;
  const size_t smem = (size_t)rows * RS + (size_t)a.n_layers * 3 * 64 * sizeof(float);
  ASW_CHECK_ARG(smem <= 160 * 1024, "resstack: image of %d rows exceeds LDS (dilation %d x %d taps too wide for a contiguous tile)",
                rows, dil0, a.taps);
  switch (a.n_layers) {
    case 1: return glu ? launch_stack<1, NW, TM, 1, false, true, 4>(k, a.precision, smem, flops, "1", s)
                       : launch_stack<1, NW, TM, 1, false, false, 4>(k, a.precision, smem, flops, "1", s);
    case 2: return glu ? launch_stack<2, NW, TM, 1, false, true, 4>(k, a.precision, smem, flops, "2", s)
                       : launch_stack<2, NW, TM, 1, false, false, 4>(k, a.precision, smem, flops, "2", s);
    default: return glu ? launch_stack<3, NW, TM, 1, false, true, 4>(k, a.precision, smem, flops, "3", s)
                        : launch_stack<3, NW, TM, 1, false, false, 4>(k, a.precision, smem, flops, "3", s);
  }
}

}  // namespace

#ifdef ASW_PHASE_TIMING
extern "C" int asw_debug_resstack_cycles(unsigned long long* out6, int reset) {
  ASW_HIP(hipMemcpyFromSymbol(out6, HIP_SYMBOL(g_rs_cycles), 6 * sizeof(unsigned long long)));
  if (reset) {
    const unsigned long long z[6] = {0, 0, 0, 0, 0, 0};
    ASW_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_rs_cycles), z, sizeof z));
  }
  return ASW_OK;
}
#endif

extern "C" int asw_resstack64_f16x3(const asw_resstack_args* args, void* stream) {
  ASW_CHECK_ARG(args != nullptr, "resstack: null args");
  const asw_resstack_args& a = *args;
  hipStream_t s = asw::as_stream(stream);
  ASW_CHECK_ARG(a.C == C, "resstack: C=%d (64 only)", a.C);
  ASW_CHECK_ARG(a.n_layers >= 1 && a.n_layers <= MAXL, "resstack: 1..%d layers", MAXL);
  ASW_CHECK_ARG(a.taps >= 3 && a.taps <= 15 && a.taps % 2 == 1, "resstack: taps=%d (odd, 3..15)", a.taps);
  ASW_CHECK_ARG(a.precision == 1 || a.precision == 2, "resstack: precision 1 (f16x3) or 2 (single-pass f16)");
  ASW_CHECK_ARG(a.B > 0 && a.B <= (1 << 20) && a.T > 0 && (int64_t)a.T * 2 * C < ((int64_t)1 << 29), "resstack: bad B / T");
  ASW_CHECK_ARG(a.out && (a.x || a.glu_raw), "resstack: null tensor");
  const bool glu = a.glu_raw != nullptr;
  if (glu) ASW_CHECK_ARG(a.glu_mr && a.glu_gamma && a.glu_beta, "resstack: GroupNorm + GLU on load needs the statistics / affine arrays");
  KArgs k = {};
  k.x = a.x; k.out = a.out; k.glu_raw = a.glu_raw; k.glu_mr = a.glu_mr; k.glu_gamma = a.glu_gamma; k.glu_beta = a.glu_beta;
  k.glu_out = glu ? a.glu_out : nullptr;
  k.B = a.B; k.T = a.T; k.taps = a.taps; k.eps = a.ln_eps;
  int halo = 0;
  for (int i = a.n_layers - 1; i >= 0; --i) {
    const asw_reslayer_desc& d = a.layer[i];
    ASW_CHECK_ARG(d.Wf_hi && d.Wf_lo && d.bias && d.ln_gamma && d.ln_beta && d.dil >= 1, "resstack: layer %d incomplete", i);
    LayerDev& L = k.L[i];
    L.Wh = reinterpret_cast<const half8*>(d.Wf_hi); L.Wl = reinterpret_cast<const half8*>(d.Wf_lo);
    L.bias = d.bias; L.gamma = d.ln_gamma; L.beta = d.ln_beta;
    L.dil = d.dil; L.pad = d.dil * (a.taps - 1) / 2; L.halo = halo;
    L.scale = ldexpf(1.0f, -d.w_shift);
    halo += L.pad;
  }
  const double flops = 2.0 * a.B * (double)a.T * C * C * a.taps * a.n_layers;
  // tile variants (waves x 32-row fragments per wave); ASW_RESSTACK_TILE=<NW><TM> picks one for measurements
  static const int tile_env = getenv("ASW_RESSTACK_TILE") ? atoi(getenv("ASW_RESSTACK_TILE")) : 0;
  const int variant = tile_env == 24 || tile_env == 44 || tile_env == 82 ? tile_env : 42;
  switch (variant) {
    case 24: return dispatch<2, 4>(a, k, glu, flops, s);
    case 44: return dispatch<4, 4>(a, k, glu, flops, s);
    case 82: return dispatch<8, 2>(a, k, glu, flops, s);
    default: return dispatch<4, 2>(a, k, glu, flops, s);
  }
}
