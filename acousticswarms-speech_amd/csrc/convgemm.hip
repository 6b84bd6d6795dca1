// convgemm.hip -- 1-D conv / transposed conv / linear as implicit GEMM on the gfx950
// matrix cores, with the layer's element-wise tail fused into the epilogue.
//
// Replaces (reference file:line): DilatedResidualLayer conv+ReLU+residual+LayerNorm
// (sep/training/SpeakerLocalization/network.py:57-68), EncoderBlock.conv1
// (:105-108), UpsamplerBlock ConvTranspose1d (:158-165,190), mask_encoder /
// reference_bypass / output_decoder contraction (:327-349,397-402) and the
// transformer linears (:254).
//
// Layout: activations channels-last [B][T][C] fp32, weights Wt[N][K] with
// K = tap*Cin + c.  One workgroup (4 waves, 8 for the 256-row f16x3 tiles) owns a BM x BN
// output tile of one batch item; K is walked in chunks of BK channels of one tap.  A and W
// chunks are staged global -> registers -> LDS (activations through a buffer descriptor, so
// padding costs no branch), the next chunk's global loads are in flight while the MFMAs of
// the current chunk run.  The 30 dilated residual layers have their own halo-staged kernel
// (resconv16_kernel below).
//
// Two arithmetic modes share the tiling and the epilogue:
//  * precision 0: v_mfma_f32_32x32x2_f32 -- an exact fp32 fmaf chain (64 FLOP/clk/SIMD).
//    Lane l supplies A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31]; a lane reads 4 consecutive
//    k (one ds_read_b128) and step s of 4 uses element s of both operands, i.e. the k
//    order is permuted identically for A and B.  LDS rows are padded to BK+4 floats so
//    those reads are bank-conflict free.
//  * precision 1 ("f16x3"): every fp32 operand is split into two halves hi = fp16(x),
//    lo = fp16(x - hi) and the product is lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_f16
//    with fp32 accumulation: operands carry ~21 bits, the dropped lo*lo term is 2^-22 of the
//    product, and the pipe nominally runs 16/3 = 5.3x the f32 MFMA rate (measured on random
//    operands: 1.63 PFLOP/s of f16 MFMA sustained, tests/micro/cu_probe.hip).  Activations are split while
//    they are staged to LDS (saturating at +-65504); weights are split once on the host,
//    pre-scaled by a power of two (undone in the epilogue) so their lo parts stay out of the
//    fp16 subnormal range.  Lane l supplies A[l&31][8(l>>5)+j], j<8: one ds_read_b128 per
//    operand half per k-step; rows padded to BK+8 halves (conflict free).
//
// Epilogue: the accumulators of one 32-row slab are written to LDS (C/D map:
// col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)), then each wave owns whole
// rows: + residual, * gate tensor, LayerNorm over the row (two-pass, wave shuffles),
// GroupNorm partial sums, coalesced row stores.
#include <cstdlib>
#include <type_traits>

#include "asw_common.h"
#include "mfma_util.h"

namespace {
using namespace asw_mfma;

// f16x3 range guard.  In the f16x3 mode activations are split into fp16 halves while they are
// staged, which saturates at +-65504.  Every tensor a GEMM reads is either normalised
// (LayerNorm / GroupNorm output, bounded by |gamma| sqrt(C) + |beta|) or the un-normalised output
// of a plain epilogue (the masked latent, the feed-forward intermediate).  The plain epilogue
// therefore counts, in f16x3 mode, the threads that wrote a value beyond the fp16 range;
// asw_f16x3_overflow_count() reads the counter.  Zero on every test and bench run with seeded
// weights; a non-zero count means the next GEMM clipped its input and the f32 mode must be used.
__device__ unsigned int g_f16x3_overflow = 0;

#ifdef ASW_PHASE_TIMING
// Diagnostic build only (tests/micro/phase_timing.py): cycles wave 0 of every workgroup spends in
// each phase of a residual-layer tile, summed over workgroups.  [0] staging (global loads, split,
// LDS writes, barrier), [1] taps x k-steps, [2] epilogue, [3] workgroups counted.
__device__ unsigned long long g_phase_cycles[4] = {0, 0, 0, 0};
// generic 256x256 f16x3 GEMM: [0] waiting at the first barrier of a chunk, [1] register -> LDS deposit
// + second barrier, [2] global loads of the next chunk + fragment reads + MFMAs, [3] epilogue, [4] workgroups
__device__ unsigned long long g_gemm_cycles[5] = {0, 0, 0, 0, 0};
#define ASW_PHASE_MARK(var) const unsigned long long var = __builtin_readcyclecounter()
#else
#define ASW_PHASE_MARK(var)
#endif
// ------------------------------------------------------------------ shared epilogue
// tile row -> output row of the batch item (or -1): contiguous tiles
struct RowsContig {
  int m0, M;
  __device__ __forceinline__ int operator()(int trow) const { const int t = m0 + trow; return t < M ? t : -1; }
};

// Row-phase geometry of the epilogue: a row is BN/4 float4; LPR lanes share a row (RPI rows per
// wave instruction, VPL float4 per lane); each wave walks its share of a WM*32-row slab in NSTEP steps.
template <int BN, int WM, int WN>
struct EpiGeom {
  static constexpr int NW = WM * WN;
  static constexpr int LPR = (BN / 4 < 64) ? BN / 4 : 64;
  static constexpr int RPI = 64 / LPR;
  static constexpr int VPL = BN / 4 / LPR;
  static constexpr int NSTEP = WM * 32 / (NW * RPI);
};

template <int BM, int BN, int WM, int WN, bool LN, bool STATS, bool RESID, bool MUL, typename RowMap, bool RESPRE = false>
__device__ __forceinline__ void epilogue(  // WM*WN waves (4 or 8)
floatx16 (&acc)[BM / WM / 32][BN / WN / 32], const asw_convgemm_args& p,
                                         float* smem, float acc_scale, const RowMap& rowmap, const dim3 tile, const int ncol,
                                         const float4* rpre = nullptr) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int LDC = BN + 4;
  float* Ct = smem;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int b = tile.z, n0 = tile.y * BN;
  float st0 = 0.f, sq0 = 0.f, st1 = 0.f, sq1 = 0.f;
  const int half_mod = p.chan_mod >> 1;
  const bool guard = !LN && !STATS && p.precision >= 1;       // un-normalised output that a later f16 GEMM may read
  float amax = 0.f;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    __syncthreads();
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = wn * (BN / WN) + tn * 32 + (lane & 31);
      const float bv = p.bias ? p.bias[n0 + col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float v = acc[tm][tn][r] * acc_scale + bv;
        if (p.relu == 1) v = fmaxf(v, 0.f);
        else if (p.relu == 2) v = v / (1.0f + expf(-v));             // Swish (Conformer feed-forward)
        Ct[row * LDC + col] = v;
      }
    }
    __syncthreads();
    // Row phase.  A row is BN/4 float4; LPR lanes share a row (RPI rows per wave
    // instruction, VPL float4 per lane), so each wave walks its 8*WM slab rows in 8 steps.
    // Steps are processed four at a time with every global load (residual / gate tensor)
    // issued before the first use: the loads of four steps overlap instead of serialising.
    using G = EpiGeom<BN, WM, WN>;
    constexpr int NW = G::NW, LPR = G::LPR, RPI = G::RPI, VPL = G::VPL;
    constexpr int NSTEP = G::NSTEP;                    // steps each wave needs for its slab rows
    constexpr int UNR = NSTEP < 4 ? NSTEP : 4;
    static_assert(NSTEP >= 1 && NSTEP % UNR == 0 && WM * 32 == NSTEP * NW * RPI, "slab rows must split evenly");
    const int sub = lane / LPR, lc = lane % LPR;
#pragma unroll
    for (int it0 = 0; it0 < NSTEP; it0 += UNR) {
      float4 v[UNR][VPL];
      long obase[UNR];
      bool ok[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int sr = ((it0 + u) * NW + wid) * RPI + sub;
        const int trow = (sr >> 5) * (BM / WM) + tm * 32 + (sr & 31);
        const int t_out = rowmap(trow);
        ok[u] = t_out >= 0;
        // rows past the end read row 0 (always valid) and are simply not stored: the loads
        // stay unconditional, so the compiler issues the whole batch before the first wait
        obase[u] = ((long)b * p.M_out + (ok[u] ? t_out : 0)) * p.N + n0;
#pragma unroll
        for (int q = 0; q < VPL; ++q) {
          const int col = (lc + q * LPR) * 4;
          if (RESID && RESPRE) v[u][q] = rpre[(tm * NSTEP + it0 + u) * VPL + q];     // residual taken from the LDS image
          else if (RESID) v[u][q] = *reinterpret_cast<const float4*>(p.resid + obase[u] + col);
          if (MUL) v[u][q] = *reinterpret_cast<const float4*>(p.mul + obase[u] + col);
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int sr = ((it0 + u) * NW + wid) * RPI + sub;
#pragma unroll
        for (int q = 0; q < VPL; ++q) {
          const int col = (lc + q * LPR) * 4;
          const float4 x = *reinterpret_cast<const float4*>(Ct + sr * LDC + col);
          if (RESID) { v[u][q].x += x.x; v[u][q].y += x.y; v[u][q].z += x.z; v[u][q].w += x.w; }
          else if (MUL) { v[u][q].x *= x.x; v[u][q].y *= x.y; v[u][q].z *= x.z; v[u][q].w *= x.w; }
          else v[u][q] = x;
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (LN) {
          float s = 0.f;
#pragma unroll
          for (int q = 0; q < VPL; ++q) s += (v[u][q].x + v[u][q].y) + (v[u][q].z + v[u][q].w);
#pragma unroll
          for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
          const float mean = s * (1.0f / BN);
          float d = 0.f;
#pragma unroll
          for (int q = 0; q < VPL; ++q) {
            const float cx = v[u][q].x - mean, cy = v[u][q].y - mean, cz = v[u][q].z - mean, cw = v[u][q].w - mean;
            d += (cx * cx + cy * cy) + (cz * cz + cw * cw);
          }
#pragma unroll
          for (int o = LPR / 2; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
          const float rstd = 1.0f / sqrtf(d * (1.0f / BN) + p.ln_eps);
#pragma unroll
          for (int q = 0; q < VPL; ++q) {
            const int col = (lc + q * LPR) * 4;
            const float4 g = *reinterpret_cast<const float4*>(p.ln_gamma + col);
            const float4 be = *reinterpret_cast<const float4*>(p.ln_beta + col);
            v[u][q].x = (v[u][q].x - mean) * rstd * g.x + be.x;
            v[u][q].y = (v[u][q].y - mean) * rstd * g.y + be.y;
            v[u][q].z = (v[u][q].z - mean) * rstd * g.z + be.z;
            v[u][q].w = (v[u][q].w - mean) * rstd * g.w + be.w;
          }
        }
        if (ok[u]) {
#pragma unroll
          for (int q = 0; q < VPL; ++q) {
            const int col = (lc + q * LPR) * 4;
            if (STATS) {
              const float4 x = v[u][q];
              const float s1 = (x.x + x.y) + (x.z + x.w), s2 = (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
              if (((n0 + col) % p.chan_mod) >= half_mod) { st1 += s1; sq1 += s2; } else { st0 += s1; sq0 += s2; }
            }
            if (!LN && !STATS) {
              const float4 x = v[u][q];
              amax = fmaxf(amax, fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w))));
            }
            *reinterpret_cast<float4*>(p.out + obase[u] + col) = v[u][q];
          }
        }
      }
    }
  }
  if (guard && !(amax <= 65504.f)) atomicAdd(&g_f16x3_overflow, 1u);      // also catches NaN
  if (STATS) {
    __syncthreads();
    st0 = wave_sum(st0); sq0 = wave_sum(sq0); st1 = wave_sum(st1); sq1 = wave_sum(sq1);
    float* red = smem;                           // Ct is dead after the barrier above
    if (lane == 0) { red[wid * 4 + 0] = st0; red[wid * 4 + 1] = sq0; red[wid * 4 + 2] = st1; red[wid * 4 + 3] = sq1; }
    __syncthreads();
    if (tid < 4) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < WM * WN; ++w) s += red[w * 4 + tid];
      // slot layout is independent of the tile shape: stats_stride slots per batch item (the
      // launcher zero-fills the buffer, smaller grids simply leave slots at zero)
      const long part = (long)b * p.stats_stride + (long)tile.x * ncol + tile.y;
      p.stats[part * 4 + tid] = s;
    }
  }
}

// address of the float4 of A this thread stages for chunk kc (or -1 when it is padding)
template <int BM, int BK>
__device__ __forceinline__ long a_elem(const asw_convgemm_args& p, int idx, int kc, int cpb, int m0) {
  constexpr int KV = BK / 4;
  const int tap = kc / cpb;
  const int c0 = (kc - tap * cpb) * BK;
  const int row = idx / KV, cv = idx - row * KV;
  const int t_out = m0 + row;
  const long e = ((long)t_out * p.stride + (long)tap * p.dil - p.pad) * p.a_row_stride + c0 + cv * 4;
  const bool ok = (idx < BM * KV) && (t_out < p.M_out) && (e >= 0) && (e + 3 < p.a_len);
  return ok ? e : -1;
}

// ------------------------------------------------------------------ exact fp32 MFMA
template <int BM, int BN, int BK, int WM, int WN, bool LN, bool STATS, bool MUL, bool A2F>
__global__ __launch_bounds__(256) void convgemm_kernel(const asw_convgemm_args p) {
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr int LDK = BK + 4;
  constexpr int KV = BK / 4;                 // float4 per staged row
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int A_VEC = (BM * KV + 255) / 256, B_VEC = (BN * KV + 255) / 256;

  extern __shared__ __align__(16) float smem[];
  float* As = smem;
  float* Bs = smem + BM * LDK;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int b = blockIdx.z, m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int K = p.taps * p.Cin;
  const int cpb = p.Cin / BK;                // chunks per tap
  const int nk = p.taps * cpb;
  const float* __restrict__ Ab = p.A + (long)b * p.a_batch_stride;
  const float* __restrict__ A2b = p.A2 ? p.A2 + (long)b * p.a_batch_stride : nullptr;

  float4 ra[A_VEC], rb[B_VEC];
  const __amdgpu_buffer_rsrc_t rA = act_rsrc(Ab, p.a_len);
  const __amdgpu_buffer_rsrc_t rA2 = act_rsrc(A2F ? A2b : Ab, p.a_len);

  // The skip-connection operand is a compile-time variant (no runtime "load or zero" branch).
  auto gload = [&](int kc) {
#pragma unroll
    for (int v = 0; v < A_VEC; ++v) {
      const long e = a_elem<BM, BK>(p, tid + v * 256, kc, cpb, m0);
      float4 x = act_load4(rA, e, e >= 0);
      if (A2F) {
        const float4 y = act_load4(rA2, e, e >= 0);
        x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
      }
      ra[v] = x;
    }
#pragma unroll
    for (int v = 0; v < B_VEC; ++v) {
      const int idx = tid + v * 256;
      const int row = idx / KV, cv = idx - row * KV;
      const bool ok = idx < BN * KV;
      const float4 x = *reinterpret_cast<const float4*>(p.Wt + (long)(n0 + (ok ? row : 0)) * K + (long)kc * BK + cv * 4);
      rb[v] = ok ? x : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int v = 0; v < A_VEC; ++v) {
      const int idx = tid + v * 256;
      const int row = idx / KV, cv = idx - row * KV;
      if (idx < BM * KV) *reinterpret_cast<float4*>(As + row * LDK + cv * 4) = ra[v];
    }
#pragma unroll
    for (int v = 0; v < B_VEC; ++v) {
      const int idx = tid + v * 256;
      const int row = idx / KV, cv = idx - row * KV;
      if (idx < BN * KV) *reinterpret_cast<float4*>(Bs + row * LDK + cv * 4) = rb[v];
    }
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const float* a_rd = As + (wm * (BM / WM) + (lane & 31)) * LDK + (lane >> 5) * 4;
  const float* b_rd = Bs + (wn * (BN / WN) + (lane & 31)) * LDK + (lane >> 5) * 4;

  gload(0);
  for (int kc = 0; kc < nk; ++kc) {
    __syncthreads();                  // previous chunk's fragment reads are done
    lstore();
    __syncthreads();
    if (kc + 1 < nk) gload(kc + 1);   // in flight under the MFMAs below
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      float4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(a_rd + i * 32 * LDK + kk * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4*>(b_rd + j * 32 * LDK + kk * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float a = s == 0 ? af[i].x : s == 1 ? af[i].y : s == 2 ? af[i].z : af[i].w;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const float bb = s == 0 ? bf[j].x : s == 1 ? bf[j].y : s == 2 ? bf[j].z : bf[j].w;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[i][j], 0, 0, 0);
          }
        }
      }
    }
  }
  epilogue<BM, BN, WM, WN, LN, STATS, LN, MUL>(acc, p, smem, 1.0f, RowsContig{m0, p.M_out}, blockIdx, gridDim.y);
}

// ------------------------------------------------------------------ f16x3 split MFMA
// Four-wave tiles whose LDS footprint lets three workgroups share a CU get a register budget for
// three waves per SIMD (hipcc otherwise spends ~180 VGPRs -> two): these are the small-K, write- and
// latency-bound layers, which gain from the third resident workgroup (128x128 tiles: strided conv
// 211 -> 242, transformer linears 232 -> 269 TFLOP/s, K = 64 / 128 transposed convs +15-20 %).
// (The 8-wave 256x128 tile squeezed into 128 VGPRs for two workgroups per CU spills and loses:
// mask encoder 293 vs 312 TFLOP/s on 256x256; 128x256 with four waves, i.e. two independent
// workgroups per CU whose deposit phases could overlap the other's MFMAs: 314 vs 310, a tie.
// Cycle counters on the 256x256 mask-encoder tile (tests/micro/phase_timing.py): per 32-wide chunk
// wave 0 spends 1900 cycles waiting at the first barrier (its SIMD mate is still multiplying), 1150
// depositing the next chunk and 2240 on loads + fragment reads + 48 MFMAs; epilogue 11 % of the tile.)
template <int BM, int BN, int BK, int WM, int WN>
constexpr int g16_waves_per_eu() {
  constexpr long stage = (long)(BM + BN) * (BK + 8) * 2 * 2, slab = (long)(WM * 32) * (BN + 4) * 4;
  if (WM * WN == 4 && (stage > slab ? stage : slab) <= 80 * 1024 && (stage > slab ? stage : slab) > 53 * 1024) return 2;
  return (WM * WN == 4 && (stage > slab ? stage : slab) <= 53 * 1024) ? 3 : 1;
}
template <int BM, int BN, int BK, int WM, int WN, bool LN, bool STATS, bool MUL, bool A2F, int NTERM = 3>
__global__ __launch_bounds__(64 * WM * WN)
__attribute__((amdgpu_waves_per_eu(g16_waves_per_eu<BM, BN, BK, WM, WN>())))
void convgemm16_kernel(const asw_convgemm_args p) {
  static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves per workgroup");
  static_assert(BK % 16 == 0, "k-step of the f16 MFMA");
  constexpr int NT = 64 * WM * WN;           // threads
  constexpr int LDH = BK + 8;                // halves per staged row
  constexpr int KV = BK / 4;                 // float4 (A, fp32) per row
  constexpr int KH = BK / 8;                 // 16-byte vectors (B, fp16) per row
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int A_VEC = (BM * KV + NT - 1) / NT, B_VEC = (BN * KH + NT - 1) / NT;

  extern __shared__ __align__(16) float smem[];
  _Float16* Ah = reinterpret_cast<_Float16*>(smem);
  _Float16* Al = Ah + BM * LDH;
  _Float16* Bh = Al + BM * LDH;
  _Float16* Bl = Bh + BN * LDH;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  // XCD-aware tile order.  The grid is 1-D; workgroup L goes to XCD L % 8 (the dispatcher deals
  // consecutive workgroups round-robin over the 8 XCDs, each with its own L2).  Slot s = L / 8 of
  // one XCD walks the column tiles of a row tile first: the ncol workgroups that read the same
  // activation rows run back to back on ONE L2, so those rows come from HBM once instead of once
  // per column tile.  The (batch item, row tile) pairs are dealt to the XCDs in groups of 8, so
  // every XCD gets the same share whatever the number of row tiles per item; padded slots exit.
  const int ncol = p.N / BN, nrt = (p.M_out + BM - 1) / BM;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int R = (slot / ncol) * 8 + xcd;                      // (batch item, row tile) index
  if (R >= p.B * nrt) return;
  const dim3 tile(R % nrt, slot % ncol, R / nrt);
  const int b = tile.z, m0 = tile.x * BM, n0 = tile.y * BN;
  const int K = p.taps * p.Cin;
  const int cpb = p.Cin / BK;
  const int nk = p.taps * cpb;
  const float* __restrict__ Ab = p.A + (long)b * p.a_batch_stride;
  const float* __restrict__ A2b = p.A2 ? p.A2 + (long)b * p.a_batch_stride : nullptr;
  const _Float16* __restrict__ Wh = reinterpret_cast<const _Float16*>(p.Wt_hi);
  const _Float16* __restrict__ Wl = reinterpret_cast<const _Float16*>(p.Wt_lo);

  // One staging register set (a second set, i.e. loads two chunks ahead, was measured: no
  // gain, and its 64 extra VGPRs cost a resident workgroup per CU).
  float4 ra0[A_VEC];
  half8 rbh0[B_VEC], rbl0[B_VEC];

  // per-thread invariants of the staging addresses
  long a_row[A_VEC];                         // element offset of (row, tap 0, c 0) + this thread's column
  bool a_ok[A_VEC];
  long b_row[B_VEC];
  bool b_ok[B_VEC];
#pragma unroll
  for (int v = 0; v < A_VEC; ++v) {
    const int idx = tid + v * NT;
    const int row = idx / KV, cv = idx - row * KV;
    a_row[v] = ((long)(m0 + row) * p.stride - p.pad) * p.a_row_stride + cv * 4;
    a_ok[v] = (idx < BM * KV) && (m0 + row < p.M_out);
  }
#pragma unroll
  for (int v = 0; v < B_VEC; ++v) {
    const int idx = tid + v * NT;
    const int row = idx / KH, cv = idx - row * KH;
    b_row[v] = (long)(n0 + row) * K + cv * 8;
    b_ok[v] = (idx < BN * KH) && (n0 + row < p.N);
  }
  const long tap_step = (long)p.dil * p.a_row_stride;
  const __amdgpu_buffer_rsrc_t rA = act_rsrc(Ab, p.a_len);
  const __amdgpu_buffer_rsrc_t rA2 = act_rsrc(A2F ? A2b : Ab, p.a_len);

  auto gload = [&](int kc, float4 (&ra)[A_VEC], half8 (&rbh)[B_VEC], half8 (&rbl)[B_VEC]) {
    const int tap = kc / cpb;
    const long koff = tap * tap_step + (kc - tap * cpb) * BK;
#pragma unroll
    for (int v = 0; v < A_VEC; ++v) {
      const long e = a_row[v] + koff;                   // padding / past-the-end offsets read as zeros
      float4 x = act_load4(rA, e, a_ok[v]);
      if (A2F) {
        const float4 y = act_load4(rA2, e, a_ok[v]);
        x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
      }
      ra[v] = x;
    }
#pragma unroll
    for (int v = 0; v < B_VEC; ++v) {
      const long o = (b_ok[v] ? b_row[v] : 0) + (long)kc * BK;
      rbh[v] = *reinterpret_cast<const half8*>(Wh + o);   // rows past BN*KH are never stored to LDS
      if (NTERM == 3) rbl[v] = *reinterpret_cast<const half8*>(Wl + o);
    }
  };
  auto lstore = [&](const float4 (&ra)[A_VEC], const half8 (&rbh)[B_VEC], const half8 (&rbl)[B_VEC]) {
#pragma unroll
    for (int v = 0; v < A_VEC; ++v) {
      const int idx = tid + v * NT;
      const int row = idx / KV, cv = idx - row * KV;
      if (idx < BM * KV) {
        half4 hi, lo;
        split4t<NTERM>(ra[v], hi, lo);
        *reinterpret_cast<half4*>(Ah + row * LDH + cv * 4) = hi;
        if (NTERM == 3) *reinterpret_cast<half4*>(Al + row * LDH + cv * 4) = lo;
      }
    }
#pragma unroll
    for (int v = 0; v < B_VEC; ++v) {
      const int idx = tid + v * NT;
      const int row = idx / KH, cv = idx - row * KH;
      if (idx < BN * KH) {
        *reinterpret_cast<half8*>(Bh + row * LDH + cv * 8) = rbh[v];
        if (NTERM == 3) *reinterpret_cast<half8*>(Bl + row * LDH + cv * 8) = rbl[v];
      }
    }
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int a_off = (wm * (BM / WM) + (lane & 31)) * LDH + (lane >> 5) * 8;
  const int b_off = (wn * (BN / WN) + (lane & 31)) * LDH + (lane >> 5) * 8;

  auto compute = [&]() {
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      half8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = *reinterpret_cast<const half8*>(Ah + a_off + i * 32 * LDH + ks * 16);
        if (NTERM == 3) al[i] = *reinterpret_cast<const half8*>(Al + a_off + i * 32 * LDH + ks * 16);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const half8*>(Bh + b_off + j * 32 * LDH + ks * 16);
        if (NTERM == 3) bl[j] = *reinterpret_cast<const half8*>(Bl + b_off + j * 32 * LDH + ks * 16);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (NTERM == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  };

#ifdef ASW_PHASE_TIMING
  unsigned long long tg_wait = 0, tg_store = 0, tg_comp = 0;
#endif
  gload(0, ra0, rbh0, rbl0);
  for (int kc = 0; kc < nk; ++kc) {
    ASW_PHASE_MARK(g0);
    __syncthreads();
    ASW_PHASE_MARK(g1);
    lstore(ra0, rbh0, rbl0);
    __syncthreads();
    ASW_PHASE_MARK(g2);
    if (kc + 1 < nk) gload(kc + 1, ra0, rbh0, rbl0);
    compute();
#ifdef ASW_PHASE_TIMING
    {
      float sink = acc[0][0][0];
      asm volatile("" ::"v"(sink));
      const unsigned long long g3 = __builtin_readcyclecounter();
      tg_wait += g1 - g0; tg_store += g2 - g1; tg_comp += g3 - g2;
    }
#endif
  }
  ASW_PHASE_MARK(ge0);
  epilogue<BM, BN, WM, WN, LN, STATS, LN, MUL>(acc, p, smem, __builtin_ldexpf(1.0f, -p.w_shift), RowsContig{m0, p.M_out}, tile, ncol);
#ifdef ASW_PHASE_TIMING
  if (threadIdx.x == 0 && BM == 256 && BN == 256) {
    const unsigned long long ge1 = __builtin_readcyclecounter();
    atomicAdd(&g_gemm_cycles[0], tg_wait);
    atomicAdd(&g_gemm_cycles[1], tg_store);
    atomicAdd(&g_gemm_cycles[2], tg_comp);
    atomicAdd(&g_gemm_cycles[3], ge1 - ge0);
    atomicAdd(&g_gemm_cycles[4], 1ull);
  }
#endif
}

// ------------------------------------------------------------------ pipelined wide-tile GEMM
// The 8-wave 256-column tiles (mask encoder, strided / transposed convolutions, big linears) as a
// software pipeline with ONE barrier per 32-wide chunk instead of two (cycle counters on the
// two-barrier kernel above, mask-encoder shape: per chunk wave 0 spent 1150 cycles depositing the
// next chunk with every MFMA pipe of the workgroup idle, tests/micro/phase_timing.py):
//  * B never touches LDS: the weights are pre-packed in MFMA-fragment order (asw_pack_fragments_f16,
//    the layout of the residual kernel), each wave pulls its two column fragments per k-step with
//    coalesced 1 KiB loads, QDB k-steps ahead of their use (L2-resident: one column tile of the
//    largest matrix is 2.1 MB);
//  * A (fp32 activations) is split to fp16 hi / lo while it is deposited, into a two-stage LDS ring:
//    the rows of chunk k+1 are fetched before, and deposited after, the MFMAs of chunk k, so the
//    deposit of one wave overlaps the MFMAs of the others and only the ring hand-over needs a barrier.
// Same tiling (wave tile BM/2 x 64), same epilogue, same XCD-aware tile order as the kernel above.
// Measured (T = 48 000, batch 64): mask encoder 312 -> 332 TFLOP/s, strided / transposed convolutions
// +3-6 %.  64-wide chunks (half the barriers, 147 KB ring) spill and lose: 304.
#ifndef ASW_PIPE_QDB
#define ASW_PIPE_QDB 2
#endif
// Main loop of one 256-column tile: picks the tile of this workgroup (false: none, the whole workgroup
// leaves), runs the K loop and returns the accumulators (wave (wm, wn) of WM x 4 holds rows
// wm*BM/WM + 32*i.., columns wn*64 + 32*j..).  Ends on a barrier: the ring is free for the epilogue.
// WM = 2: eight waves on a 256-row tile, one workgroup per CU.  WM = 1: four waves on a 128-row tile with the
// SAME wave tile (128 x 64), two independent workgroups per CU -- one's epilogue under the other's main loop.
template <int BM, bool A2F, int BK, int NTERM = 3, int WM = 2>
__device__ __forceinline__ bool pipe_mainloop(const asw_convgemm_args& p, float* smem, floatx16 (&acc)[BM / WM / 32][2],
                                              dim3& tile_out, int& ncol_out) {
  constexpr int BN = 256, WN = 4, NT = 64 * WM * WN, QDB = ASW_PIPE_QDB;
  constexpr int LDH = BK + 8, KV = BK / 4;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int A_VEC = (BM * KV + NT - 1) / NT;
  constexpr int STAGE = 2 * BM * LDH;              // halves per ring stage (hi image + lo image)

  _Float16* ring = reinterpret_cast<_Float16*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int ncol = p.N / BN, nrt = (p.M_out + BM - 1) / BM;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int R = (slot / ncol) * 8 + xcd;                      // (batch item, row tile) index, XCD-aware order
  if (R >= p.B * nrt) return false;
  const dim3 tile(R % nrt, slot % ncol, R / nrt);
  tile_out = tile;
  ncol_out = ncol;
  const int b = tile.z, m0 = tile.x * BM, n0 = tile.y * BN;
  const int cpb = p.Cin / BK;
  const int nk = p.taps * cpb;
  const float* __restrict__ Ab = p.A + (long)b * p.a_batch_stride;
  const float* __restrict__ A2b = p.A2 ? p.A2 + (long)b * p.a_batch_stride : nullptr;
  const half8* __restrict__ Wh = reinterpret_cast<const half8*>(p.Wf_hi);
  const half8* __restrict__ Wl = reinterpret_cast<const half8*>(p.Wf_lo);
  const int NTF = p.N / 32;                        // column fragments across N
  const int nt0 = n0 / 32 + wn * TN;

  float4 ra[A_VEC];
  long a_row[A_VEC];
  bool a_ok[A_VEC];
#pragma unroll
  for (int v = 0; v < A_VEC; ++v) {
    const int idx = tid + v * NT;
    const int row = idx / KV, cv = idx - row * KV;
    a_row[v] = ((long)(m0 + row) * p.stride - p.pad) * p.a_row_stride + cv * 4;
    a_ok[v] = (idx < BM * KV) && (m0 + row < p.M_out);
  }
  const long tap_step = (long)p.dil * p.a_row_stride;
  const __amdgpu_buffer_rsrc_t rA = act_rsrc(Ab, p.a_len);
  const __amdgpu_buffer_rsrc_t rA2 = act_rsrc(A2F ? A2b : Ab, p.a_len);

  auto gload = [&](int kc) {
    const int tap = kc / cpb;
    const long koff = tap * tap_step + (kc - tap * cpb) * BK;
#pragma unroll
    for (int v = 0; v < A_VEC; ++v) {
      const long e = a_row[v] + koff;
      float4 x = act_load4(rA, e, a_ok[v]);
      if (A2F) {
        const float4 y = act_load4(rA2, e, a_ok[v]);
        x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
      }
      ra[v] = x;
    }
  };
  auto deposit = [&](int stage) {
    _Float16* Ah = ring + stage * STAGE;
    _Float16* Al = Ah + BM * LDH;
#pragma unroll
    for (int v = 0; v < A_VEC; ++v) {
      const int idx = tid + v * NT;
      const int row = idx / KV, cv = idx - row * KV;
      if (idx < BM * KV) {
        half4 hi, lo;
        split4t<NTERM>(ra[v], hi, lo);
        *reinterpret_cast<half4*>(Ah + row * LDH + cv * 4) = hi;
        if (NTERM == 3) *reinterpret_cast<half4*>(Al + row * LDH + cv * 4) = lo;
      }
    }
  };
  auto bload = [&](int kg, half8 (&bh)[TN], half8 (&bl)[TN]) {          // kg = global k-step (16 K each)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const long o = ((long)kg * NTF + nt0 + j) * 64 + lane;
      bh[j] = Wh[o];
      if (NTERM == 3) bl[j] = Wl[o];
    }
  };

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int a_off = (wm * (BM / WM) + (lane & 31)) * LDH + (lane >> 5) * 8;
  half8 qh[QDB][TN], ql[QDB][TN];                  // B fragments of the next QDB k-steps
  const int nks = nk * (BK / 16);                  // k-steps in all
#pragma unroll
  for (int q = 0; q < QDB; ++q) bload(q, qh[q], ql[q]);
  gload(0);
  deposit(0);
  __syncthreads();
  constexpr int KS = BK / 16;
  static_assert(KS % QDB == 0 || QDB == 2 * KS, "the B ring is one or two chunks deep");
  // One chunk; PAR = chunk parity, compile-time so that ring stage and B buffer indices are static
  // (the loop below is unrolled by two).
  auto chunk = [&](int kc, auto par) {
    constexpr int PAR = decltype(par)::value;
    if (kc + 1 < nk) gload(kc + 1);                // in flight under the MFMAs of this chunk
    const _Float16* Ah = ring + PAR * STAGE;
    const _Float16* Al = Ah + BM * LDH;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      half8 ah[TM], al[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = *reinterpret_cast<const half8*>(Ah + a_off + i * 32 * LDH + ks * 16);
        if (NTERM == 3) al[i] = *reinterpret_cast<const half8*>(Al + a_off + i * 32 * LDH + ks * 16);
      }
      const int q = (PAR * KS + ks) % QDB;         // B register buffer of this k-step
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (NTERM == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], qh[q][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], ql[q][j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], qh[q][j], acc[i][j], 0, 0, 0);
        }
      const int kg = kc * KS + ks + QDB;           // same slot, QDB k-steps ahead
      if (kg < nks) bload(kg, qh[q], ql[q]);
      // deposit of the next chunk between the k-steps: its conversions and LDS writes issue in the
      // shadow of this wave's own MFMAs (the other stage was last read one chunk ago, before the
      // previous barrier)
      if (ks == KS / 2 - 1 && kc + 1 < nk) deposit(PAR ^ 1);
    }
    __syncthreads();
  };
  for (int kc = 0; kc < nk; kc += 2) {
    chunk(kc, std::integral_constant<int, 0>{});
    if (kc + 1 < nk) chunk(kc + 1, std::integral_constant<int, 1>{});
  }
  return true;
}

template <int BM, bool STATS, bool MUL, bool A2F, int BK = 32, int NTERM = 3, int WM = 2>
__global__ __launch_bounds__(256 * WM) __attribute__((amdgpu_waves_per_eu(2)))
void convgemm16p_kernel(const asw_convgemm_args p) {
  constexpr int BN = 256, WN = 4;
  extern __shared__ __align__(16) float smem[];
  floatx16 acc[BM / WM / 32][2];
  dim3 tile;
  int ncol;
  if (!pipe_mainloop<BM, A2F, BK, NTERM, WM>(p, smem, acc, tile, ncol)) return;
  epilogue<BM, BN, WM, WN, false, STATS, false, MUL>(acc, p, smem, __builtin_ldexpf(1.0f, -p.w_shift),
                                                     RowsContig{(int)tile.x * BM, p.M_out}, tile, ncol);
}

// ------------------------------------------------------------------ mask path in one kernel
// reference_bypass, mask_encoder and the output_decoder taps (network.py:327-349,397-405) without the
// 2048-channel latents ever reaching memory:
//   taps[f][j] = sum_e relu(mask_enc(x)[f][e] + b_e) * relu(bypass(ref)[f][e] + c_e) * D[e][j]
// The main loop is the pipelined GEMM above (mask encoder, K = taps*Cin).  Epilogue, per 256 x 256 tile:
//  A. each wave computes the bypass tile of its own 32 x 32 accumulator blocks with nine more MFMAs
//     (K = 33 padded to 48; the frames of the reference channel are read straight from global memory
//     in A-fragment order) and gates the accumulators in registers;
//  B. the gated latent goes through an LDS slab, 128 rows at a time, and comes back in A-fragment
//     order for the decoder contraction over the tile's 256 latent channels: eight waves = four
//     32-row blocks x two 32-tap blocks, 48 MFMAs each.  The result is a PARTIAL tap product (this
//     column tile's share of the sum over e); the overlap-add kernel adds the N/256 partials.
// Per candidate (T = 48 000) this writes 8 x 3008 x 33 floats instead of writing the bypass latent,
// reading it, writing the gated latent and reading that again (4 x 24.6 MB).
template <int BM, int KSB, int NTERM = 3>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2)))
void maskpath16p_kernel(const asw_convgemm_args p, const asw_maskpath_args mf) {
  constexpr int BN = 256, WN = 4, TM = BM / 64, TN = 2, LDC = BN + 4, BK = 32;
  static_assert(BM == 256, "slab passes are written for 2 x 128 rows");
  extern __shared__ __align__(16) float smem[];
  floatx16 acc[TM][TN];
  dim3 tile;
  int ncol;
  if (!pipe_mainloop<BM, false, BK, NTERM>(p, smem, acc, tile, ncol)) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int b = tile.z, m0 = tile.x * BM, n0 = tile.y * BN;
  const float acc_scale = __builtin_ldexpf(1.0f, -p.w_shift);
  const float byp_scale = __builtin_ldexpf(1.0f, -mf.byp_shift), dec_scale = __builtin_ldexpf(1.0f, -mf.dec_shift);
  // ---- A: bypass tile + gating, in registers
  const __amdgpu_buffer_rsrc_t rR = act_rsrc(mf.ref + (long)b * mf.ref_batch_stride, mf.ref_len);
  const half8* __restrict__ Bh = reinterpret_cast<const half8*>(mf.byp_hi);
  const half8* __restrict__ Bl = reinterpret_cast<const half8*>(mf.byp_lo);
  const int NTF = p.N / 32;
  float amax = 0.f;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int nt = n0 / 32 + wn * TN + tn;
    const int col = nt * 32 + (lane & 31);
    const float bm = p.bias ? p.bias[col] : 0.f, bb = mf.byp_bias ? mf.byp_bias[col] : 0.f;
    half8 wh[KSB], wl[KSB];
#pragma unroll
    for (int ks = 0; ks < KSB; ++ks) {
      wh[ks] = Bh[((long)ks * NTF + nt) * 64 + lane];
      wl[ks] = Bl[((long)ks * NTF + nt) * 64 + lane];
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int f = m0 + wm * (BM / 2) + tm * 32 + (lane & 31);
      const bool ok = f < p.M_out;
      floatx16 bp;
#pragma unroll
      for (int r = 0; r < 16; ++r) bp[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KSB; ++ks) {
        const long e = (long)f * mf.ref_hop + ks * 16 + (lane >> 5) * 8;
        const float4 x0 = act_load4(rR, e, ok), x1 = act_load4(rR, e + 4, ok);
        half4 h0, l0, h1, l1;
        split4t<NTERM>(x0, h0, l0);
        split4t<NTERM>(x1, h1, l1);
        const half8 ah = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        const half8 al = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
        if (NTERM == 3) {
          bp = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh[ks], bp, 0, 0, 0);
          bp = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl[ks], bp, 0, 0, 0);
        }
        bp = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh[ks], bp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = fmaxf(acc[tm][tn][r] * acc_scale + bm, 0.f) * fmaxf(bp[r] * byp_scale + bb, 0.f);
        amax = fmaxf(amax, v);
        acc[tm][tn][r] = v;
      }
    }
  }
  // the latent is split to fp16 halves below: same range guard as a latent written for a later GEMM
  if (!(amax <= 65504.f)) atomicAdd(&g_f16x3_overflow, 1u);
  // ---- B: decoder contraction through the slab, rows [pass*128, pass*128 + 128) of the tile per pass
  float* Ct = smem;
  const half8* __restrict__ Dh = reinterpret_cast<const half8*>(mf.dec_hi);
  const half8* __restrict__ Dl = reinterpret_cast<const half8*>(mf.dec_lo);
  const int ft = wid & 3, tt = wid >> 2;                       // 32-row block, 32-tap block of this wave
  float* __restrict__ outp = mf.taps + ((long)tile.y * p.B + b) * p.M_out * 64;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if (pass) __syncthreads();                                 // (pass 0: the main loop ended on a barrier)
    if (wm == pass) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const int col = wn * 64 + tn * 32 + (lane & 31);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            Ct[row * LDC + col] = acc[tm][tn][r];
          }
        }
    }
    __syncthreads();
    floatx16 tp;
#pragma unroll
    for (int r = 0; r < 16; ++r) tp[r] = 0.f;
    const float* src = Ct + (ft * 32 + (lane & 31)) * LDC + (lane >> 5) * 8;
#pragma unroll 4
    for (int ks = 0; ks < BN / 16; ++ks) {
      const float4 x0 = *reinterpret_cast<const float4*>(src + ks * 16);
      const float4 x1 = *reinterpret_cast<const float4*>(src + ks * 16 + 4);
      const long o = ((long)(n0 / 16 + ks) * 2 + tt) * 64 + lane;
      const half8 dh = Dh[o], dl = Dl[o];
      half4 h0, l0, h1, l1;
      split4t<NTERM>(x0, h0, l0);
      split4t<NTERM>(x1, h1, l1);
      const half8 ah = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
      const half8 al = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
      if (NTERM == 3) {
        tp = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, dh, tp, 0, 0, 0);
        tp = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, dl, tp, 0, 0, 0);
      }
      tp = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, dh, tp, 0, 0, 0);
    }
    const int j = tt * 32 + (lane & 31);
    if (j < mf.dec_taps) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = m0 + pass * 128 + ft * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (f < p.M_out) outp[(long)f * 64 + j] = tp[r] * dec_scale;
      }
    }
  }
}

int launch_mask_path(const asw_maskpath_args* args, void* stream) {
  ASW_CHECK_ARG(args, "mask_path: null argument block");
  const asw_maskpath_args& m = *args;
  asw_convgemm_args a = m.enc;
  hipStream_t s = asw::as_stream(stream);
  constexpr int BM = 256, BN = 256, BK = 32, KSB = 3;
  ASW_CHECK_ARG(a.A && a.Wf_hi && a.Wf_lo && m.ref && m.byp_hi && m.byp_lo && m.dec_hi && m.dec_lo && m.taps,
                "mask_path: null pointer (fragment-order weights are required)");
  ASW_CHECK_ARG(a.B > 0 && a.M_out > 0 && a.N % BN == 0 && a.Cin % BK == 0 && a.taps > 0 && a.stride > 0,
                "mask_path: shape (N %% 256 == 0, Cin %% 32 == 0)");
  ASW_CHECK_ARG(a.A2 == nullptr && a.mul == nullptr && a.resid == nullptr && a.ln_gamma == nullptr && a.stats == nullptr,
                "mask_path: the encoder block takes A, weights and bias only");
  ASW_CHECK_ARG(m.byp_k == 16 * KSB, "mask_path: bypass kernel padded to %d taps, %d given", 16 * KSB, m.byp_k);
  ASW_CHECK_ARG(m.dec_taps > 0 && m.dec_taps <= 64 && m.ref_hop > 0 && m.ref_hop % 4 == 0 && m.ref_len > 0,
                "mask_path: decoder taps 1..64, reference hop a multiple of 4 samples");
  ASW_CHECK_ARG((reinterpret_cast<uintptr_t>(m.ref) & 15) == 0 && m.ref_batch_stride % 4 == 0,
                "mask_path: reference rows must be 16-byte aligned");
  ASW_CHECK_ARG(a.precision == 1 || a.precision == 2, "mask_path: precision 1 (f16x3) or 2 (single-pass f16)");
  const bool x1 = a.precision == 2;
  a.relu = 1;
  constexpr size_t ring = (size_t)2 * 2 * BM * (BK + 8) * sizeof(_Float16);
  constexpr size_t slab = (size_t)128 * (BN + 4) * sizeof(float);
  constexpr size_t smem = ring > slab ? ring : slab;
  static_assert(smem <= 160 * 1024, "LDS budget");
  const void* kern = x1 ? reinterpret_cast<const void*>(maskpath16p_kernel<BM, KSB, 1>)
                        : reinterpret_cast<const void*>(maskpath16p_kernel<BM, KSB, 3>);
  static asw::SmemAttr attr[2];                         // per device and instantiation
  if (int rc = attr[x1].ensure(kern, smem)) return rc;
  const long nrt = asw::cdiv(a.M_out, BM);
  dim3 grid(((nrt * a.B + 7) / 8) * 8 * (a.N / BN), 1, 1);
  std::string pn = "maskpath16p<256,256,32>";
  if (asw::prof_detail()) {
    char sh[64];
    snprintf(sh, sizeof sh, "[B%d M%d N%d K%d s%d]", a.B, a.M_out, a.N, a.taps * a.Cin, a.stride);
    pn += sh;
  }
  // mask encoder + bypass + decoder taps
  asw::ProfScope prof(s, pn, 2.0 * a.B * (double)a.M_out * a.N * ((double)a.taps * a.Cin + m.byp_taps + m.dec_taps));
  if (x1) hipLaunchKernelGGL((maskpath16p_kernel<BM, KSB, 1>), grid, dim3(512), smem, s, a, m);
  else hipLaunchKernelGGL((maskpath16p_kernel<BM, KSB, 3>), grid, dim3(512), smem, s, a, m);
  ASW_LAUNCH_CHECK();
  return ASW_OK;
}

template <int BM, bool STATS, bool MUL, bool A2F, int BK = 32, int WM = 2>
int launch_pipe(const asw_convgemm_args& a, hipStream_t s) {
  constexpr int BN = 256;
  constexpr size_t ring = (size_t)2 * 2 * BM * (BK + 8) * sizeof(_Float16);
  constexpr size_t slab = (size_t)(WM * 32) * (BN + 4) * sizeof(float);
  constexpr size_t smem = ring > slab ? ring : slab;
  static_assert(smem <= 160 * 1024, "LDS budget");
  static_assert(BM == 128 * WM, "wave tile 128 x 64");
  const bool x1 = a.precision == 2;                     // single-pass f16: the one-term instantiation
  const void* kern = x1 ? reinterpret_cast<const void*>(convgemm16p_kernel<BM, STATS, MUL, A2F, BK, 1, WM>)
                        : reinterpret_cast<const void*>(convgemm16p_kernel<BM, STATS, MUL, A2F, BK, 3, WM>);
  static asw::SmemAttr attr[2];                         // per device and instantiation
  if (int rc = attr[x1].ensure(kern, smem)) return rc;
  ASW_CHECK_ARG(A2F == (a.A2 != nullptr), "convgemm: skip operand variant mismatch");
  ASW_CHECK_ARG(a.Cin % BK == 0 && a.N % BN == 0, "convgemm: pipelined tile needs Cin %% BK == 0 and N %% 256 == 0");
  const long nrt = asw::cdiv(a.M_out, BM);
  dim3 grid(((nrt * a.B + 7) / 8) * 8 * (a.N / BN), 1, 1);
  std::string pn = asw::prof_name(MUL ? "convgemm16pm" : "convgemm16p", BM, BN, BK, false, STATS);
  if (asw::prof_detail()) {
    char sh[64];
    snprintf(sh, sizeof sh, "[B%d M%d N%d K%d s%d]", a.B, a.M_out, a.N, a.taps * a.Cin, a.stride);
    pn += sh;
  }
  asw::ProfScope prof(s, pn, 2.0 * a.B * (double)a.M_out * a.N * (double)a.taps * a.Cin);
  if (x1) hipLaunchKernelGGL((convgemm16p_kernel<BM, STATS, MUL, A2F, BK, 1, WM>), grid, dim3(256 * WM), smem, s, a);
  else hipLaunchKernelGGL((convgemm16p_kernel<BM, STATS, MUL, A2F, BK, 3, WM>), grid, dim3(256 * WM), smem, s, a);
  ASW_LAUNCH_CHECK();
  return ASW_OK;
}

// ------------------------------------------------------------------ halo-staged residual conv
// DilatedResidualLayer (network.py:57-68) in f16x3 arithmetic: out = LN(ReLU(conv_d(x)+b) + x).
// The workgroup owns BM output rows x all C channels.  For each 64-channel slice of the
// input it stages the rows its taps touch ONCE into LDS, already split into fp16 hi/lo
// (row = 128 B hi + 128 B lo + 16 B pad: the per-lane 16-byte fragment reads of 32
// consecutive rows are bank-conflict free); every tap reads that image at a row offset, so
// the input is fetched and converted once per workgroup instead of once per tap.
//
// Row sets.  PH == 1: BM consecutive rows, image = rows [m0 - pad, m0 + BM + pad), tap step
// = dil rows.  PH > 1 (large dilation, 49): a dilated convolution is `dil` independent
// dilation-1 convolutions on the polyphase sub-sequences x[phase + dil*j]; the workgroup
// takes PH phases x BM/PH consecutive j, image = PH x (BM/PH + taps-1) rows, tap step = 1
// row.  The halo is then K-1 rows per phase instead of (K-1)*dil, which keeps the image at
// ~40 KB and lets 3-4 workgroups share a CU (the kernel is latency-bound otherwise).
//
// Weights never touch LDS: they are pre-packed in MFMA-fragment order, so each wave pulls
// its B operand with one coalesced 1 KiB load per fragment, one k-step ahead of the MFMAs
// (they are L2/L1-resident: a layer's weights are at most 7.3 MB and shared by every
// workgroup).  No barrier inside the taps x k-steps of a slice.
#ifndef ASW_RES128_WAVES
#define ASW_RES128_WAVES 3
#endif
template <int BM, int PH, bool POLY>
struct ResRows {
  static constexpr int BMJ = BM / PH;
  int m0, jb, pb, dil, T;                    // contiguous tiles use m0; polyphase tiles (jb, pb)
  __device__ __forceinline__ int operator()(int trow) const {
    if (!POLY) { const int t = m0 + trow; return t < T ? t : -1; }
    const int ph = pb * PH + trow / BMJ;
    const int t = dil * (jb * BMJ + trow % BMJ) + ph;
    return (ph < dil && t < T) ? t : -1;
  }
};

template <int BM, int C, int WM, int WN, int PH, int QD = 4, bool POLY = (PH > 1), bool GLU = false, int NTERM = 3>
__global__ __launch_bounds__(64 * WM * WN)
__attribute__((amdgpu_waves_per_eu(C == 64 ? 4 : WM * WN == 8 ? 2 : (QD == 2 ? (C >= 512 || (C == 256 && BM == 128) ? 2 : (C == 128 ? ASW_RES128_WAVES : 3)) : (C == 64 && WM * WN == 4 ? 4 : 1)))))
void resconv16_kernel(const asw_convgemm_args p) {
  static_assert(QD == 2 || QD == 4, "B prefetch depth in k-steps");
  static_assert(!GLU || (PH == 1 && !POLY), "GroupNorm + GLU on load: contiguous tiles only");
  static_assert(WM * WN == 2 || WM * WN == 4 || WM * WN == 8, "2, 4 or 8 waves per workgroup");
  constexpr int NTHR = 64 * WM * WN, SROWS = NTHR / 16;   // staging: 16 threads per row
  constexpr int TM = BM / WM / 32, TN = C / WN / 32;
  constexpr int RS = 272;                    // bytes per staged row: 128 hi + 128 lo + 16 pad
  constexpr int NT = C / 32;                 // 32-column fragments across N
  constexpr int BMJ = BM / PH;
  constexpr int SU = (GLU && C > 64) ? 4 : 8;   // staging rows per thread in flight
  static_assert(BMJ % 32 == 0, "an MFMA row tile must stay inside one phase");

  extern __shared__ __align__(16) float smem[];
  char* img = reinterpret_cast<char*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int b = blockIdx.z;
  const int taps = p.taps, dil = p.dil, pad = p.pad;
  const int T = p.M_out;
  // contiguous: blockIdx.x = row tile.  polyphase: blockIdx.x = jb * n_pb + pb (PH phases per workgroup).
  const int n_pb = (dil + PH - 1) / PH;
  const int jb = !POLY ? 0 : blockIdx.x / n_pb, pb = !POLY ? 0 : blockIdx.x % n_pb;
  const int m0 = blockIdx.x * BM;
  const int RJ = BMJ + (!POLY ? (taps - 1) * dil : taps - 1);      // image rows per phase
  const int R = PH * RJ;
  const int tapstep = !POLY ? dil : 1;
  // GLU: the input row g is GLU(GroupNorm(raw row g)), raw = [T][value half C | gate half C]
  const __amdgpu_buffer_rsrc_t rX = GLU ? act_rsrc(p.glu_raw + (long)b * T * 2 * C, (long)T * 2 * C)
                                        : act_rsrc(p.A + (long)b * p.a_batch_stride, (long)T * C);
  const half8* __restrict__ Wh = reinterpret_cast<const half8*>(p.Wf_hi);
  const half8* __restrict__ Wl = reinterpret_cast<const half8*>(p.Wf_lo);

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int srow = tid >> 4, sc4 = tid & 15;          // staging: 16 threads per row, SROWS rows per pass
  int a_base[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int trow = wm * (BM / WM) + i * 32;          // first row of this MFMA tile
    a_base[i] = ((trow / BMJ) * RJ + trow % BMJ + (lane & 31)) * RS + (lane >> 5) * 16;
  }
  const int nt0 = wn * TN;                            // first N fragment of this wave

  float gm0 = 0.f, gr0 = 0.f, gm1 = 0.f, gr1 = 0.f;
  float4 gga, gba, ggg, gbg;
  if (GLU) {
    gm0 = p.glu_mr[b * 4 + 0]; gr0 = p.glu_mr[b * 4 + 1]; gm1 = p.glu_mr[b * 4 + 2]; gr1 = p.glu_mr[b * 4 + 3];
  }
  ASW_PHASE_MARK(t_begin);
#ifdef ASW_PHASE_TIMING
  unsigned long long t_stage = 0, t_loop = 0;
#endif
  for (int cc = 0; cc < C / 64; ++cc) {
    ASW_PHASE_MARK(t_s0);
    if (GLU) {
      gga = *reinterpret_cast<const float4*>(p.glu_gamma + cc * 64 + sc4 * 4);
      gba = *reinterpret_cast<const float4*>(p.glu_beta + cc * 64 + sc4 * 4);
      ggg = *reinterpret_cast<const float4*>(p.glu_gamma + C + cc * 64 + sc4 * 4);
      gbg = *reinterpret_cast<const float4*>(p.glu_beta + C + cc * 64 + sc4 * 4);
    }
    __syncthreads();                                   // previous slice fully consumed
    // ---- stage + split the image of this channel slice (8 loads per thread in flight: 8 rows, or 4 rows of value + gate
    // halves where the accumulators leave no room for more)
    for (int r0 = 0; r0 < R; r0 += SROWS * SU) {
      float4 buf[SU];
      float4 gate[GLU ? SU : 1];
      bool okr[GLU ? SU : 1];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int row = r0 + u * SROWS + srow;
        int g;
        bool ok = row < R;
        if (!POLY) {
          g = m0 - pad + row;
        } else {
          const int ph = pb * PH + row / RJ;
          g = dil * (jb * BMJ + row % RJ - (taps - 1) / 2) + ph;
          ok = ok && ph < dil && (jb * BMJ + row % RJ - (taps - 1) / 2) >= 0;
        }
        ok = ok && g >= 0 && g < T;
        if (GLU) {
          buf[u] = act_load4(rX, (long)g * 2 * C + cc * 64 + sc4 * 4, ok);
          gate[u] = act_load4(rX, (long)g * 2 * C + C + cc * 64 + sc4 * 4, ok);
          okr[u] = ok;
        } else {
          buf[u] = act_load4(rX, (long)g * C + cc * 64 + sc4 * 4, ok);
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
          // the normalised rows of the tile's own output range go out once as well: the skip connection of an
          // encoder block, and (C > 64, where the image holds one channel slice at a time) this layer's residual
          const int g = m0 - pad + r0 + u * SROWS + srow;
          if (p.glu_out && okr[u] && g >= m0 && g < m0 + BM)
            *reinterpret_cast<float4*>(p.glu_out + ((long)b * T + g) * C + cc * 64 + sc4 * 4) = o;
        }
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int row = r0 + u * SROWS + srow;
        if (row < R) {
          half4 hi, lo;
          split4t<NTERM>(buf[u], hi, lo);
          *reinterpret_cast<half4*>(img + row * RS + sc4 * 8) = hi;
          if (NTERM == 3 || C == 64) *reinterpret_cast<half4*>(img + row * RS + 128 + sc4 * 8) = lo;
        }
      }
    }
    __syncthreads();
    // ---- taps x k-steps, B fragments double-buffered in registers
    auto bload = [&](int tap, int ks, half8 (&bh)[TN], half8 (&bl)[TN]) {
      const long kg = (long)tap * (C / 16) + cc * 4 + ks;          // global k-step
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const long o = (kg * NT + nt0 + j) * 64 + lane;
        bh[j] = Wh[o];
        if (NTERM == 3) bl[j] = Wl[o];
      }
    };
    // A fragments are double buffered in registers, one k-step ahead: left to itself the compiler
    // keeps ONE fragment register and waits for every ds_read right before its MFMA
    // (ds_read -> s_waitcnt lgkmcnt(0) -> mfma, four times per k-step), i.e. no LDS read of a wave
    // ever overlaps its own MFMAs.
    auto aload = [&](int tap, int ks, half8 (&ah)[TM], half8 (&al)[TM]) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const char* q = img + a_base[i] + tap * tapstep * RS + ks * 32;
        ah[i] = *reinterpret_cast<const half8*>(q);
        if (NTERM == 3) al[i] = *reinterpret_cast<const half8*>(q + 128);
      }
    };
    auto mma = [&](const half8 (&ah)[TM], const half8 (&al)[TM], const half8 (&bh)[TN], const half8 (&bl)[TN]) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (NTERM == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    };
    // One B register buffer per k-step of a tap: the fragment for (tap+1, ks) is requested
    // right after (tap, ks) has been multiplied, i.e. three k-steps (600-1200 MFMA cycles)
    // before its use -- enough to cover an L2 hit without the register cost of a second
    // whole-tap set (which halves occupancy; measured slower for C <= 128).
    // (measured, T = 48 000 batch 64: C = 64 268 -> 280 TFLOP/s, C = 512 361 -> 368, C = 256 unchanged;
    // at C = 128 the 32 extra registers cost more than the overlap gains, 305 -> 301, so it keeps
    // the single buffer)
    constexpr bool ADB = C != 128;
    half8 qh[QD][TN], ql[QD][TN];
    half8 ah[ADB ? 2 : 1][TM], al[ADB ? 2 : 1][TM];
    ASW_PHASE_MARK(t_s1);
#pragma unroll
    for (int ks = 0; ks < QD; ++ks) bload(0, ks, qh[ks], ql[ks]);
    if (ADB) aload(0, 0, ah[0], al[0]);
    // (Measured in round 3 and dropped here: the same loop with every load unconditional -- clamped past-the-end
    // taps -- and the k-steps pinned by sched_barrier, which lifts the transposed C = 64 kernel of resstack.hip by
    // 5-8 %: C = 128 +0.7 %, C = 256 +-0, C = 512 -2 %; unconditional loads without the pinning -3...-10 %.)
    for (int tap = 0; tap < taps; ++tap) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ADB) {                          // next k-step's A fragments
          const int nks = (ks + 1) & 3, ntp = tap + (ks == 3 ? 1 : 0);
          if (ntp < taps) aload(ntp, nks, ah[(ks + 1) & 1], al[(ks + 1) & 1]);
        } else {
          aload(tap, ks, ah[0], al[0]);
        }
        mma(ah[ADB ? (ks & 1) : 0], al[ADB ? (ks & 1) : 0], qh[ks % QD], ql[ks % QD]);
        const int nk = ks + QD, ntap = tap + nk / 4;               // QD k-steps ahead
        if (ntap < taps) bload(ntap, nk % 4, qh[ks % QD], ql[ks % QD]);
      }
    }
#ifdef ASW_PHASE_TIMING
    {
      // make the timestamp wait for the MFMAs: read one accumulator lane
      float sink = acc[0][0][0];
      asm volatile("" ::"v"(sink));
      const unsigned long long t_s2 = __builtin_readcyclecounter();
      t_stage += t_s1 - t_s0;
      t_loop += t_s2 - t_s1;
    }
#endif
  }
  ASW_PHASE_MARK(t_epi0);
  if constexpr (C == 64) {
    // The residual of this layer is its own input, and at C = 64 the whole input row of every
    // output row still sits in the LDS image (one channel slice) as fp16 hi + lo.  Taking it from
    // there (x = hi + lo, 2^-22 relative) instead of re-loading it from global memory removes the
    // load latency from the epilogue, which is 44 % of a workgroup's time at this width
    // (tests/micro/phase_timing.py).  Read before the first slab barrier: the slab aliases the image.
    using G = EpiGeom<C, WM, WN>;
    float4 rpre[TM * G::NSTEP * G::VPL];
    const int sub = lane / G::LPR, lc = lane % G::LPR;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int st = 0; st < G::NSTEP; ++st) {
        const int sr = (st * G::NW + wid) * G::RPI + sub;
        const int trow = (sr >> 5) * (BM / WM) + tm * 32 + (sr & 31);
        const int irow = POLY ? (trow / BMJ) * RJ + trow % BMJ + (taps - 1) / 2 : trow + pad;
#pragma unroll
        for (int q = 0; q < G::VPL; ++q) {
          const int col = (lc + q * G::LPR) * 4;
          const half4 hi = *reinterpret_cast<const half4*>(img + irow * RS + col * 2);
          const half4 lo = *reinterpret_cast<const half4*>(img + irow * RS + 128 + col * 2);
          rpre[(tm * G::NSTEP + st) * G::VPL + q] = make_float4((float)hi[0] + (float)lo[0], (float)hi[1] + (float)lo[1],
                                                                (float)hi[2] + (float)lo[2], (float)hi[3] + (float)lo[3]);
        }
      }
    epilogue<BM, C, WM, WN, true, false, true, false, ResRows<BM, PH, POLY>, true>(
        acc, p, smem, __builtin_ldexpf(1.0f, -p.w_shift), ResRows<BM, PH, POLY>{m0, jb, pb, dil, T}, blockIdx, gridDim.y, rpre);
  } else {
    epilogue<BM, C, WM, WN, true, false, true, false>(acc, p, smem, __builtin_ldexpf(1.0f, -p.w_shift),
                                                      ResRows<BM, PH, POLY>{m0, jb, pb, dil, T}, blockIdx, gridDim.y);
  }
#ifdef ASW_PHASE_TIMING
  {
    const unsigned long long t_end = __builtin_readcyclecounter();
    if (threadIdx.x == 0) {
      atomicAdd(&g_phase_cycles[0], t_stage);
      atomicAdd(&g_phase_cycles[1], t_loop);
      atomicAdd(&g_phase_cycles[2], t_end - t_epi0);
      atomicAdd(&g_phase_cycles[3], 1ull);
    }
    (void)t_begin;
  }
#endif
}

template <int BM, int C, int WM, int WN, int PH, int QD = 4, bool POLY = (PH > 1), bool GLU = false>
int launch_res(const asw_convgemm_args& a, hipStream_t s) {
  constexpr int BMJ = BM / PH;
  const int RJ = BMJ + (!POLY ? (a.taps - 1) * a.dil : a.taps - 1);
  const size_t img = (size_t)PH * RJ * 272;
  const size_t slab = (size_t)(WM * 32) * (C + 4) * sizeof(float);
  const size_t smem = img > slab ? img : slab;
  if (smem > 160 * 1024) return 1;                     // caller falls back to the generic kernel
  const bool x1 = a.precision == 2;                     // single-pass f16: the one-term instantiation
  auto kern = x1 ? resconv16_kernel<BM, C, WM, WN, PH, QD, POLY, GLU, 1> : resconv16_kernel<BM, C, WM, WN, PH, QD, POLY, GLU, 3>;
  static asw::SmemAttr attr[2];                         // per device and instantiation
  if (int rc = attr[x1].ensure(reinterpret_cast<const void*>(kern), 160 * 1024)) return rc;
  const int gx = !POLY ? asw::cdiv(a.M_out, BM)
                       : asw::cdiv(asw::cdiv(a.M_out, a.dil), BMJ) * asw::cdiv(a.dil, PH);
  dim3 grid(gx, 1, a.B);
  char nm[96];
  int nl = snprintf(nm, sizeof nm, "resconv16<%d,%d,%s%d%s%s>", BM, C, POLY ? "poly" : "ph", PH, QD == 2 ? ",q2" : "", GLU ? ",glu" : "");
  if (asw::prof_detail()) snprintf(nm + nl, sizeof nm - nl, "[B%d M%d N%d K%d d%d]", a.B, a.M_out, a.N, a.taps * a.Cin, a.dil);
  asw::ProfScope prof(s, nm, 2.0 * a.B * (double)a.M_out * a.N * (double)a.taps * a.Cin);
  asw_convgemm_args k = a;
  // C > 64: the image holds one 64-channel slice at a time, so the residual (= the normalised input) is read back
  // from glu_out: the rows a workgroup reads in its epilogue are the ones it stored while staging (same CU, after
  // the barriers of the k-loop)
  if (GLU && C > 64) k.resid = a.glu_out;
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), smem, s, k);
  ASW_LAUNCH_CHECK();
  return ASW_OK;
}



// returns 1 when the layer is not a halo-kernel case (or does not fit LDS)
int try_resconv(const asw_convgemm_args& a, hipStream_t s) {
  const bool shape = a.precision >= 1 && a.Wf_hi && a.Wf_lo && a.ln_gamma && a.stride == 1 && a.taps > 1 &&
                     a.taps % 2 == 1 && a.Cin == a.N && a.a_row_stride == a.Cin && a.resid == a.A && !a.A2 &&
                     !a.mul && !a.stats && a.pad * 2 == (a.taps - 1) * a.dil &&
                     a.a_len == (int64_t)a.M_out * a.Cin && a.a_batch_stride == a.a_len;
  if (!shape) return 1;
  if (a.glu_raw) {
    ASW_CHECK_ARG(a.dil == 1 && a.glu_mr && a.glu_gamma && a.glu_beta,
                  "convgemm: GroupNorm + GLU on load needs dilation 1 and the statistics / affine arrays");
    ASW_CHECK_ARG(a.N == 64 || a.glu_out, "convgemm: GroupNorm + GLU on load at %d channels takes the residual from glu_out", a.N);
  }
  // large dilation: polyphase row sets -- but only while every phase still fills a 32-row
  // MFMA tile; on short sequences (T/dil < 32, e.g. T = 752 at dil 49) most of each tile
  // would be empty (measured: 141 vs 243 TFLOP/s), so those stay contiguous
  const int rows_per_phase = a.M_out / a.dil;
  const bool poly = a.dil >= 16 && rows_per_phase >= 32;
  // Tile / prefetch choices are measured (tests/perf_layers.py, T = 48 000, batch 64):
  //  C = 64  : waves 2x2 (64 rows x 32 columns each) halves the weight fragments every wave
  //            pulls through L1 compared with 4x1 -> 222 -> 257 TFLOP/s.  (A persistent variant
  //            with the weights stationary in registers, 224 VGPRs per wave, was tried: one wave
  //            per SIMD leaves the LDS reads of the A operand exposed -> 160 TFLOP/s.);
  //  C = 128 : B prefetch depth 2 fits 3 waves/SIMD -> 294 -> 312 TFLOP/s (dil 49: 259 -> 279);
  //  C = 256 : depth 4 and depth 2 tied in round 1; with the A fragments double buffered depth 2
  //            (3 waves/SIMD) is 1-2 % ahead (322 -> 325, dilation 49: 298 -> 305);
  //  C = 512 : depth 2 fits 2 workgroups per CU -> 346 -> 366 TFLOP/s, except dilation 49 whose
  //            contiguous halo image (294 extra rows) leaves room for one workgroup anyway.
  // Also measured and dropped: 8-wave 128-row tiles for C >= 256 (305 vs 368), and a persistent
  // variant that double-buffers the image slices (prefetch under the MFMAs, one barrier per
  // slice): 338 vs 330 at C = 256 but 146 vs 239 where the doubled image costs a resident
  // workgroup.  With the epilogue removed the same loops run at 355-385 TFLOP/s, the level of an
  // idealised k-step loop fed from L2 on random data (tests/micro/cu_probe.hip: 400).
  switch (a.N) {
    // Dilation 49 as polyphase dilation-1 convolutions, PH phases per workgroup (measured at
    // T = 48 000, batch 64, TFLOP/s for PH = 1 / 2 / 4): C = 64: 243 / 238 / 205; C = 128: 292 / 307 /
    // 282; C = 256: 300 / 301 / -.  Fewer phases per workgroup mean fewer halo rows in the image
    // (134 / 140 / 152 rows for 128 outputs) and longer runs of one phase -- as long as a phase
    // (M_out / dil rows) still fills the BM / PH rows the workgroup gives it.
    // Round 2 also measured, for C = 64: 256-row tiles with 4 x 1 waves (221 vs 262 TFLOP/s at
    // dilation 1), 2 waves of 128 x 64 (155) and 8 waves 4 x 2 on 256 rows (same wave tile, weight
    // fragments shared by four waves through L1: 266 vs 268): neither LDS, L2 nor the weight path
    // is the limit.  Cycle counters per phase (tests/micro/phase_timing.py): a workgroup spends 16 %
    // staging, 40 % in the k-loop, 44 % in the epilogue; taking the residual from the LDS image
    // instead of global memory and budgeting registers for 4 waves per SIMD gave +3 %.
    case 64:
      // at C = 64 even dilation 7 is better off as 7 single-phase tiles (halo 6 instead of 42 rows per
      // 128 outputs, image 36 instead of 46 KB -> 4 resident workgroups): 238 -> 257 TFLOP/s
      if (a.glu_raw) return launch_res<128, 64, 2, 2, 1, 4, false, true>(a, s);
      if (a.dil >= 7 && a.dil < 16 && rows_per_phase >= 96) return launch_res<128, 64, 2, 2, 1, 4, true>(a, s);
      if (!poly) return launch_res<128, 64, 2, 2, 1>(a, s);
      if (rows_per_phase >= 96) return launch_res<128, 64, 2, 2, 1, 4, true>(a, s);
      return rows_per_phase >= 48 ? launch_res<128, 64, 2, 2, 2>(a, s) : launch_res<128, 64, 2, 2, 4>(a, s);
    case 128:
      if (a.glu_raw) return launch_res<128, 128, 2, 2, 1, 2, false, true>(a, s);
      if (!poly) return launch_res<128, 128, 2, 2, 1, 2>(a, s);
      return rows_per_phase >= 48 ? launch_res<128, 128, 2, 2, 2, 2>(a, s) : launch_res<128, 128, 2, 2, 4, 2>(a, s);
    case 256: {
      // 128-row tiles (wave tile 128 x 64: half the weight-fragment traffic per MFMA, two waves per SIMD
      // instead of three) once they still fill the chip twice over: 313 -> 335 TFLOP/s at T = 48 000,
      // batch 64 (same box).  The same step at C = 128 (256-row tiles) loses, 300 -> 292.
      if (a.glu_raw)
        return (long)asw::cdiv(a.M_out, 128) * a.B >= 512 ? launch_res<128, 256, 1, 4, 1, 2, false, true>(a, s)
                                                          : launch_res<64, 256, 1, 4, 1, 2, false, true>(a, s);
      if (!poly && (long)asw::cdiv(a.M_out, 128) * a.B >= 512) return launch_res<128, 256, 1, 4, 1, 2>(a, s);
      // (polyphase, two phases of 64 rows: 302 -> 307)
      if (poly && rows_per_phase >= 48 && (long)asw::cdiv(a.M_out, 128) * a.B >= 512) return launch_res<128, 256, 1, 4, 2, 2>(a, s);
      return poly ? launch_res<64, 256, 1, 4, 2, 2>(a, s) : launch_res<64, 256, 1, 4, 1, 2>(a, s);
    }
    case 512:
      // polyphase at C = 512 pays only for long phases: 45 rows per phase (T = 144 000) measured 243
      // TFLOP/s against 307 for the contiguous halo image on the same layer shape at T = 48 000
      if (a.glu_raw) return launch_res<64, 512, 1, 4, 1, 2, false, true>(a, s);
      if (poly && a.M_out / a.dil >= 64) return launch_res<64, 512, 1, 4, 2>(a, s);
      return a.dil >= 16 ? launch_res<64, 512, 1, 4, 1>(a, s) : launch_res<64, 512, 1, 4, 1, 2>(a, s);
    default: return 1;
  }
}

template <int BM, int BN, int BK, int WM, int WN, bool LN, bool STATS, bool MUL, bool F16, bool A2F = false>
int launch(const asw_convgemm_args& a, hipStream_t s) {
  constexpr size_t stage = F16 ? (size_t)(BM + BN) * (BK + 8) * 2 * sizeof(_Float16)
                               : (size_t)(BM + BN) * (BK + 4) * sizeof(float);
  constexpr size_t slab = (size_t)(WM * 32) * (BN + 4) * sizeof(float);
  constexpr size_t smem = stage > slab ? stage : slab;
  static_assert(smem <= 160 * 1024, "LDS budget");
  const bool x1 = F16 && a.precision == 2;              // single-pass f16: the one-term instantiation
  const void* kern;
  if constexpr (F16)
    kern = x1 ? reinterpret_cast<const void*>(convgemm16_kernel<BM, BN, BK, WM, WN, LN, STATS, MUL, A2F, 1>)
              : reinterpret_cast<const void*>(convgemm16_kernel<BM, BN, BK, WM, WN, LN, STATS, MUL, A2F, 3>);
  else kern = reinterpret_cast<const void*>(convgemm_kernel<BM, BN, BK, WM, WN, LN, STATS, MUL, A2F>);
  static asw::SmemAttr attr[2];                         // per device and instantiation
  if (int rc = attr[x1].ensure(kern, smem)) return rc;
  ASW_CHECK_ARG(A2F == (a.A2 != nullptr), "convgemm: the skip operand is fused only in the 128-wide statistics tile");
  ASW_CHECK_ARG(a.Cin % BK == 0, "convgemm: Cin=%d not a multiple of BK=%d", a.Cin, BK);
  ASW_CHECK_ARG(a.N % BN == 0, "convgemm: N=%d not a multiple of BN=%d", a.N, BN);
  dim3 grid(asw::cdiv(a.M_out, BM), a.N / BN, a.B);
  if constexpr (F16) grid = dim3((((long)grid.x * a.B + 7) / 8) * 8 * grid.y, 1, 1);   // XCD-aware 1-D order, see the kernel
  std::string pn = asw::prof_name(F16 ? (MUL ? "convgemm16m" : "convgemm16") : (MUL ? "convgemm_m" : "convgemm"),
                                  BM, BN, BK, LN, STATS);
  if (asw::prof_detail()) {
    char sh[64];
    snprintf(sh, sizeof sh, "[B%d M%d N%d K%d s%d]", a.B, a.M_out, a.N, a.taps * a.Cin, a.stride);
    pn += sh;
  }
  asw::ProfScope prof(s, pn, 2.0 * a.B * (double)a.M_out * a.N * (double)a.taps * a.Cin);
  if constexpr (F16) {
    if (x1) hipLaunchKernelGGL((convgemm16_kernel<BM, BN, BK, WM, WN, LN, STATS, MUL, A2F, 1>), grid, dim3(64 * WM * WN), smem, s, a);
    else hipLaunchKernelGGL((convgemm16_kernel<BM, BN, BK, WM, WN, LN, STATS, MUL, A2F, 3>), grid, dim3(64 * WM * WN), smem, s, a);
  } else
    hipLaunchKernelGGL((convgemm_kernel<BM, BN, BK, WM, WN, LN, STATS, MUL, A2F>), grid, dim3(256), smem, s, a);
  ASW_LAUNCH_CHECK();
  return ASW_OK;
}

template <int BM, int BN, int BK, int WM, int WN, bool LN, bool STATS, bool MUL = false>
int launch_prec(const asw_convgemm_args& a, hipStream_t s) {
  return a.precision >= 1 ? launch<BM, BN, BK, WM, WN, LN, STATS, MUL, true>(a, s)
                          : launch<BM, BN, BK, WM, WN, LN, STATS, MUL, false>(a, s);
}

// tile choice for the non-LayerNorm variants; must match asw_convgemm_stats_tiles
inline bool wide_tile(int N) { return N % 128 == 0; }
// f16x3 only: 2 = 256x256 (8 waves), 0 = 128x128.  The big tile halves the bytes pulled per
// MAC but needs a grid of >= 2 workgroups per CU to stay balanced (measured: mask encoder
// 242 -> 267 TFLOP/s, deep down/up convs 169 -> 201, but the 288-workgroup QKV GEMM is
// faster on 128x128).
// 3 = 192x256: the same 8-wave kernel with three MFMA row tiles per wave, for sequence lengths
// that leave a 256-row tile a quarter or more empty (the bottleneck-side convolutions: 188 rows at
// T = 48 000, 563 at T = 144 000 -> 2 % instead of 27 % of the MFMAs on padding rows).
inline int wide_tile_kind(int B, int M_out, int N, int K) {
  static const int min_k = getenv("ASW_WIDE_MIN_K") ? atoi(getenv("ASW_WIDE_MIN_K")) : 256;   // A/B measurements
  if (M_out <= 128 || N % 256 != 0 || K < min_k) return 0;
  const long blocks = (long)asw::cdiv(M_out, 256) * (N / 256) * B;
  if (blocks < 512) return 0;
  const long pad256 = (long)asw::cdiv(M_out, 256) * 256, pad192 = (long)asw::cdiv(M_out, 192) * 192;
  return pad192 * 10 <= pad256 * 9 ? 3 : 2;                  // at least 10 % fewer padded rows
}

}  // namespace

#ifdef ASW_PHASE_TIMING
extern "C" int asw_debug_gemm_cycles(unsigned long long* out5, int reset) {
  ASW_HIP(hipMemcpyFromSymbol(out5, HIP_SYMBOL(g_gemm_cycles), 5 * sizeof(unsigned long long)));
  if (reset) {
    const unsigned long long z[5] = {0, 0, 0, 0, 0};
    ASW_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_cycles), z, sizeof z));
  }
  return ASW_OK;
}
extern "C" int asw_debug_phase_cycles(unsigned long long* out4, int reset) {
  ASW_HIP(hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_phase_cycles), 4 * sizeof(unsigned long long)));
  if (reset) {
    const unsigned long long z[4] = {0, 0, 0, 0};
    ASW_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof z));
  }
  return ASW_OK;
}
#endif

namespace asw { int try_downconv64(const asw_convgemm_args& a, hipStream_t s); }

extern "C" int asw_f16x3_overflow_count(int reset, uint32_t* count) {
  ASW_CHECK_ARG(count != nullptr, "f16x3_overflow_count: null pointer");
  unsigned int v = 0;
  ASW_HIP(hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_f16x3_overflow), sizeof v));      // waits for the device
  if (reset && v) {
    const unsigned int z = 0;
    ASW_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_f16x3_overflow), &z, sizeof z));
  }
  *count = v;
  return ASW_OK;
}

extern "C" int asw_convgemm_stats_tiles(int M_out, int N) {
  // upper bound over the tile choices of both precisions (the smallest tiles)
  if (wide_tile(N)) return asw::cdiv(M_out, 128) * (N / 128);
  return asw::cdiv(M_out, 256) * (N / 64);
}

extern "C" int asw_split_weights_f16(const float* w, size_t n, uint16_t* hi, uint16_t* lo, int32_t* w_shift) {
  ASW_CHECK_ARG(w && hi && lo && w_shift, "split_weights: null pointer");
  float mx = 0.f;
  for (size_t i = 0; i < n; ++i) { const float a = w[i] < 0 ? -w[i] : w[i]; if (a > mx) mx = a; }
  ASW_CHECK_ARG(mx == mx && mx < 3.0e38f, "split_weights: non-finite weight");
  // largest power of two with max|w| * 2^shift < 2048 (fp16 keeps 11 significant bits there and
  // typical weights, 10-100x below the maximum, still have normal lo parts); bounded to +-24.
  int shift = 0;
  if (mx > 0.f) {
    int e;
    (void)frexpf(mx, &e);                  // mx = f * 2^e, f in [0.5,1)
    shift = 11 - e;
    if (shift > 24) shift = 24;
    if (shift < -24) shift = -24;
  }
  const float sc = ldexpf(1.0f, shift);
  for (size_t i = 0; i < n; ++i) {
    float c = w[i] * sc;
    if (c > 65504.f) c = 65504.f;
    if (c < -65504.f) c = -65504.f;
    const _Float16 h = (_Float16)c;
    const _Float16 l = (_Float16)(c - (float)h);
    memcpy(hi + i, &h, 2);
    memcpy(lo + i, &l, 2);
  }
  *w_shift = shift;
  return ASW_OK;
}

extern "C" int asw_pack_fragments_f16(const float* Wt, int N, int K, uint16_t* hi, uint16_t* lo, int32_t* w_shift) {
  ASW_CHECK_ARG(Wt && hi && lo && w_shift, "pack_fragments: null pointer");
  ASW_CHECK_ARG(N > 0 && K > 0 && N % 32 == 0 && K % 16 == 0, "pack_fragments: N %% 32, K %% 16 required (N=%d K=%d)", N, K);
  const size_t n = (size_t)N * K;
  uint16_t* th = new uint16_t[2 * n];
  uint16_t* tl = th + n;
  int rc = asw_split_weights_f16(Wt, n, th, tl, w_shift);
  if (rc == ASW_OK) {
    const int NT = N / 32;
    for (int ks = 0; ks < K / 16; ++ks)
      for (int nt = 0; nt < NT; ++nt)
        for (int l = 0; l < 64; ++l) {
          const size_t src = (size_t)(nt * 32 + (l & 31)) * K + ks * 16 + 8 * (l >> 5);
          const size_t dst = (((size_t)ks * NT + nt) * 64 + l) * 8;
          memcpy(hi + dst, th + src, 16);
          memcpy(lo + dst, tl + src, 16);
        }
  }
  delete[] th;
  return rc;
}

extern "C" int asw_mask_path_f16x3(const asw_maskpath_args* args, void* stream) { return launch_mask_path(args, stream); }

extern "C" int asw_convgemm_f32(const asw_convgemm_args* args, void* stream) {
  ASW_CHECK_ARG(args != nullptr, "convgemm: null args");
  asw_convgemm_args a = *args;
  hipStream_t s = asw::as_stream(stream);
  if (a.stats) {
    a.stats_stride = asw_convgemm_stats_tiles(a.M_out, a.N);
    ASW_HIP(hipMemsetAsync(a.stats, 0, (size_t)a.B * a.stats_stride * 4 * sizeof(float), s));
  }
  ASW_CHECK_ARG(a.A && a.out, "convgemm: null tensor");
  ASW_CHECK_ARG(a.precision >= 0 && a.precision <= 2, "convgemm: precision %d", a.precision);
  if (a.precision >= 1) ASW_CHECK_ARG(a.Wt_hi && a.Wt_lo, "convgemm: the f16 modes need Wt_hi/Wt_lo");
  else ASW_CHECK_ARG(a.Wt != nullptr, "convgemm: null weights");
  ASW_CHECK_ARG(a.B > 0 && a.M_out > 0 && a.N > 0 && a.Cin > 0 && a.taps > 0, "convgemm: bad dims");
  ASW_CHECK_ARG(a.a_len > 0 && a.a_len < (int64_t)1 << 29, "convgemm: a_len=%lld per batch item exceeds the 2 GiB buffer descriptor",
                (long long)a.a_len);
  ASW_CHECK_ARG(a.a_row_stride % 4 == 0 && a.a_batch_stride % 4 == 0 && a.a_len % 4 == 0 && a.Cin % 8 == 0,
                "convgemm: strides must be multiples of 4 floats, Cin of 8");
  ASW_CHECK_ARG(a.B <= 65535, "convgemm: batch %d exceeds grid.z", a.B);
  const bool stats = a.stats != nullptr;
  if (stats) ASW_CHECK_ARG(a.chan_mod >= 2 && a.chan_mod % 2 == 0, "convgemm: stats need even chan_mod");
  ASW_CHECK_ARG(!(a.resid && !a.ln_gamma), "convgemm: a residual is fused only together with LayerNorm");
  ASW_CHECK_ARG(!(a.mul && (a.ln_gamma || stats || !wide_tile(a.N))),
                "convgemm: the gate tensor is fused only in the plain 128-wide tile");
  if (a.ln_gamma) {
    ASW_CHECK_ARG(a.ln_beta != nullptr && a.resid != nullptr, "convgemm: LayerNorm needs beta and a residual");
    ASW_CHECK_ARG(!stats, "convgemm: LayerNorm + stats epilogue is not a reference layer");
    {
      const int rc = try_resconv(a, s);
      if (rc != 1) return rc;
    }
    ASW_CHECK_ARG(!a.glu_raw, "convgemm: GroupNorm + GLU on load exists only in the halo-staged f16x3 layer "
                              "(C_in == N == 64, dilation 1, fragment-order weights)");
    switch (a.N) {
      case 64: return launch_prec<256, 64, 32, 4, 1, true, false>(a, s);
      case 128: return launch_prec<128, 128, 32, 2, 2, true, false>(a, s);
      case 256: return launch_prec<64, 256, 32, 1, 4, true, false>(a, s);
      case 512: return launch_prec<64, 512, 16, 1, 4, true, false>(a, s);
      case 1024: return launch_prec<32, 1024, 16, 1, 4, true, false>(a, s);
      default:
        return asw::set_error(ASW_ERR_ARG, "convgemm: LayerNorm width %d unsupported (64..1024, power of 2)", a.N);
    }
  }
  ASW_CHECK_ARG(!a.glu_raw, "convgemm: GroupNorm + GLU on load belongs to a residual layer (LayerNorm + residual)");
  {
    const int rc = asw::try_downconv64(a, s);          // stride-2 convolutions of a 64-channel input (downconv.hip)
    if (rc != 1) return rc;
  }
  if (wide_tile(a.N)) {
    if (a.precision >= 1) {
      // f16x3 is bound by the bytes each CU can pull per cycle, so take the largest tile the
      // shape fills: 256x256 (8 waves, 1/32 B per MAC), 256x128, else 128x128 (1/16 B per MAC)
      const int t = wide_tile_kind(a.B, a.M_out, a.N, a.taps * a.Cin);
      static const bool no_pipe = getenv("ASW_NO_PIPE") != nullptr;           // A/B switch for measurements
      // (192-row tiles, i.e. a single row tile per item and a long K, stay on the two-barrier kernel:
      // with so little reuse of a weight fragment the global B path loses, 296 vs 322 TFLOP/s)
      if (!no_pipe && a.Wf_hi && a.Wf_lo && t == 2 && (a.taps * a.Cin) % 16 == 0 && a.N % 32 == 0) {
        // Short K (the decoder's transposed convolutions): two independent 4-wave workgroups of 128 rows per CU, same
        // wave tile -- one drains its tile while the other computes (K = 512: 234 -> 260 TFLOP/s, K = 256: 186 -> 193;
        // at K >= 896 and in the mask path the 8-wave tile is 2-4 % ahead).  ASW_PIPE_HALF_TILE=0 / 1 forces one form.
        static const int force = getenv("ASW_PIPE_HALF_TILE") ? atoi(getenv("ASW_PIPE_HALF_TILE")) : -1;
        const bool half_tile = force >= 0 ? force == 1 : a.taps * a.Cin <= 512;
        if (half_tile && !a.mul) {
          if (stats && a.A2) return launch_pipe<128, true, false, true, 32, 1>(a, s);
          return stats ? launch_pipe<128, true, false, false, 32, 1>(a, s) : launch_pipe<128, false, false, false, 32, 1>(a, s);
        }
        if (a.mul) return launch_pipe<256, false, true, false>(a, s);
        if (stats && a.A2) return launch_pipe<256, true, false, true>(a, s);
        return stats ? launch_pipe<256, true, false, false>(a, s) : launch_pipe<256, false, false, false>(a, s);
      }
      if (t == 3) {
        if (a.mul) return launch<192, 256, 32, 2, 4, false, false, true, true>(a, s);
        if (stats && a.A2) return launch<192, 256, 32, 2, 4, false, true, false, true, true>(a, s);
        return stats ? launch<192, 256, 32, 2, 4, false, true, false, true>(a, s)
                     : launch<192, 256, 32, 2, 4, false, false, false, true>(a, s);
      }
      if (t == 2) {
        if (a.mul) return launch<256, 256, 32, 2, 4, false, false, true, true>(a, s);
        if (stats && a.A2) return launch<256, 256, 32, 2, 4, false, true, false, true, true>(a, s);
        return stats ? launch<256, 256, 32, 2, 4, false, true, false, true>(a, s)
                     : launch<256, 256, 32, 2, 4, false, false, false, true>(a, s);
      }
      if (t == 1) {
        if (a.mul) return launch<256, 128, 32, 4, 2, false, false, true, true>(a, s);
        if (stats && a.A2) return launch<256, 128, 32, 4, 2, false, true, false, true, true>(a, s);
        return stats ? launch<256, 128, 32, 4, 2, false, true, false, true>(a, s)
                     : launch<256, 128, 32, 4, 2, false, false, false, true>(a, s);
      }
    }
    if (a.mul) return launch_prec<128, 128, 32, 2, 2, false, false, true>(a, s);
    if (stats && a.A2)
      return a.precision >= 1 ? launch<128, 128, 32, 2, 2, false, true, false, true, true>(a, s)
                              : launch<128, 128, 32, 2, 2, false, true, false, false, true>(a, s);
    return stats ? launch_prec<128, 128, 32, 2, 2, false, true>(a, s) : launch_prec<128, 128, 32, 2, 2, false, false>(a, s);
  }
  ASW_CHECK_ARG(a.N % 64 == 0, "convgemm: N=%d must be a multiple of 64", a.N);
  return stats ? launch_prec<256, 64, 32, 4, 1, false, true>(a, s) : launch_prec<256, 64, 32, 4, 1, false, false>(a, s);
}
