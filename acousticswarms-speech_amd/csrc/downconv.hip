// downconv.hip -- the stride-2 encoder convolutions that read a 64-channel tensor (EncoderBlock.conv1 of the two
// full-rate levels: sep/training/SpeakerLocalization/network.py:105-108, and the same class of the separation network)
// on the gfx950 matrix cores, f16x3 split-operand arithmetic, with the GroupNorm partial sums in the epilogue.
//
// The generic chunked GEMM (convgemm16 / convgemm16p) re-stages every input row once per tap and runs these
// K = 448 layers at 235 TFLOP/s.  Here the workgroup stages the rows of its 128 outputs ONCE, split to fp16 hi / lo,
// as two LDS images -- even input rows and odd input rows -- so that a tap of the stride-2 convolution reads
// consecutive image rows (row stride 272 B: conflict-free 16-byte fragment reads, as in resstack.hip); all taps read
// those images.  Accumulators are transposed (weight fragment = A operand): a lane holds one output row and four
// consecutive channels per register group, i.e. float4 stores; a wave owns 32 channels x 128 rows, so a weight
// fragment pulled from L2 feeds 12 MFMAs (four row fragments) instead of 6.
#include <cstdlib>

#include "asw_common.h"
#include "mfma_util.h"

namespace {
using namespace asw_mfma;

constexpr int RS = 272;                  // bytes per image row: 64 channels hi | lo + pad
constexpr int C = 64;
constexpr int R0 = 128;                  // output rows per workgroup
constexpr int TM = R0 / 32;              // row fragments per wave
constexpr int NW = 4;                    // waves = 32-channel blocks per workgroup (128 output channels)

struct DArgs {
  const float* x;
  float* out;
  float* stats;
  const half8* Wh;
  const half8* Wl;
  const float* bias;
  int B, T_in, M_out, N, taps, pad, chan_mod, stats_stride, ncol, nrt, img_rows;
  float scale;
};

template <int NTERM, int QD>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2)))
void downconv64_kernel(const DArgs p) {
  constexpr int NTHR = 64 * NW, SROWS = NTHR / 16;
  extern __shared__ __align__(16) char smem[];
  char* imgE = smem;                                   // even input rows
  char* imgO = smem + (size_t)p.img_rows * RS;          // odd input rows
  float* red = reinterpret_cast<float*>(smem + (size_t)2 * p.img_rows * RS);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // XCD-aware order (workgroup L runs on XCD L % 8): the column tiles of one row tile, then the next row tile of the
  // same XCD's contiguous run -- neighbours share their input rows in one L2
  const int total = p.B * p.nrt * p.ncol;
  const int per = (total + 7) >> 3;
  const int idx = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (idx >= total) return;
  const int ct = idx % p.ncol, rt = (idx / p.ncol) % p.nrt, b = idx / (p.ncol * p.nrt);
  const int m0 = rt * R0, n0 = ct * (32 * NW);
  const int taps = p.taps, pad = p.pad, he = pad >> 1, ho = (pad + 1) >> 1;
  const int T = p.T_in;
  // ---- stage both parity images: image row i of parity q is input row 2 (m0 + i - h_q) + q
  {
    const __amdgpu_buffer_rsrc_t rX = act_rsrc(p.x + (long)b * T * C, (long)T * C);
    const int srow = tid >> 4, sc4 = tid & 15;
    const int R_img = 2 * p.img_rows;
    constexpr int SU = 18;
    for (int r0 = 0; r0 < R_img; r0 += SROWS * SU) {
      float4 buf[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int row = r0 + u * SROWS + srow;
        const int q = row >= p.img_rows ? 1 : 0, i = row - q * p.img_rows;
        const int g = 2 * (m0 + i - (q ? ho : he)) + q;
        buf[u] = act_load4(rX, (long)g * C + sc4 * 4, row < R_img && g >= 0 && g < T);
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int row = r0 + u * SROWS + srow;
        if (row < R_img) {
          half4 hi, lo;
          split4t<NTERM>(buf[u], hi, lo);
          *reinterpret_cast<half4*>(smem + row * RS + sc4 * 8) = hi;
          if (NTERM == 3) *reinterpret_cast<half4*>(smem + row * RS + 128 + sc4 * 8) = lo;
        }
      }
    }
  }
  __syncthreads();
  // ---- taps x 4 k-steps: this wave's 32 output channels x 128 rows
  floatx16 acc[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int h = lane >> 5;
  const int NT = p.N / 32, nt = n0 / 32 + wid;
  auto wload = [&](int kg, half8& wh, half8& wl) {
    const long o = ((long)kg * NT + nt) * 64 + lane;
    wh = p.Wh[o];
    if (NTERM == 3) wl = p.Wl[o];
  };
  const int lane_off = (lane & 31) * RS + h * 16;
  auto xload = [&](int tap, int ks, half8 (&xh)[TM], half8 (&xl)[TM]) {
    const int o = tap - pad;                              // input row 2 j + o
    const char* base = (o & 1) ? imgO + (((o - 1) >> 1) + ho) * RS : imgE + ((o >> 1) + he) * RS;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const char* q = base + i * 32 * RS + lane_off + ks * 32;
      xh[i] = *reinterpret_cast<const half8*>(q);
      if (NTERM == 3) xl[i] = *reinterpret_cast<const half8*>(q + 128);
    }
  };
  half8 wh[QD], wl[QD];
  half8 xh[2][TM], xl[2][TM];
#pragma unroll
  for (int q = 0; q < QD; ++q) wload(q, wh[q], wl[q]);
  xload(0, 0, xh[0], xl[0]);
  const int nks = taps * 4;
  // every load unconditional (clamped), k-steps pinned in program order: see resstack.hip kloop
  for (int tap = 0; tap < taps; ++tap) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      int ntp = tap + (ks == 3 ? 1 : 0);
      ntp = ntp < taps ? ntp : taps - 1;
      xload(ntp, (ks + 1) & 3, xh[(ks + 1) & 1], xl[(ks + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      const int s = ks % QD, xb = ks & 1;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (NTERM == 3) {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], xl[xb][i], acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[s], xh[xb][i], acc[i], 0, 0, 0);
        }
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], xh[xb][i], acc[i], 0, 0, 0);
      }
      int kg = tap * 4 + ks + QD;
      kg = kg < nks ? kg : nks - 1;
      wload(kg, wh[s], wl[s]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- epilogue in registers: + bias, GroupNorm partial sums (group = channel half of chan_mod), float4 stores
  const int ch0 = n0 + wid * 32;                        // first channel of this wave
  float s1 = 0.f, s2 = 0.f;
  float* __restrict__ ob = p.out + (long)b * p.M_out * p.N;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = m0 + i * 32 + (lane & 31);
    const bool ok = row < p.M_out;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c0 = ch0 + q * 8 + h * 4;
      const float4 bi = p.bias ? *reinterpret_cast<const float4*>(p.bias + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 v = make_float4(acc[i][q * 4 + 0] * p.scale + bi.x, acc[i][q * 4 + 1] * p.scale + bi.y,
                             acc[i][q * 4 + 2] * p.scale + bi.z, acc[i][q * 4 + 3] * p.scale + bi.w);
      if (ok) {
        s1 += (v.x + v.y) + (v.z + v.w);
        s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        *reinterpret_cast<float4*>(ob + (long)row * p.N + c0) = v;
      }
    }
  }
  if (p.stats) {
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    __syncthreads();                                    // (the images are dead: red aliases nothing, but keep the order)
    if (lane == 0) { red[wid * 2 + 0] = s1; red[wid * 2 + 1] = s2; }
    __syncthreads();
    if (tid < 4) {
      // slot value tid: (sum0, sumsq0, sum1, sumsq1); a wave's 32 channels lie in one group (chan_mod / 2 is a multiple of 32)
      const int grp = tid >> 1, which = tid & 1, half_mod = p.chan_mod >> 1;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w)
        if ((((n0 + w * 32) % p.chan_mod) >= half_mod ? 1 : 0) == grp) s += red[w * 2 + which];
      const long part = (long)b * p.stats_stride + (long)rt * p.ncol + ct;
      p.stats[part * 4 + tid] = s;
    }
  }
}

}  // namespace

namespace asw {
// returns 1 when the layer is not a case of this kernel (the caller falls back to the generic GEMM)
int try_downconv64(const asw_convgemm_args& a, hipStream_t s) {
  static const bool off = getenv("ASW_NO_DOWNCONV") != nullptr;      // A/B switch for measurements
  const bool shape = !off && a.precision >= 1 && a.Wf_hi && a.Wf_lo && a.Cin == C && a.stride == 2 && a.dil == 1 &&
                     a.taps % 2 == 1 && a.taps >= 3 && a.taps <= 15 && a.pad == a.taps / 2 && a.a_row_stride == C &&
                     !a.A2 && !a.mul && !a.resid && !a.ln_gamma && !a.glu_raw && a.relu == 0 && a.N % (32 * NW) == 0 &&
                     a.a_batch_stride == a.a_len && a.a_len % C == 0 &&
                     (!a.stats || (a.chan_mod % 64 == 0 && a.chan_mod >= 64));
  if (!shape) return 1;
  const int T_in = (int)(a.a_len / C);
  if (a.M_out != (T_in + 2 * a.pad - a.taps) / 2 + 1) return 1;
  DArgs k = {};
  k.x = a.A; k.out = a.out; k.stats = a.stats; k.bias = a.bias;
  k.Wh = reinterpret_cast<const half8*>(a.Wf_hi); k.Wl = reinterpret_cast<const half8*>(a.Wf_lo);
  k.B = a.B; k.T_in = T_in; k.M_out = a.M_out; k.N = a.N; k.taps = a.taps; k.pad = a.pad; k.chan_mod = a.chan_mod;
  k.stats_stride = a.stats_stride; k.ncol = a.N / (32 * NW); k.nrt = cdiv(a.M_out, R0);
  k.img_rows = R0 + a.pad + 1;
  k.scale = ldexpf(1.0f, -a.w_shift);
  const size_t smem = (size_t)2 * k.img_rows * RS + 64;
  const bool x1 = a.precision == 2;
  auto kern = x1 ? downconv64_kernel<1, 4> : downconv64_kernel<3, 4>;
  static SmemAttr attr[2];
  if (int rc = attr[x1].ensure(reinterpret_cast<const void*>(kern), smem)) return rc;
  const int total = k.B * k.nrt * k.ncol;
  char nm[96];
  int nl = snprintf(nm, sizeof nm, "downconv64<128x128,%s>", a.stats ? "stats" : "plain");
  if (prof_detail()) snprintf(nm + nl, sizeof nm - nl, "[B%d M%d N%d K%d s2]", a.B, a.M_out, a.N, a.taps * C);
  ProfScope prof(s, nm, 2.0 * a.B * (double)a.M_out * a.N * (double)a.taps * C);
  hipLaunchKernelGGL(kern, dim3(((total + 7) / 8) * 8), dim3(64 * NW), smem, s, k);
  ASW_LAUNCH_CHECK();
  return ASW_OK;
}
}  // namespace asw
