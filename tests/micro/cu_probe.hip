// cu_probe.hip -- ground-truth throughput of one gfx950 CU for the operand paths the f16x3
// conv-GEMM kernels use: MFMA issue (independent / dependent accumulators), LDS fragment
// reads, direct-to-register global fragment loads from an L2-resident array, and their mixes.
// Build: hipcc -O3 --offload-arch=gfx950 cu_probe.hip -o cu_probe ; run on the GPU box.
// Diagnostic only (DESIGN.md section 5 quotes its numbers).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// ---- MFMA only: NACC accumulators used round-robin, CH consecutive MFMAs per accumulator
template <int NACC, int CH>
__global__ __launch_bounds__(256) void k_mfma(float* out, int iters) {
  floatx16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  half8 a, b;
  // pseudo-random operands: the matrix pipe's power (and with it the sustained clock) depends on
  // how many operand bits toggle; zeros or tiny integers overstate what real data can reach
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int r = 0; r < 8; ++r) {
    h = h * 1664525u + 1013904223u; a[r] = (_Float16)(((int)(h >> 16) & 4095) * (1.0f / 2048.0f) - 1.0f);
    h = h * 1664525u + 1013904223u; b[r] = (_Float16)(((int)(h >> 16) & 4095) * (1.0f / 2048.0f) - 1.0f);
  }
  if (iters < 0) { for (int r = 0; r < 8; ++r) { a[r] = 0; b[r] = 0; } iters = -iters; }   // zeros variant
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---- the 16x16x32 shape of the same pipe (4 accumulator registers instead of 16)
typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma16(float* out, int iters) {
  floatx4 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  half8 a, b;
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int r = 0; r < 8; ++r) {
    h = h * 1664525u + 1013904223u; a[r] = (_Float16)(((int)(h >> 16) & 4095) * (1.0f / 2048.0f) - 1.0f);
    h = h * 1664525u + 1013904223u; b[r] = (_Float16)(((int)(h >> 16) & 4095) * (1.0f / 2048.0f) - 1.0f);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---- LDS fragment reads only (ds_read_b128, rows of 272 B like the halo image)
template <int NR>
__global__ __launch_bounds__(256) void k_lds(float* out, int iters) {
  extern __shared__ __align__(16) char lds[];
  for (int i = threadIdx.x; i < 40 * 1024 / 4; i += 256) reinterpret_cast<float*>(lds)[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const char* base = lds + (wid * 32 + (lane & 31)) * 272 + (lane >> 5) * 16;
  half8 s[NR];
  for (int i = 0; i < NR; ++i) for (int r = 0; r < 8; ++r) s[i][r] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const half8 v = *reinterpret_cast<const half8*>(base + ((it + i) & 3) * 32 + (i & 1) * 128);
      s[i] += v;
    }
  }
  float t = 0.f;
  for (int i = 0; i < NR; ++i) for (int r = 0; r < 8; ++r) t += (float)s[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}

// ---- global fragment loads only: every wave streams 1 KiB fragments of a `span`-byte array
template <int NL>
__global__ __launch_bounds__(256) void k_gld(const half8* __restrict__ w, long span_frag, float* out, int iters, int shared) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  // shared = 1: the four waves of a workgroup read the same fragments (L1 reuse)
  long f = ((long)blockIdx.x * 7 + (shared ? 0 : wid) * 1237) % span_frag;
  half8 s[NL];
  for (int i = 0; i < NL; ++i) for (int r = 0; r < 8; ++r) s[i][r] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      s[i] += w[f * 64 + lane];
      f += 1; if (f >= span_frag) f = 0;
    }
  }
  float t = 0.f;
  for (int i = 0; i < NL; ++i) for (int r = 0; r < 8; ++r) t += (float)s[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}

// ---- the residual-conv k-step: TM x {hi,lo} A fragments from LDS, TN x {hi,lo} B fragments
//      from global (BSRC 1) or LDS (BSRC 2) or registers (BSRC 0), 3*TM*TN MFMAs
template <int TM, int TN, int BSRC>
__global__ __launch_bounds__(256) void k_step(const half8* __restrict__ w, long span_frag, float* out, int iters, int lockstep) {
  extern __shared__ __align__(16) char lds[];
  for (int i = threadIdx.x; i < 64 * 1024 / 2; i += 256) {
    unsigned h = (i + 7919u * blockIdx.x) * 2654435761u;
    reinterpret_cast<_Float16*>(lds)[i] = (_Float16)(((int)(h >> 16) & 4095) * (1.0f / 2048.0f) - 1.0f);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const char* abase = lds + ((lane & 31)) * 272 + (lane >> 5) * 16;
  const char* bbase = lds + 36 * 1024 + lane * 16;
  floatx16 acc[TM][TN];
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  half8 bh[TN], bl[TN];
  for (int j = 0; j < TN; ++j) for (int r = 0; r < 8; ++r) { bh[j][r] = (_Float16)(r + j); bl[j][r] = (_Float16)(0.01f * r); }
  // lockstep = 1: every workgroup walks the same fragments in the same order at the same time,
  // as the layers of a real network do (all tiles read one weight matrix front to back)
  long f = lockstep ? (long)wid * 2 * TN : ((long)blockIdx.x * 7 + wid * 1237) % span_frag;
  half8 nh[TN], nl[TN];
  if (BSRC == 1) for (int j = 0; j < TN; ++j) { nh[j] = w[(f + 2 * j) * 64 + lane]; nl[j] = w[(f + 2 * j + 1) * 64 + lane]; }
  for (int it = 0; it < iters; ++it) {
    half8 ah[TM], al[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const char* q = abase + ((i * 32 + (it & 7)) * 272) + (it & 3) * 32;
      ah[i] = *reinterpret_cast<const half8*>(q);
      al[i] = *reinterpret_cast<const half8*>(q + 128);
    }
    if (BSRC == 1) {
#pragma unroll
      for (int j = 0; j < TN; ++j) { bh[j] = nh[j]; bl[j] = nl[j]; }
      f += (lockstep ? 8 : 2) * TN; if (f + 8 * TN >= span_frag) f = lockstep ? (long)wid * 2 * TN : 0;
#pragma unroll
      for (int j = 0; j < TN; ++j) { nh[j] = w[(f + 2 * j) * 64 + lane]; nl[j] = w[(f + 2 * j + 1) * 64 + lane]; }
    } else if (BSRC == 2) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const half8*>(bbase + ((it + j) & 7) * 2048);
        bl[j] = *reinterpret_cast<const half8*>(bbase + ((it + j) & 7) * 2048 + 1024);
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---- the same k-step on the 16x16x32 shape (same operand bytes per FLOP for the same 64 x 64 wave tile:
//      per K = 32, 4 x {hi,lo} A fragments of 16 rows from LDS, 4 x {hi,lo} B fragments from global, 48 MFMAs)
template <int TM, int TN>
__global__ __launch_bounds__(256) void k_step16(const half8* __restrict__ w, long span_frag, float* out, int iters, int lockstep) {
  extern __shared__ __align__(16) char lds[];
  for (int i = threadIdx.x; i < 64 * 1024 / 2; i += 256) {
    unsigned h = (i + 7919u * blockIdx.x) * 2654435761u;
    reinterpret_cast<_Float16*>(lds)[i] = (_Float16)(((int)(h >> 16) & 4095) * (1.0f / 2048.0f) - 1.0f);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const char* abase = lds + lane * 16;                 // conflict-free synthetic layout: the question here is the clock
  floatx4 acc[TM][TN];
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  half8 bh[TN], bl[TN], nh[TN], nl[TN];
  long f = lockstep ? (long)wid * 2 * TN : ((long)blockIdx.x * 7 + wid * 1237) % span_frag;
  for (int j = 0; j < TN; ++j) { nh[j] = w[(f + 2 * j) * 64 + lane]; nl[j] = w[(f + 2 * j + 1) * 64 + lane]; }
  for (int it = 0; it < iters; ++it) {
    half8 ah[TM], al[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const char* q = abase + ((i * 2 + (it & 7) * 8) * 1024);
      ah[i] = *reinterpret_cast<const half8*>(q);
      al[i] = *reinterpret_cast<const half8*>(q + 1024);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) { bh[j] = nh[j]; bl[j] = nl[j]; }
    f += (lockstep ? 8 : 2) * TN; if (f + 8 * TN >= span_frag) f = lockstep ? (long)wid * 2 * TN : 0;
#pragma unroll
    for (int j = 0; j < TN; ++j) { nh[j] = w[(f + 2 * j) * 64 + lane]; nl[j] = w[(f + 2 * j + 1) * 64 + lane]; }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static double time_ms(void (*launch)(), int reps = 3) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch();
  CHECK(hipDeviceSynchronize());
  double best = 1e30;
  for (int r = 0; r < reps; ++r) {
    CHECK(hipEventRecord(e0));
    launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best;
}

static float* g_out; static half8* g_w; static int g_cus; static int g_wgs; static int g_iters; static long g_span;
static double g_ghz;

template <int NACC, int CH> void run_mfma(const char* tag) {
  auto L = []() { hipLaunchKernelGGL((k_mfma<NACC, CH>), dim3(g_cus * g_wgs), dim3(256), 0, 0, g_out, g_iters); };
  const double ms = time_ms(L);
  const double mf = (double)g_cus * g_wgs * 4 * (g_iters < 0 ? -g_iters : g_iters) * NACC * CH;        // wave-level MFMAs
  const double cyc = ms * 1e-3 * g_ghz * 1e9;
  printf("%-44s wgs/cu %d: %.1f cycles per MFMA per SIMD (%.0f TFLOP/s)\n", tag, g_wgs, cyc / (mf / (g_cus * 4)), mf * 32768 / ms * 1e-9);
}
template <int NACC> void run_mfma16(const char* tag) {
  auto L = []() { hipLaunchKernelGGL((k_mfma16<NACC>), dim3(g_cus * g_wgs), dim3(256), 0, 0, g_out, g_iters); };
  const double ms = time_ms(L);
  const double mf = (double)g_cus * g_wgs * 4 * g_iters * NACC;
  const double cyc = ms * 1e-3 * g_ghz * 1e9;
  printf("%-44s wgs/cu %d: %.1f cycles per MFMA per SIMD (%.0f TFLOP/s)\n", tag, g_wgs, cyc / (mf / (g_cus * 4)), mf * 16384 / ms * 1e-9);
}
template <int NR> void run_lds() {
  auto L = []() { hipLaunchKernelGGL((k_lds<NR>), dim3(g_cus * g_wgs), dim3(256), 48 * 1024, 0, g_out, g_iters); };
  const double ms = time_ms(L);
  const double bytes = (double)g_cus * g_wgs * 4 * g_iters * NR * 1024;
  printf("LDS ds_read_b128 x%d                         wgs/cu %d: %.1f B/clk/CU\n", NR, g_wgs, bytes / g_cus / (ms * 1e-3 * g_ghz * 1e9));
}
template <int NL> void run_gld(int shared) {
  static int sh; sh = shared;
  auto L = []() { hipLaunchKernelGGL((k_gld<NL>), dim3(g_cus * g_wgs), dim3(256), 0, 0, g_w, g_span, g_out, g_iters, sh); };
  const double ms = time_ms(L);
  const double bytes = (double)g_cus * g_wgs * 4 * g_iters * NL * 1024;
  printf("global 1 KiB fragment loads x%d span %5.1f MB %s wgs/cu %d: %.1f B/clk/CU (%.2f TB/s)\n", NL, g_span * 1024 / 1e6,
         shared ? "shared " : "private", g_wgs, bytes / g_cus / (ms * 1e-3 * g_ghz * 1e9), bytes / ms * 1e-9);
}
static int g_lock = 0;
template <int TM, int TN, int BSRC> void run_step() {
  auto L = []() { hipLaunchKernelGGL((k_step<TM, TN, BSRC>), dim3(g_cus * g_wgs), dim3(256), 64 * 1024, 0, g_w, g_span, g_out, g_iters, g_lock); };
  const double ms = time_ms(L);
  const double mf = (double)g_cus * g_wgs * 4 * g_iters * TM * TN * 3;
  const double cyc = ms * 1e-3 * g_ghz * 1e9;
  printf("k-step%s TM=%d TN=%d B from %-7s span %5.1f MB wgs/cu %d: %.1f cycles/MFMA/SIMD = %.0f%% of peak (%.0f f16x3-TFLOP/s)\n", g_lock ? " LOCKSTEP" : "", TM, TN,
         BSRC == 0 ? "regs" : BSRC == 1 ? "global" : "LDS", g_span * 1024 / 1e6, g_wgs, cyc / (mf / (g_cus * 4)),
         100.0 * 32.0 / (cyc / (mf / (g_cus * 4))), mf * 32768 / 3 / ms * 1e-9);
}

template <int TM, int TN> void run_step16() {
  auto L = []() { hipLaunchKernelGGL((k_step16<TM, TN>), dim3(g_cus * g_wgs), dim3(256), 64 * 1024, 0, g_w, g_span, g_out, g_iters, g_lock); };
  const double ms = time_ms(L);
  const double mf = (double)g_cus * g_wgs * 4 * g_iters * TM * TN * 3;
  const double cyc = ms * 1e-3 * g_ghz * 1e9;
  printf("k-step16x16x32%s TM=%d TN=%d B from global span %5.1f MB wgs/cu %d: %.1f cycles/MFMA/SIMD = %.0f%% of peak (%.0f f16x3-TFLOP/s)\n", g_lock ? " LOCKSTEP" : "", TM, TN,
         g_span * 1024 / 1e6, g_wgs, cyc / (mf / (g_cus * 4)), 100.0 * 16.0 / (cyc / (mf / (g_cus * 4))), mf * 16384 / 3 / ms * 1e-9);
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  g_cus = prop.multiProcessorCount;
  g_ghz = prop.clockRate * 1e-6;
  printf("device %s: %d CUs, %.2f GHz\n", prop.name, g_cus, g_ghz);
  CHECK(hipMalloc(&g_out, (size_t)g_cus * 8 * 256 * 4));
  const size_t wbytes = 64u << 20;
  CHECK(hipMalloc(&g_w, wbytes));
  {
    std::vector<unsigned short> hw(wbytes / 2);
    unsigned h = 1u;
    for (size_t i = 0; i < hw.size(); ++i) {           // random fp16 in (-2, 2): sign, exponent 13..15, mantissa
      h = h * 1664525u + 1013904223u;
      hw[i] = (unsigned short)(((h >> 31) << 15) | ((13u + ((h >> 20) % 3u)) << 10) | ((h >> 8) & 1023u));
    }
    if (getenv("PROBE_ZERO")) std::fill(hw.begin(), hw.end(), 0);
    CHECK(hipMemcpy(g_w, hw.data(), wbytes, hipMemcpyHostToDevice));
  }
  g_iters = 2000;
  for (int w : {1, 2}) {
    g_wgs = w;
    run_mfma<4, 3>("MFMA 4 accumulators, 3 in a row, random operands");
    run_mfma16<8>("MFMA 16x16x32, 8 accumulators, random operands");
    g_iters = -2000;
    run_mfma<4, 3>("MFMA 4 accumulators, 3 in a row, ZERO operands");
    g_iters = 2000;
  }
  if (!getenv("PROBE_QUICK"))
  for (int w : {1, 2}) {
    g_wgs = w;
    run_mfma<4, 1>("MFMA 4 independent accumulators");
    run_mfma<1, 1>("MFMA 1 accumulator (dependent chain)");
    run_mfma<4, 3>("MFMA 4 accumulators, 3 in a row each");
    run_mfma<2, 3>("MFMA 2 accumulators, 3 in a row each");
  }
  if (!getenv("PROBE_QUICK"))
  for (int w : {1, 2, 4}) { g_wgs = w; run_lds<4>(); run_lds<8>(); }
  if (!getenv("PROBE_QUICK"))
  for (long mb : {1, 7, 32}) {
    g_span = mb * 1024;                      // fragments of 1 KiB
    for (int w : {1, 2, 4}) { g_wgs = w; run_gld<4>(0); run_gld<8>(0); }
    g_wgs = 2; run_gld<8>(1);
  }
  const bool quick = getenv("PROBE_QUICK") != nullptr;
  for (long mb : {2, 7}) {
    g_span = mb * 1024;
    for (int w : {1, 2}) {
      g_wgs = w;
      for (int lock : {0, 1}) {
        g_lock = lock;
        run_step<2, 2, 1>(); run_step<2, 4, 1>(); run_step<2, 1, 1>();
        run_step16<4, 4>(); run_step16<4, 2>();          // wave tiles 64 x 64 and 64 x 32 on the 16x16x32 shape
      }
      g_lock = 0;
      if (quick) continue;
      run_step<2, 2, 0>(); run_step<2, 2, 2>(); run_step<2, 4, 0>(); run_step<1, 2, 1>(); run_step<4, 2, 2>();
    }
  }
  return 0;
}
