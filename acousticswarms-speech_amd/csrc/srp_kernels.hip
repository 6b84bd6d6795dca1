// srp_kernels.hip -- SRP-PHAT pruning map on the device (K10 of SURVEY.md §2.2).
//
// Replaces SRP_PHAT.SRP_Map_WINDOW_torch (sep/Traditional_SP/SRP_Prunning.py:387-434):
//   per analysis window: STFT (nfft 2048, hop 512, rectangular, :404-409) -> PHAT
//   normalisation X/max(|X|,tol) (:414-416) -> per-bin cross-spectrum of every mic pair
//   averaged over frames (:421-426) -> steered response sum over (bin, pair) against
//   exp(+j w (tau_i - tau_j)) (:428-429) -> running maximum over windows starting from a
//   zero map (:248-256,430).
//
// Only bins bin0..bin0+nbins-1 are ever consumed, so the STFT is a DFT-as-GEMM on the
// f32 MFMA pipe restricted to those bins: rows = hop-strided frames of the mixture
// (asw_convgemm_f32 with a_row_stride = hop), columns = cos / -sin twiddles.  The
// reference's [G][nbins][P] complex128 steering table (1.16 GB at G = 17 438) is never
// built: the map kernel regenerates exp(j w dtau) from the G x M propagation delays
// (phase formed and range-reduced in double, sincos in float) and reads only the
// KB-sized cross-spectra.
#include "asw_common.h"

namespace {

// grid (nbins), block 256.  xf: [M][F][ld] with Re at column k, Im at column nb_pad + k.
__global__ __launch_bounds__(256) void phat_cc_kernel(const float* __restrict__ xf, int M, int F, int ld, int nb_pad,
                                                      float tol, const int32_t* __restrict__ pair_i,
                                                      const int32_t* __restrict__ pair_j, int P,
                                                      float* __restrict__ cc /* [nbins][P][2] */) {
  extern __shared__ float sm[];           // [F][M][2]
  const int k = blockIdx.x;
  for (int i = threadIdx.x; i < F * M; i += blockDim.x) {
    const int f = i / M, m = i - f * M;
    const float re = xf[((long)m * F + f) * ld + k];
    const float im = xf[((long)m * F + f) * ld + nb_pad + k];
    float a = hypotf(re, im);
    if (a < tol) a = tol;
    sm[2 * i] = re / a;
    sm[2 * i + 1] = im / a;
  }
  __syncthreads();
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    const int a = pair_i[p], b = pair_j[p];
    float sr = 0.f, si = 0.f;
    for (int f = 0; f < F; ++f) {
      const float ar = sm[2 * (f * M + a)], ai = sm[2 * (f * M + a) + 1];
      const float br = sm[2 * (f * M + b)], bi = sm[2 * (f * M + b) + 1];
      sr += ar * br + ai * bi;            // a * conj(b)
      si += ai * br - ar * bi;
    }
    cc[((long)k * P + p) * 2] = sr / (float)F;
    cc[((long)k * P + p) * 2 + 1] = si / (float)F;
  }
}

constexpr int MAP_W = 8;                   // windows per pass of the map kernel

// grid (ceil(G/256), KS): thread = one grid cluster g, k-slice blockIdx.y.
// part[ks][w][g] = sum_{k in slice} sum_p Re(CC[w][k][p] * exp(j w_k (tau_gi - tau_gj))).
__global__ __launch_bounds__(256) void srp_partial_kernel(const float* __restrict__ cc, int W, int nbins, int P,
                                                          const double* __restrict__ tau, int G, int M,
                                                          const double* __restrict__ omega,
                                                          const int32_t* __restrict__ pair_i,
                                                          const int32_t* __restrict__ pair_j, int k_per_slice,
                                                          float* __restrict__ part) {
  extern __shared__ double taus[];         // [M][256]
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int gg = g < G ? g : G - 1;
  for (int m = 0; m < M; ++m) taus[m * blockDim.x + threadIdx.x] = tau[(long)gg * M + m];
  const int k0 = blockIdx.y * k_per_slice;
  const int k1 = k0 + k_per_slice < nbins ? k0 + k_per_slice : nbins;
  float acc[MAP_W];
#pragma unroll
  for (int w = 0; w < MAP_W; ++w) acc[w] = 0.f;
  const double inv2pi = 0.15915494309189535, twopi = 6.283185307179586;
  for (int p = 0; p < P; ++p) {
    const double dt = taus[pair_i[p] * blockDim.x + threadIdx.x] - taus[pair_j[p] * blockDim.x + threadIdx.x];
    for (int k = k0; k < k1; ++k) {
      const double ph = omega[k] * dt;
      const float r = (float)(ph - twopi * rint(ph * inv2pi));
      float sn, cs;
      sincosf(r, &sn, &cs);
      const float* c = cc + ((long)k * P + p) * 2;
#pragma unroll
      for (int w = 0; w < MAP_W; ++w)
        if (w < W) {
          const float* cw = c + (long)w * nbins * P * 2;
          acc[w] += cw[0] * cs - cw[1] * sn;
        }
    }
  }
  if (g < G)
    for (int w = 0; w < W && w < MAP_W; ++w) part[((long)blockIdx.y * MAP_W + w) * G + g] = acc[w];
}

// out[g] = max(out_in[g], max_w sum_ks part / (nbins*P))
__global__ void srp_finish_kernel(const float* __restrict__ part, int KS, int W, int G, float scale, int first,
                                  float* __restrict__ out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  float best = first ? 0.f : out[g];       // the reference's map starts from zeros (SRP_Prunning.py:252)
  for (int w = 0; w < W; ++w) {
    float s = 0.f;
    for (int ks = 0; ks < KS; ++ks) s += part[((long)ks * MAP_W + w) * G + g];
    best = fmaxf(best, s * scale);
  }
  out[g] = best;
}

}  // namespace

extern "C" int asw_srp_frames(int window, int nfft, int hop) { return (window - nfft) / hop + 1; }

extern "C" int asw_srp_cross_spectra(const float* mix, int M, int T, int window, int step, int n_windows, int nfft,
                                     int hop, int nbins, int nb_pad, float tol, const float* twiddle,
                                     const int32_t* pair_i, const int32_t* pair_j, int P, float* xf_scratch,
                                     float* cc, void* stream) {
  ASW_CHECK_ARG(mix && twiddle && pair_i && pair_j && xf_scratch && cc, "srp_cross_spectra: null pointer");
  ASW_CHECK_ARG(M >= 2 && M <= 32 && window >= nfft && nfft % 32 == 0 && hop % 4 == 0 && step % 4 == 0 && T % 4 == 0,
                "srp_cross_spectra: bad shape (T, step and hop must be multiples of 4)");
  ASW_CHECK_ARG(nb_pad % 64 == 0 && nbins <= nb_pad && P == M * (M - 1) / 2, "srp_cross_spectra: bad bin/pair count");
  const int F = asw_srp_frames(window, nfft, hop);
  hipStream_t s = asw::as_stream(stream);
  const size_t smem = (size_t)F * M * 2 * sizeof(float);
  ASW_CHECK_ARG(smem <= 64 * 1024, "srp_cross_spectra: %d frames x %d mics exceed the LDS tile", F, M);
  for (int w = 0; w < n_windows; ++w) {
    const long start = (long)w * step;
    ASW_CHECK_ARG(start + window <= T, "srp_cross_spectra: window %d exceeds the signal", w);
    asw_convgemm_args a = {};
    a.A = mix + start; a.Wt = twiddle; a.out = xf_scratch;
    a.B = M; a.M_out = F; a.N = 2 * nb_pad; a.Cin = nfft; a.taps = 1; a.stride = 1; a.dil = 1; a.pad = 0;
    a.a_row_stride = hop; a.a_batch_stride = T; a.a_len = window;
    int rc = asw_convgemm_f32(&a, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(phat_cc_kernel, dim3(nbins), dim3(256), smem, s, xf_scratch, M, F, 2 * nb_pad, nb_pad, tol,
                       pair_i, pair_j, P, cc + (size_t)w * nbins * P * 2);
    ASW_LAUNCH_CHECK();
  }
  return ASW_OK;
}

extern "C" int asw_srp_map(const float* cc, int n_windows, int nbins, int P, const double* tau, int G, int M,
                           const double* omega, const int32_t* pair_i, const int32_t* pair_j, float* part_scratch,
                           float* out, void* stream) {
  ASW_CHECK_ARG(cc && tau && omega && pair_i && pair_j && part_scratch && out, "srp_map: null pointer");
  ASW_CHECK_ARG(G > 0 && M >= 2 && M <= 32 && nbins > 0 && P > 0 && n_windows > 0, "srp_map: bad shape");
  hipStream_t s = asw::as_stream(stream);
  const int KS = 8, kps = asw::cdiv(nbins, KS);
  const float scale = 1.0f / ((float)nbins * (float)P);
  const size_t smem = (size_t)M * 256 * sizeof(double);
  // the steered-response map: G x nbins x P sincos + complex MACs per window (VALU-bound); algorithmic bytes
  // = cross spectra + delays read, map written
  asw::ProfScope prof(s, "srp_map", 8.0 * (double)G * nbins * P * n_windows,
                      (double)n_windows * nbins * P * 8 + (double)G * M * 8 + (double)G * 4);
  for (int w0 = 0; w0 < n_windows; w0 += MAP_W) {
    const int W = n_windows - w0 < MAP_W ? n_windows - w0 : MAP_W;
    hipLaunchKernelGGL(srp_partial_kernel, dim3(asw::cdiv(G, 256), KS), dim3(256), smem, s,
                       cc + (size_t)w0 * nbins * P * 2, W, nbins, P, tau, G, M, omega, pair_i, pair_j, kps,
                       part_scratch);
    ASW_LAUNCH_CHECK();
    hipLaunchKernelGGL(srp_finish_kernel, dim3(asw::cdiv(G, 256)), dim3(256), 0, s, part_scratch, KS, W, G, scale,
                       w0 == 0 ? 1 : 0, out);
    ASW_LAUNCH_CHECK();
  }
  return ASW_OK;
}
