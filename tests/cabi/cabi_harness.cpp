// cabi_harness.cpp -- the drop-in boundary without Python or torch: a plain C++ program that
// links libasw_hip.so through include/asw_hip.h only, runs one DilatedResidualLayer-shaped
// convolution (network.py:57-68) on the GPU in both arithmetic modes, checks it against a
// double-precision host loop, exercises the error path and the host-side asw_search_area.
// Built and run by tests/test_gpu_cabi.py:  hipcc cabi_harness.cpp -I include -L <pkg> -lasw_hip
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "asw_hip.h"

#define HIPOK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("FAIL hip %s line %d\n", hipGetErrorString(e), __LINE__); return 2; } } while (0)
#define ASWOK(x) do { int rc = (x); if (rc != 0) { printf("FAIL asw rc=%d (%s) line %d\n", rc, asw_last_error(), __LINE__); return 3; } } while (0)

static unsigned g_seed = 12345u;
static float rnd() { g_seed = g_seed * 1664525u + 1013904223u; return ((g_seed >> 8) & 0xFFFF) / 32768.0f - 1.0f; }

int main() {
  printf("abi version %d\n", asw_abi_version());
  // ---- error path: null arguments come back as a status + message, nothing is thrown
  if (asw_convgemm_f32(nullptr, nullptr) == 0 || std::strlen(asw_last_error()) == 0) { printf("FAIL error path\n"); return 1; }

  const int B = 2, T = 300, C = 64, K = 7, dil = 7;
  std::vector<float> x((size_t)B * T * C), w((size_t)C * K * C), bias(C), gam(C), bet(C);
  for (auto& v : x) v = rnd();
  for (auto& v : w) v = rnd() * 0.05f;                      // Wt[n][tap*C + c]
  for (int i = 0; i < C; ++i) { bias[i] = rnd() * 0.1f; gam[i] = 1.0f + rnd() * 0.1f; bet[i] = rnd() * 0.1f; }
  // host reference in double: LN(ReLU(conv_d(x) + b) + x)
  std::vector<double> want((size_t)B * T * C);
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < T; ++t) {
      double row[64];
      for (int n = 0; n < C; ++n) {
        double acc = bias[n];
        for (int k = 0; k < K; ++k) {
          const int ti = t + (k - K / 2) * dil;
          if (ti < 0 || ti >= T) continue;
          for (int c = 0; c < C; ++c) acc += (double)w[(size_t)n * K * C + k * C + c] * x[((size_t)b * T + ti) * C + c];
        }
        row[n] = (acc > 0 ? acc : 0) + x[((size_t)b * T + t) * C + n];
      }
      double mu = 0, var = 0;
      for (int n = 0; n < C; ++n) mu += row[n];
      mu /= C;
      for (int n = 0; n < C; ++n) var += (row[n] - mu) * (row[n] - mu);
      var /= C;
      for (int n = 0; n < C; ++n) want[((size_t)b * T + t) * C + n] = (row[n] - mu) / std::sqrt(var + 1e-5) * gam[n] + bet[n];
    }

  float *dx, *dw, *db, *dg, *dbe, *dout;
  HIPOK(hipMalloc((void**)&dx, x.size() * 4)); HIPOK(hipMalloc((void**)&dw, w.size() * 4));
  HIPOK(hipMalloc((void**)&db, C * 4)); HIPOK(hipMalloc((void**)&dg, C * 4)); HIPOK(hipMalloc((void**)&dbe, C * 4));
  HIPOK(hipMalloc((void**)&dout, x.size() * 4));
  HIPOK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
  HIPOK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
  HIPOK(hipMemcpy(db, bias.data(), C * 4, hipMemcpyHostToDevice));
  HIPOK(hipMemcpy(dg, gam.data(), C * 4, hipMemcpyHostToDevice));
  HIPOK(hipMemcpy(dbe, bet.data(), C * 4, hipMemcpyHostToDevice));
  // f16x3 operands: split + fragment order on the host, as asw_spot_finalize does
  std::vector<uint16_t> hi(w.size()), lo(w.size()), fhi(w.size()), flo(w.size());
  int32_t shift = 0, fshift = 0;
  ASWOK(asw_split_weights_f16(w.data(), w.size(), hi.data(), lo.data(), &shift));
  ASWOK(asw_pack_fragments_f16(w.data(), C, K * C, fhi.data(), flo.data(), &fshift));
  if (shift != fshift) { printf("FAIL shift mismatch\n"); return 1; }
  void *dhi, *dlo, *dfhi, *dflo;
  HIPOK(hipMalloc(&dhi, w.size() * 2)); HIPOK(hipMalloc(&dlo, w.size() * 2));
  HIPOK(hipMalloc(&dfhi, w.size() * 2)); HIPOK(hipMalloc(&dflo, w.size() * 2));
  HIPOK(hipMemcpy(dhi, hi.data(), w.size() * 2, hipMemcpyHostToDevice));
  HIPOK(hipMemcpy(dlo, lo.data(), w.size() * 2, hipMemcpyHostToDevice));
  HIPOK(hipMemcpy(dfhi, fhi.data(), w.size() * 2, hipMemcpyHostToDevice));
  HIPOK(hipMemcpy(dflo, flo.data(), w.size() * 2, hipMemcpyHostToDevice));

  std::vector<float> got(x.size());
  for (int prec = 0; prec < 2; ++prec) {
    asw_convgemm_args a;
    std::memset(&a, 0, sizeof a);
    a.A = dx; a.Wt = dw; a.bias = db; a.resid = dx; a.ln_gamma = dg; a.ln_beta = dbe; a.out = dout;
    a.B = B; a.M_out = T; a.N = C; a.Cin = C; a.taps = K; a.stride = 1; a.dil = dil; a.pad = (K / 2) * dil;
    a.a_row_stride = C; a.a_batch_stride = (int64_t)T * C; a.a_len = (int64_t)T * C;
    a.relu = 1; a.ln_eps = 1e-5f; a.precision = prec; a.w_shift = shift;
    a.Wt_hi = dhi; a.Wt_lo = dlo; a.Wf_hi = dfhi; a.Wf_lo = dflo;
    HIPOK(hipMemset(dout, 0, x.size() * 4));
    ASWOK(asw_convgemm_f32(&a, nullptr));
    HIPOK(hipDeviceSynchronize());
    HIPOK(hipMemcpy(got.data(), dout, x.size() * 4, hipMemcpyDeviceToHost));
    double num = 0, den = 0;
    for (size_t i = 0; i < got.size(); ++i) { num += (got[i] - want[i]) * (got[i] - want[i]); den += want[i] * want[i]; }
    const double rel = std::sqrt(num / den);
    printf("residual layer, precision %d: relative L2 error %.3e\n", prec, rel);
    if (!(rel < (prec == 0 ? 2e-6 : 2e-5))) { printf("FAIL accuracy\n"); return 1; }
  }

  // ---- host-side entry point: one width-8 hypercube around a point, subdivided
  {
    const int M = 4, n = 600;
    const double mic[12] = {0, 0, 0.02, 0.3, 0.1, 0.02, -0.3, 0.1, 0.02, 0.0, 0.4, 0.02};
    std::vector<double> pts(3 * n);
    for (int j = 0; j < n; ++j) { pts[j] = 1.0 + 0.2 * rnd(); pts[n + j] = 1.5 + 0.2 * rnd(); pts[2 * n + j] = 0.3 + 0.1 * rnd(); }
    double off[3], wid[3] = {8, 8, 8};
    for (int i = 0; i < 3; ++i) {
      const double* m = mic + 3 * (i + 1);
      const double d0 = std::sqrt(1.0 + 2.25 + (0.3 - 0.02) * (0.3 - 0.02)), di = std::sqrt((1.0 - m[0]) * (1.0 - m[0]) + (1.5 - m[1]) * (1.5 - m[1]) + 0.28 * 0.28);
      off[i] = std::round((di - d0) / 343.0 * 48000);
    }
    int nc = 0; double *co = nullptr, *cw = nullptr; int *cc = nullptr, *ci = nullptr;
    ASWOK(asw_search_area(pts.data(), n, mic, M, off, wid, nullptr, 343.0, 48000.0, &nc, &co, &cw, &cc, &ci));
    long total = 0;
    for (int k = 0; k < nc; ++k) { total += cc[k]; for (int i = 0; i < 3; ++i) if (cw[k * 3 + i] > 4.0) { printf("FAIL child width\n"); return 1; } }
    printf("search_area: %d children, %ld member points of %d\n", nc, total, n);
    if (nc < 1) { printf("FAIL search_area\n"); return 1; }
    asw_free(co); asw_free(cw); asw_free(cc); asw_free(ci);
  }
  printf("CABI OK\n");
  return 0;
}
