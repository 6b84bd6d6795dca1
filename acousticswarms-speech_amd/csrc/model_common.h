// model_common.h -- host-side pieces shared by the two device-resident networks
// (spot_model.hip: sep/training/SpeakerLocalization/network.py; sep_model.hip:
// sep/training/SpeakerSeparation/network.py): device buffers, GEMM weights in both arithmetic
// forms, the dilated-residual stack and the linear-layer launch helper.
#pragma once
#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "asw_common.h"

namespace asw_model {

struct DevBuf {
  float* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  ~DevBuf() { if (p) (void)hipFree(p); }
  int upload(const std::vector<float>& h) {
    if (p) { (void)hipFree(p); p = nullptr; }
    n = h.size();
    if (hipMalloc(&p, n * sizeof(float)) != hipSuccess) return asw::set_error(ASW_ERR_NOMEM, "hipMalloc(%zu floats)", n);
    ASW_HIP(hipMemcpy(p, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    return ASW_OK;
  }
};

// A GEMM weight in both arithmetic forms: fp32 (exact f32 MFMA) and the fp16 hi/lo split
// with its power-of-two pre-scale (f16x3 MFMA), see convgemm.hip.
struct WBuf {
  DevBuf f32;
  uint16_t* hi = nullptr;
  uint16_t* lo = nullptr;
  int32_t shift = 0;
  WBuf() = default;
  WBuf(const WBuf&) = delete;
  WBuf& operator=(const WBuf&) = delete;
  WBuf(WBuf&& o) noexcept : f32(std::move(o.f32)), hi(o.hi), lo(o.lo), shift(o.shift), fhi(o.fhi), flo(o.flo) {
    o.hi = o.lo = o.fhi = o.flo = nullptr;
  }
  ~WBuf() {
    if (hi) (void)hipFree(hi);
    if (lo) (void)hipFree(lo);
    if (fhi) (void)hipFree(fhi);
    if (flo) (void)hipFree(flo);
  }
  // GEMM weight Wt[N][K] in every form the kernels take: fp32, fp16 hi / lo row-major, and -- when the
  // shape allows (N % 32 == 0, K % 16 == 0) -- hi / lo in MFMA-fragment order for the kernels that
  // pull their B operand straight from global memory (residual layers, pipelined wide tiles)
  int upload_gemm(const std::vector<float>& h, int N, int K) {
    int rc = upload(h);
    if (rc) return rc;
    if (N % 32 == 0 && K % 16 == 0 && (size_t)N * K == h.size()) return upload_frags(h, N, K);
    return ASW_OK;
  }
  int upload(const std::vector<float>& h) {
    int rc = f32.upload(h);
    if (rc) return rc;
    std::vector<uint16_t> vh(h.size()), vl(h.size());
    if ((rc = asw_split_weights_f16(h.data(), h.size(), vh.data(), vl.data(), &shift))) return rc;
    if (hi) { (void)hipFree(hi); hi = nullptr; }
    if (lo) { (void)hipFree(lo); lo = nullptr; }
    if (hipMalloc(&hi, h.size() * 2) != hipSuccess || hipMalloc(&lo, h.size() * 2) != hipSuccess)
      return asw::set_error(ASW_ERR_NOMEM, "hipMalloc(%zu halves)", h.size());
    ASW_HIP(hipMemcpy(hi, vh.data(), h.size() * 2, hipMemcpyHostToDevice));
    ASW_HIP(hipMemcpy(lo, vl.data(), h.size() * 2, hipMemcpyHostToDevice));
    return ASW_OK;
  }
  // fragment-major copy for the halo-staged residual conv (Wt is [N][K])
  uint16_t* fhi = nullptr;
  uint16_t* flo = nullptr;
  int upload_frags(const std::vector<float>& h, int N, int K) {
    std::vector<uint16_t> vh(h.size()), vl(h.size());
    int32_t sh = 0;
    int rc = asw_pack_fragments_f16(h.data(), N, K, vh.data(), vl.data(), &sh);
    if (rc) return rc;
    if (sh != shift) return asw::set_error(ASW_ERR_STATE, "fragment pack: inconsistent weight shift");
    if (fhi) { (void)hipFree(fhi); fhi = nullptr; }
    if (flo) { (void)hipFree(flo); flo = nullptr; }
    if (hipMalloc(&fhi, h.size() * 2) != hipSuccess || hipMalloc(&flo, h.size() * 2) != hipSuccess)
      return asw::set_error(ASW_ERR_NOMEM, "hipMalloc(%zu halves)", h.size());
    ASW_HIP(hipMemcpy(fhi, vh.data(), h.size() * 2, hipMemcpyHostToDevice));
    ASW_HIP(hipMemcpy(flo, vl.data(), h.size() * 2, hipMemcpyHostToDevice));
    return ASW_OK;
  }
  void bind(asw_convgemm_args& a, int precision) const {
    a.Wt = f32.p; a.Wt_hi = hi; a.Wt_lo = lo; a.w_shift = shift; a.precision = precision;
    a.Wf_hi = fhi; a.Wf_lo = flo;
  }
};

struct ResLayer { WBuf wt; DevBuf bias, g, b; int dil = 1; };

// conv weight [N][Cin][K] -> Wt[N][tap*Cin + c] (optionally scaled per input channel)
inline std::vector<float> pack_conv(const std::vector<float>& w, int N, int Cin, int K, const float* in_gate) {
  std::vector<float> o((size_t)N * Cin * K);
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < Cin; ++c)
      for (int k = 0; k < K; ++k)
        o[((size_t)n * K + k) * Cin + c] = w[((size_t)n * Cin + c) * K + k] * (in_gate ? in_gate[c] : 1.f);
  return o;
}


// DilatedResidualSequence weights (network.py:70-82 of either network): conv weights in GEMM
// layout (+ MFMA-fragment order for the halo-staged f16x3 kernel), bias, LayerNorm affine.
inline int pack_res_layers(const std::map<std::string, std::vector<float>>& raw, const std::string& p, int ch, int K,
                           int n_layers, int dil_factor, std::vector<ResLayer>& out) {
  out.clear();
  out.resize(n_layers);
  int dil = 1;
  for (int j = 0; j < n_layers; ++j) {
    const std::string q = p + ".res.seq." + std::to_string(j);
    int rc;
    {
      const std::vector<float> packed = pack_conv(raw.at(q + ".conv.weight"), ch, ch, K, nullptr);
      if ((rc = out[j].wt.upload(packed))) return rc;
      if (ch % 32 == 0 && (ch * K) % 16 == 0 && (rc = out[j].wt.upload_frags(packed, ch, ch * K))) return rc;
    }
    if ((rc = out[j].bias.upload(raw.at(q + ".conv.bias")))) return rc;
    if ((rc = out[j].g.upload(raw.at(q + ".norm.weight")))) return rc;
    if ((rc = out[j].b.upload(raw.at(q + ".norm.bias")))) return rc;
    out[j].dil = dil;
    dil *= dil_factor;
  }
  return ASW_OK;
}

// ---- workspace arena --------------------------------------------------------
struct Arena {
  char* base;
  size_t off = 0, cap;
  bool dry;
  Arena(char* b, size_t c, bool d) : base(b), cap(c), dry(d) {}
  template <typename T>
  T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = reinterpret_cast<T*>(base + off);
    off += n * sizeof(T);
    return p;
  }
};


// Input of a residual stack given as the un-normalised output of the preceding (transposed) convolution:
// the first layer applies GroupNorm + GLU while it stages its rows (asw_convgemm_args.glu_raw).
struct GluSrc { const float* raw; const float* mr; const float* gamma; const float* beta; float* side_out = nullptr; };

// f16 arithmetic, 64..512 channels, first layer of dilation 1 with fragment-order weights: the layers that can do it.
// Above 64 channels the layer needs GluSrc.side_out (it reads its residual from there).  ASW_GLU_ON_LOAD_MAX_C=64
// keeps the wider blocks on the separate asw_gn_glu pass (A/B measurements).
inline bool glu_on_load_ok(const std::vector<ResLayer>& res, int prec, int ch) {
  static const int max_c = getenv("ASW_GLU_ON_LOAD_MAX_C") ? atoi(getenv("ASW_GLU_ON_LOAD_MAX_C")) : 512;
  return prec >= 1 && (ch == 64 || ch == 128 || ch == 256 || ch == 512) && ch <= max_c && !res.empty() &&
         res[0].dil == 1 && res[0].wt.fhi && res[0].wt.flo;
}

// 64-channel stacks in f16x3 arithmetic run through asw_resstack64_f16x3 (resstack.hip): consecutive layers whose
// later dilations leave most of a 256-row tile (summed halo <= 64 rows per side) share ONE launch, the rest
// (dilation 49) run as single layers of the same kernel.  ASW_NO_RESSTACK=1 keeps the per-layer kernels
// of convgemm.hip (A/B measurements).
inline bool resstack_ok(const std::vector<ResLayer>& res, int prec, int ch, int K) {
  static const bool off = getenv("ASW_NO_RESSTACK") != nullptr;
  if (off || prec < 1 || ch != 64 || K % 2 == 0 || K < 3 || K > 15) return false;
  for (const ResLayer& r : res)
    if (!r.wt.fhi || !r.wt.flo) return false;
  return true;
}

inline int run_res(const std::vector<ResLayer>& res, int prec, int B, int T, int ch, int K, float* x, float* p, float* q,
            float** final_out, hipStream_t s, const GluSrc* glu = nullptr) {
  // ping-pong: layer 0 reads x (kept intact), later layers alternate p/q
  const float* in = glu ? glu->raw : x;
  float* outb = p;
  if (resstack_ok(res, prec, ch, K)) {
    static const bool nofuse = getenv("ASW_RESSTACK_NOFUSE") != nullptr;
    size_t j = 0;
    while (j < res.size()) {
      size_t n = 1;
      int halo = 0;
      while (!nofuse && j + n < res.size() && n < 3) {
        const int pad = res[j + n].dil * (K - 1) / 2;
        if (halo + pad > 64) break;
        halo += pad;
        ++n;
      }
      asw_resstack_args a = {};
      a.x = (glu && j == 0) ? nullptr : in;
      a.out = outb;
      a.B = B; a.T = T; a.C = ch; a.taps = K; a.n_layers = (int)n; a.precision = prec; a.ln_eps = 1e-5f;
      for (size_t i = 0; i < n; ++i) {
        const ResLayer& r = res[j + i];
        a.layer[i].Wf_hi = r.wt.fhi; a.layer[i].Wf_lo = r.wt.flo; a.layer[i].w_shift = r.wt.shift;
        a.layer[i].bias = r.bias.p; a.layer[i].ln_gamma = r.g.p; a.layer[i].ln_beta = r.b.p; a.layer[i].dil = r.dil;
      }
      if (glu && j == 0) {
        a.glu_raw = glu->raw; a.glu_mr = glu->mr; a.glu_gamma = glu->gamma; a.glu_beta = glu->beta; a.glu_out = glu->side_out;
      }
      int rc = asw_resstack64_f16x3(&a, s);
      if (rc) return rc;
      in = outb;
      outb = (outb == p) ? q : p;
      j += n;
    }
    *final_out = const_cast<float*>(in);
    return ASW_OK;
  }
  for (size_t j = 0; j < res.size(); ++j) {
    asw_convgemm_args a = {};
    a.A = in; res[j].wt.bind(a, prec); a.bias = res[j].bias.p; a.resid = in;
    if (glu && j == 0) {
      a.glu_raw = glu->raw; a.glu_mr = glu->mr; a.glu_gamma = glu->gamma; a.glu_beta = glu->beta; a.glu_out = glu->side_out;
    }
    a.ln_gamma = res[j].g.p; a.ln_beta = res[j].b.p; a.out = outb;
    a.B = B; a.M_out = T; a.N = ch; a.Cin = ch; a.taps = K; a.stride = 1; a.dil = res[j].dil;
    a.pad = (res[j].dil * (K - 1) + 1) / 2;
    a.a_row_stride = ch; a.a_batch_stride = (int64_t)T * ch; a.a_len = (int64_t)T * ch;
    a.relu = 1; a.ln_eps = 1e-5f;
    int rc = asw_convgemm_f32(&a, s);
    if (rc) return rc;
    in = outb;
    outb = (outb == p) ? q : p;
  }
  *final_out = const_cast<float*>(in);
  return ASW_OK;
}

inline int linear(const float* A, const WBuf& W, int prec, const float* bias, int rows, int N, int K, int relu, const float* resid,
           const float* g, const float* b, float* out, hipStream_t s) {
  asw_convgemm_args a = {};
  a.A = A; W.bind(a, prec); a.bias = bias; a.resid = resid; a.ln_gamma = g; a.ln_beta = b; a.out = out;
  a.B = 1; a.M_out = rows; a.N = N; a.Cin = K; a.taps = 1; a.stride = 1; a.dil = 1; a.pad = 0;
  a.a_row_stride = K; a.a_batch_stride = (int64_t)rows * K; a.a_len = (int64_t)rows * K;
  a.relu = relu; a.ln_eps = 1e-5f;
  if (g && N >= 1024 && N % 256 == 0 && N <= 2048) {
    // a LayerNorm-fused tile must hold the whole row: at d = 1024 that leaves 32 rows per
    // workgroup and every workgroup re-reads all of W (measured 75 TFLOP/s).  Run the GEMM on
    // the wide tile instead and normalise in one extra pass over the rows (240 TFLOP/s + 30 us).
    a.resid = nullptr; a.ln_gamma = nullptr; a.ln_beta = nullptr;
    int rc = asw_convgemm_f32(&a, s);
    if (rc) return rc;
    return asw_add_layernorm(out, resid, g, b, rows, N, 1e-5f, out, s);
  }
  return asw_convgemm_f32(&a, s);
}


}  // namespace asw_model
