// sep_model.hip -- device-resident joint separation network ("separation by localization"):
// weight packing + the layer schedule of Network.forward / infer_sample
// (sep/training/SpeakerSeparation/network.py:418-548) as launches of the kernels in
// prep_kernels.hip / convgemm.hip / misc_kernels.hip / sep_kernels.hip on one HIP stream.
//
// One call separates the S talkers the search found: every speaker s is one "sequence" --
// the mixture aligned to that speaker (zero-filled integer shift), normalised with statistics
// shared by all S*M channels -- so the U-Net encoder / decoder and the mask path run with
// batch S on the same channels-last [S][T_l][C] layout and the same MFMA kernels as the spot
// network (kernel size 5, strides 2,2,4,4, dilations 1,2,4, latent 4096).  The bottleneck
// ([S][L][d], L = T/64, d = 512) alternates a Conformer layer along time per speaker with a
// transformer layer across the S speakers of each time step (:270-321).
//
// The Conformer follows the published speechbrain definitions (the library is absent from the
// image; oracle/sep_ref.py restates it and states what is pinned).  Weight transformations done
// once at finalize: speechbrain's per-head (q,k,v) interleave of in_proj_weight is permuted to
// Q | K | V; the macaron factor 1/2 is folded into the second feed-forward linear; the
// depthwise kernel is stored tap-major.
#include <cmath>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "model_common.h"

using namespace asw_model;

extern "C" int asw_joint_shift_stats_scratch_doubles(void);

namespace {

struct EncBlock { std::vector<ResLayer> res; WBuf down_wt; DevBuf bias, gn_g, gn_b; };
struct DecBlock { std::vector<ResLayer> res; WBuf up_wt; DevBuf up_bias, gn_g, gn_b; };
struct Ffn { DevBuf lng, lnb, b1, b2; WBuf w1, w2; };                       // w2 / b2 carry the macaron 1/2
struct ConfLayer {
  Ffn f1, f2;
  DevBuf n1g, n1b, n2g, n2b, fng, fnb;                                     // norm1, norm2, encoder-final norm
  WBuf w_in, w_pos, w_out;
  DevBuf b_out, bu, bv;
  DevBuf cm_lng, cm_lnb, cm_pwb, dw_wT, dw_b, ac_lng, ac_lnb, ac_b;
  WBuf cm_pw, ac_w;
};
struct InterLayer { WBuf w_in, w_out, w1, w2; DevBuf b_in, b_out, b1, b2, n1g, n1b, n2g, n2b; };
struct Tap { const float* p; size_t numel; };

}  // namespace

struct asw_sep {
  asw_sep_config cfg;
  std::map<std::string, std::vector<float>> raw;
  bool finalized = false;
  int device = 0;
  int precision = 0;

  std::vector<int> enc_cin, enc_cout, dec_cin, dec_cout, dec_stride;
  int stride_product = 1;

  DevBuf pre_w, pre_b;
  std::vector<EncBlock> enc;
  std::vector<DecBlock> dec;
  std::vector<ConfLayer> conf;
  std::vector<InterLayer> inter;
  WBuf byp_wt, mask_wt, dec_wt;
  WBuf byp_wt48;                           // bypass kernel padded to 48 taps, fragment order (fused mask path)
  DevBuf byp_b, mask_b;
  float out_bias = 0.f;
  int byp_k = 0;
  std::vector<float> inv_freq;
  DevBuf pe;                                // sinusoid table [2L-1][d] of the last sequence length
  int pe_L = 0;

  char* ws = nullptr;
  size_t ws_bytes = 0;
  std::map<std::string, Tap> taps;

  ~asw_sep() { if (ws) (void)hipFree(ws); }
};

namespace {

const std::vector<float>& P(const asw_sep* m, const std::string& k) { return m->raw.at(k); }

std::vector<std::pair<std::string, size_t>> expected_params(const asw_sep* m) {
  const asw_sep_config& c = m->cfg;
  std::vector<std::pair<std::string, size_t>> v;
  const size_t K = c.kernel_size;
  v.push_back({"preproc.weight", (size_t)c.channels * c.n_mics});
  v.push_back({"preproc.bias", (size_t)c.channels});
  auto res = [&](const std::string& p, size_t ch) {
    for (int j = 0; j < c.residual_layers; ++j) {
      const std::string q = p + ".res.seq." + std::to_string(j);
      v.push_back({q + ".conv.weight", ch * ch * K});
      v.push_back({q + ".conv.bias", ch});
      v.push_back({q + ".norm.weight", ch});
      v.push_back({q + ".norm.bias", ch});
    }
  };
  for (int i = 0; i < c.depth; ++i) {
    const std::string p = "encoder.module_list." + std::to_string(i);
    const size_t ci = m->enc_cin[i], co = m->enc_cout[i];
    res(p, ci);
    v.push_back({p + ".conv1.weight", 2 * co * ci * K});
    v.push_back({p + ".conv1.bias", 2 * co});
    v.push_back({p + ".norm1.weight", 2 * co});
    v.push_back({p + ".norm1.bias", 2 * co});
  }
  const size_t d = m->enc_cout.back(), f = c.ffw_dim, H = c.num_head, BK = c.bottleneck_ksize;
  v.push_back({"bottleneck.pe_single.inv_freq", d / 2});
  for (int l = 0; l < c.bottleneck_layers; ++l) {
    const std::string cl = "bottleneck.module_list." + std::to_string(l) + ".intra.layers.0";
    v.push_back({cl + ".mha_layer.in_proj_weight", 3 * d * d});
    v.push_back({cl + ".mha_layer.pos_bias_u", d});
    v.push_back({cl + ".mha_layer.pos_bias_v", d});
    v.push_back({cl + ".mha_layer.out_proj.weight", d * d});
    v.push_back({cl + ".mha_layer.out_proj.bias", d});
    v.push_back({cl + ".mha_layer.linear_pos.weight", d * d});
    v.push_back({cl + ".convolution_module.layer_norm.weight", d});
    v.push_back({cl + ".convolution_module.layer_norm.bias", d});
    v.push_back({cl + ".convolution_module.bottleneck.0.weight", 2 * d * d});
    v.push_back({cl + ".convolution_module.bottleneck.0.bias", 2 * d});
    v.push_back({cl + ".convolution_module.conv.weight", d * BK});
    v.push_back({cl + ".convolution_module.conv.bias", d});
    v.push_back({cl + ".convolution_module.after_conv.0.weight", d});
    v.push_back({cl + ".convolution_module.after_conv.0.bias", d});
    v.push_back({cl + ".convolution_module.after_conv.2.weight", d * d});
    v.push_back({cl + ".convolution_module.after_conv.2.bias", d});
    for (const char* fm : {".ffn_module1", ".ffn_module2"}) {
      v.push_back({cl + fm + ".0.weight", d});
      v.push_back({cl + fm + ".0.bias", d});
      v.push_back({cl + fm + ".1.ffn.0.weight", f * d});
      v.push_back({cl + fm + ".1.ffn.0.bias", f});
      v.push_back({cl + fm + ".1.ffn.3.weight", d * f});
      v.push_back({cl + fm + ".1.ffn.3.bias", d});
    }
    v.push_back({cl + ".norm1.norm.weight", d});
    v.push_back({cl + ".norm1.norm.bias", d});
    v.push_back({cl + ".norm2.norm.weight", d});
    v.push_back({cl + ".norm2.norm.bias", d});
    const std::string ce = "bottleneck.module_list." + std::to_string(l) + ".intra.norm.norm";
    v.push_back({ce + ".weight", d});
    v.push_back({ce + ".bias", d});
    const std::string t = "bottleneck.module_list." + std::to_string(l) + ".inter.layers.0";
    v.push_back({t + ".self_attn.in_proj_weight", 3 * d * d});
    v.push_back({t + ".self_attn.in_proj_bias", 3 * d});
    v.push_back({t + ".self_attn.out_proj.weight", d * d});
    v.push_back({t + ".self_attn.out_proj.bias", d});
    v.push_back({t + ".linear1.weight", f * d});
    v.push_back({t + ".linear1.bias", f});
    v.push_back({t + ".linear2.weight", d * f});
    v.push_back({t + ".linear2.bias", d});
    v.push_back({t + ".norm1.weight", d});
    v.push_back({t + ".norm1.bias", d});
    v.push_back({t + ".norm2.weight", d});
    v.push_back({t + ".norm2.bias", d});
  }
  (void)H;
  for (int i = 0; i < c.depth; ++i) {
    const std::string p = "decoder.module_list." + std::to_string(i);
    const size_t ci = m->dec_cin[i], co = m->dec_cout[i], s = m->dec_stride[i];
    v.push_back({p + ".upsample.conv.weight", ci * 2 * co * s});
    v.push_back({p + ".upsample.conv.bias", 2 * co});
    v.push_back({p + ".norm1.weight", 2 * co});
    v.push_back({p + ".norm1.bias", 2 * co});
    res(p, co);
  }
  const size_t E = c.encoder_channels, EK = c.encoder_kernel_size;
  v.push_back({"reference_bypass.weight", E * EK});
  v.push_back({"reference_bypass.bias", E});
  v.push_back({"mask_encoder.weight", E * c.channels * EK});
  v.push_back({"mask_encoder.bias", E});
  v.push_back({"output_decoder.weight", E * EK});
  v.push_back({"output_decoder.bias", 1});
  return v;
}

std::vector<float> scaled(const std::vector<float>& w, float a) {
  std::vector<float> o(w.size());
  for (size_t i = 0; i < w.size(); ++i) o[i] = w[i] * a;
  return o;
}

// ---- workspace ---------------------------------------------------------------------------
struct Plan {
  int NB, S, BS, T, Tp, F, RL, depth, L, d;
  std::vector<int> Tl;
  float *mean, *stdv, *refn;
  double* jscr;
  std::vector<float*> X, Pb, Qb, raw_dn, raw_up, st_dn, st_up, mr_up, mr_dn;
  std::vector<float*> intra_out, inter_out;   // per bottleneck layer (kept apart so every tap stays readable)
  float *h, *g, *x1, *x2, *x3, *qkv, *ctx, *raw2, *u, *v2, *y, *f, *pos;
  float *Y, *D, *ywave;
  const int32_t* counts = nullptr;            // host [NB]: speakers present per item (NULL: S everywhere), forward() only
};

// The one-launch mask path (asw_mask_path_f16x3) applies in f16x3 mode when the shapes fit its tiles.
bool fused_mask_path(const asw_sep* m) {
  const asw_sep_config& c = m->cfg;
  return m->precision >= 1 && c.encoder_channels % 256 == 0 && c.channels % 32 == 0 && c.encoder_kernel_size <= 48 &&
         c.encoder_stride % 4 == 0 && m->byp_wt48.fhi && m->dec_wt.fhi && m->mask_wt.fhi;
}

void layout(const asw_sep* m, int NB, int S, int T, Arena& a, Plan& pl) {
  const asw_sep_config& c = m->cfg;
  const int BS = NB * S;
  pl.NB = NB; pl.S = S; pl.BS = BS; pl.T = T; pl.depth = c.depth;
  pl.Tp = ((T - 1) / m->stride_product + 1) * m->stride_product;
  const int EK = c.encoder_kernel_size, ES = c.encoder_stride;
  pl.F = (pl.Tp + 2 * (EK / 2) - EK) / ES + 1;
  pl.RL = ((EK / 2 + pl.Tp + m->byp_k + 64) + 3) & ~3;
  pl.Tl.assign(c.depth + 1, pl.Tp);
  for (int i = 0; i < c.depth; ++i) pl.Tl[i + 1] = pl.Tl[i] / c.stride_list[i];
  pl.mean = a.take<float>(BS);
  pl.stdv = a.take<float>(BS);
  pl.jscr = a.take<double>(asw_joint_shift_stats_scratch_doubles());
  pl.refn = a.take<float>((size_t)BS * pl.RL);
  pl.X.resize(c.depth + 1); pl.Pb.resize(c.depth); pl.Qb.resize(c.depth);
  pl.raw_dn.resize(c.depth); pl.raw_up.resize(c.depth); pl.st_dn.resize(c.depth); pl.st_up.resize(c.depth);
  pl.mr_up.resize(c.depth); pl.mr_dn.resize(c.depth);
  for (int i = 0; i <= c.depth; ++i) {
    const int ch = i == 0 ? c.channels : m->enc_cout[i - 1];
    pl.X[i] = a.take<float>((size_t)BS * pl.Tl[i] * ch);
  }
  for (int i = 0; i < c.depth; ++i) {
    const size_t n = (size_t)BS * pl.Tl[i] * m->enc_cin[i];
    pl.Pb[i] = a.take<float>(n);
    pl.Qb[i] = a.take<float>(n);
    pl.raw_dn[i] = a.take<float>((size_t)BS * pl.Tl[i + 1] * 2 * m->enc_cout[i]);
    pl.st_dn[i] = a.take<float>((size_t)BS * 4 * asw_convgemm_stats_tiles(pl.Tl[i + 1], 2 * m->enc_cout[i]));
    pl.mr_dn[i] = a.take<float>((size_t)BS * 4);
  }
  for (int j = 0; j < c.depth; ++j) {
    const int lvl = c.depth - j;
    const int s = m->dec_stride[j], co2 = 2 * m->dec_cout[j];
    pl.raw_up[j] = a.take<float>((size_t)BS * pl.Tl[lvl] * s * co2);
    pl.st_up[j] = a.take<float>((size_t)BS * 4 * asw_convgemm_stats_tiles(pl.Tl[lvl], s * co2));
    pl.mr_up[j] = a.take<float>((size_t)BS * 4);
  }
  const size_t L = pl.Tl[c.depth], d = m->enc_cout.back(), rows = (size_t)BS * L;
  pl.L = (int)L; pl.d = (int)d;
  pl.intra_out.resize(c.bottleneck_layers); pl.inter_out.resize(c.bottleneck_layers);
  for (int l = 0; l < c.bottleneck_layers; ++l) { pl.intra_out[l] = a.take<float>(rows * d); pl.inter_out[l] = a.take<float>(rows * d); }
  float** dbufs[] = {&pl.h, &pl.g, &pl.x1, &pl.x2, &pl.x3, &pl.ctx, &pl.u, &pl.v2, &pl.y};
  for (float** b : dbufs) *b = a.take<float>(rows * d);
  pl.qkv = a.take<float>(rows * 3 * d);
  pl.raw2 = a.take<float>(rows * 2 * d);
  pl.f = a.take<float>(rows * c.ffw_dim);
  pl.pos = a.take<float>((2 * L - 1) * d);
  const bool fused = fused_mask_path(m);
  pl.Y = fused ? nullptr : a.take<float>((size_t)BS * pl.F * c.encoder_channels);
  pl.D = a.take<float>((size_t)(fused ? c.encoder_channels / 256 : 1) * BS * pl.F * 64);
  pl.ywave = a.take<float>((size_t)BS * T);
}

int ensure_ws(asw_sep* m, int NB, int S, int T, Plan& pl) {
  Arena dry(nullptr, 0, true);
  layout(m, NB, S, T, dry, pl);
  const size_t need = dry.off + 4096;
  if (need > m->ws_bytes) {
    // the number of talkers changes from mixture to mixture: grow by half again, not to the exact size, so that a
    // stream of mixtures re-allocates (device synchronisation + hipFree / hipMalloc of GBs) a few times, not every
    // time one more talker than ever before turns up
    size_t want = need + need / 2;
    if (m->ws) { ASW_HIP(hipDeviceSynchronize()); (void)hipFree(m->ws); m->ws = nullptr; m->ws_bytes = 0; }
    if (hipMalloc(&m->ws, want) != hipSuccess) {
      (void)hipGetLastError();
      want = need;
      if (hipMalloc(&m->ws, want) != hipSuccess)
        return asw::set_error(ASW_ERR_NOMEM, "workspace of %.1f MiB for %d sequences, T=%d", need / 1048576.0, NB * S, T);
    }
    m->ws_bytes = want;
  }
  Arena real(m->ws, m->ws_bytes, false);
  layout(m, NB, S, T, real, pl);
  return ASW_OK;
}

// RelPosEncXL table for sequence length L: row r stands for relative position (L-1) - r; even
// columns sin(|pos| * inv_freq), odd columns cos (the published table uses the same sinusoid for
// past and future).  float32 arithmetic like the torch module.  Cached per L.
int ensure_pos_table(asw_sep* m, int L) {
  if (m->pe_L == L && m->pe.p) return ASW_OK;
  const int d = m->enc_cout.back();
  std::vector<float> pe((size_t)(2 * L - 1) * d);
  for (int r = 0; r < 2 * L - 1; ++r) {
    const float pos = (float)std::abs(r - (L - 1));
    for (int k = 0; k < d / 2; ++k) {
      const float ang = pos * m->inv_freq[k];
      pe[(size_t)r * d + 2 * k] = sinf(ang);
      pe[(size_t)r * d + 2 * k + 1] = cosf(ang);
    }
  }
  ASW_HIP(hipDeviceSynchronize());            // launches that read the previous table have finished
  int rc = m->pe.upload(pe);
  if (rc) return rc;
  m->pe_L = L;
  return ASW_OK;
}

int ffn_first(const Ffn& f, int prec, const float* hin, int rows, int d, int ffw, float* fbuf, hipStream_t s) {
  return linear(hin, f.w1, prec, f.b1.p, rows, ffw, d, /*swish*/ 2, nullptr, nullptr, nullptr, fbuf, s);
}

// one Conformer layer along time for the BS sequences: x -> out (both [BS*L][d])
int run_conformer(asw_sep* m, Plan& pl, ConfLayer& c, const float* x, float* out, hipStream_t s) {
  const asw_sep_config& cfg = m->cfg;
  const int rows = pl.BS * pl.L, d = pl.d, ffw = cfg.ffw_dim, prec = m->precision;
  int rc;
  // x1 = x + FFN1(x)/2
  if ((rc = asw_add_layernorm2(x, nullptr, 0.f, c.f1.lng.p, c.f1.lnb.p, rows, d, 1e-5f, 0, nullptr, pl.h, s))) return rc;
  if ((rc = ffn_first(c.f1, prec, pl.h, rows, d, ffw, pl.f, s))) return rc;
  if ((rc = linear(pl.f, c.f1.w2, prec, c.f1.b2.p, rows, d, ffw, 0, nullptr, nullptr, nullptr, pl.g, s))) return rc;
  if ((rc = asw_add_layernorm2(x, pl.g, 1.f, c.n1g.p, c.n1b.p, rows, d, 1e-5f, 0, pl.x1, pl.h, s))) return rc;   // h = norm1(x1)
  // x2 = x1 + MHA(norm1(x1))
  if ((rc = linear(pl.h, c.w_in, prec, nullptr, rows, 3 * d, d, 0, nullptr, nullptr, nullptr, pl.qkv, s))) return rc;
  if ((rc = linear(m->pe.p, c.w_pos, prec, nullptr, 2 * pl.L - 1, d, d, 0, nullptr, nullptr, nullptr, pl.pos, s))) return rc;
  if ((rc = asw_relpos_attention(pl.qkv, pl.pos, c.bu.p, c.bv.p, pl.BS, pl.L, d, cfg.num_head, 1.0f / sqrtf((float)d), pl.ctx, s)))
    return rc;
  if ((rc = linear(pl.ctx, c.w_out, prec, c.b_out.p, rows, d, d, 0, nullptr, nullptr, nullptr, pl.g, s))) return rc;
  if ((rc = asw_add_layernorm2(pl.x1, pl.g, 1.f, c.cm_lng.p, c.cm_lnb.p, rows, d, 1e-5f, 0, pl.x2, pl.h, s))) return rc;
  // x3 = x2 + ConvolutionModule(x2)
  if ((rc = linear(pl.h, c.cm_pw, prec, c.cm_pwb.p, rows, 2 * d, d, 0, nullptr, nullptr, nullptr, pl.raw2, s))) return rc;
  if ((rc = asw_glu_rows(pl.raw2, rows, d, pl.u, s))) return rc;
  if ((rc = asw_dwconv_ln_swish(pl.u, c.dw_wT.p, c.dw_b.p, c.ac_lng.p, c.ac_lnb.p, pl.BS, pl.L, d, cfg.bottleneck_ksize, 1e-5f,
                                pl.v2, s)))
    return rc;
  if ((rc = linear(pl.v2, c.ac_w, prec, c.ac_b.p, rows, d, d, 0, nullptr, nullptr, nullptr, pl.g, s))) return rc;
  if ((rc = asw_add_layernorm2(pl.x2, pl.g, 1.f, c.f2.lng.p, c.f2.lnb.p, rows, d, 1e-5f, 0, pl.x3, pl.h, s))) return rc;
  // y = norm2(x3 + FFN2(x3)/2); out = final norm (eps 1e-6)
  if ((rc = ffn_first(c.f2, prec, pl.h, rows, d, ffw, pl.f, s))) return rc;
  if ((rc = linear(pl.f, c.f2.w2, prec, c.f2.b2.p, rows, d, ffw, 0, pl.x3, c.n2g.p, c.n2b.p, pl.y, s))) return rc;
  return asw_add_layernorm2(pl.y, nullptr, 0.f, c.fng.p, c.fnb.p, rows, d, 1e-6f, 0, nullptr, out, s);
}

// post-norm transformer layer across the S speakers of every time step: x -> out
int run_inter(asw_sep* m, Plan& pl, InterLayer& t, const float* x, float* out, hipStream_t s) {
  const int rows = pl.BS * pl.L, d = pl.d, ffw = m->cfg.ffw_dim, prec = m->precision;
  int rc;
  if ((rc = linear(x, t.w_in, prec, t.b_in.p, rows, 3 * d, d, 0, nullptr, nullptr, nullptr, pl.qkv, s))) return rc;
  if ((rc = asw_inter_attention(pl.qkv, pl.NB, pl.S, pl.L, d, m->cfg.num_head, pl.ctx, s))) return rc;
  if ((rc = linear(pl.ctx, t.w_out, prec, t.b_out.p, rows, d, d, 0, x, t.n1g.p, t.n1b.p, pl.x1, s))) return rc;
  if ((rc = linear(pl.x1, t.w1, prec, t.b1.p, rows, ffw, d, 1, nullptr, nullptr, nullptr, pl.f, s))) return rc;
  return linear(pl.f, t.w2, prec, t.b2.p, rows, d, ffw, 0, pl.x1, t.n2g.p, t.n2b.p, out, s);
}

// everything after the preproc stage; pl.X[0] / pl.refn are filled
int run_network(asw_sep* m, Plan& pl, const float* mean, const float* stdv, float* out_wave, hipStream_t s) {
  const asw_sep_config& c = m->cfg;
  const int B = pl.BS, K = c.kernel_size;
  m->taps.clear();
  int rc;
  // ---- encoder (:84-156)
  GluSrc enc_src = {};
  bool enc_glu = false;                       // as in spot_model.hip: block i normalises raw_dn[i-1] while it stages
  for (int i = 0; i < c.depth; ++i) {
    float* r = nullptr;
    if ((rc = run_res(m->enc[i].res, m->precision, B, pl.Tl[i], m->enc_cin[i], K, pl.X[i], pl.Pb[i], pl.Qb[i], &r, s,
                      enc_glu ? &enc_src : nullptr)))
      return rc;
    asw_convgemm_args a = {};
    a.A = r; m->enc[i].down_wt.bind(a, m->precision); a.bias = m->enc[i].bias.p; a.out = pl.raw_dn[i]; a.stats = pl.st_dn[i];
    a.B = B; a.M_out = pl.Tl[i + 1]; a.N = 2 * m->enc_cout[i]; a.Cin = m->enc_cin[i]; a.taps = K;
    a.stride = c.stride_list[i]; a.dil = 1; a.pad = K / 2;
    a.a_row_stride = a.Cin; a.a_batch_stride = (int64_t)pl.Tl[i] * a.Cin; a.a_len = a.a_batch_stride;
    a.chan_mod = a.N;
    if ((rc = asw_convgemm_f32(&a, s))) return rc;
    enc_glu = i + 1 < c.depth && glu_on_load_ok(m->enc[i + 1].res, m->precision, m->enc_cout[i]);
    if (enc_glu) {
      if ((rc = asw_gn_finalize(pl.st_dn[i], asw_convgemm_stats_tiles(a.M_out, a.N), B, pl.Tl[i + 1], m->enc_cout[i], 1e-5f,
                                pl.mr_dn[i], s)))
        return rc;
      enc_src = {pl.raw_dn[i], pl.mr_dn[i], m->enc[i].gn_g.p, m->enc[i].gn_b.p, pl.X[i + 1]};
    } else if ((rc = asw_gn_glu(pl.raw_dn[i], pl.st_dn[i], asw_convgemm_stats_tiles(a.M_out, a.N), m->enc[i].gn_g.p,
                                m->enc[i].gn_b.p, B, pl.Tl[i + 1], m->enc_cout[i], 1e-5f, pl.X[i + 1], s))) {
      return rc;
    }
    m->taps["enc" + std::to_string(i)] = {pl.X[i + 1], (size_t)B * pl.Tl[i + 1] * m->enc_cout[i]};
  }
  // ---- bottleneck (:296-321): [BS][L][d] is already the (B*S, T, F) layout of the Conformer and,
  // row for row, the (N*T, S, F) layout of the inter-speaker layer
  if ((rc = ensure_pos_table(m, pl.L))) return rc;
  const float* x = pl.X[c.depth];
  for (int l = 0; l < c.bottleneck_layers; ++l) {
    if ((rc = run_conformer(m, pl, m->conf[l], x, pl.intra_out[l], s))) return rc;
    m->taps["intra" + std::to_string(l)] = {pl.intra_out[l], (size_t)B * pl.L * pl.d};
    if (pl.counts) {
      // items with fewer speakers: batches_to_speakers (:250-268) re-inserts the missing ones as ZERO sequences, which
      // take part in the inter-speaker attention below; whatever the Conformer made of those rows is discarded
      for (int n = 0; n < pl.NB; ++n)
        if (pl.counts[n] < pl.S)
          ASW_HIP(hipMemsetAsync(pl.intra_out[l] + ((size_t)n * pl.S + pl.counts[n]) * pl.L * pl.d, 0,
                                 (size_t)(pl.S - pl.counts[n]) * pl.L * pl.d * sizeof(float), s));
    }
    if ((rc = run_inter(m, pl, m->inter[l], pl.intra_out[l], pl.inter_out[l], s))) return rc;
    m->taps["inter" + std::to_string(l)] = {pl.inter_out[l], (size_t)B * pl.L * pl.d};
    x = pl.inter_out[l];
  }
  m->taps["bottleneck"] = {x, (size_t)B * pl.L * pl.d};
  // ---- decoder (:204-238)
  for (int j = 0; j < c.depth; ++j) {
    const int lvl = c.depth - j, ci = m->dec_cin[j], co = m->dec_cout[j], st = m->dec_stride[j];
    asw_convgemm_args a = {};
    a.A = x; a.A2 = pl.X[lvl]; m->dec[j].up_wt.bind(a, m->precision); a.bias = m->dec[j].up_bias.p; a.out = pl.raw_up[j];
    a.stats = pl.st_up[j];
    a.B = B; a.M_out = pl.Tl[lvl]; a.N = st * 2 * co; a.Cin = ci; a.taps = 1; a.stride = 1; a.dil = 1; a.pad = 0;
    a.a_row_stride = ci; a.a_batch_stride = (int64_t)pl.Tl[lvl] * ci; a.a_len = a.a_batch_stride;
    a.chan_mod = 2 * co;
    if ((rc = asw_convgemm_f32(&a, s))) return rc;
    const int To = pl.Tl[lvl] * st;
    float* g = pl.Qb[lvl - 1];
    float* r = nullptr;
    if (true && glu_on_load_ok(m->dec[j].res, m->precision, co)) {
      // GroupNorm + GLU happen while the first residual layer stages its rows (spot_model.hip, same place)
      if ((rc = asw_gn_finalize(pl.st_up[j], asw_convgemm_stats_tiles(a.M_out, a.N), B, To, co, 1e-5f, pl.mr_up[j], s))) return rc;
      const GluSrc src = {pl.raw_up[j], pl.mr_up[j], m->dec[j].gn_g.p, m->dec[j].gn_b.p, co > 64 ? g : nullptr};
      if ((rc = run_res(m->dec[j].res, m->precision, B, To, co, K, g, pl.Pb[lvl - 1], g, &r, s, &src))) return rc;
    } else {
      if ((rc = asw_gn_glu(pl.raw_up[j], pl.st_up[j], asw_convgemm_stats_tiles(a.M_out, a.N), m->dec[j].gn_g.p,
                           m->dec[j].gn_b.p, B, To, co, 1e-5f, g, s)))
        return rc;
      // residual ping-pong: g -> P -> g -> P ...
      if ((rc = run_res(m->dec[j].res, m->precision, B, To, co, K, g, pl.Pb[lvl - 1], g, &r, s))) return rc;
    }
    x = r;
    m->taps["dec" + std::to_string(j)] = {x, (size_t)B * To * co};
  }
  // ---- mask path (:457-484): every speaker's mask gates the latent of the shared reference channel
  const int E = c.encoder_channels, EK = c.encoder_kernel_size, ES = c.encoder_stride;
  if (fused_mask_path(m)) {
    asw_maskpath_args f = {};
    asw_convgemm_args& a = f.enc;
    a.A = x; m->mask_wt.bind(a, m->precision); a.bias = m->mask_b.p;
    a.B = B; a.M_out = pl.F; a.N = E; a.Cin = c.channels; a.taps = EK; a.stride = ES; a.dil = 1; a.pad = EK / 2;
    a.a_row_stride = c.channels; a.a_batch_stride = (int64_t)pl.Tp * c.channels; a.a_len = a.a_batch_stride;
    f.ref = pl.refn; f.ref_batch_stride = pl.RL; f.ref_len = pl.RL; f.ref_hop = ES;
    f.byp_k = 48; f.byp_taps = EK; f.byp_shift = m->byp_wt48.shift; f.byp_hi = m->byp_wt48.fhi; f.byp_lo = m->byp_wt48.flo;
    f.byp_bias = m->byp_b.p;
    f.dec_hi = m->dec_wt.fhi; f.dec_lo = m->dec_wt.flo; f.dec_shift = m->dec_wt.shift; f.dec_taps = EK;
    f.taps = pl.D;
    if ((rc = asw_mask_path_f16x3(&f, s))) return rc;
    return asw_overlap_add_parts(pl.D, E / 256, B, pl.F, 64, EK, EK / 2, pl.T, 9, 8, m->out_bias, mean, stdv, out_wave, s);
  }
  {
    asw_convgemm_args a = {};
    a.A = pl.refn; m->byp_wt.bind(a, m->precision); a.bias = m->byp_b.p; a.out = pl.Y;
    a.B = B; a.M_out = pl.F; a.N = E; a.Cin = m->byp_k; a.taps = 1; a.stride = 1; a.dil = 1; a.pad = 0;
    a.a_row_stride = ES; a.a_batch_stride = pl.RL; a.a_len = pl.RL; a.relu = 1;
    if ((rc = asw_convgemm_f32(&a, s))) return rc;
  }
  {
    asw_convgemm_args a = {};
    a.A = x; m->mask_wt.bind(a, m->precision); a.bias = m->mask_b.p; a.mul = pl.Y; a.out = pl.Y;
    a.B = B; a.M_out = pl.F; a.N = E; a.Cin = c.channels; a.taps = EK; a.stride = ES; a.dil = 1; a.pad = EK / 2;
    a.a_row_stride = c.channels; a.a_batch_stride = (int64_t)pl.Tp * c.channels; a.a_len = a.a_batch_stride;
    a.relu = 1;
    if ((rc = asw_convgemm_f32(&a, s))) return rc;
  }
  {
    asw_convgemm_args a = {};
    a.A = pl.Y; m->dec_wt.bind(a, m->precision); a.out = pl.D;
    a.B = B; a.M_out = pl.F; a.N = 64; a.Cin = E; a.taps = 1; a.stride = 1; a.dil = 1; a.pad = 0;
    a.a_row_stride = E; a.a_batch_stride = (int64_t)pl.F * E; a.a_len = a.a_batch_stride;
    if ((rc = asw_convgemm_f32(&a, s))) return rc;
  }
  return asw_overlap_add_unnorm(pl.D, B, pl.F, 64, EK, EK / 2, pl.T, 9, 8, m->out_bias, mean, stdv, out_wave, s);
}

int check_ready(const asw_sep* m) {
  if (!m) return asw::set_error(ASW_ERR_ARG, "null model handle");
  if (!m->finalized) return asw::set_error(ASW_ERR_STATE, "asw_sep_finalize() has not been called");
  int dev = -1;
  ASW_HIP(hipGetDevice(&dev));
  if (dev != m->device)
    return asw::set_error(ASW_ERR_STATE, "model lives on HIP device %d but the current device is %d", m->device, dev);
  return ASW_OK;
}

}  // namespace

extern "C" int asw_sep_create(const asw_sep_config* cfg, asw_sep** out) {
  ASW_CHECK_ARG(cfg && out, "sep_create: null pointer");
  const asw_sep_config& c = *cfg;
  ASW_CHECK_ARG(c.depth >= 1 && c.depth <= 8, "sep_create: depth %d", c.depth);
  ASW_CHECK_ARG(c.n_mics >= 1 && c.n_mics <= 32, "sep_create: n_mics %d", c.n_mics);
  ASW_CHECK_ARG(c.max_speakers >= 1 && c.max_speakers <= 64, "sep_create: max_speakers %d", c.max_speakers);
  ASW_CHECK_ARG(c.channels % 64 == 0, "sep_create: channels=%d must be a multiple of 64 for the MFMA tiles", c.channels);
  ASW_CHECK_ARG(c.growth >= 1 && c.residual_layers >= 1 && c.bottleneck_layers >= 0, "sep_create: bad config");
  ASW_CHECK_ARG(c.kernel_size % 2 == 1 && c.bottleneck_ksize % 2 == 1, "sep_create: kernel sizes must be odd");
  ASW_CHECK_ARG(c.encoder_channels % 128 == 0, "sep_create: encoder_channels must be a multiple of 128");
  ASW_CHECK_ARG(c.encoder_stride % 4 == 0 && c.encoder_kernel_size / 2 == c.encoder_stride && c.encoder_kernel_size <= 64,
                "sep_create: encoder kernel/stride %d/%d unsupported (the reference's trim [9:-8] assumes 33/16)",
                c.encoder_kernel_size, c.encoder_stride);
  ASW_CHECK_ARG(c.ffw_dim % 128 == 0, "sep_create: ffw_dim must be a multiple of 128");
  std::unique_ptr<asw_sep> m(new asw_sep());
  m->cfg = c;
  ASW_HIP(hipGetDevice(&m->device));
  int cin = c.channels, ch = c.channels;
  for (int i = 0; i < c.depth; ++i) {
    ASW_CHECK_ARG(c.stride_list[i] >= 1, "sep_create: stride");
    m->enc_cin.push_back(cin);
    m->enc_cout.push_back(ch);
    m->stride_product *= c.stride_list[i];
    cin = ch;
    ch *= c.growth;
  }
  cin = c.channels; ch = c.channels;
  for (int i = 0; i < c.depth; ++i) {                   // decoder blocks in execution order (:221-231)
    m->dec_cin.insert(m->dec_cin.begin(), ch);
    m->dec_cout.insert(m->dec_cout.begin(), cin);
    m->dec_stride.insert(m->dec_stride.begin(), c.stride_list[i]);
    cin = ch;
    ch *= c.growth;
  }
  const int d = m->enc_cout.back();
  ASW_CHECK_ARG(d <= 1024 && (d & (d - 1)) == 0 && d >= 128, "sep_create: bottleneck width %d must be a power of two in 128..1024", d);
  const int hd = c.num_head > 0 && d % c.num_head == 0 ? d / c.num_head : 0;
  ASW_CHECK_ARG(hd == 16 || hd == 32 || hd == 64, "sep_create: head_dim %d unsupported (16, 32, 64)", hd);
  for (int i = 0; i < c.depth; ++i)
    ASW_CHECK_ARG(m->enc_cin[i] <= 512 && (m->enc_cin[i] & (m->enc_cin[i] - 1)) == 0,
                  "sep_create: level width %d must be a power of two <= 512", m->enc_cin[i]);
  *out = m.release();
  return ASW_OK;
}

extern "C" void asw_sep_destroy(asw_sep* m) { delete m; }

extern "C" int asw_sep_set_precision(asw_sep* m, int precision) {
  ASW_CHECK_ARG(m && (precision >= 0 && precision <= 2), "sep_set_precision: 0 (f32), 1 (f16x3) or 2 (single-pass f16)");
  m->precision = precision;
  return ASW_OK;
}

extern "C" int asw_sep_set_param(asw_sep* m, const char* key, const float* host_data, size_t numel) {
  ASW_CHECK_ARG(m && key && host_data, "sep_set_param: null pointer");
  m->raw[key].assign(host_data, host_data + numel);
  m->finalized = false;
  return ASW_OK;
}

extern "C" int asw_sep_finalize(asw_sep* m) {
  ASW_CHECK_ARG(m, "sep_finalize: null handle");
  {
    int dev = -1;
    ASW_HIP(hipGetDevice(&dev));
    if (dev != m->device)
      return asw::set_error(ASW_ERR_STATE, "sep_finalize: model was created on HIP device %d, current device is %d", m->device, dev);
  }
  const asw_sep_config& c = m->cfg;
  const auto want = expected_params(m);
  for (const auto& kv : want) {
    auto it = m->raw.find(kv.first);
    if (it == m->raw.end()) return asw::set_error(ASW_ERR_STATE, "state dict is missing key %s", kv.first.c_str());
    if (it->second.size() != kv.second)
      return asw::set_error(ASW_ERR_ARG, "%s: %zu elements, expected %zu", kv.first.c_str(), it->second.size(), kv.second);
  }
  if (m->raw.size() != want.size())
    return asw::set_error(ASW_ERR_ARG, "state dict has %zu unexpected keys", m->raw.size() - want.size());
  int rc;
#define UP(buf, vec) if ((rc = (buf).upload(vec))) return rc
  UP(m->pre_w, P(m, "preproc.weight"));
  UP(m->pre_b, P(m, "preproc.bias"));
  const int K = c.kernel_size;
  m->enc.clear(); m->enc.resize(c.depth);
  m->dec.clear(); m->dec.resize(c.depth);
  for (int i = 0; i < c.depth; ++i) {
    const std::string p = "encoder.module_list." + std::to_string(i);
    EncBlock& e = m->enc[i];
    if ((rc = pack_res_layers(m->raw, p, m->enc_cin[i], K, c.residual_layers, c.residual_dilation_factor, e.res))) return rc;
    if ((rc = e.down_wt.upload_gemm(pack_conv(P(m, p + ".conv1.weight"), 2 * m->enc_cout[i], m->enc_cin[i], K, nullptr),
                                    2 * m->enc_cout[i], m->enc_cin[i] * K)))
      return rc;
    UP(e.bias, P(m, p + ".conv1.bias"));
    UP(e.gn_g, P(m, p + ".norm1.weight"));
    UP(e.gn_b, P(m, p + ".norm1.bias"));
  }
  for (int i = 0; i < c.depth; ++i) {
    const std::string p = "decoder.module_list." + std::to_string(i);
    DecBlock& dd = m->dec[i];
    const int ci = m->dec_cin[i], co2 = 2 * m->dec_cout[i], st = m->dec_stride[i];
    if ((rc = pack_res_layers(m->raw, p, m->dec_cout[i], K, c.residual_layers, c.residual_dilation_factor, dd.res))) return rc;
    const std::vector<float>& w = P(m, p + ".upsample.conv.weight");   // [ci][co2][st]
    const std::vector<float>& b = P(m, p + ".upsample.conv.bias");
    // ConvTranspose1d with kernel == stride is a plain GEMM whose output row t_in holds the st
    // output frames t_in*st .. t_in*st+st-1 back to back: column n' = r*co2 + n.
    std::vector<float> wt((size_t)st * co2 * ci), bb((size_t)st * co2);
    for (int r = 0; r < st; ++r)
      for (int n = 0; n < co2; ++n) {
        bb[(size_t)r * co2 + n] = b[n];
        for (int cc = 0; cc < ci; ++cc) wt[((size_t)r * co2 + n) * ci + cc] = w[((size_t)cc * co2 + n) * st + r];
      }
    if ((rc = dd.up_wt.upload_gemm(wt, st * co2, ci))) return rc;
    UP(dd.up_bias, bb);
    UP(dd.gn_g, P(m, p + ".norm1.weight"));
    UP(dd.gn_b, P(m, p + ".norm1.bias"));
  }
  const int d = m->enc_cout.back(), H = c.num_head, hd = d / H, BK = c.bottleneck_ksize;
  m->inv_freq = P(m, "bottleneck.pe_single.inv_freq");
  m->pe_L = 0;
  m->conf.clear(); m->conf.resize(c.bottleneck_layers);
  m->inter.clear(); m->inter.resize(c.bottleneck_layers);
  for (int l = 0; l < c.bottleneck_layers; ++l) {
    const std::string cl = "bottleneck.module_list." + std::to_string(l) + ".intra.layers.0";
    ConfLayer& q = m->conf[l];
    auto ffn = [&](Ffn& f, const std::string& p) -> int {
      UP(f.lng, P(m, p + ".0.weight")); UP(f.lnb, P(m, p + ".0.bias"));
      UP(f.w1, P(m, p + ".1.ffn.0.weight")); UP(f.b1, P(m, p + ".1.ffn.0.bias"));
      UP(f.w2, scaled(P(m, p + ".1.ffn.3.weight"), 0.5f)); UP(f.b2, scaled(P(m, p + ".1.ffn.3.bias"), 0.5f));
      return ASW_OK;
    };
    if ((rc = ffn(q.f1, cl + ".ffn_module1"))) return rc;
    if ((rc = ffn(q.f2, cl + ".ffn_module2"))) return rc;
    UP(q.n1g, P(m, cl + ".norm1.norm.weight")); UP(q.n1b, P(m, cl + ".norm1.norm.bias"));
    UP(q.n2g, P(m, cl + ".norm2.norm.weight")); UP(q.n2b, P(m, cl + ".norm2.norm.bias"));
    const std::string ce = "bottleneck.module_list." + std::to_string(l) + ".intra.norm.norm";
    UP(q.fng, P(m, ce + ".weight")); UP(q.fnb, P(m, ce + ".bias"));
    {
      // RelPosMHAXL cuts the in_proj output per head into (q, k, v): source row h*3*hd + part*hd + c
      // -> row part*d + h*hd + c of the standard Q | K | V layout the attention kernel reads
      const std::vector<float>& w = P(m, cl + ".mha_layer.in_proj_weight");
      std::vector<float> o(w.size());
      for (int h = 0; h < H; ++h)
        for (int part = 0; part < 3; ++part)
          for (int cc = 0; cc < hd; ++cc)
            memcpy(&o[((size_t)part * d + h * hd + cc) * d], &w[((size_t)h * 3 * hd + part * hd + cc) * d], sizeof(float) * d);
      UP(q.w_in, o);
    }
    UP(q.w_pos, P(m, cl + ".mha_layer.linear_pos.weight"));
    UP(q.w_out, P(m, cl + ".mha_layer.out_proj.weight")); UP(q.b_out, P(m, cl + ".mha_layer.out_proj.bias"));
    // pos_bias_* are stored [hd][H] and read through .view(1,1,H,hd): the flat buffer, head-major
    UP(q.bu, P(m, cl + ".mha_layer.pos_bias_u")); UP(q.bv, P(m, cl + ".mha_layer.pos_bias_v"));
    UP(q.cm_lng, P(m, cl + ".convolution_module.layer_norm.weight"));
    UP(q.cm_lnb, P(m, cl + ".convolution_module.layer_norm.bias"));
    UP(q.cm_pw, P(m, cl + ".convolution_module.bottleneck.0.weight"));       // [2d][d][1] == [2d][d]
    UP(q.cm_pwb, P(m, cl + ".convolution_module.bottleneck.0.bias"));
    {
      const std::vector<float>& w = P(m, cl + ".convolution_module.conv.weight");   // [d][1][BK] -> [BK][d]
      std::vector<float> o((size_t)BK * d);
      for (int cc = 0; cc < d; ++cc)
        for (int k = 0; k < BK; ++k) o[(size_t)k * d + cc] = w[(size_t)cc * BK + k];
      UP(q.dw_wT, o);
    }
    UP(q.dw_b, P(m, cl + ".convolution_module.conv.bias"));
    UP(q.ac_lng, P(m, cl + ".convolution_module.after_conv.0.weight"));
    UP(q.ac_lnb, P(m, cl + ".convolution_module.after_conv.0.bias"));
    UP(q.ac_w, P(m, cl + ".convolution_module.after_conv.2.weight"));
    UP(q.ac_b, P(m, cl + ".convolution_module.after_conv.2.bias"));
    const std::string t = "bottleneck.module_list." + std::to_string(l) + ".inter.layers.0";
    InterLayer& il = m->inter[l];
    UP(il.w_in, P(m, t + ".self_attn.in_proj_weight")); UP(il.b_in, P(m, t + ".self_attn.in_proj_bias"));
    UP(il.w_out, P(m, t + ".self_attn.out_proj.weight")); UP(il.b_out, P(m, t + ".self_attn.out_proj.bias"));
    UP(il.w1, P(m, t + ".linear1.weight")); UP(il.b1, P(m, t + ".linear1.bias"));
    UP(il.w2, P(m, t + ".linear2.weight")); UP(il.b2, P(m, t + ".linear2.bias"));
    UP(il.n1g, P(m, t + ".norm1.weight")); UP(il.n1b, P(m, t + ".norm1.bias"));
    UP(il.n2g, P(m, t + ".norm2.weight")); UP(il.n2b, P(m, t + ".norm2.bias"));
  }
  const int E = c.encoder_channels, EK = c.encoder_kernel_size;
  m->byp_k = ((EK + 31) / 32) * 32;
  {
    const std::vector<float>& w = P(m, "reference_bypass.weight");   // [E][1][EK]
    std::vector<float> wt((size_t)E * m->byp_k, 0.f);
    for (int n = 0; n < E; ++n)
      for (int k = 0; k < EK; ++k) wt[(size_t)n * m->byp_k + k] = w[(size_t)n * EK + k];
    UP(m->byp_wt, wt);
    UP(m->byp_b, P(m, "reference_bypass.bias"));
    if (E % 32 == 0 && EK <= 48) {
      std::vector<float> w48((size_t)E * 48, 0.f);
      for (int n = 0; n < E; ++n)
        for (int k = 0; k < EK; ++k) w48[(size_t)n * 48 + k] = w[(size_t)n * EK + k];
      if ((rc = m->byp_wt48.upload_gemm(w48, E, 48))) return rc;
    }
  }
  if ((rc = m->mask_wt.upload_gemm(pack_conv(P(m, "mask_encoder.weight"), E, c.channels, EK, nullptr), E, c.channels * EK))) return rc;
  UP(m->mask_b, P(m, "mask_encoder.bias"));
  {
    const std::vector<float>& w = P(m, "output_decoder.weight");     // [E][1][EK]
    std::vector<float> wt((size_t)64 * E, 0.f);
    for (int j = 0; j < EK; ++j)
      for (int e = 0; e < E; ++e) wt[(size_t)j * E + e] = w[(size_t)e * EK + j];
    if ((rc = m->dec_wt.upload_gemm(wt, 64, E))) return rc;
    m->out_bias = P(m, "output_decoder.bias")[0];
  }
#undef UP
  m->finalized = true;
  return ASW_OK;
}

extern "C" int asw_sep_infer(asw_sep* m, const float* mix, int M, int T, const int32_t* offsets, int S, float* out,
                             void* stream) {
  int rc = check_ready(m);
  if (rc) return rc;
  ASW_CHECK_ARG(S >= 0 && S <= 64, "sep_infer: S=%d speakers (at most 64 per call)", S);
  if (S == 0) return ASW_OK;
  ASW_CHECK_ARG(mix && offsets && out, "sep_infer: null pointer");
  ASW_CHECK_ARG(M == m->cfg.n_mics, "sep_infer: mixture has %d channels, model expects %d", M, m->cfg.n_mics);
  ASW_CHECK_ARG(T >= 2, "sep_infer: T=%d", T);
  hipStream_t s = asw::as_stream(stream);
  Plan pl;
  if ((rc = ensure_ws(m, 1, S, T, pl))) return rc;
  const int C = m->cfg.channels, pad_l = m->cfg.encoder_kernel_size / 2;
  ASW_HIP(hipMemsetAsync(pl.refn, 0, (size_t)S * pl.RL * sizeof(float), s));
  if ((rc = asw_joint_shift_stats(mix, M, T, offsets, S, pl.jscr, pl.mean, pl.stdv, s))) return rc;
  if ((rc = asw_shift_norm_preproc(mix, M, T, pl.Tp, offsets, S, /*circular*/ 0, pl.mean, pl.stdv, m->pre_w.p, m->pre_b.p, C,
                                   pl.X[0], pl.refn + pad_l, pl.RL, s)))
    return rc;
  return run_network(m, pl, pl.mean, pl.stdv, out, s);
}

extern "C" int asw_sep_forward(asw_sep* m, const float* mix_norm, int B, int S, int M, int t, float* out, void* stream) {
  return asw_sep_forward_counts(m, mix_norm, B, S, M, t, nullptr, out, stream);
}

extern "C" int asw_sep_forward_counts(asw_sep* m, const float* mix_norm, int B, int S, int M, int t, const int32_t* counts,
                                      float* out, void* stream) {
  int rc = check_ready(m);
  if (rc) return rc;
  if (counts) {
    int mx = 0;
    for (int n = 0; n < B; ++n) {
      ASW_CHECK_ARG(counts[n] >= 1 && counts[n] <= S, "sep_forward: item %d holds %d speakers of a %d-wide stack", n, counts[n], S);
      mx = counts[n] > mx ? counts[n] : mx;
    }
    ASW_CHECK_ARG(B == 0 || mx == S, "sep_forward: the stack is %d speakers wide but the largest item holds %d "
                                     "(the reference pads to the largest count, :250-268)", S, mx);
  }
  ASW_CHECK_ARG(B >= 0 && S >= 1 && (long)B * S <= 64, "sep_forward: B=%d S=%d (at most 64 sequences per call)", B, S);
  if (B == 0) return ASW_OK;
  ASW_CHECK_ARG(mix_norm && out, "sep_forward: null pointer");
  ASW_CHECK_ARG(M == m->cfg.n_mics && t >= 1, "sep_forward: bad shape");
  hipStream_t s = asw::as_stream(stream);
  Plan pl;
  if ((rc = ensure_ws(m, B, S, t, pl))) return rc;
  const int BS = B * S, C = m->cfg.channels, pad_l = m->cfg.encoder_kernel_size / 2;
  ASW_HIP(hipMemsetAsync(pl.refn, 0, (size_t)BS * pl.RL * sizeof(float), s));
  // [B][S*M][t] is [B*S][M][t]: every speaker block of M channels is one sequence (:436-437)
  if ((rc = asw_pad_preproc(mix_norm, BS, M, t, pl.Tp, m->pre_w.p, m->pre_b.p, C, pl.X[0], pl.refn + pad_l, pl.RL, s))) return rc;
  // the reference channel of an item is the FIRST channel of its stack (:430): give every
  // speaker of item b the padded channel 0 of sequence (b, 0)
  for (int sp = 1; sp < S; ++sp)
    ASW_HIP(hipMemcpy2DAsync(pl.refn + (size_t)sp * pl.RL, (size_t)S * pl.RL * sizeof(float), pl.refn,
                             (size_t)S * pl.RL * sizeof(float), (size_t)pl.RL * sizeof(float), B, hipMemcpyDeviceToDevice, s));
  pl.counts = counts;
  if ((rc = run_network(m, pl, nullptr, nullptr, pl.ywave, s))) return rc;
  if (counts) {
    // a missing speaker has a zero mask, so its output is the bare output_decoder bias (:474-484)
    uint32_t bits;
    memcpy(&bits, &m->out_bias, sizeof bits);
    for (int n = 0; n < B; ++n)
      if (counts[n] < S)
        ASW_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(pl.ywave + ((size_t)n * S + counts[n]) * t), (int)bits,
                                  (size_t)(S - counts[n]) * t, s));
  }
  // rows padded with zeros to max_speakers (:486-488)
  const int R = S > m->cfg.max_speakers ? S : m->cfg.max_speakers;
  ASW_HIP(hipMemsetAsync(out, 0, (size_t)B * R * t * sizeof(float), s));
  ASW_HIP(hipMemcpy2DAsync(out, (size_t)R * t * sizeof(float), pl.ywave, (size_t)S * t * sizeof(float),
                           (size_t)S * t * sizeof(float), B, hipMemcpyDeviceToDevice, s));
  return ASW_OK;
}

extern "C" int asw_sep_get_config(const asw_sep* m, asw_sep_config* out) {
  ASW_CHECK_ARG(m && out, "sep_get_config: null pointer");
  *out = m->cfg;
  return ASW_OK;
}

extern "C" int asw_sep_get_tap(asw_sep* m, const char* name, float* dst, size_t capacity, size_t* numel, void* stream) {
  ASW_CHECK_ARG(m && name && numel, "sep_get_tap: null pointer");
  auto it = m->taps.find(name);
  if (it == m->taps.end()) return asw::set_error(ASW_ERR_ARG, "sep_get_tap: no activation named %s", name);
  *numel = it->second.numel;
  if (dst) {
    ASW_CHECK_ARG(capacity >= it->second.numel, "sep_get_tap: buffer too small");
    ASW_HIP(hipMemcpyAsync(dst, it->second.p, it->second.numel * sizeof(float), hipMemcpyDeviceToDevice,
                           asw::as_stream(stream)));
  }
  return ASW_OK;
}
