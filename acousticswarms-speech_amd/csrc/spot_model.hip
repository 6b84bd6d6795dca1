// spot_model.hip -- device-resident spot network: weight packing + the layer schedule of
// Network.forward (sep/training/SpeakerLocalization/network.py:363-405) and of the
// candidate hot loop DataParallelSpotModel.shift_and_sep
// (sep/training/JointModel/network.py:37-104), expressed as launches of the kernels in
// prep_kernels.hip / convgemm.hip / misc_kernels.hip on one HIP stream.
//
// Data layout in HBM: every activation is channels-last [B][T_l][C] fp32, so a
// LayerNorm row, a GLU pair and a GEMM A-row are each contiguous.  Weights are packed
// once (finalize) as Wt[N][tap*Cin + c]; the window gate of an encoder/decoder block
// (embed1, network.py:101,186) is folded into the adjacent convolution's weights per
// window embedding, so it costs nothing at run time.
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "model_common.h"

using namespace asw_model;

namespace {


struct EncBlock { std::vector<ResLayer> res; DevBuf bias, gn_g, gn_b; int cin = 0, cout = 0, stride = 1; };
struct DecBlock { std::vector<ResLayer> res; DevBuf gn_g, gn_b; int cin = 0, cout = 0, stride = 1; };
struct TfLayer { WBuf w_in, w_out, w1, w2; DevBuf b_in, b_out, b1, b2, n1g, n1b, n2g, n2b; };
// window-embedding dependent weights (gate folded in)
struct GateSet {
  std::vector<std::unique_ptr<WBuf>> down_wt;   // per encoder block
  std::vector<std::unique_ptr<WBuf>> up_wt;     // per decoder block
  std::vector<std::unique_ptr<DevBuf>> up_bias;
  uint64_t stamp = 0;                           // last use (LRU eviction)
};
struct Tap { const float* p; size_t numel; };

}  // namespace

struct asw_spot {
  asw_spot_config cfg;
  std::map<std::string, std::vector<float>> raw;
  bool finalized = false;
  int batch = 32;
  int device = 0;                          // HIP device the weights and the workspace live on
  uint64_t clock = 0;

  // derived
  std::vector<int> enc_cin, enc_cout;      // per encoder block
  std::vector<int> dec_cin, dec_cout, dec_stride;
  int stride_product = 1;

  DevBuf pre_w, pre_b;
  std::vector<EncBlock> enc;
  std::vector<DecBlock> dec;
  std::vector<TfLayer> tf;
  WBuf byp_wt, mask_wt, dec_wt;
  WBuf byp_wt48;                           // bypass kernel padded to 48 taps, fragment order (fused mask path)
  bool fuse_mask = true;                   // f16x3: bypass + mask encoder + decoder taps in one launch
  DevBuf byp_b, mask_b;
  int precision = 0;                       // 0 = exact f32 MFMA, 1 = f16x3 split MFMA
  float out_bias = 0.f;
  int byp_k = 0;                           // padded K of the bypass GEMM
  std::map<std::pair<float, float>, std::unique_ptr<GateSet>> gates;

  // workspace: one arena per lane.  With two lanes consecutive internal batches run on two HIP
  // streams (the caller's and a side stream), so the memory-bound passes and the launch tails of
  // one batch overlap the MFMA kernels of the other.
  char* ws[2] = {nullptr, nullptr};
  size_t ws_bytes[2] = {0, 0};
  int lanes = 1;
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  std::map<std::string, Tap> taps;

  ~asw_spot() {
    for (char* w : ws) if (w) (void)hipFree(w);
    if (side) (void)hipStreamDestroy(side);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
  }
};

namespace {

const std::vector<float>& P(const asw_spot* m, const std::string& k) { return m->raw.at(k); }

std::vector<std::pair<std::string, size_t>> expected_params(const asw_spot* m) {
  const asw_spot_config& c = m->cfg;
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
    v.push_back({p + ".embed1.weight", ci * 2});
    v.push_back({p + ".embed1.bias", ci});
  }
  for (int i = 0; i < c.depth; ++i) {
    const std::string p = "decoder.module_list." + std::to_string(i);
    const size_t ci = m->dec_cin[i], co = m->dec_cout[i], s = m->dec_stride[i];
    v.push_back({p + ".upsample.conv.weight", ci * 2 * co * s});
    v.push_back({p + ".upsample.conv.bias", 2 * co});
    v.push_back({p + ".norm1.weight", 2 * co});
    v.push_back({p + ".norm1.bias", 2 * co});
    res(p, co);
    v.push_back({p + ".embed1.weight", 2 * co * 2});
    v.push_back({p + ".embed1.bias", 2 * co});
  }
  const size_t E = c.encoder_channels, EK = c.encoder_kernel_size;
  v.push_back({"reference_bypass.weight", E * EK});
  v.push_back({"reference_bypass.bias", E});
  v.push_back({"mask_encoder.weight", E * c.channels * EK});
  v.push_back({"mask_encoder.bias", E});
  v.push_back({"output_decoder.weight", E * EK});
  v.push_back({"output_decoder.bias", 1});
  const size_t d = m->enc_cout.back(), f = c.ffw_dim;
  for (int l = 0; l < c.num_transformer_layers; ++l) {
    const std::string p = "bottleneck.transf.layers." + std::to_string(l);
    v.push_back({p + ".self_attn.in_proj_weight", 3 * d * d});
    v.push_back({p + ".self_attn.in_proj_bias", 3 * d});
    v.push_back({p + ".self_attn.out_proj.weight", d * d});
    v.push_back({p + ".self_attn.out_proj.bias", d});
    v.push_back({p + ".linear1.weight", f * d});
    v.push_back({p + ".linear1.bias", f});
    v.push_back({p + ".linear2.weight", d * f});
    v.push_back({p + ".linear2.bias", d});
    v.push_back({p + ".norm1.weight", d});
    v.push_back({p + ".norm1.bias", d});
    v.push_back({p + ".norm2.weight", d});
    v.push_back({p + ".norm2.bias", d});
  }
  return v;
}

int pack_res(asw_spot* m, const std::string& p, int ch, std::vector<ResLayer>& out) {
  const asw_spot_config& c = m->cfg;
  return pack_res_layers(m->raw, p, ch, c.kernel_size, c.residual_layers, c.residual_dilation_factor, out);
}

// gate[c] = W[c][0]*w0 + W[c][1]*w1 + b[c]   (embed1 is Conv1d(2->C, k=1))
std::vector<float> gate_of(const std::vector<float>& w, const std::vector<float>& b, float w0, float w1) {
  std::vector<float> g(b.size());
  for (size_t c = 0; c < b.size(); ++c) g[c] = w[2 * c] * w0 + w[2 * c + 1] * w1 + b[c];
  return g;
}

int get_gates(asw_spot* m, float w0, float w1, GateSet** out) {
  auto key = std::make_pair(w0, w1);
  auto it = m->gates.find(key);
  if (it != m->gates.end()) { it->second->stamp = ++m->clock; *out = it->second.get(); return ASW_OK; }
  if (m->gates.size() >= 8) {
    // bounded cache: drop the least recently used set.  Launches that read it may still be queued;
    // its buffers are released with hipFree, which waits for the device, so they finish first.
    auto lru = m->gates.begin();
    for (auto g = m->gates.begin(); g != m->gates.end(); ++g)
      if (g->second->stamp < lru->second->stamp) lru = g;
    m->gates.erase(lru);
  }
  std::unique_ptr<GateSet> gs(new GateSet());
  gs->stamp = ++m->clock;
  const asw_spot_config& c = m->cfg;
  for (int i = 0; i < c.depth; ++i) {
    const std::string p = "encoder.module_list." + std::to_string(i);
    const std::vector<float> g = gate_of(P(m, p + ".embed1.weight"), P(m, p + ".embed1.bias"), w0, w1);
    gs->down_wt.emplace_back(new WBuf());
    int rc = gs->down_wt.back()->upload_gemm(
        pack_conv(P(m, p + ".conv1.weight"), 2 * m->enc_cout[i], m->enc_cin[i], c.kernel_size, g.data()),
        2 * m->enc_cout[i], m->enc_cin[i] * c.kernel_size);
    if (rc) return rc;
  }
  for (int i = 0; i < c.depth; ++i) {
    const std::string p = "decoder.module_list." + std::to_string(i);
    const int ci = m->dec_cin[i], co2 = 2 * m->dec_cout[i], s = m->dec_stride[i];
    const std::vector<float> g = gate_of(P(m, p + ".embed1.weight"), P(m, p + ".embed1.bias"), w0, w1);
    const std::vector<float>& w = P(m, p + ".upsample.conv.weight");   // [ci][co2][s]
    const std::vector<float>& b = P(m, p + ".upsample.conv.bias");
    // ConvTranspose1d with kernel == stride is a plain GEMM whose output row t_in holds the
    // s output frames t_in*s..t_in*s+s-1 back to back: column n' = r*co2 + n.
    std::vector<float> wt((size_t)s * co2 * ci), bb((size_t)s * co2);
    for (int r = 0; r < s; ++r)
      for (int n = 0; n < co2; ++n) {
        bb[(size_t)r * co2 + n] = b[n] * g[n];
        for (int cc = 0; cc < ci; ++cc)
          wt[((size_t)r * co2 + n) * ci + cc] = w[((size_t)cc * co2 + n) * s + r] * g[n];
      }
    gs->up_wt.emplace_back(new WBuf());
    gs->up_bias.emplace_back(new DevBuf());
    int rc;
    if ((rc = gs->up_wt.back()->upload_gemm(wt, s * co2, ci))) return rc;
    if ((rc = gs->up_bias.back()->upload(bb))) return rc;
  }
  *out = gs.get();
  m->gates[key] = std::move(gs);
  return ASW_OK;
}

// The one-launch mask path (asw_mask_path_f16x3) applies in f16x3 mode when the shapes fit its tiles.
bool fused_mask_path(const asw_spot* m) {
  const asw_spot_config& c = m->cfg;
  return m->fuse_mask && m->precision >= 1 && c.encoder_channels % 256 == 0 && c.channels % 32 == 0 &&
         c.encoder_kernel_size <= 48 && c.encoder_stride % 4 == 0 && m->byp_wt48.fhi && m->dec_wt.fhi && m->mask_wt.fhi;
}

struct Plan {
  int B, T, Tp, F, RL, depth;
  std::vector<int> Tl;                     // length at level 0..depth
  float *mean, *stdv, *refn;
  std::vector<float*> X, Pb, Qb;           // level tensors: X[i] input of enc block i (X[0]=preproc out)
  std::vector<float*> raw_dn, raw_up, st_dn, st_up, mr_up, mr_dn;
  float *qkv, *ctx, *x1, *ff, *ha, *hb, *Y, *D, *ywave;
  double *escr;
};

void layout(const asw_spot* m, int B, int T, Arena& a, Plan& pl) {
  const asw_spot_config& c = m->cfg;
  pl.B = B; pl.T = T; pl.depth = c.depth;
  pl.Tp = ((T - 1) / m->stride_product + 1) * m->stride_product;
  const int EK = c.encoder_kernel_size, ES = c.encoder_stride;
  pl.F = (pl.Tp + 2 * (EK / 2) - EK) / ES + 1;
  pl.RL = ((EK / 2 + pl.Tp + m->byp_k + 64) + 3) & ~3;
  pl.Tl.assign(c.depth + 1, pl.Tp);
  for (int i = 0; i < c.depth; ++i) pl.Tl[i + 1] = pl.Tl[i] / c.stride_list[i];
  pl.mean = a.take<float>(B);
  pl.stdv = a.take<float>(B);
  pl.refn = a.take<float>((size_t)B * pl.RL);
  pl.X.resize(c.depth + 1); pl.Pb.resize(c.depth); pl.Qb.resize(c.depth);
  pl.raw_dn.resize(c.depth); pl.raw_up.resize(c.depth); pl.st_dn.resize(c.depth); pl.st_up.resize(c.depth);
  pl.mr_up.resize(c.depth); pl.mr_dn.resize(c.depth);
  for (int i = 0; i <= c.depth; ++i) {
    const int ch = i == 0 ? c.channels : m->enc_cout[i - 1];
    pl.X[i] = a.take<float>((size_t)B * pl.Tl[i] * ch);
  }
  for (int i = 0; i < c.depth; ++i) {
    const size_t n = (size_t)B * pl.Tl[i] * m->enc_cin[i];
    pl.Pb[i] = a.take<float>(n);
    pl.Qb[i] = a.take<float>(n);
    pl.raw_dn[i] = a.take<float>((size_t)B * pl.Tl[i + 1] * 2 * m->enc_cout[i]);
    pl.st_dn[i] = a.take<float>((size_t)B * 4 * asw_convgemm_stats_tiles(pl.Tl[i + 1], 2 * m->enc_cout[i]));
    pl.mr_dn[i] = a.take<float>((size_t)B * 4);
  }
  for (int j = 0; j < c.depth; ++j) {
    const int lvl = c.depth - j;            // input level of decoder block j
    const int s = m->dec_stride[j], co2 = 2 * m->dec_cout[j];
    pl.raw_up[j] = a.take<float>((size_t)B * pl.Tl[lvl] * s * co2);
    pl.st_up[j] = a.take<float>((size_t)B * 4 * asw_convgemm_stats_tiles(pl.Tl[lvl], s * co2));
    pl.mr_up[j] = a.take<float>((size_t)B * 4);
  }
  const size_t L = pl.Tl[c.depth], d = m->enc_cout.back();
  pl.qkv = a.take<float>((size_t)B * L * 3 * d);
  pl.ctx = a.take<float>((size_t)B * L * d);
  pl.x1 = a.take<float>((size_t)B * L * d);
  pl.ff = a.take<float>((size_t)B * L * c.ffw_dim);
  pl.ha = a.take<float>((size_t)B * L * d);
  pl.hb = a.take<float>((size_t)B * L * d);
  const bool fused = fused_mask_path(m);
  // fused mask path: no latent, one partial tap tensor per 256-channel column tile
  pl.Y = fused ? nullptr : a.take<float>((size_t)B * pl.F * c.encoder_channels);
  pl.D = a.take<float>((size_t)(fused ? c.encoder_channels / 256 : 1) * B * pl.F * 64);
  pl.ywave = a.take<float>((size_t)B * T);
  pl.escr = a.take<double>((size_t)B * (T + 1));
}

int ensure_ws(asw_spot* m, int B, int T, Plan& pl, int lane = 0) {
  Arena dry(nullptr, 0, true);
  layout(m, B, T, dry, pl);
  const size_t need = dry.off + 4096;
  if (need > m->ws_bytes[lane]) {
    // A workspace that has to grow grows to the model's full internal batch at once: a search issues calls of 30,
    // then 50 ... 256 candidates, and every growth is a device synchronisation plus a hipFree / hipMalloc of tens of
    // GB (about 0.2 GB per candidate at T = 48 000) -- 2-3 s each in the kernel trace of the 64-mixture run.
    size_t want = need;
    if (B < m->batch) {
      Plan full;
      Arena dry_full(nullptr, 0, true);
      layout(m, m->batch, T, dry_full, full);
      want = dry_full.off + 4096;
    }
    if (m->ws[lane]) { ASW_HIP(hipDeviceSynchronize()); (void)hipFree(m->ws[lane]); m->ws[lane] = nullptr; m->ws_bytes[lane] = 0; }
    if (hipMalloc(&m->ws[lane], want) != hipSuccess) {
      (void)hipGetLastError();
      want = need;                                         // not enough memory for the full batch: what this call needs
      if (hipMalloc(&m->ws[lane], want) != hipSuccess)
        return asw::set_error(ASW_ERR_NOMEM, "workspace of %.1f MiB for batch %d, T=%d", need / 1048576.0, B, T);
    }
    m->ws_bytes[lane] = want;
  }
  Arena real(m->ws[lane], m->ws_bytes[lane], false);
  layout(m, B, T, real, pl);
  return ASW_OK;
}

// everything after the preproc stage; pl.X[0] / pl.refn are filled
int run_network(asw_spot* m, Plan& pl, GateSet* gs, const float* mean, const float* stdv, float* out_wave,
                hipStream_t s) {
  const asw_spot_config& c = m->cfg;
  const int B = pl.B, K = c.kernel_size;
  m->taps.clear();
  m->taps["preproc"] = {pl.X[0], (size_t)B * pl.Tl[0] * c.channels};
  int rc;
  // ---- encoder (network.py:98-113,146-156)
  GluSrc enc_src = {};
  bool enc_glu = false;                       // X[i] is still un-normalised in raw_dn[i-1]: block i applies GroupNorm + GLU
  for (int i = 0; i < c.depth; ++i) {
    float* r = nullptr;
    if ((rc = run_res(m->enc[i].res, m->precision, B, pl.Tl[i], m->enc_cin[i], K, pl.X[i], pl.Pb[i], pl.Qb[i], &r, s,
                      enc_glu ? &enc_src : nullptr)))
      return rc;
    asw_convgemm_args a = {};
    a.A = r; gs->down_wt[i]->bind(a, m->precision); a.bias = m->enc[i].bias.p; a.out = pl.raw_dn[i]; a.stats = pl.st_dn[i];
    a.B = B; a.M_out = pl.Tl[i + 1]; a.N = 2 * m->enc_cout[i]; a.Cin = m->enc_cin[i]; a.taps = K;
    a.stride = c.stride_list[i]; a.dil = 1; a.pad = K / 2;
    a.a_row_stride = a.Cin; a.a_batch_stride = (int64_t)pl.Tl[i] * a.Cin; a.a_len = a.a_batch_stride;
    a.chan_mod = a.N;
    if ((rc = asw_convgemm_f32(&a, s))) return rc;
    // the next block normalises while its first layer stages rows and writes X[i+1] (the skip connection the
    // decoder reads) from the same registers: one pass over raw_dn less
    enc_glu = m->fuse_mask && i + 1 < c.depth && glu_on_load_ok(m->enc[i + 1].res, m->precision, m->enc_cout[i]);
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
  // ---- bottleneck (network.py:240-265): post-norm transformer layers, batch-first rows
  const int L = pl.Tl[c.depth], d = m->enc_cout.back(), rows = B * L;
  const float* h = pl.X[c.depth];
  for (int l = 0; l < c.num_transformer_layers; ++l) {
    TfLayer& t = m->tf[l];
    float* hout = (l % 2 == 0) ? pl.ha : pl.hb;
    if ((rc = linear(h, t.w_in, m->precision, t.b_in.p, rows, 3 * d, d, 0, nullptr, nullptr, nullptr, pl.qkv, s))) return rc;
    if ((rc = asw_attention_prec(pl.qkv, B, L, d, c.num_head, m->precision, pl.ctx, s))) return rc;
    if ((rc = linear(pl.ctx, t.w_out, m->precision, t.b_out.p, rows, d, d, 0, h, t.n1g.p, t.n1b.p, pl.x1, s))) return rc;
    if ((rc = linear(pl.x1, t.w1, m->precision, t.b1.p, rows, c.ffw_dim, d, 1, nullptr, nullptr, nullptr, pl.ff, s))) return rc;
    if ((rc = linear(pl.ff, t.w2, m->precision, t.b2.p, rows, d, c.ffw_dim, 0, pl.x1, t.n2g.p, t.n2b.p, hout, s))) return rc;
    h = hout;
  }
  m->taps["bottleneck"] = {h, (size_t)rows * d};
  // ---- decoder (network.py:180-200,233-238)
  const float* x = h;
  for (int j = 0; j < c.depth; ++j) {
    const int lvl = c.depth - j, ci = m->dec_cin[j], co = m->dec_cout[j], st = m->dec_stride[j];
    asw_convgemm_args a = {};
    a.A = x; a.A2 = pl.X[lvl]; gs->up_wt[j]->bind(a, m->precision); a.bias = gs->up_bias[j]->p; a.out = pl.raw_up[j];
    a.stats = pl.st_up[j];
    a.B = B; a.M_out = pl.Tl[lvl]; a.N = st * 2 * co; a.Cin = ci; a.taps = 1; a.stride = 1; a.dil = 1; a.pad = 0;
    a.a_row_stride = ci; a.a_batch_stride = (int64_t)pl.Tl[lvl] * ci; a.a_len = a.a_batch_stride;
    a.chan_mod = 2 * co;
    if ((rc = asw_convgemm_f32(&a, s))) return rc;
    const int To = pl.Tl[lvl] * st;          // == pl.Tl[lvl-1]
    float* g = pl.Qb[lvl - 1];
    float* r = nullptr;
    if (m->fuse_mask && glu_on_load_ok(m->dec[j].res, m->precision, co)) {
      // GroupNorm + GLU happen while the first residual layer stages its rows: at 64 channels the normalised
      // tensor is neither written nor read back (P -> g -> P are the stack's own buffers), above that it is
      // written once for the layer's residual instead of written and read twice
      if ((rc = asw_gn_finalize(pl.st_up[j], asw_convgemm_stats_tiles(a.M_out, a.N), B, To, co, 1e-5f, pl.mr_up[j], s))) return rc;
      // (g: free until the second layer writes it; the wide layers read their residual from there)
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
  // ---- mask path (network.py:327-349,397-405)
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
    m->taps.erase("latent");                                   // not materialised on this path
    return asw_overlap_add_parts(pl.D, E / 256, B, pl.F, 64, EK, EK / 2, pl.T, 9, 8, m->out_bias, mean, stdv, out_wave, s);
  }
  {
    asw_convgemm_args a = {};   // reference_bypass: rows of the padded reference channel, hop ES
    a.A = pl.refn; m->byp_wt.bind(a, m->precision); a.bias = m->byp_b.p; a.out = pl.Y;
    a.B = B; a.M_out = pl.F; a.N = E; a.Cin = m->byp_k; a.taps = 1; a.stride = 1; a.dil = 1; a.pad = 0;
    a.a_row_stride = ES; a.a_batch_stride = pl.RL; a.a_len = pl.RL; a.relu = 1;
    if ((rc = asw_convgemm_f32(&a, s))) return rc;
  }
  {
    asw_convgemm_args a = {};   // mask_encoder, ReLU, times the bypass latent (in place)
    a.A = x; m->mask_wt.bind(a, m->precision); a.bias = m->mask_b.p; a.mul = pl.Y; a.out = pl.Y;
    a.B = B; a.M_out = pl.F; a.N = E; a.Cin = c.channels; a.taps = EK; a.stride = ES; a.dil = 1; a.pad = EK / 2;
    a.a_row_stride = c.channels; a.a_batch_stride = (int64_t)pl.Tp * c.channels; a.a_len = a.a_batch_stride;
    a.relu = 1;
    if ((rc = asw_convgemm_f32(&a, s))) return rc;
  }
  m->taps["latent"] = {pl.Y, (size_t)B * pl.F * E};
  {
    asw_convgemm_args a = {};   // output_decoder taps: D[f][j] = sum_e latent[f][e] * w[e][j]
    a.A = pl.Y; m->dec_wt.bind(a, m->precision); a.out = pl.D;
    a.B = B; a.M_out = pl.F; a.N = 64; a.Cin = E; a.taps = 1; a.stride = 1; a.dil = 1; a.pad = 0;
    a.a_row_stride = E; a.a_batch_stride = (int64_t)pl.F * E; a.a_len = a.a_batch_stride;
    if ((rc = asw_convgemm_f32(&a, s))) return rc;
  }
  return asw_overlap_add_unnorm(pl.D, B, pl.F, 64, EK, EK / 2, pl.T, 9, 8, m->out_bias, mean, stdv, out_wave, s);
}

int check_ready(const asw_spot* m) {
  if (!m) return asw::set_error(ASW_ERR_ARG, "null model handle");
  if (!m->finalized) return asw::set_error(ASW_ERR_STATE, "asw_spot_finalize() has not been called");
  int dev = -1;
  ASW_HIP(hipGetDevice(&dev));
  if (dev != m->device)
    return asw::set_error(ASW_ERR_STATE, "model lives on HIP device %d but the current device is %d", m->device, dev);
  return ASW_OK;
}

}  // namespace

extern "C" int asw_spot_create(const asw_spot_config* cfg, asw_spot** out) {
  ASW_CHECK_ARG(cfg && out, "spot_create: null pointer");
  const asw_spot_config& c = *cfg;
  ASW_CHECK_ARG(c.depth >= 1 && c.depth <= 8, "spot_create: depth %d", c.depth);
  ASW_CHECK_ARG(c.n_mics >= 1 && c.n_mics <= 32, "spot_create: n_mics %d", c.n_mics);
  ASW_CHECK_ARG(c.channels % 64 == 0, "spot_create: channels=%d must be a multiple of 64 for the MFMA tiles", c.channels);
  ASW_CHECK_ARG(c.growth >= 1 && c.residual_layers >= 1 && c.num_transformer_layers >= 0, "spot_create: bad config");
  ASW_CHECK_ARG(c.kernel_size % 2 == 1, "spot_create: kernel_size must be odd");
  ASW_CHECK_ARG(c.encoder_channels % 128 == 0, "spot_create: encoder_channels must be a multiple of 128");
  ASW_CHECK_ARG(c.encoder_stride % 4 == 0 && c.encoder_kernel_size / 2 == c.encoder_stride &&
                    c.encoder_kernel_size <= 64,
                "spot_create: encoder kernel/stride %d/%d unsupported (the reference's trim [9:-8] assumes 33/16)",
                c.encoder_kernel_size, c.encoder_stride);
  ASW_CHECK_ARG(c.ffw_dim % 128 == 0, "spot_create: ffw_dim must be a multiple of 128");
  std::unique_ptr<asw_spot> m(new asw_spot());
  m->cfg = c;
  ASW_HIP(hipGetDevice(&m->device));
  int cin = c.channels, ch = c.channels;
  for (int i = 0; i < c.depth; ++i) {
    ASW_CHECK_ARG(c.stride_list[i] >= 1, "spot_create: stride");
    m->enc_cin.push_back(cin);
    m->enc_cout.push_back(ch);
    m->stride_product *= c.stride_list[i];
    cin = ch;
    ch *= c.growth;
  }
  // decoder blocks in execution order (network.py:221-231 inserts at the front)
  cin = c.channels; ch = c.channels;
  for (int i = 0; i < c.depth; ++i) {
    m->dec_cin.insert(m->dec_cin.begin(), ch);
    m->dec_cout.insert(m->dec_cout.begin(), cin);
    m->dec_stride.insert(m->dec_stride.begin(), c.stride_list[i]);
    cin = ch;
    ch *= c.growth;
  }
  const int d = m->enc_cout.back();
  ASW_CHECK_ARG(d <= 1024 && (d & (d - 1)) == 0, "spot_create: bottleneck width %d must be a power of two <= 1024", d);
  ASW_CHECK_ARG(d % c.num_head == 0 && (d / c.num_head) % 16 == 0 && d / c.num_head <= 128,
                "spot_create: head_dim %d unsupported", d / (c.num_head ? c.num_head : 1));
  for (int i = 0; i < c.depth; ++i)
    ASW_CHECK_ARG(m->enc_cin[i] <= 512 && (m->enc_cin[i] & (m->enc_cin[i] - 1)) == 0,
                  "spot_create: level width %d must be a power of two <= 512", m->enc_cin[i]);
  *out = m.release();
  return ASW_OK;
}

extern "C" void asw_spot_destroy(asw_spot* m) { delete m; }

extern "C" int asw_spot_set_precision(asw_spot* m, int precision) {
  ASW_CHECK_ARG(m && (precision >= 0 && precision <= 2), "set_precision: 0 (f32), 1 (f16x3) or 2 (single-pass f16)");
  m->precision = precision;
  return ASW_OK;
}

extern "C" int asw_spot_set_batch(asw_spot* m, int batch) {
  ASW_CHECK_ARG(m && batch >= 1 && batch <= 4096, "set_batch: bad argument");
  m->batch = batch;
  return ASW_OK;
}

extern "C" int asw_spot_set_param(asw_spot* m, const char* key, const float* host_data, size_t numel) {
  ASW_CHECK_ARG(m && key && host_data, "set_param: null pointer");
  m->raw[key].assign(host_data, host_data + numel);
  m->finalized = false;
  return ASW_OK;
}

extern "C" int asw_spot_finalize(asw_spot* m) {
  ASW_CHECK_ARG(m, "finalize: null handle");
  {
    int dev = -1;
    ASW_HIP(hipGetDevice(&dev));
    if (dev != m->device)
      return asw::set_error(ASW_ERR_STATE, "finalize: model was created on HIP device %d, current device is %d", m->device, dev);
  }
  const asw_spot_config& c = m->cfg;
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
  m->enc.clear(); m->enc.resize(c.depth);
  m->dec.clear(); m->dec.resize(c.depth);
  for (int i = 0; i < c.depth; ++i) {
    const std::string p = "encoder.module_list." + std::to_string(i);
    EncBlock& e = m->enc[i];
    e.cin = m->enc_cin[i]; e.cout = m->enc_cout[i]; e.stride = c.stride_list[i];
    if ((rc = pack_res(m, p, e.cin, e.res))) return rc;
    UP(e.bias, P(m, p + ".conv1.bias"));
    UP(e.gn_g, P(m, p + ".norm1.weight"));
    UP(e.gn_b, P(m, p + ".norm1.bias"));
  }
  for (int i = 0; i < c.depth; ++i) {
    const std::string p = "decoder.module_list." + std::to_string(i);
    DecBlock& dd = m->dec[i];
    dd.cin = m->dec_cin[i]; dd.cout = m->dec_cout[i]; dd.stride = m->dec_stride[i];
    if ((rc = pack_res(m, p, dd.cout, dd.res))) return rc;
    UP(dd.gn_g, P(m, p + ".norm1.weight"));
    UP(dd.gn_b, P(m, p + ".norm1.bias"));
  }
  m->tf.clear(); m->tf.resize(c.num_transformer_layers);
  for (int l = 0; l < c.num_transformer_layers; ++l) {
    const std::string p = "bottleneck.transf.layers." + std::to_string(l);
    TfLayer& t = m->tf[l];
    {
      const int dm = m->enc_cout.back(), ff = c.ffw_dim;
      if ((rc = t.w_in.upload_gemm(P(m, p + ".self_attn.in_proj_weight"), 3 * dm, dm))) return rc;
      if ((rc = t.w_out.upload_gemm(P(m, p + ".self_attn.out_proj.weight"), dm, dm))) return rc;
      if ((rc = t.w1.upload_gemm(P(m, p + ".linear1.weight"), ff, dm))) return rc;
      if ((rc = t.w2.upload_gemm(P(m, p + ".linear2.weight"), dm, ff))) return rc;
    }
    UP(t.b_in, P(m, p + ".self_attn.in_proj_bias"));
    UP(t.b_out, P(m, p + ".self_attn.out_proj.bias"));
    UP(t.b1, P(m, p + ".linear1.bias"));
    UP(t.b2, P(m, p + ".linear2.bias"));
    UP(t.n1g, P(m, p + ".norm1.weight")); UP(t.n1b, P(m, p + ".norm1.bias"));
    UP(t.n2g, P(m, p + ".norm2.weight")); UP(t.n2b, P(m, p + ".norm2.bias"));
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
  m->gates.clear();
  m->finalized = true;
  return ASW_OK;
}

extern "C" int asw_spot_shift_and_sep(asw_spot* m, const float* mix, int M, int T, const int32_t* offsets, int N,
                                      int strict, int circular, float* out_wave, double* out_energy,
                                      int energy_window, void* stream) {
  return asw_spot_shift_and_sep_multi(m, mix, 1, M, T, offsets, nullptr, N, strict, circular, out_wave, out_energy,
                                      energy_window, stream);
}

extern "C" int asw_spot_shift_and_sep_multi(asw_spot* m, const float* mix, int K, int M, int T, const int32_t* offsets,
                                            const int32_t* mix_index, int N, int strict, int circular, float* out_wave,
                                            double* out_energy, int energy_window, void* stream) {
  int rc = check_ready(m);
  ASW_CHECK_ARG(K >= 1 && (K == 1 || mix_index != nullptr), "shift_and_sep: K=%d mixtures need a mix_index array", K);
  if (rc) return rc;
  ASW_CHECK_ARG(N >= 0, "shift_and_sep: N=%d", N);
  if (N == 0) return ASW_OK;
  ASW_CHECK_ARG(mix && offsets, "shift_and_sep: null pointer");
  ASW_CHECK_ARG(M == m->cfg.n_mics, "shift_and_sep: mixture has %d channels, model expects %d", M, m->cfg.n_mics);
  ASW_CHECK_ARG(T >= 2, "shift_and_sep: T=%d", T);
  ASW_CHECK_ARG(out_energy == nullptr || energy_window > 0, "shift_and_sep: energy_window");
  hipStream_t s = asw::as_stream(stream);
  GateSet* gs = nullptr;
  if ((rc = get_gates(m, strict == 1 ? 1.f : 0.f, strict == 1 ? 0.f : 1.f, &gs))) return rc;
  const int Bmax = N < m->batch ? N : m->batch;
  const int n_batches = (N + Bmax - 1) / Bmax;
  const int lanes = (m->lanes == 2 && n_batches >= 2) ? 2 : 1;
  Plan pl[2];
  for (int l = 0; l < lanes; ++l)
    if ((rc = ensure_ws(m, Bmax, T, pl[l], l))) return rc;
  hipStream_t st[2] = {s, s};
  if (lanes == 2) {
    if (!m->side) {
      ASW_HIP(hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking));
      ASW_HIP(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
      ASW_HIP(hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming));
    }
    st[1] = m->side;
    ASW_HIP(hipEventRecord(m->ev_fork, s));                // the side lane starts after everything queued before this call
    ASW_HIP(hipStreamWaitEvent(m->side, m->ev_fork, 0));
  }
  const int C = m->cfg.channels, pad_l = m->cfg.encoder_kernel_size / 2;
  // The caller's stream continues after BOTH lanes on every exit path: when a launch fails half way the side
  // lane may still be writing out_wave / out_energy (the caller's buffers), so the join is not skipped -- and if
  // the join itself cannot be queued the side lane is drained before the status is returned.
  auto batches = [&]() -> int {
    int k = 0;
    for (int i0 = 0; i0 < N; i0 += Bmax, ++k) {
      const int B = N - i0 < Bmax ? N - i0 : Bmax;
      Plan& p = pl[k % lanes];
      hipStream_t q = st[k % lanes];
      p.B = B;
      const int32_t* off = offsets + (size_t)i0 * (M - 1);
      const int32_t* mi = mix_index ? mix_index + i0 : nullptr;
      ASW_HIP(hipMemsetAsync(p.refn, 0, (size_t)B * p.RL * sizeof(float), q));
      int r;
      if ((r = asw_shift_stats_multi(mix, M, T, off, mi, B, circular, p.mean, p.stdv, q))) return r;
      if ((r = asw_shift_norm_preproc_multi(mix, M, T, p.Tp, off, mi, B, circular, p.mean, p.stdv, m->pre_w.p, m->pre_b.p, C,
                                            p.X[0], p.refn + pad_l, p.RL, q)))
        return r;
      float* y = out_wave ? out_wave + (size_t)i0 * T : p.ywave;
      if ((r = run_network(m, p, gs, p.mean, p.stdv, y, q))) return r;
      if (out_energy && (r = asw_energies(y, B, T, energy_window, p.escr, out_energy + (size_t)i0 * 2, q))) return r;
    }
    return ASW_OK;
  };
  rc = batches();
  if (lanes == 2) {
    if (hipEventRecord(m->ev_join, m->side) != hipSuccess || hipStreamWaitEvent(s, m->ev_join, 0) != hipSuccess) {
      (void)hipStreamSynchronize(m->side);
      if (!rc) rc = asw::set_error(ASW_ERR_HIP, "shift_and_sep: joining the side lane failed");
    }
  }
  return rc;
}

extern "C" int asw_spot_set_lanes(asw_spot* m, int lanes) {
  ASW_CHECK_ARG(m && (lanes == 1 || lanes == 2), "set_lanes: 1 or 2");
  m->lanes = lanes;
  return ASW_OK;
}

extern "C" int asw_spot_forward(asw_spot* m, const float* mix_norm, int B, int M, int t,
                                const float* window_embedding_host, float* out, void* stream) {
  int rc = check_ready(m);
  if (rc) return rc;
  ASW_CHECK_ARG(B >= 0, "forward: B=%d", B);
  if (B == 0) return ASW_OK;
  ASW_CHECK_ARG(mix_norm && window_embedding_host && out, "forward: null pointer");
  ASW_CHECK_ARG(M == m->cfg.n_mics && t >= 1, "forward: bad shape");
  hipStream_t s = asw::as_stream(stream);
  GateSet* gs = nullptr;
  if ((rc = get_gates(m, window_embedding_host[0], window_embedding_host[1], &gs))) return rc;
  const int Bmax = B < m->batch ? B : m->batch;
  Plan pl;
  if ((rc = ensure_ws(m, Bmax, t, pl))) return rc;
  const int C = m->cfg.channels, pad_l = m->cfg.encoder_kernel_size / 2;
  for (int i0 = 0; i0 < B; i0 += Bmax) {
    const int b = B - i0 < Bmax ? B - i0 : Bmax;
    pl.B = b;
    ASW_HIP(hipMemsetAsync(pl.refn, 0, (size_t)b * pl.RL * sizeof(float), s));
    if ((rc = asw_pad_preproc(mix_norm + (size_t)i0 * M * t, b, M, t, pl.Tp, m->pre_w.p, m->pre_b.p, C, pl.X[0],
                              pl.refn + pad_l, pl.RL, s)))
      return rc;
    if ((rc = run_network(m, pl, gs, nullptr, nullptr, out + (size_t)i0 * t, s))) return rc;
  }
  return ASW_OK;
}

extern "C" int asw_spot_set_fused_mask(asw_spot* m, int on) {
  ASW_CHECK_ARG(m, "set_fused_mask: null model handle");
  m->fuse_mask = on != 0;
  return ASW_OK;
}

extern "C" int asw_spot_get_tap(asw_spot* m, const char* name, float* dst, size_t capacity, size_t* numel,
                                void* stream) {
  ASW_CHECK_ARG(m && name && numel, "get_tap: null pointer");
  auto it = m->taps.find(name);
  if (it == m->taps.end()) return asw::set_error(ASW_ERR_ARG, "get_tap: no activation named %s", name);
  *numel = it->second.numel;
  if (dst) {
    ASW_CHECK_ARG(capacity >= it->second.numel, "get_tap: buffer too small");
    ASW_HIP(hipMemcpyAsync(dst, it->second.p, it->second.numel * sizeof(float), hipMemcpyDeviceToDevice,
                           asw::as_stream(stream)));
  }
  return ASW_OK;
}
