// torch_ops.cpp -- PyTorch-ROCm custom ops `torch.ops.asw.*` over the C ABI of libasw_hip.so.
//
// the hot path is "exposed as PyTorch-ROCm custom ops".  Every op is a thin
// adapter: TORCH_CHECK of dtype / shape / device / contiguity, output allocation with torch's
// allocator on the input's device, then ONE call of the C-ABI entry point (include/asw_hip.h)
// on that device and on torch's CURRENT HIP stream -- no hidden synchronisation, no arithmetic
// here.  A failed C-ABI call becomes a Python RuntimeError carrying asw_last_error().
//
// Replaces (reference): DataParallelSpotModel.shift_and_sep and the nn.DataParallel forward
// (sep/training/JointModel/network.py:27-104), Network.forward
// (sep/training/SpeakerLocalization/network.py:363-405), the host energy loops
// (sep/helpers/local_utils_3d.py:13-17,349-354), si_sdr pairs (sep/helpers/eval_utils.py:11-82),
// SRP_Map_WINDOW_torch (sep/Traditional_SP/SRP_Prunning.py:387-434) and the joint separation
// network's infer_sample / forward (sep/training/SpeakerSeparation/network.py:418-548).
//
// Model handles (asw_spot*, asw_sep*) are created and loaded through the C ABI
// (asw_spot_create / set_param / finalize) and passed to the ops as int64.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>

#include <tuple>

#include "../../include/asw_hip.h"

namespace {

using at::Tensor;

void check_status(int rc, const char* what) {
  TORCH_CHECK(rc == ASW_OK, "libasw_hip: ", what, " failed with status ", rc, ": ", asw_last_error());
}

void need(const Tensor& t, const char* name, at::ScalarType dtype, int64_t dim) {
  TORCH_CHECK(t.is_cuda(), name, " must be a HIP (cuda) tensor");
  TORCH_CHECK(t.scalar_type() == dtype, name, " must be ", dtype, ", got ", t.scalar_type());
  TORCH_CHECK(t.dim() == dim, name, " must have ", dim, " dimensions, got ", t.dim());
  TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
}

void same_device(const Tensor& a, const Tensor& b, const char* what) {
  TORCH_CHECK(a.device() == b.device(), what, " must live on the same device");
}

// Device guard + torch's current stream on the tensor's device.  PyTorch-ROCm calls its HIP
// devices "cuda", so the guard / stream accessors are the "masquerading as CUDA" ones.
struct Launch {
  c10::hip::HIPGuardMasqueradingAsCUDA guard;
  void* stream;
  explicit Launch(const Tensor& t)
      : guard(t.device()), stream(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream()) {}
};

int checked_int(int64_t v, const char* name) {
  TORCH_CHECK(v >= INT32_MIN && v <= INT32_MAX, name, " out of int32 range");
  return static_cast<int>(v);
}

// ---- spot network -------------------------------------------------------------------------
std::tuple<Tensor, Tensor> spot_shift_and_sep(int64_t model, const Tensor& mix, const Tensor& offsets, int64_t strict,
                                              bool circular, bool want_wave, bool want_energy, int64_t window) {
  TORCH_CHECK(model != 0, "spot_shift_and_sep: null model handle");
  need(mix, "mix", at::kFloat, 2);
  need(offsets, "offsets", at::kInt, 2);
  same_device(mix, offsets, "mix and offsets");
  const int M = checked_int(mix.size(0), "M"), T = checked_int(mix.size(1), "T"), N = checked_int(offsets.size(0), "N");
  TORCH_CHECK(offsets.size(1) == M - 1, "offsets must be [N, M-1] = [N, ", M - 1, "], got [N, ", offsets.size(1), "]");
  Tensor wave = at::empty({want_wave ? N : 0, T}, mix.options());
  Tensor energy = at::empty({want_energy ? N : 0, 2}, mix.options().dtype(at::kDouble));
  if (N == 0) return {wave, energy};
  Launch l(mix);
  check_status(asw_spot_shift_and_sep(reinterpret_cast<asw_spot*>(model), mix.data_ptr<float>(), M, T,
                                      offsets.data_ptr<int32_t>(), N, checked_int(strict, "strict"), circular ? 1 : 0,
                                      want_wave ? wave.data_ptr<float>() : nullptr,
                                      want_energy ? energy.data_ptr<double>() : nullptr, checked_int(window, "window"),
                                      l.stream),
               "asw_spot_shift_and_sep");
  return {wave, energy};
}

// candidates of several mixtures in one call: mix [K, M, T], mix_index [N] (values in [0, K): built and checked on the host
// by the caller -- reading them back here would stall the stream)
std::tuple<Tensor, Tensor> spot_shift_and_sep_multi(int64_t model, const Tensor& mix, const Tensor& offsets,
                                                    const Tensor& mix_index, int64_t strict, bool circular, bool want_wave,
                                                    bool want_energy, int64_t window) {
  TORCH_CHECK(model != 0, "spot_shift_and_sep_multi: null model handle");
  need(mix, "mix", at::kFloat, 3);
  need(offsets, "offsets", at::kInt, 2);
  need(mix_index, "mix_index", at::kInt, 1);
  same_device(mix, offsets, "mix and offsets");
  same_device(mix, mix_index, "mix and mix_index");
  const int K = checked_int(mix.size(0), "K"), M = checked_int(mix.size(1), "M"), T = checked_int(mix.size(2), "T");
  const int N = checked_int(offsets.size(0), "N");
  TORCH_CHECK(K >= 1 && offsets.size(1) == M - 1 && mix_index.size(0) == N, "offsets must be [N, M-1] and mix_index [N]");
  Tensor wave = at::empty({want_wave ? N : 0, T}, mix.options());
  Tensor energy = at::empty({want_energy ? N : 0, 2}, mix.options().dtype(at::kDouble));
  if (N == 0) return {wave, energy};
  Launch l(mix);
  check_status(asw_spot_shift_and_sep_multi(reinterpret_cast<asw_spot*>(model), mix.data_ptr<float>(), K, M, T,
                                            offsets.data_ptr<int32_t>(), mix_index.data_ptr<int32_t>(), N,
                                            checked_int(strict, "strict"), circular ? 1 : 0,
                                            want_wave ? wave.data_ptr<float>() : nullptr,
                                            want_energy ? energy.data_ptr<double>() : nullptr, checked_int(window, "window"),
                                            l.stream),
               "asw_spot_shift_and_sep_multi");
  return {wave, energy};
}

Tensor spot_forward(int64_t model, const Tensor& mix_norm, double w0, double w1) {
  TORCH_CHECK(model != 0, "spot_forward: null model handle");
  need(mix_norm, "mix", at::kFloat, 3);
  const int B = checked_int(mix_norm.size(0), "B"), M = checked_int(mix_norm.size(1), "M"), t = checked_int(mix_norm.size(2), "t");
  Tensor out = at::empty({B, t}, mix_norm.options());
  if (B == 0) return out;
  const float w[2] = {static_cast<float>(w0), static_cast<float>(w1)};
  Launch l(mix_norm);
  check_status(asw_spot_forward(reinterpret_cast<asw_spot*>(model), mix_norm.data_ptr<float>(), B, M, t, w,
                                out.data_ptr<float>(), l.stream),
               "asw_spot_forward");
  return out;
}

std::tuple<Tensor, Tensor, Tensor, Tensor> shift_norm_preproc(const Tensor& mix, const Tensor& offsets, const Tensor& w,
                                                              const Tensor& b, int64_t T_pad, bool circular) {
  need(mix, "mix", at::kFloat, 2);
  need(offsets, "offsets", at::kInt, 2);
  need(w, "w", at::kFloat, 2);
  need(b, "b", at::kFloat, 1);
  same_device(mix, offsets, "mix and offsets");
  same_device(mix, w, "mix and w");
  same_device(mix, b, "mix and b");
  const int M = checked_int(mix.size(0), "M"), T = checked_int(mix.size(1), "T"), N = checked_int(offsets.size(0), "N");
  const int C = checked_int(w.size(0), "C"), Tp = checked_int(T_pad, "T_pad");
  TORCH_CHECK(offsets.size(1) == M - 1, "offsets must be [N, M-1]");
  TORCH_CHECK(w.size(1) == M && b.size(0) == C, "w must be [C, M] and b [C]");
  TORCH_CHECK(Tp >= T, "T_pad must be >= T");
  Tensor mean = at::empty({N}, mix.options()), stdv = at::empty({N}, mix.options());
  Tensor x0 = at::empty({N, Tp, C}, mix.options());
  Tensor refn = at::zeros({N, Tp}, mix.options());
  if (N == 0) return {x0, refn, mean, stdv};
  Launch l(mix);
  check_status(asw_shift_stats(mix.data_ptr<float>(), M, T, offsets.data_ptr<int32_t>(), N, circular ? 1 : 0,
                               mean.data_ptr<float>(), stdv.data_ptr<float>(), l.stream),
               "asw_shift_stats");
  check_status(asw_shift_norm_preproc(mix.data_ptr<float>(), M, T, Tp, offsets.data_ptr<int32_t>(), N, circular ? 1 : 0,
                                      mean.data_ptr<float>(), stdv.data_ptr<float>(), w.data_ptr<float>(), b.data_ptr<float>(), C,
                                      x0.data_ptr<float>(), refn.data_ptr<float>(), Tp, l.stream),
               "asw_shift_norm_preproc");
  return {x0, refn, mean, stdv};
}

// ---- energies / SI-SDR ----------------------------------------------------------------------
Tensor energies(const Tensor& y, int64_t window) {
  need(y, "y", at::kFloat, 2);
  const int B = checked_int(y.size(0), "B"), T = checked_int(y.size(1), "T");
  Tensor out = at::empty({B, 2}, y.options().dtype(at::kDouble));
  if (B == 0) return out;
  Tensor scratch = at::empty({B, T + 1}, y.options().dtype(at::kDouble));
  Launch l(y);
  check_status(asw_energies(y.data_ptr<float>(), B, T, checked_int(window, "window"), scratch.data_ptr<double>(),
                            out.data_ptr<double>(), l.stream),
               "asw_energies");
  return out;
}

Tensor pair_sisdr(const Tensor& y) {
  need(y, "y", at::kFloat, 2);
  const int n = checked_int(y.size(0), "n"), T = checked_int(y.size(1), "T");
  Tensor out = at::empty({n, n}, y.options().dtype(at::kDouble));
  if (n == 0) return out;
  Launch l(y);
  check_status(asw_pair_sisdr(y.data_ptr<float>(), n, T, out.data_ptr<double>(), l.stream), "asw_pair_sisdr");
  return out;
}

Tensor segment_sisdr(const Tensor& y, const Tensor& segments, const Tensor& counts) {
  need(y, "y", at::kFloat, 2);
  need(segments, "segments", at::kInt, 3);
  need(counts, "counts", at::kInt, 1);
  same_device(y, segments, "y and segments");
  same_device(y, counts, "y and counts");
  const int n = checked_int(y.size(0), "n"), T = checked_int(y.size(1), "T"), kmax = checked_int(segments.size(1), "kmax");
  TORCH_CHECK(segments.size(0) == n && segments.size(2) == 2 && counts.size(0) == n && kmax >= 1,
              "segments must be [n, kmax, 2] and counts [n]");
  Tensor out = at::full({n, n, kmax}, std::numeric_limits<double>::quiet_NaN(), y.options().dtype(at::kDouble));
  if (n == 0) return out;
  Launch l(y);
  check_status(asw_segment_sisdr(y.data_ptr<float>(), n, T, segments.data_ptr<int32_t>(), counts.data_ptr<int32_t>(), kmax,
                                 out.data_ptr<double>(), l.stream),
               "asw_segment_sisdr");
  return out;
}

Tensor center_rows_(Tensor y) {
  need(y, "y", at::kFloat, 2);
  if (y.size(0) == 0) return y;
  Launch l(y);
  check_status(asw_center_rows(y.data_ptr<float>(), checked_int(y.size(0), "B"), checked_int(y.size(1), "T"), l.stream),
               "asw_center_rows");
  return y;
}

// ---- SRP-PHAT map -------------------------------------------------------------------------------
Tensor srp_phat_map(const Tensor& mix, const Tensor& twiddle, const Tensor& pair_i, const Tensor& pair_j, const Tensor& tau,
                    const Tensor& omega, int64_t window, int64_t step, int64_t n_windows, int64_t nfft, int64_t hop,
                    double tol) {
  need(mix, "mix", at::kFloat, 2);
  need(twiddle, "twiddle", at::kFloat, 2);
  need(pair_i, "pair_i", at::kInt, 1);
  need(pair_j, "pair_j", at::kInt, 1);
  need(tau, "tau", at::kDouble, 2);
  need(omega, "omega", at::kDouble, 1);
  for (const Tensor* t : {&twiddle, &pair_i, &pair_j, &tau, &omega}) same_device(mix, *t, "all SRP-PHAT operands");
  const int M = checked_int(mix.size(0), "M"), T = checked_int(mix.size(1), "T");
  const int P = checked_int(pair_i.size(0), "P"), G = checked_int(tau.size(0), "G"), nbins = checked_int(omega.size(0), "nbins");
  const int nb_pad = checked_int(twiddle.size(0) / 2, "nb_pad"), nw = checked_int(n_windows, "n_windows");
  TORCH_CHECK(T % 4 == 0, "mix length must be a multiple of 4 (pad with zeros)");
  TORCH_CHECK(twiddle.size(1) == nfft && twiddle.size(0) == 2 * (int64_t)nb_pad && nb_pad >= nbins, "twiddle must be [2*nb_pad, nfft]");
  TORCH_CHECK(pair_j.size(0) == P && tau.size(1) == M, "pair_j must be [P], tau [G, M]");
  Tensor out = at::zeros({G}, mix.options());
  if (nw <= 0 || G == 0) return out;
  TORCH_CHECK((nw - 1) * step + window <= T, "windows run past the end of the mixture");
  Launch l(mix);
  const int F = asw_srp_frames(checked_int(window, "window"), checked_int(nfft, "nfft"), checked_int(hop, "hop"));
  Tensor xf = at::empty({M, F, 2 * nb_pad}, mix.options());
  Tensor cc = at::empty({nw, nbins, P, 2}, mix.options());
  Tensor part = at::empty({8 * 8 * (int64_t)G}, mix.options());
  check_status(asw_srp_cross_spectra(mix.data_ptr<float>(), M, T, (int)window, (int)step, nw, (int)nfft, (int)hop, nbins, nb_pad,
                                     static_cast<float>(tol), twiddle.data_ptr<float>(), pair_i.data_ptr<int32_t>(),
                                     pair_j.data_ptr<int32_t>(), P, xf.data_ptr<float>(), cc.data_ptr<float>(), l.stream),
               "asw_srp_cross_spectra");
  check_status(asw_srp_map(cc.data_ptr<float>(), nw, nbins, P, tau.data_ptr<double>(), G, M, omega.data_ptr<double>(),
                           pair_i.data_ptr<int32_t>(), pair_j.data_ptr<int32_t>(), part.data_ptr<float>(), out.data_ptr<float>(),
                           l.stream),
               "asw_srp_map");
  return out;
}

// ---- joint separation network ---------------------------------------------------------------------
Tensor sep_infer(int64_t model, const Tensor& mix, const Tensor& offsets) {
  TORCH_CHECK(model != 0, "sep_infer: null model handle");
  need(mix, "mix", at::kFloat, 2);
  need(offsets, "offsets", at::kInt, 2);
  same_device(mix, offsets, "mix and offsets");
  const int M = checked_int(mix.size(0), "M"), T = checked_int(mix.size(1), "T"), S = checked_int(offsets.size(0), "S");
  TORCH_CHECK(offsets.size(1) == M - 1, "offsets must be [S, M-1]");
  Tensor out = at::empty({S, T}, mix.options());
  if (S == 0) return out;
  Launch l(mix);
  check_status(asw_sep_infer(reinterpret_cast<asw_sep*>(model), mix.data_ptr<float>(), M, T, offsets.data_ptr<int32_t>(), S,
                             out.data_ptr<float>(), l.stream),
               "asw_sep_infer");
  return out;
}

Tensor sep_forward(int64_t model, const Tensor& mix_norm, int64_t n_speakers, int64_t n_mics, int64_t max_speakers) {
  TORCH_CHECK(model != 0, "sep_forward: null model handle");
  need(mix_norm, "mix", at::kFloat, 3);
  const int B = checked_int(mix_norm.size(0), "B"), t = checked_int(mix_norm.size(2), "t");
  const int S = checked_int(n_speakers, "S"), M = checked_int(n_mics, "M");
  TORCH_CHECK(S >= 1 && mix_norm.size(1) == (int64_t)S * M, "mix must be [B, S*M, t]");
  // the library pads the rows of `out` to the HANDLE's max_speakers: the caller's view of the model must agree,
  // or the copy would run past the tensor allocated here
  asw_sep_config cfg;
  check_status(asw_sep_get_config(reinterpret_cast<asw_sep*>(model), &cfg), "asw_sep_get_config");
  TORCH_CHECK(cfg.n_mics == M, "sep_forward: n_mics=", M, " but the model was built for ", cfg.n_mics);
  TORCH_CHECK(cfg.max_speakers == max_speakers, "sep_forward: max_speakers=", max_speakers, " but the model was built for ",
              cfg.max_speakers);
  const int64_t R = S > cfg.max_speakers ? S : cfg.max_speakers;
  Tensor out = at::empty({B, R, t}, mix_norm.options());
  if (B == 0) return out;
  Launch l(mix_norm);
  check_status(asw_sep_forward(reinterpret_cast<asw_sep*>(model), mix_norm.data_ptr<float>(), B, S, M, t, out.data_ptr<float>(),
                               l.stream),
               "asw_sep_forward");
  return out;
}

// Network.forward with a speaker count per item: mix [B, S*M, t] with S = max(counts)
Tensor sep_forward_counts(int64_t model, const Tensor& mix_norm, at::IntArrayRef counts, int64_t n_mics, int64_t max_speakers) {
  TORCH_CHECK(model != 0, "sep_forward_counts: null model handle");
  need(mix_norm, "mix", at::kFloat, 3);
  const int B = checked_int(mix_norm.size(0), "B"), t = checked_int(mix_norm.size(2), "t"), M = checked_int(n_mics, "M");
  TORCH_CHECK((int64_t)counts.size() == B && B >= 1, "counts must hold one entry per batch item");
  std::vector<int32_t> c(B);
  int S = 0;
  for (int b = 0; b < B; ++b) {
    c[b] = checked_int(counts[b], "count");
    S = c[b] > S ? c[b] : S;
  }
  TORCH_CHECK(S >= 1 && mix_norm.size(1) == (int64_t)S * M, "mix must be [B, max(counts)*M, t]");
  asw_sep_config cfg;
  check_status(asw_sep_get_config(reinterpret_cast<asw_sep*>(model), &cfg), "asw_sep_get_config");
  TORCH_CHECK(cfg.n_mics == M && cfg.max_speakers == max_speakers, "sep_forward_counts: n_mics / max_speakers differ from the model's");
  const int64_t R = S > cfg.max_speakers ? S : cfg.max_speakers;
  Tensor out = at::empty({B, R, t}, mix_norm.options());
  Launch l(mix_norm);
  check_status(asw_sep_forward_counts(reinterpret_cast<asw_sep*>(model), mix_norm.data_ptr<float>(), B, S, M, t, c.data(),
                                      out.data_ptr<float>(), l.stream),
               "asw_sep_forward_counts");
  return out;
}

}  // namespace

TORCH_LIBRARY(asw, m) {
  m.def("spot_shift_and_sep(int model, Tensor mix, Tensor offsets, int strict, bool circular, bool want_wave, "
        "bool want_energy, int window) -> (Tensor, Tensor)");
  m.def("spot_shift_and_sep_multi(int model, Tensor mix, Tensor offsets, Tensor mix_index, int strict, bool circular, "
        "bool want_wave, bool want_energy, int window) -> (Tensor, Tensor)");
  m.def("spot_forward(int model, Tensor mix, float w0, float w1) -> Tensor");
  m.def("shift_norm_preproc(Tensor mix, Tensor offsets, Tensor w, Tensor b, int T_pad, bool circular) -> "
        "(Tensor, Tensor, Tensor, Tensor)");
  m.def("energies(Tensor y, int window) -> Tensor");
  m.def("pair_sisdr(Tensor y) -> Tensor");
  m.def("segment_sisdr(Tensor y, Tensor segments, Tensor counts) -> Tensor");
  m.def("center_rows_(Tensor(a!) y) -> Tensor(a!)");
  m.def("srp_phat_map(Tensor mix, Tensor twiddle, Tensor pair_i, Tensor pair_j, Tensor tau, Tensor omega, int window, "
        "int step, int n_windows, int nfft, int hop, float tol) -> Tensor");
  m.def("sep_infer(int model, Tensor mix, Tensor offsets) -> Tensor");
  m.def("sep_forward(int model, Tensor mix, int n_speakers, int n_mics, int max_speakers) -> Tensor");
  m.def("sep_forward_counts(int model, Tensor mix, int[] counts, int n_mics, int max_speakers) -> Tensor");
}

TORCH_LIBRARY_IMPL(asw, CUDA, m) {
  m.impl("spot_shift_and_sep", &spot_shift_and_sep);
  m.impl("spot_shift_and_sep_multi", &spot_shift_and_sep_multi);
  m.impl("spot_forward", &spot_forward);
  m.impl("shift_norm_preproc", &shift_norm_preproc);
  m.impl("energies", &energies);
  m.impl("pair_sisdr", &pair_sisdr);
  m.impl("segment_sisdr", &segment_sisdr);
  m.impl("center_rows_", &center_rows_);
  m.impl("srp_phat_map", &srp_phat_map);
  m.impl("sep_infer", &sep_infer);
  m.impl("sep_forward", &sep_forward);
  m.impl("sep_forward_counts", &sep_forward_counts);
}
