/*
 * asw_hip.h -- C ABI of libasw_hip.so: the MI355X (gfx950) implementation of the
 * Spotforming localization-by-separation hot path.
 *
 * Every entry point is `extern "C"`, takes plain pointers and sizes (device
 * pointers unless a parameter says "host"), enqueues its work on the given HIP
 * stream (hipStream_t passed as void*; NULL = default stream) and returns 0 or a
 * negative asw_status.  No exception crosses the ABI; asw_last_error() returns a
 * thread-local message for the last failure.  Inputs are borrowed and never
 * mutated; outputs are caller-owned buffers.
 *
 * Each function cites the reference interface (file:line under the upstream
 * repo) it replaces.  INTEGRATION.md shows the ctypes binding a maintainer of
 * the reference would add.
 */
#ifndef ASW_HIP_H
#define ASW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum asw_status {
  ASW_OK = 0,
  ASW_ERR_ARG = -1,       /* bad shape / null pointer / unsupported configuration */
  ASW_ERR_HIP = -2,       /* a HIP runtime call failed                             */
  ASW_ERR_NOMEM = -3,     /* device allocation failed                              */
  ASW_ERR_STATE = -4      /* handle used before weights were loaded                */
} asw_status;

const char* asw_last_error(void);
int asw_abi_version(void);

/* Optional launch profiler: while enabled, every GEMM-class launch is bracketed by HIP
 * events on its own stream.  asw_profile_enable(on) also clears earlier records;
 * asw_profile_report() waits for the recorded events and writes a JSON object
 * {"kernel<tile>": {"launches": n, "ms": total, "work": flops}, ...} into buf. */
int asw_profile_enable(int on);
int asw_profile_report(char* buf, size_t cap);

/* ------------------------------------------------------------------------
 * Spot-network hyper-parameters.  Mirrors Network.__init__
 * (sep/training/SpeakerLocalization/network.py:268-292).
 * ---------------------------------------------------------------------- */
typedef struct asw_spot_config {
  int32_t n_mics;                 /* 7 */
  int32_t kernel_size;            /* 7 */
  int32_t depth;                  /* len(stride_list) <= 8 */
  int32_t stride_list[8];         /* 2,2,4,4,4 */
  int32_t channels;               /* 64 */
  int32_t growth;                 /* 2 */
  int32_t encoder_channels;       /* 2048 */
  int32_t encoder_kernel_size;    /* 33 */
  int32_t encoder_stride;         /* 16 */
  int32_t residual_layers;        /* 3 */
  int32_t residual_dilation_factor; /* 7 */
  int32_t num_head;               /* 8 */
  int32_t ffw_dim;                /* 1024 */
  int32_t num_transformer_layers; /* 2 */
} asw_spot_config;

typedef struct asw_spot asw_spot;   /* opaque: device-resident weights + workspace */

/* Create a model for `cfg` on the current HIP device.  Replaces Network(**model_params)
 * + .to(device) (sep/helpers/utils.py:176-183, sep/training/base_network.py:41-45). */
int asw_spot_create(const asw_spot_config* cfg, asw_spot** out);
void asw_spot_destroy(asw_spot* m);

/* Upload one tensor of a reference-format state dict (host float32, contiguous, in the
 * reference's own layout and key name, SURVEY.md §8 a-N).  Replaces
 * model.load_state_dict(..., strict=True) (sep/helpers/utils.py:196-198).
 * asw_spot_finalize() checks that every key arrived (strict) and packs the weights
 * into the kernels' layouts. */
int asw_spot_set_param(asw_spot* m, const char* key, const float* host_data, size_t numel);
int asw_spot_finalize(asw_spot* m);

/* Maximum number of candidates processed per internal batch (the reference's
 * spot_batch_size, sep/training/JointModel/network.py:28,75). */
int asw_spot_set_batch(asw_spot* m, int batch);

/* Number of execution lanes (1 or 2, default 1).  With 2, consecutive internal batches of one
 * asw_spot_shift_and_sep call run on two HIP streams (the caller's stream and a library-owned
 * side stream, forked and joined with events inside the call) with one workspace each, so the
 * memory-bound passes and the launch tails of one batch overlap the MFMA kernels of the other.
 * Results are identical; the call still only depends on, and is ordered by, the caller's stream. */
int asw_spot_set_lanes(asw_spot* m, int lanes);

/* Arithmetic of the GEMM-class layers: 0 = exact fp32 MFMA (default), 1 = "f16x3"
 * split-operand half MFMA with fp32 accumulation, 2 = optional single-pass f16 (reduced
 * precision; see asw_convgemm_args.precision). */
int asw_spot_set_precision(asw_spot* m, int precision);

/* The hot loop: replaces DataParallelSpotModel.shift_and_sep
 * (sep/training/JointModel/network.py:37-104): for each of the N candidates,
 * circularly advance channel m>=1 of `mix` by offsets[n][m-1] samples, int16-quantise
 * and normalise (network.py:28-40), run the spot network with the window one-hot
 * selected by `strict` (1 -> [1,0], else [0,1]), un-normalise (network.py:42-47).
 *   mix      [M][T] float32 device
 *   offsets  [N][M-1] int32 device (already rounded, JointModel/network.py:81-82)
 *   out_wave [N][T] float32 device, or NULL
 *   out_energy [N][2] float64 device, or NULL: (power, power2) of the mean-removed
 *              output as computed by the stage loops (sep/helpers/local_utils_3d.py:13-17,
 *              349-354; sep/Mic_Array.py:290-295) with window `energy_window` samples. */
int asw_spot_shift_and_sep(asw_spot* m, const float* mix, int M, int T,
                           const int32_t* offsets, int N, int strict, int circular,
                           float* out_wave, double* out_energy, int energy_window,
                           void* stream);

/* The same call over the candidates of SEVERAL mixtures in one stream (the reference fills its 128-wide
 * batches from one mixture at a time, sep/training/JointModel/network.py:75-96; a batch of mixtures --
 * BASELINE configs[3] -- keeps the internal batches full when its searches are interleaved):
 *   mix [K][M][T] float32 device, mix_index [N] int32 device (candidate n reads mixture mix_index[n], values in
 *   [0, K); may be NULL when K == 1).  The mix_index values are NOT range-checked on the device: the caller
 *   guarantees them.  Everything else as asw_spot_shift_and_sep; each candidate's result is the one it has in a
 *   single-mixture call of the same internal batch size. */
int asw_spot_shift_and_sep_multi(asw_spot* m, const float* mix, int K, int M, int T, const int32_t* offsets,
                                 const int32_t* mix_index, int N, int strict, int circular, float* out_wave,
                                 double* out_energy, int energy_window, void* stream);

/* Network.forward (sep/training/SpeakerLocalization/network.py:363-405) on already
 * normalised input: mix [B][M][t], window_embedding host [2] shared by the batch ->
 * out [B][t]. */
int asw_spot_forward(asw_spot* m, const float* mix_norm, int B, int M, int t,
                     const float* window_embedding_host, float* out, void* stream);

/* f16x3 mode runs the mask path (reference_bypass, mask_encoder, product, output_decoder taps) as ONE
 * launch (asw_mask_path_f16x3) when the shapes fit (encoder_channels % 256 == 0, channels % 32 == 0,
 * kernel <= 48): the latent is then never written and the "latent" tap does not exist; and the
 * 64-channel decoder blocks apply GroupNorm + GLU while their first residual layer loads its rows
 * (asw_convgemm_args.glu_raw).  on = 0 selects the three-GEMM mask path and the separate GroupNorm + GLU
 * pass everywhere (default: on). */
int asw_spot_set_fused_mask(asw_spot* m, int on);

/* Debug/parity tap: copy an intermediate activation of the LAST forward to `dst`
 * (channels-last [B][T_l][C] float32).  names: "preproc", "enc0".., "bottleneck",
 * "dec0".., "latent" (three-GEMM mask path only).  Returns the element count through *numel (dst may be NULL). */
int asw_spot_get_tap(asw_spot* m, const char* name, float* dst, size_t capacity, size_t* numel,
                     void* stream);

/* ------------------------------------------------------------------------
 * Joint separation network ("separation by localization").  Mirrors Network.__init__
 * (sep/training/SpeakerSeparation/network.py:323-416; experiments/separation/
 * description.json:4-10).  The bottleneck's Conformer follows the published speechbrain
 * definitions (the library is absent from the build image: parity of that part is pinned to
 * this repository's restatement only, see oracle/sep_ref.py).
 * ---------------------------------------------------------------------- */
typedef struct asw_sep_config {
  int32_t n_mics;                 /* 7 */
  int32_t max_speakers;           /* 5 (constructor default 6): forward() pads its rows to this */
  int32_t kernel_size;            /* 5 */
  int32_t depth;                  /* len(stride_list) <= 8 */
  int32_t stride_list[8];         /* 2,2,4,4 */
  int32_t channels;               /* 64 */
  int32_t growth;                 /* 2 */
  int32_t encoder_channels;       /* 4096 */
  int32_t encoder_kernel_size;    /* 33 */
  int32_t encoder_stride;         /* 16 */
  int32_t residual_layers;        /* 3 */
  int32_t residual_dilation_factor; /* 2 */
  int32_t num_head;               /* 8 */
  int32_t ffw_dim;                /* 1024 */
  int32_t bottleneck_layers;      /* 3 */
  int32_t bottleneck_ksize;       /* 31 */
} asw_sep_config;

typedef struct asw_sep asw_sep;     /* opaque: device-resident weights + workspace */

/* Network(**model_params).to(device) + load_state_dict(strict=True)
 * (sep/helpers/utils.py:176-198); same protocol as the asw_spot_* calls above. */
int asw_sep_create(const asw_sep_config* cfg, asw_sep** out);
void asw_sep_destroy(asw_sep* m);
int asw_sep_set_param(asw_sep* m, const char* key, const float* host_data, size_t numel);
int asw_sep_finalize(asw_sep* m);
int asw_sep_set_precision(asw_sep* m, int precision);

/* Network.infer_sample (SpeakerSeparation/network.py:496-548): for each of the S speakers
 * advance channel m>=1 of `mix` by offsets[s][m-1] samples with ZERO fill (:510-522), stack to
 * S*M channels, int16-quantise and normalise with ONE mean / std over all of them (:534),
 * run the network and un-normalise.
 *   mix [M][T] float32 device; offsets [S][M-1] int32 device (already rounded, :507);
 *   out [S][T] float32 device.  S <= 64. */
int asw_sep_infer(asw_sep* m, const float* mix, int M, int T, const int32_t* offsets, int S, float* out,
                  void* stream);

/* Network.forward (:418-490) on already normalised input, every item with the same number of
 * speakers: mix_norm [B][S*M][t] -> out [B][max(S, max_speakers)][t] (rows beyond S are zeros,
 * :486-488).  B*S <= 64. */
int asw_sep_forward(asw_sep* m, const float* mix_norm, int B, int S, int M, int t, float* out, void* stream);
/* Network.forward with DIFFERENT speaker counts per item (speakers_to_batches / batches_to_speakers, :236-268):
 * mix_norm [B][S*M][t] with S = the largest count; counts host int32 [B], 1 <= counts[b] <= S (NULL: S everywhere).
 * A missing speaker enters every inter-speaker layer as a zero sequence and leaves as the bare output_decoder bias; the
 * channels of its block in mix_norm are ignored. */
int asw_sep_forward_counts(asw_sep* m, const float* mix_norm, int B, int S, int M, int t, const int32_t* counts,
                           float* out, void* stream);
/* The hyper-parameters the handle was created with (a caller that sizes `out` of asw_sep_forward from
 * max_speakers must use the handle's value, not its own). */
int asw_sep_get_config(const asw_sep* m, asw_sep_config* out);

/* Debug/parity tap of the LAST call (channels-last [B*S][T_l][C] float32): "enc0".., "intra0"..,
 * "inter0".., "bottleneck", "dec0".. */
int asw_sep_get_tap(asw_sep* m, const char* name, float* dst, size_t capacity, size_t* numel, void* stream);

/* Joint normalisation statistics of the S*M zero-fill-shifted, int16-quantised channels
 * (SpeakerSeparation/network.py:510-534 with :28-35): mean / unbiased std over time of the
 * all-channel average.  scratch: asw_joint_shift_stats_scratch_doubles() doubles (device).
 * mean, std: [S] float32 (the same value S times, the layout the preproc kernel takes). */
int asw_joint_shift_stats(const float* mix, int M, int T, const int32_t* offsets, int S, double* scratch,
                          float* mean, float* std, void* stream);
int asw_joint_shift_stats_scratch_doubles(void);

/* Row kernel of the Conformer layer: s = x + alpha*y (y may be NULL); sum_out = s (may be NULL);
 * ln_out = act(LayerNorm(s)*gamma + beta) (may be NULL; act 0 none, 2 Swish).  rows x N floats,
 * N % 4 == 0, N <= 2048.  Replaces the residual adds and LayerNorms of ConformerEncoderLayer. */
int asw_add_layernorm2(const float* x, const float* y, float alpha, const float* gamma, const float* beta,
                       int rows, int N, float eps, int act, float* sum_out, float* ln_out, void* stream);

/* nn.GLU over channels-last rows: raw [rows][2C] -> out [rows][C] = raw[:, :C] * sigmoid(raw[:, C:]). */
int asw_glu_rows(const float* raw, long rows, int C, float* out, void* stream);

/* ConvolutionModule tail: depthwise Conv1d(d, d, K, padding (K-1)/2, groups d) over time within
 * each of the BS sequences + bias, LayerNorm over channels, Swish.  u, out [BS][L][d];
 * wT [K][d] (tap-major copy of the [d][1][K] weight). */
int asw_dwconv_ln_swish(const float* u, const float* wT, const float* bias, const float* gamma, const float* beta,
                        int BS, int L, int d, int K, float eps, float* out, void* stream);

/* Relative-position multi-head self-attention core (speechbrain RelPosMHAXL as published):
 * score[i][j] = scale*((q_i+u).k_j + (q_i+v).P[(L-1)+j-i]); ctx = softmax(score) V.
 * qkv [BS][L][3d] in Q|K|V layout (head h at columns h*hd), P [2L-1][d] = linear_pos(table),
 * bias_u / bias_v [d] head-major; ctx [BS][L][d].  head_dim in {16, 32, 64}. */
int asw_relpos_attention(const float* qkv, const float* P, const float* bias_u, const float* bias_v, int BS, int L,
                         int d, int nhead, float scale, float* ctx, void* stream);

/* Inter-speaker attention core: for every (item, time step, head) softmax(QK^T/sqrt(hd))V over the
 * S speakers (nn.TransformerEncoderLayer on x.reshape(N*T, S, F), :311-316).
 * qkv [NB][S][L][3d] (in_proj bias included) -> ctx [NB][S][L][d].  S <= 64, head_dim <= 64. */
int asw_inter_attention(const float* qkv, int NB, int S, int L, int d, int nhead, float* ctx, void* stream);

/* ------------------------------------------------------------------------
 * Individual kernels (unit-testable; the model above is built from these).
 * ---------------------------------------------------------------------- */

/* Shift + quantise + per-candidate mean / unbiased std of the mic-average
 * (JointModel/network.py:80-83 + network.py:28-35).  mean,std: [N] float32. */
int asw_shift_stats(const float* mix, int M, int T, const int32_t* offsets, int N,
                    int circular, float* mean, float* std, void* stream);

/* Shift + quantise + normalise + left zero-pad to T_pad + 1x1 preproc conv
 * (network.py:36-38,377-378,385).  w [C][M], b [C];
 * x0 [N][T_pad][C] channels-last, refn [N][refn_stride] (normalised padded mic 0; the
 * first T_pad entries of each row are written). */
int asw_shift_norm_preproc(const float* mix, int M, int T, int T_pad, const int32_t* offsets,
                           int N, int circular, const float* mean, const float* std,
                           const float* w, const float* b, int C, float* x0, float* refn,
                           long refn_stride, void* stream);

/* The two calls above over candidates of several mixtures: mix [K][M][T], candidate n reads mixture
 * mix_index[n] (int32 device [N], values in [0, K), not range-checked; NULL = mixture 0 for every candidate). */
int asw_shift_stats_multi(const float* mix, int M, int T, const int32_t* offsets, const int32_t* mix_index, int N,
                          int circular, float* mean, float* std, void* stream);
int asw_shift_norm_preproc_multi(const float* mix, int M, int T, int T_pad, const int32_t* offsets,
                                 const int32_t* mix_index, int N, int circular, const float* mean, const float* std,
                                 const float* w, const float* b, int C, float* x0, float* refn, long refn_stride,
                                 void* stream);

/* Normalised input variant used by asw_spot_forward: x [B][M][t] -> x0, refn. */
int asw_pad_preproc(const float* x, int B, int M, int t, int T_pad, const float* w,
                    const float* b, int C, float* x0, float* refn, long refn_stride,
                    void* stream);

/* 1-D convolution / transposed convolution / linear layer as an implicit GEMM on the
 * f32 MFMA pipe with a fused epilogue.  Activations are channels-last.
 *   out[b][r][n] = epi( sum_{tap,c} A[b][(r*stride + tap*dil - pad)*a_row_stride + c]
 *                                    * Wt[n][tap*Cin + c] )
 * epi: +bias[n]; activation (relu: 0 none, 1 ReLU, 2 Swish x*sigmoid(x)); +resid; *mul; LayerNorm over n (ln_gamma!=NULL,
 * requires N in {64,128,256,512,1024}); group statistics partials (stats!=NULL).
 * Replaces nn.Conv1d / nn.ConvTranspose1d / nn.Linear + ReLU / residual / LayerNorm of
 * network.py:57-68,105-113,190-198 and the transformer linears.
 * Zero-initialise the block (memset / = {}) before filling it: every optional pointer is tested against NULL,
 * and the struct only ever grows at its end. */
typedef struct asw_convgemm_args {
  const float* A;         /* [B][a_batch_stride] */
  const float* A2;        /* optional tensor added to A while loading (skip connection) */
  const float* Wt;        /* [N][taps*Cin] */
  const float* bias;      /* [N] or NULL */
  const float* resid;     /* [B][M_out][N] or NULL */
  const float* mul;       /* [B][M_out][N] or NULL */
  const float* ln_gamma;  /* [N] or NULL */
  const float* ln_beta;   /* [N] */
  float* out;             /* [B][M_out][N] */
  float* stats;           /* [B][tiles_m*tiles_n][4] or NULL: (sum0,sumsq0,sum1,sumsq1) */
  int32_t B, M_out, N, Cin, taps, stride, dil, pad;
  int32_t a_row_stride;   /* floats between consecutive input rows (normally Cin) */
  int64_t a_batch_stride; /* floats between batch items of A */
  int64_t a_len;          /* valid floats per batch item (bounds for zero padding) */
  int32_t chan_mod;       /* stats: group = ((n % chan_mod) >= chan_mod/2) */
  int32_t relu;
  float ln_eps;
  /* precision 0: Wt is fp32 and the products run on the exact f32 MFMA.
   * precision 1 ("f16x3"): every fp32 operand x is split into two halves
   *   hi = fp16(x), lo = fp16(x - hi) and the product is hi*hi + hi*lo + lo*hi on the f16
   *   MFMA with fp32 accumulation (operands good to ~2^-21, 5.3x the f32 MFMA rate).
   *   Weights arrive pre-split: Wt_hi / Wt_lo are fp16 [N][taps*Cin] of (w * 2^w_shift);
   *   activations are split on the fly (saturated at +-65504).
   * precision 2 ("f16", optional, reduced precision): the same kernels with ONE MFMA per product, hi * hi on
   *   round-to-nearest halves (same weight arrays; the lo halves are ignored): ~2e-4 per layer, 47-48 dB
   *   end to end against the reference, 1.5x the f16x3 throughput.  Never the default. */
  int32_t precision;
  int32_t w_shift;
  const void* Wt_hi;
  const void* Wt_lo;
  /* Optional (precision 1): the same split weights in MFMA-fragment order (see
   * asw_pack_fragments_f16).  When given and the layer is a stride-1 "same" convolution
   * with C_in == N <= 512, a residual that is the input itself and a LayerNorm epilogue
   * (the reference's DilatedResidualLayer), the halo-staged kernel is used: each input row
   * is fetched and split once per workgroup instead of once per tap. */
  const void* Wf_hi;
  const void* Wf_lo;
  int32_t stats_stride;   /* set by the library: partial-statistics slots per batch item */
  /* Optional (precision >= 1, the halo-staged residual layer with C_in == N in {64, 128, 256, 512}, dilation 1,
   * i.e. the first layer of a block's residual stack): take the layer's input -- and residual -- from the
   * un-normalised output of the preceding transposed convolution instead of A, applying GroupNorm(2) + GLU
   * while the rows are staged (the arithmetic of asw_gn_glu, bit for bit), so that tensor is neither written
   * nor read back.  glu_raw [B][M_out][2N] (value half | gate half of every output row), glu_mr [B][4] =
   * (mean0, rstd0, mean1, rstd1) from asw_gn_finalize, glu_gamma / glu_beta [2N].  A is ignored. */
  const float* glu_raw;
  const float* glu_mr;
  const float* glu_gamma;
  const float* glu_beta;
  /* [B][M_out][N]: receives GLU(GroupNorm(glu_raw)), the layer's input, as a tensor (an encoder block's skip
   * connection).  Optional at N == 64; required at N = 128 / 256 / 512, where the layer also reads its residual
   * from it (each workgroup the rows it wrote itself). */
  float* glu_out;
} asw_convgemm_args;
/* Host helper: fp32 Wt[N][K] -> fragment-major fp16 hi/lo [K/16][N/32][64 lanes][8]:
 * lane l of fragment (ks, nt) holds Wt[nt*32 + (l&31)][ks*16 + 8*(l>>5) + j], j < 8, i.e.
 * exactly the B operand of v_mfma_f32_32x32x16_f16, so a wave fetches a fragment with one
 * coalesced 1 KiB load.  Pre-scaled by 2^w_shift like asw_split_weights_f16.  N % 32 == 0,
 * K % 16 == 0; hi/lo: N*K uint16 each (host). */
int asw_pack_fragments_f16(const float* Wt, int N, int K, uint16_t* hi, uint16_t* lo, int32_t* w_shift);
int asw_convgemm_f32(const asw_convgemm_args* args, void* stream);
/* The mask path of the spot / separation network in ONE launch (f16x3 arithmetic), replacing
 * reference_bypass -> ReLU, mask_encoder -> ReLU, their product and the output_decoder tap products
 * (SpeakerLocalization/network.py:327-349,397-405; SpeakerSeparation/network.py:385-416) without the
 * E-channel latents ever being written:
 *   taps[c][b][f][j] = sum_{e in column tile c} relu(enc(x)[b][f][e] + bias_e) * relu(byp(ref)[b][f][e] + byp_bias_e) * D[e][j]
 * `enc` describes the mask_encoder GEMM exactly as for asw_convgemm_f32 (A, Wf_hi / Wf_lo fragment-order
 * weights, w_shift, bias, B, M_out = frames, N = E, Cin, taps, stride, pad, a_*; out / mul / stats unused;
 * N % 256 == 0, Cin % 32 == 0).  Frame f of item b reads the bypass window
 * ref[b*ref_batch_stride + f*ref_hop + k], k < byp_k (zero beyond ref_len); byp_hi / byp_lo: bypass weights
 * [E][byp_k = 48] (taps beyond byp_taps zero) packed by asw_pack_fragments_f16 (shift byp_shift); dec_hi /
 * dec_lo: decoder weights Wt[64][E] (row j = tap j, rows >= dec_taps zero) packed likewise (dec_shift).
 * Output: N/256 partial tap tensors [N/256][B][M_out][64] (columns < dec_taps written) to be summed by
 * asw_overlap_add_parts. */
typedef struct asw_maskpath_args {
  asw_convgemm_args enc;
  const float* ref;
  int64_t ref_batch_stride;
  int64_t ref_len;
  int32_t ref_hop;
  int32_t byp_k;          /* padded bypass kernel length (48) */
  int32_t byp_taps;       /* true bypass kernel length (33), for the FLOP count only */
  int32_t byp_shift;
  const void* byp_hi;
  const void* byp_lo;
  const float* byp_bias;  /* [E] or NULL */
  const void* dec_hi;
  const void* dec_lo;
  int32_t dec_shift;
  int32_t dec_taps;       /* 33 */
  float* taps;            /* [N/256][B][M_out][64] */
} asw_maskpath_args;
int asw_mask_path_f16x3(const asw_maskpath_args* args, void* stream);
/* A stack of 1..3 consecutive 64-channel DilatedResidualLayers (DilatedResidualSequence,
 * sep/training/SpeakerLocalization/network.py:50-82; the separation network uses the same classes) in
 * ONE launch, f16x3 arithmetic: out_i = LayerNorm(ReLU(conv_{dil_i}(x_i) + bias_i) + x_i), x_{i+1} = out_i.
 * The workgroup stages the rows the whole stack needs once; the intermediate tensors live in LDS only
 * (halo recomputation: a tile of 256 rows of layer 0 yields 256 - 2 * sum_{i>0} dil_i (taps-1)/2 finished
 * rows, so the later layers' dilations must be small: the call fails when fewer than 128 would be left).
 * A single layer of dilation >= 7 runs on polyphase row sets.  Results agree with n_layers calls of
 * asw_convgemm_f32 to fp32 rounding (the residual is taken from the fp16 hi + lo image: 2^-22 relative).
 *   x / out [B][T][64] float32 (out must not alias x); layer[i]: weights Wt[64][taps*64] in fragment order
 *   (asw_pack_fragments_f16), their shift, conv bias, LayerNorm affine.  glu_raw (optional, instead of x):
 *   as asw_convgemm_args.glu_raw -- GroupNorm(2) + GLU applied while layer 0 stages its rows. */
typedef struct asw_reslayer_desc {
  const void* Wf_hi;
  const void* Wf_lo;
  const float* bias;
  const float* ln_gamma;
  const float* ln_beta;
  int32_t dil;
  int32_t w_shift;
} asw_reslayer_desc;
typedef struct asw_resstack_args {
  const float* x;
  float* out;
  int32_t B, T, C, taps, n_layers;
  int32_t precision;      /* 1 = f16x3, 2 = single-pass f16 */
  float ln_eps;
  asw_reslayer_desc layer[3];
  const float* glu_raw;
  const float* glu_mr;
  const float* glu_gamma;
  const float* glu_beta;
  float* glu_out;         /* optional with glu_raw: [B][T][64], receives GLU(GroupNorm(glu_raw)) -- the stack's input --
                             for a caller that needs it as a tensor as well (the encoder's skip connection) */
} asw_resstack_args;
int asw_resstack64_f16x3(const asw_resstack_args* args, void* stream);
/* Host helper: split n fp32 weights into the fp16 hi/lo pair used by precision 1 with the
 * power-of-two pre-scale that keeps the lo parts out of the fp16 subnormal range; returns
 * the shift through *w_shift.  hi/lo: n uint16 each (host). */
int asw_split_weights_f16(const float* w, size_t n, uint16_t* hi, uint16_t* lo, int32_t* w_shift);
/* f16x3 range guard: activations are split to fp16 halves (saturating at +-65504) when a GEMM
 * stages them.  Normalised tensors are bounded; the un-normalised ones (outputs of epilogues
 * without LayerNorm / GroupNorm statistics: masked latent, feed-forward intermediate, attention
 * projections) are checked where they are produced.  Returns through *count how many threads
 * of f16x3 launches on the current device wrote a value beyond the fp16 range (or a NaN) since
 * the last reset; waits for the device.  Non-zero means a later GEMM clipped its input: rerun in
 * precision 0.  The device must be the one the launches ran on. */
int asw_f16x3_overflow_count(int reset, uint32_t* count);
/* Number of stats partials per batch item the call above will write. */
int asw_convgemm_stats_tiles(int M_out, int N);

/* GroupNorm(2 groups) + GLU over channels-last raw [B][T][2C] using the partial
 * statistics written by asw_convgemm_f32 (network.py:107-113,194-197). */
int asw_gn_glu(const float* raw, const float* stats, int n_partials, const float* gamma,
               const float* beta, int B, int T, int C, float eps, float* out, void* stream);
/* The statistics half of asw_gn_glu alone: reduces the same partial sums the same way and writes
 * mr [B][4] = (mean0, rstd0, mean1, rstd1) (float32), for a consumer that normalises while it loads
 * (asw_convgemm_args.glu_raw). */
int asw_gn_finalize(const float* stats, int n_partials, int B, int T, int C, float eps, float* mr, void* stream);

/* Multi-head self-attention core: qkv [B][L][3*d] (in_proj output) -> ctx [B][L][d]
 * (softmax(QK^T/sqrt(hd))V per head); nn.MultiheadAttention inside
 * nn.TransformerEncoderLayer (network.py:254). */
int asw_attention(const float* qkv, int B, int L, int d, int nhead, float* ctx, void* stream);
/* The same with the arithmetic of the two products chosen like asw_convgemm_args.precision: 0 = exact f32 MFMA
 * (asw_attention), 1 / 2 = f16x3 split operands on the f16 MFMA (head_dim 128 and L <= 352; the softmax stays
 * fp32; other shapes run as precision 0). */
int asw_attention_prec(const float* qkv, int B, int L, int d, int nhead, int precision, float* ctx, void* stream);

/* output_decoder ConvTranspose1d overlap-add + trim + un-normalise
 * (network.py:346-349,400-405; JointModel/network.py:96).
 * D [B][F][ldd] (per-frame tap products); the transposed convolution has
 * (F-1)*hop + taps samples, of which [trim_left : -trim_right] and then the last t are
 * kept; out [B][t]. */
int asw_overlap_add_unnorm(const float* D, int B, int F, int ldd, int taps, int hop,
                           int t, int trim_left, int trim_right, float bias, const float* mean,
                           const float* std, float* out, void* stream);
/* Same with the tap products given as `nparts` partial tensors [nparts][B][F][ldd] that are added
 * first (asw_mask_path_f16x3 writes one per 256-channel column tile of the latent). */
int asw_overlap_add_parts(const float* D, int nparts, int B, int F, int ldd, int taps, int hop,
                          int t, int trim_left, int trim_right, float bias, const float* mean,
                          const float* std, float* out, void* stream);

/* Per-candidate energies: mean removal, power = sum x^2, power2 = max windowed RMS
 * (local_utils_3d.py:13-17,349-354).  out [B][2] float64.  scratch is unused since ABI 3 (the prefix
 * sums stay in registers / LDS) and may be NULL. */
int asw_energies(const float* y, int B, int T, int window, double* scratch, double* out,
                 void* stream);

/* out = LayerNorm(x + resid) * gamma + beta over rows of N floats (N % 256 == 0, N <= 2048):
 * norm1 / norm2 of the post-norm nn.TransformerEncoderLayer
 * (sep/training/SpeakerLocalization/network.py:254) when the row is too wide to fuse into the
 * producing GEMM's tile.  out may alias x. */
int asw_add_layernorm(const float* x, const float* resid, const float* gamma, const float* beta,
                      int rows, int N, float eps, float* out, void* stream);

/* In-place mean removal of every row (sep/Mic_Array.py:291, local_utils_3d.py:350): the
 * stage loops centre each candidate output before comparing waveforms. */
int asw_center_rows(float* y, int B, int T, void* stream);

/* SI-SDR of every ordered pair (est=i, ref=j) of n waveforms (eval_utils.py:11-39;
 * call sites Mic_Array.py:353,432).  out [n][n] float64. */
int asw_pair_sisdr(const float* y, int n, int T, double* out, void* stream);

/* Segment-wise SI-SDR of every ordered pair (split_wise_sisdr, eval_utils.py:73-82; call site
 * Mic_Array.py:432-458): segments [n][kmax][2] int32 = the [start,end) voiced segments of
 * waveform i (split_wav), seg_count [n]; out [n][n][kmax] float64, entry (i,j,k) = SI-SDR of
 * est = y[i][seg k of i] against ref = y[j][same samples]; entries k >= seg_count[i] are left
 * untouched.  Segment bounds must lie in [0,T] (device arrays). */
int asw_segment_sisdr(const float* y, int n, int T, const int32_t* segments, const int32_t* seg_count,
                      int kmax, double* out, void* stream);

/* HOST function (no GPU): breadth-first subdivision of one coarse hypercube into the fine
 * candidate hypercubes -- search_area / binary_area_divide_width
 * (sep/helpers/local_utils_3d.py:212-335) with Patch.check_out / hyperbola_sample
 * (sep/Traditional_SP/Patch_3D.py:40-47,69-87).  points [3][n_pts] float64 (the coarse patch's
 * area_points), mic [M][3]; offset/width [M-1] are IN/OUT (check_out mutates the caller's
 * patch, as the reference does); ub [M-1] physical TDoA bounds or NULL.  Results are
 * malloc'ed: child_offset/child_width [n_children][M-1], child_count [n_children] and the
 * concatenated point indices child_index; release each with asw_free. */
int asw_search_area(const double* points, int n_pts, const double* mic, int M, double* offset,
                    double* width, const double* ub, double sound_speed, double fs, int* n_children,
                    double** child_offset, double** child_width, int** child_count, int** child_index);
void asw_free(void* p);

/* HOST function (no GPU): flat indices ((y*nx + x)*nz + z), in scan order, of the points of the
 * sub-box [y0,y1) x [x0,x1) x all z of a TDoA lookup table offsets[ny][nx][nz][P] (float64)
 * whose every pair offset lies in [lo[p], hi[p]] -- hyperbola_offset / hyperbola_area_sample
 * (sep/Traditional_SP/SRP_Prunning.py:19-61). */
int asw_cube_select(const double* offsets, int ny, int nx, int nz, int P, int y0, int y1, int x0, int x1,
                    const double* lo, const double* hi, int32_t* out_idx, int64_t cap, int64_t* count);
/* The same scan over a pair-major table planes[P][ny][nx][nz] (8 bytes streamed per point instead of 8 P: the
 * first pair rejects almost every point); identical comparisons, order and result. */
int asw_cube_select_planes(const double* planes, int ny, int nx, int nz, int P, int y0, int y1, int x0, int x1,
                    const double* lo, const double* hi, int32_t* out_idx, int64_t cap, int64_t* count);

/* SRP-PHAT pruning map (sep/Traditional_SP/SRP_Prunning.py:387-434), two stages.
 *
 * asw_srp_cross_spectra: for each of n_windows analysis windows (start w*step, length
 * `window`): STFT restricted to `nbins` bins (rectangular window, hop `hop`, :404-409) as
 * a DFT-GEMM against `twiddle` [2*nb_pad][nfft] (row k: cos(2 pi (bin0+k) n / nfft),
 * row nb_pad+k: -sin(...)), PHAT normalisation X/max(|X|,tol) (:414-416), frame-averaged
 * cross-spectrum of the P = M(M-1)/2 pairs (pair_i < pair_j, :421-426).
 *   mix [M][T]; xf_scratch [M][frames][2*nb_pad]; cc [n_windows][nbins][P][2] (re,im).
 *
 * asw_srp_map: out[g] = max(0, max_w (1/(nbins*P)) sum_{k,p} Re(cc[w][k][p] *
 * exp(+j omega[k] (tau[g][pair_i[p]] - tau[g][pair_j[p]])))) (:428-430; the map starts
 * from zeros, :248-256).  tau [G][M] float64 seconds (mic z ignored by the caller,
 * :368-381); omega [nbins] float64 rad/s; part_scratch [8][8][G] float32; out [G]. */
int asw_srp_frames(int window, int nfft, int hop);
int asw_srp_cross_spectra(const float* mix, int M, int T, int window, int step, int n_windows,
                          int nfft, int hop, int nbins, int nb_pad, float tol,
                          const float* twiddle, const int32_t* pair_i, const int32_t* pair_j,
                          int P, float* xf_scratch, float* cc, void* stream);
int asw_srp_map(const float* cc, int n_windows, int nbins, int P, const double* tau, int G, int M,
                const double* omega, const int32_t* pair_i, const int32_t* pair_j,
                float* part_scratch, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ASW_HIP_H */
