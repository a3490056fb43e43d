/*
 * basic_hip.h -- C ABI of libbasic_hip.so, the MI355X (gfx950) native replacement for the
 * native layer of worldlife123/cbench_BaSIC on the encode/decode hot path.
 *
 * Every entry point is `extern "C"`, takes plain pointers / sizes, returns an int status and
 * never throws.  Pointers named d_* are DEVICE pointers (HBM); everything else is host
 * memory.  `hip_stream` is a hipStream_t passed as void* (NULL = default stream).
 * Citations (file:line) are relative to the reference checkout /root/reference.
 */
#ifndef BASIC_HIP_H
#define BASIC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (the Python shim maps them to the reference's exceptions) ---------- */
#define BASIC_OK 0
#define BASIC_ERR_INVALID (-1)   /* py::value_error in the reference (bad shape/argument)      */
#define BASIC_ERR_NOT_INIT (-2)  /* "ANS not initialized!"  csrc/ans/rans64.cpp:209,395,506   */
#define BASIC_ERR_HIP (-3)       /* HIP runtime failure; text in basic_last_error()            */
#define BASIC_ERR_OVERFLOW (-4)  /* caller buffer too small (reference: UB, rans64.cpp:240)    */
#define BASIC_ERR_NO_DEVICE (-5) /* no gfx950 device visible: the library never falls back     */

const char *basic_last_error(void);
/* Number of visible HIP devices; BASIC_ERR_NO_DEVICE if none. */
int basic_device_count(int *count);
int basic_set_device(int ordinal);
int basic_stream_synchronize(void *hip_stream);

/* ======================================================================================
 * 1. rANS tables  -- replaces Rans64Base::init_params / init_cdf_params / get_cdfs
 *    (csrc/ans/rans64.cpp:69-182, rans64.hpp:38-49) and pmf_to_quantized_cdf
 *    (csrc/ans/rans64.cpp:69-126 == csrc/rans/rans_interface.cpp:450-519).
 *    Tables are built on the host in IEEE float32 with the reference's operation order and
 *    uploaded once; the object owns both copies.
 * ==================================================================================== */
typedef struct basic_rans_tables basic_rans_tables;

int basic_pmf_to_quantized_cdf(const float *pmf, int n, int precision, int32_t *cdf_out /* n+1 */);

/* freqs: int32 [rows][freq_stride]; nsym/offsets: int32 [rows]. */
int basic_rans_tables_from_freqs(const int32_t *freqs, int rows, int freq_stride, const int32_t *nsym,
                                 const int32_t *offsets, int freq_precision, int bypass_coding,
                                 int bypass_precision, basic_rans_tables **out);
/* cdfs: int32 [rows][cdf_stride]; cdf_sizes/offsets: int32 [rows]. */
int basic_rans_tables_from_cdfs(const int32_t *cdfs, int rows, int cdf_stride, const int32_t *cdf_sizes,
                                const int32_t *offsets, int freq_precision, int bypass_coding,
                                int bypass_precision, basic_rans_tables **out);
/* AR index-remap tables, ANSBase::init_ar_params (csrc/ans/ans_interface.cpp:75-137):
 * ar_tab int32 [k][rows][s1] (order 1) or [k][rows][s1][s1] (order 2). */
int basic_rans_tables_set_ar(basic_rans_tables *t, const int32_t *ar_tab, int k, int rows, int order, int s1);
/* Custom AR ops, ANSBase::init_custom_ar_ops (csrc/ans/ans_interface.hpp:40-48; the op: ar_limited_scaled_add_linear_op,
 * csrc/ans/ar_funcs.hpp:58-87): ops float32 [k][7] = (w0, w1, w2, bias, scale, min, max).  The table row of an element becomes
 * op(index, previous symbols) -- RAW symbol values at the back distances of the call's ar_offsets rows (1..3 of them). */
int basic_rans_tables_set_ar_ops(basic_rans_tables *t, const float *ops, int k);
int basic_rans_tables_info(const basic_rans_tables *t, int *rows, int *max_cdf_len);
/* get_cdfs(): out int32 [rows][out_stride], padding written as 0. */
int basic_rans_tables_get_cdfs(const basic_rans_tables *t, int32_t *out, int out_stride);
void basic_rans_tables_destroy(basic_rans_tables *t);

/* ======================================================================================
 * 2. Host-buffer coder -- the drop-in for the pybind11 methods
 *      Rans64Encoder::encode_with_indexes   csrc/ans/rans64.cpp:203-361
 *      Rans64Decoder::decode_with_indexes   csrc/ans/rans64.cpp:389-499
 *      Rans64Decoder::set_stream/decode_stream  rans64.hpp:104-111, rans64.cpp:501-598
 *      BufferedRansEncoder/RansDecoder      csrc/rans/rans_interface.cpp:109-424
 *    Arrays are staged to HBM, coded by the HIP kernel, and the result copied back.
 *    ar_indexes may be NULL (treated as 0); ar_off0/ar_off1 are the per-element back
 *    distances (rows of the reference's ar_offsets array); pass NULL when no AR table is set.
 * ==================================================================================== */
int basic_rans_encode_host(const basic_rans_tables *t, const int32_t *symbols, const int32_t *indexes,
                           int64_t n, const int32_t *ar_indexes, const int32_t *ar_off0,
                           const int32_t *ar_off1, uint8_t *out, int64_t out_capacity, int64_t *out_len);
/* The same with a third ar_offsets row (custom AR ops take up to three predecessors). */
int basic_rans_encode_host_ex(const basic_rans_tables *t, const int32_t *symbols, const int32_t *indexes, int64_t n,
                              const int32_t *ar_indexes, const int32_t *ar_off0, const int32_t *ar_off1,
                              const int32_t *ar_off2, uint8_t *out, int64_t out_capacity, int64_t *out_len);
/* The same with the table rows taken as given even when the set carries an AR remap: Rans64Encoder::flush() of symbols that AR
 * calls cached -- their rows were remapped when they were cached (rans64.cpp:258-263,343,363-386). */
int basic_rans_encode_host_rows(const basic_rans_tables *t, const int32_t *symbols, const int32_t *indexes, int64_t n,
                                uint8_t *out, int64_t out_capacity, int64_t *out_len);
int basic_rans_decode_host_ex(const basic_rans_tables *t, const uint8_t *stream, int64_t stream_len, const int32_t *indexes,
                              int64_t n, const int32_t *ar_indexes, const int32_t *ar_off0, const int32_t *ar_off1,
                              const int32_t *ar_off2, int32_t *out_symbols);
/* Upper bound of the encoded size in bytes for n symbols (always sufficient). */
int64_t basic_rans_encode_bound(int64_t n);
int basic_rans_decode_host(const basic_rans_tables *t, const uint8_t *stream, int64_t stream_len,
                           const int32_t *indexes, int64_t n, const int32_t *ar_indexes,
                           const int32_t *ar_off0, const int32_t *ar_off1, int32_t *out_symbols);

/* Host framing of a batch of per-image streams, the wire format of CompressAI-style coders
 * (reference: write_body / read_body, cbench/modules/prior_model/prior_coder/compressai_coder.py:63-84):
 * ">III" (h, w, n) then per stream ">I" byte length + payload.  `words` holds the n streams back to
 * back, stream i = words[word_off[i] .. word_off[i+1]).  frame: returns BASIC_ERR_OVERFLOW if
 * out_capacity is too small (needed size in *out_len).  unframe: word_off needs n+1 entries and
 * words_out (len - 12 - 4n) / 4 words; pass words_out = NULL to query h, w, n only. */
int basic_frame_streams(const uint32_t *words, const int64_t *word_off, int n, uint32_t h, uint32_t w,
                        uint8_t *out, int64_t out_capacity, int64_t *out_len);
int basic_unframe_streams(const uint8_t *data, int64_t len, uint32_t *h, uint32_t *w, int *n,
                          int64_t *word_off, int n_capacity, uint32_t *words_out);

typedef struct basic_rans_stream basic_rans_stream;
int basic_rans_stream_open(const basic_rans_tables *t, const uint8_t *stream, int64_t stream_len,
                           basic_rans_stream **out);
int basic_rans_stream_decode(basic_rans_stream *s, const int32_t *indexes, int64_t n, int32_t *out_symbols);
void basic_rans_stream_close(basic_rans_stream *s);

/* ======================================================================================
 * 3. Batched device-pointer coder (the hot path: one independent stream per image).
 *    d_seg[nstreams+1] (int64) gives each stream's symbol range inside d_symbols/d_indexes.
 *    Encoder: stream i is written right-aligned into its slot
 *       d_out_words[i*slot_words .. (i+1)*slot_words), its length in d_out_nwords[i]
 *       (-1 = slot overflow).  Bytes = little-endian u32 words, exactly the reference's
 *       py::bytes (rans64.cpp:352-354).
 *    Decoder: stream i is the words d_words[d_word_off[i] .. d_word_off[i+1]) (d_word_off has
 *       nstreams+1 entries; reads past a stream's end return 0 instead of faulting);
 *       d_state/d_pos (per stream) carry the coder between calls (decode_stream semantics);
 *       set d_pos[i] = -1 to initialise from the stream head.
 * ==================================================================================== */
int basic_rans_encode_batch_dev(const basic_rans_tables *t, const int32_t *d_symbols,
                                const int32_t *d_indexes, const int64_t *d_seg, int nstreams,
                                uint32_t *d_out_words, int64_t slot_words, int32_t *d_out_nwords,
                                void *hip_stream);
/* Packs the right-aligned encoder slots into one contiguous buffer: stream i (d_nwords[i] words)
 * goes to d_out + d_out_off[i]; streams with d_nwords[i] <= 0 are skipped. */
int basic_rans_compact_streams_dev(const uint32_t *d_slots, int64_t slot_words, const int32_t *d_nwords,
                                   const int64_t *d_out_off, int nstreams, uint32_t *d_out, void *hip_stream);
int basic_rans_decode_batch_dev(const basic_rans_tables *t, const uint32_t *d_words,
                                const int64_t *d_word_off, const int32_t *d_indexes, const int64_t *d_seg,
                                int nstreams, int32_t *d_out_symbols, uint64_t *d_state, int64_t *d_pos,
                                void *hip_stream);

/* Same decoder with arithmetic segments instead of a d_seg array: stream b decodes the `count` symbols whose
 * indexes sit at d_indexes[first + b*stride ...] and writes them at the same positions of d_out_symbols.  This is
 * the per-topo-group step of the AR coder (decode_stream, pgm_coder.py:971) on dense [B][n] arrays. */
int basic_rans_decode_batch_strided_dev(const basic_rans_tables *t, const uint32_t *d_words,
                                        const int64_t *d_word_off, const int32_t *d_indexes, int64_t first,
                                        int64_t stride, int64_t count, int nstreams, int32_t *d_out_symbols,
                                        uint64_t *d_state, int64_t *d_pos, void *hip_stream);

/* ======================================================================================
 * 4. Entropy-parameter kernels (coalesced elementwise, fused quantise + table index).
 * ==================================================================================== */
/* CompressAI GaussianConditional path (compressai_coder.py:377-393; upstream build_indexes
 * quoted at pgm_coder.py:814-818): idx = (T-1) - #{j<T-1 : max(scale,bound) <= table[j]},
 * sym = round_half_even(y).  d_table: float32 [T] on device.  d_yhat (optional) = float(sym). */
int basic_gc_quantize_index_dev(const float *d_y, const float *d_scales, int64_t n, const float *d_table,
                                int table_len, float scale_bound, int32_t *d_symbols, int32_t *d_indexes,
                                float *d_yhat, void *hip_stream);
/* EntropyBottleneck path (compressai_coder.py:230-245): sym = round(z - median[c]),
 * idx = c, zhat = sym + median[c].  Layout [B][C][HW]. */
int basic_eb_quantize_index_dev(const float *d_z, const float *d_medians, int batch, int channels, int hw,
                                int32_t *d_symbols, int32_t *d_indexes, float *d_zhat, void *hip_stream);
int basic_eb_dequantize_dev(const int32_t *d_symbols, const float *d_medians, int batch, int channels,
                            int hw, float *d_zhat, void *hip_stream);
/* int32 symbols -> float32 (GaussianConditional.decompress/dequantize without means). */
int basic_i32_to_f32_dev(const int32_t *d_in, int64_t n, float *d_out, void *hip_stream);

/* Gaussian PGM coder (pgm_coder.py:735-821, torch_ans.py:279-282) evaluated on the element
 * list of ONE topo group.  Every image of the batch is an independent stream and shares the
 * list d_elems (int32 per-image element ids c*HW+p in coding order = ascending flat index,
 * pgm_coder.py:898-900).  params are "split_interleave" (channel 2c = mean, 2c+1 = scale)
 * in a [B][2C][HW] tensor; y / ybuf are [B][C][HW].  For element k of image b:
 *    idx = argmin_j |scale - table[j]| (first minimum),  mu = mean,
 *    sym = round_half_even(y - mu),  ybuf = sym + mu          (pgm_coder.py:927-941)
 * Outputs are dense per image: d_symbols/d_indexes[b*per_image + out_base + k]. */
int basic_pgm_gauss_encode_group_dev(const float *d_y, const float *d_params, int batch, int channels, int hw,
                                     const int32_t *d_elems, int64_t n_elems, const float *d_table,
                                     int table_len, int32_t *d_symbols, int32_t *d_indexes, int64_t per_image,
                                     int64_t out_base, float *d_ybuf, void *hip_stream);
/* Decoder half: indexes only (before rANS, pgm_coder.py:962-966) ... */
int basic_pgm_gauss_index_group_dev(const float *d_params, int batch, int channels, int hw,
                                    const int32_t *d_elems, int64_t n_elems, const float *d_table,
                                    int table_len, int32_t *d_indexes, int64_t per_image, int64_t out_base,
                                    void *hip_stream);
/* ... and reconstruction (after rANS, pgm_coder.py:973-978): ybuf = sym + mu. */
int basic_pgm_gauss_scatter_group_dev(const int32_t *d_symbols, const float *d_params, int batch, int channels,
                                      int hw, const int32_t *d_elems, int64_t n_elems, int64_t per_image,
                                      int64_t in_base, float *d_ybuf, void *hip_stream);

/* Rate estimate of the forward() pass ("prior_entropy", nats per image): d_nll[b] = -sum log(max(P(q), likelihood_bound)).
 *   interleaved_mean_scale == 0: CompressAI GaussianConditional likelihood (call site compressai_coder.py:352-375),
 *       zero mean, d_scales_or_params = scales [B][C][HW];
 *   interleaved_mean_scale == 1: PGM coder likelihood (pgm_coder.py:374-389), d_scales_or_params = [B][2C][HW] with
 *       channel 2c = mean, 2c+1 = scale.  d_q = the quantised latent [B][C][HW] (or, for the train-mode proxy, the latent
 *       plus uniform noise);
 *   interleaved_mean_scale == 2: the same parameters, likelihood of the ROUNDED RESIDUAL round(d_q - mean) under the zero-mean
 *       density (training_no_quantize_for_likelihood in eval mode, pgm_coder.py:376-387); d_q = the UNQUANTISED latent. */
int basic_gauss_nll_per_image_dev(const float *d_q, const float *d_scales_or_params, int batch, int channels, int hw,
                                  int interleaved_mean_scale, float scale_bound, float likelihood_bound, float *d_nll,
                                  void *hip_stream);
/* EntropyBottleneck likelihood (call site compressai_coder.py:203-228); d_coef float32 [C][58]: the pre-activated
 * 1-3-3-3-3-1 cumulative-logit network of every channel (softplus(matrix), bias, tanh(factor) per layer). */
int basic_eb_nll_per_image_dev(const float *d_zq, const float *d_coef, int batch, int channels, int hw,
                               float likelihood_bound, float *d_nll, void *hip_stream);

/* ======================================================================================
 * 5. Transforms: implicit-GEMM convolution on fp32 MFMA (v_mfma_f32_32x32x2_f32) with fused
 *    bias + activation / (I)GDN epilogue.  Replaces the ATen calls behind
 *    nn/models/google.py:25-101, nn/layers/slimmable_layers.py:157-183,258-282.
 * ==================================================================================== */
typedef struct basic_conv_plan basic_conv_plan;

#define BASIC_ACT_NONE 0
#define BASIC_ACT_RELU 1
#define BASIC_ACT_LEAKY_RELU 2 /* slope 0.01 (torch default) */
#define BASIC_ACT_GDN 3        /* y = x * rsqrt(beta + gamma . x^2)   */
#define BASIC_ACT_IGDN 4       /* y = x * sqrt (beta + gamma . x^2)   */

/* Create a plan from host weights in PyTorch layout.
 *   transposed == 0: weight [cout][cin][k][k]  (nn.Conv2d),   out = floor((in+2p-k)/s)+1
 *   transposed == 1: weight [cin][cout][k][k]  (nn.ConvTranspose2d), out = (in-1)*s-2p+k+output_padding
 * bias may be NULL.  For GDN/IGDN, gamma [cout][cout] and beta [cout] are the EFFECTIVE
 * (already re-parametrised, non-negative) values.  cin_active/cout_active <= cin/cout select
 * the slimmable sub-network (weight slicing W[:co,:ci], slimmable_layers.py:142-170). */
int basic_conv_plan_create(const float *weight, const float *bias, int cin, int cout, int ksize, int stride,
                           int padding, int output_padding, int transposed, int activation,
                           const float *gamma, const float *beta, int cin_active, int cout_active,
                           basic_conv_plan **out);
int basic_conv_plan_out_hw(const basic_conv_plan *p, int in_h, int in_w, int *out_h, int *out_w);
/* Active (sliced) channel counts of the plan's input and output tensors. */
int basic_conv_plan_channels(const basic_conv_plan *p, int *cin_active, int *cout_active);
/* d_in: float32 [batch][cin_active][in_h][in_w]; d_out: float32 [batch][cout_active][out_h][out_w]. */
int basic_conv_forward_dev(const basic_conv_plan *p, const float *d_in, int batch, int in_h, int in_w,
                           float *d_out, void *hip_stream);
void basic_conv_plan_destroy(basic_conv_plan *p);
/* 2*MACs of one forward (conv + GDN), for roofline accounting. */
int64_t basic_conv_plan_flops(const basic_conv_plan *p, int batch, int in_h, int in_w);
/* Kernel launches one forward call makes for this input (sub-pixel phases of transposed convolutions, fused or not;
 * 32-channel slices count once: they share a launch). -1 on bad arguments. */
int basic_conv_plan_launches(const basic_conv_plan *p, int batch, int in_h, int in_w);

/* ======================================================================================
 * 6. Topo-group masked convolution (TopoGroupDynamicMaskConv2d.forward,
 *    nn/layers/masked_conv.py:102-228) evaluated ONLY at a list of positions -- the positions
 *    of the topo group being coded -- instead of the full map the reference recomputes for
 *    every group (pgm_coder.py:922-924,958-961).  One operator serves the 5x5 context conv
 *    and the 1x1 "param merger" layers (pgm_coder.py:1215-1239, masked_conv.py:262-300):
 *      y[b,co,p] = bias[co] + sum_{ci,tap} W[co,ci,tap] * x[b,ci,p+tap]
 *                    * [ topo_in[g_in(ci), p+tap]  (< | <=)  topo_out[g_out(co), p] ]
 * ==================================================================================== */
typedef struct basic_mconv_plan basic_mconv_plan;
/* weight [cout][cin][k][k] (k in {1,3,5}, padding k/2), bias [cout] or NULL.
 * in_groups/out_groups: contiguous channel groups of x / y; allow_same_topogroup selects <=. */
int basic_mconv_plan_create(const float *weight, const float *bias, int cin, int cout, int ksize,
                            int in_groups, int out_groups, int allow_same_topogroup, int activation,
                            basic_mconv_plan **out);
/* d_x [B][cin][H][W]; d_topo_in int32 [in_groups][H][W]; d_topo_out int32 [out_groups][H][W];
 * d_pos int32 [n_pos] flat ids b*H*W + y*W + x.  Writes channels
 * [out_channel_offset, out_channel_offset+cout) of d_y [B][out_channels_total][H][W] at the
 * listed positions only. */
int basic_mconv_forward_pos_dev(const basic_mconv_plan *p, const float *d_x, const int32_t *d_topo_in,
                                const int32_t *d_topo_out, int batch, int h, int w, const int32_t *d_pos,
                                int64_t n_pos, float *d_y, int out_channels_total, int out_channel_offset,
                                void *hip_stream);
/* The same operator inside the group-sequential coding loop (pgm_coder.py:921-941 / :958-978), where d_y persists
 * across the steps of one encode / decode: only (output group, position) pairs whose topo id equals `step` are
 * evaluated -- plus groups without a topo id (-1) at the first step that visits the position (d_first_step int32
 * [H][W] = min over channel groups of the topo ids).  Everything else in d_y is either already exact (earlier steps;
 * the mask makes a value depend only on elements coded before its own step) or not needed yet.  The reference
 * recomputes the full map at every step instead. */
int basic_mconv_forward_step_dev(const basic_mconv_plan *p, const float *d_x, const int32_t *d_topo_in,
                                 const int32_t *d_topo_out, int batch, int h, int w, const int32_t *d_pos,
                                 int64_t n_pos, float *d_y, int out_channels_total, int out_channel_offset,
                                 int step, const int32_t *d_first_step, void *hip_stream);
/* Everything at once: use_step != 0 = the coding-loop variant above; d_in_perm / d_out_perm (int32 [H*W] or NULL) permute
 * the positions inside the planes of x / y -- element (b, c, p) at (b * C + c) * H*W + perm[p] -- for buffers that are
 * PRIVATE to a chain of 1x1 layers (the merger's hidden activations): with a coding step's positions contiguous its gathers
 * and stores cover whole cache lines.  d_in_perm needs a 1x1 layer.  Values are unaffected. */
int basic_mconv_forward_ex_dev(const basic_mconv_plan *p, const float *d_x, const int32_t *d_topo_in,
                               const int32_t *d_topo_out, int batch, int h, int w, const int32_t *d_pos, int64_t n_pos,
                               float *d_y, int out_channels_total, int out_channel_offset, int use_step, int step,
                               const int32_t *d_first_step, const int32_t *d_in_perm, const int32_t *d_out_perm,
                               void *hip_stream);
void basic_mconv_plan_destroy(basic_mconv_plan *p);

/* ======================================================================================
 * 7. Distortion metric: per-image PSNR pieces (benchmark/metrics/pytorch_distortion.py:12-15):
 *    d_mse[b] = mean((a-b)^2) over C*H*W in float32 accumulate-by-tree.
 * ==================================================================================== */
int basic_mse_per_image_dev(const float *d_a, const float *d_b, int batch, int64_t elems_per_image,
                            float *d_mse, void *hip_stream);

/* ======================================================================================
 * 8. Fused image entry points of the plain hyperprior latent graph (SURVEY 8b "basic_encode_image / decode_image"):
 *    ONE call runs what GeneralCodec.compress / decompress run for that graph
 *      compress   general_codec.py:44-89 -> LatentGraphicalANSEntropyCoder.encode latent_graph.py:1232-1264
 *                 x -> g_a -> y -> h_a -> z ; EntropyBottleneck coder (compressai_coder.py:230-236) ; h_s ;
 *                 GaussianConditional coder (:377-385) ; merge_bytes([z body, y body], num_segments=2)
 *      decompress general_codec.py:91-130 -> .decode latent_graph.py:1266-1295 ; g_s
 *    with the same kernels in the same order as the module-by-module path, so the bytes are identical; the host
 *    side of a call is ~60 kernel launches from C++ instead of a Python interpreter walking the graph.
 *    A session BORROWS the layer plans and table sets (they must outlive it), owns its device workspace and pinned
 *    staging buffers, and serves one call at a time (one session per host thread / HIP stream).
 *    Wire format: [u32 native-endian len(z body)] [z body] [y body]; a body is write_body's ">III" (h, w, n) followed by
 *    ">I" length + rANS words per image (compressai_coder.py:63-84).
 * ==================================================================================== */
typedef struct basic_hp_session basic_hp_session;
/* g_a / h_a / h_s / g_s: the fused layer plans of each transform in execution order.  eb_medians: float32
 * [z_channels] (host).  scale_table: float32 [n_scales] (host), scale_bound = GaussianConditional's lower bound. */
int basic_hp_session_create(const basic_conv_plan *const *g_a, int n_g_a, const basic_conv_plan *const *h_a, int n_h_a,
                            const basic_conv_plan *const *h_s, int n_h_s, const basic_conv_plan *const *g_s, int n_g_s,
                            const float *eb_medians, int z_channels, const basic_rans_tables *z_tables,
                            const float *scale_table, int n_scales, float scale_bound,
                            const basic_rans_tables *y_tables, basic_hp_session **out);
/* Upper bound of basic_hp_encode_images' output for this input shape (always sufficient). */
int64_t basic_hp_encode_bound(const basic_hp_session *s, int batch, int h, int w);
/* x: float32 [batch][cin][h][w], on the device (x_on_host == 0) or in host memory (x_on_host != 0: uploaded inside the
 * call on hip_stream -- asynchronously when the memory is page-locked -- as GeneralCodec.compress does,
 * general_codec.py:46-47).  Returns with the bytes complete in `out` (the stream has been synchronised). */
int basic_hp_encode_images(basic_hp_session *s, const float *x, int x_on_host, int batch, int h, int w, uint8_t *out,
                           int64_t out_capacity, int64_t *out_len, void *hip_stream);
/* With out == NULL basic_hp_encode_images only reports the size in *out_len and keeps the encoded batch in the session
 * (page-locked host memory); this call frames it into caller memory (any number of times, until the next encode). */
int basic_hp_encode_result(basic_hp_session *s, uint8_t *out, int64_t out_capacity, int64_t *out_len);
/* Shape of the reconstruction a stream decodes to (from its headers only). */
int basic_hp_decoded_shape(const basic_hp_session *s, const uint8_t *data, int64_t len, int *batch, int *channels, int *h,
                           int *w);
/* d_xhat: float32 [batch][channels][h][w] on the device (the un-clamped output of g_s).  The work is ENQUEUED on
 * hip_stream; the call returns once the stream words have left `data` (no final synchronisation). */
int basic_hp_decode_images(basic_hp_session *s, const uint8_t *data, int64_t len, float *d_xhat,
                           int64_t xhat_capacity_floats, void *hip_stream);
/* Wavefronts (image streams) per workgroup of this session's rANS launches: 1, 2, 4, 8 or 16 (0 = library default). */
int basic_hp_session_set_rans_waves(basic_hp_session *s, int waves_per_block);
/* Opt this session into the process-wide "transform token": the MFMA-heavy phases (compress: everything up to the y
 * rANS encoder; decompress: g_s) of all such sessions then run one after another in GPU time, in host enqueue order,
 * through stream-wait events -- so that with several sessions on several HIP streams one session's rANS chains always
 * run beside another's transforms instead of all sessions falling into lock-step.  enable = 1: one phase at a time (full
 * batches fill the chip on their own); 2: two at a time (small batches, whose launches leave compute units idle); 0: off.
 * The order is kept per device.  No effect on results. */
int basic_hp_session_set_transform_token(basic_hp_session *s, int enable);
void basic_hp_session_destroy(basic_hp_session *s);

/* ======================================================================================
 * 9. Persistent scan-line AR coding loop (csrc/scanline.hip): ONE launch walks all H*W coding steps of
 *    TopoGroupPGMPriorCoder._encode_with_pgm / _pgm_generate (pgm_coder.py:912-981) for the "scanline" topo groups
 *    (one position per group, raster order) with one channel group: masked k x k context convolution at the coded
 *    position (masked_conv.py:102-228: the causal raster neighbours), the dense 1x1 merger layers on cat(ctx, prior)
 *    (masked_conv.py:262-300 / pgm_coder.py:1606-1638), Gaussian index + quantise (pgm_coder.py:735-821,927-941).
 *    The layers' weights stay resident in the LDS of `workgroups` compute units for the whole launch.
 * ==================================================================================== */
typedef struct basic_scanline_plan basic_scanline_plan;
/* ctx_weight [ctx_out][channels][k][k] (PyTorch layout; only the causal taps are used), ctx_bias [ctx_out] or NULL.
 * Dense layer i (i < n_dense): weight [dense_out[i]][in_i], in_0 = ctx_out + prior_channels (input = cat(ctx, prior)),
 * in_i = dense_out[i-1]; bias or NULL.  act_after[0] belongs to the context layer, act_after[1 + i] to dense layer i
 * (1 = LeakyReLU(0.01)).  The last layer must have 2 * channels rows: (mean, scale) pairs, channel 2c = mean.
 * dense_in_groups[i] (NULL = all 1): the input channel groups of the masked convolution dense layer i stands for
 * (masked_conv.py:170-172; cat(ctx, prior) of the first merger layer = 2): the sums are taken in that operator's canonical
 * block order (csrc/mconv.hip), so the integers coded here equal the per-step path's at any batch size. */
int basic_scanline_plan_create(const float *ctx_weight, const float *ctx_bias, int channels, int ctx_out, int ksize,
                               int prior_channels, int n_dense, const float *const *dense_weight,
                               const float *const *dense_bias, const int *dense_out, const int *act_after,
                               const int *dense_in_groups, basic_scanline_plan **out);
int basic_scanline_plan_info(const basic_scanline_plan *p, int *workgroups, int *lds_weight_bytes);
/* d_y [B][C][H][W], d_prior [B][prior_channels][H][W] (NULL when prior_channels == 0), d_table float32 [table_len]:
 * writes d_symbols / d_indexes int32 [B][H*W*C] in coding order (element p * C + c) and d_ybuf [B][C][H][W]
 * (= round(y - mu) + mu).  Enqueued on hip_stream; basic_scanline_status() afterwards tells whether every in-kernel
 * barrier completed (a launch whose grid was not fully resident gives up after a bounded spin instead of hanging). */
int basic_scanline_encode_dev(basic_scanline_plan *p, const float *d_y, const float *d_prior, int batch, int h, int w,
                              const float *d_table, int table_len, int32_t *d_symbols, int32_t *d_indexes, float *d_ybuf,
                              void *hip_stream);
/* Decoder: `tables` = the rANS table set of the coder; stream b = d_words[d_word_off[b] .. d_word_off[b+1]).  The
 * launch adds ceil(B / 4) decoder workgroups (one wavefront per image stream, the table set's search image in LDS) to the
 * compute workgroups; a coding step is: parameters (as in the encoder) -> table rows -> rANS decode of the C symbols of
 * that position -> y_hat = symbol + mean.  Writes d_symbols / d_indexes (coding order) and d_ybuf [B][C][H][W]. */
int basic_scanline_decode_dev(basic_scanline_plan *p, const basic_rans_tables *tables, const uint32_t *d_words,
                              const int64_t *d_word_off, const float *d_prior, int batch, int h, int w, const float *d_table,
                              int table_len, int32_t *d_symbols, int32_t *d_indexes, float *d_ybuf, void *hip_stream);
/* *ok = 1 when basic_scanline_decode_dev can serve `batch` streams of `tables` on the current device (fast search image that
 * fits the LDS; compute + decoder workgroups <= compute units); otherwise the caller decodes with the per-step path, which
 * codes the same integers. */
int basic_scanline_can_decode(const basic_scanline_plan *p, const basic_rans_tables *tables, int batch, int *ok);
/* Batches of 3 .. 64 images take the BATCHED persistent kernel when the layers have its shape (whole 32-row tiles and 64-channel
 * canonical blocks, <= 3 dense layers of <= 768 inputs, a context window of <= 36 blocks): the batch is the N dimension of
 * v_mfma_f32_32x32x2_f32 tiles, a workgroup keeps one row tile of a layer as A fragments in registers, 32 images per set of
 * workgroups.  *max_batch = the largest batch it serves for a latent `w` columns wide on the current device (0 = never);
 * decode != 0 counts the decoder workgroups too.  Which kernel serves a call never changes the coded integers. */
int basic_scanline_batched_max(const basic_scanline_plan *p, int w, int decode, int *max_batch);
int basic_scanline_status(basic_scanline_plan *p, void *hip_stream, int *poisoned);
void basic_scanline_plan_destroy(basic_scanline_plan *p);

/* ======================================================================================
 * 10. Table ANS (csrc/tans.hip) -- the drop-in for the reference's TansEncoder / TansDecoder
 *       class surface            csrc/ans/tans.hpp:78-157 (PYBIND11_TANS_CLASSES :146-157)
 *       TansBase::init_params    csrc/ans/tans.cpp:368-383, init_tables :385-525
 *       encode_with_indexes      csrc/ans/tans.cpp:527-680      flush :682-713
 *       decode_with_indexes      csrc/ans/tans.cpp:715-815
 *     Tables (count normalisation, state / symbol-transform / decode tables, tans.cpp:27-318) are built on the host
 *     and uploaded once; row `rows` of the images is the uniform bypass alphabet when bypass coding is on.
 *     freqs int32 [rows][freq_stride]; nsym / offsets int32 [rows]; table_log in [5, 12].
 * ==================================================================================== */
typedef struct basic_tans_tables basic_tans_tables;
int basic_tans_tables_create(const int32_t *freqs, int rows, int freq_stride, const int32_t *nsym, const int32_t *offsets,
                             int table_log, int max_symbol_value, int bypass_coding, int bypass_precision,
                             basic_tans_tables **out);
/* ANSBase::init_ar_params (csrc/ans/ans_interface.cpp:75-137), as basic_rans_tables_set_ar. */
int basic_tans_tables_set_ar(basic_tans_tables *t, const int32_t *ar_tab, int k, int rows, int order, int s1);
/* One row of the device images back on the host: next_state u16 [2^L], delta_bits / delta_state [nsym],
 * dec_packed u32 [2^L] = base | bits << 12 | symbol << 16 (any pointer may be NULL). */
int basic_tans_tables_get_row(const basic_tans_tables *t, int row, uint16_t *next_state, uint32_t *delta_bits,
                              int32_t *delta_state, uint32_t *dec_packed);
void basic_tans_tables_destroy(basic_tans_tables *t);
/* Host-buffer coder (arrays staged to HBM, coded by the HIP kernel, result copied back).  capacity_syms: the symbol
 * count the reference sizes its output with (-1 = n; flush(): the cached count incl. bypass digits): capacity
 * capacity_syms * table_log / 8 <= 8 bytes is its "Destination buffer is too small" error, a stream of capacity - 8
 * whole bytes or more comes back EMPTY there (bitstream.h:192,245) and here (*out_len = 0). */
int basic_tans_encode_host(const basic_tans_tables *t, const int32_t *symbols, const int32_t *indexes, int64_t n,
                           const int32_t *ar_indexes, const int32_t *ar_off0, const int32_t *ar_off1,
                           int64_t capacity_syms, uint8_t *out, int64_t out_capacity, int64_t *out_len, int64_t *coded_syms);
int basic_tans_decode_host(const basic_tans_tables *t, const uint8_t *stream, int64_t stream_len, const int32_t *indexes,
                           int64_t n, const int32_t *ar_indexes, const int32_t *ar_off0, const int32_t *ar_off1,
                           int32_t *out_symbols);
/* Batched device-pointer coder, one wavefront per stream (no AR remap).  Encoder: stream i goes LEFT-aligned into
 * d_out_words[i * slot_words ..] (slot_words >= basic_tans_encode_bound_words(t, longest stream));
 * d_out_info[2i] = its length in BITS incl. final state and end mark (-1 = slot overflow), d_out_info[2i+1] = symbols
 * coded incl. bypass digits; bytes = ceil(bits / 8).  Decoder: stream i = d_bytes[d_byte_off[i] .. d_byte_off[i+1]);
 * d_status[i] = 1 for an empty stream / missing end mark. */
int64_t basic_tans_encode_bound_words(const basic_tans_tables *t, int64_t n);
int basic_tans_encode_batch_dev(const basic_tans_tables *t, const int32_t *d_symbols, const int32_t *d_indexes,
                                const int64_t *d_seg, int nstreams, uint32_t *d_out_words, int64_t slot_words,
                                int64_t *d_out_info, void *hip_stream);
int basic_tans_decode_batch_dev(const basic_tans_tables *t, const uint8_t *d_bytes, const int64_t *d_byte_off,
                                const int32_t *d_indexes, const int64_t *d_seg, int nstreams, int32_t *d_out_symbols,
                                int32_t *d_status, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* BASIC_HIP_H */
