/* seva_hip.h -- C-ABI of libseva_hip.so: the MI355X (gfx950) kernels behind the
 * `seva.model` / `seva.sampling` operator API of Stable Virtual Camera.
 *
 * The reference (atakan-topaloglu/stable-virtual-camera) has no FFI layer: below its Python
 * operator API sit PyTorch ATen ops.  This library sits exactly there.
 * Each entry point names the reference lines whose arithmetic it replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless stated; no allocation, no ownership transfer;
 *  - `seva_stream_t` is a hipStream_t; all work is enqueued on it, nothing synchronises;
 *  - return 0 on success, a negative code on error; `seva_last_error()` gives the message
 *    (thread-local);
 *  - activations are channels-last: a (n, c, h, w) tensor of the reference is stored as
 *    [n][h*w][c]; "f16" is IEEE binary16, "f32" binary32;
 *  - all entry points are graph-capture safe (no sync, no malloc).
 */
#ifndef SEVA_HIP_H
#define SEVA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* seva_stream_t;

#define SEVA_OK 0
#define SEVA_ERR_ARG (-1)
#define SEVA_ERR_LAUNCH (-2)
#define SEVA_ERR_UNSUPPORTED (-3)

const char* seva_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int seva_abi_version(void);
/* Name of the code object's target, e.g. "gfx950". */
const char* seva_target_arch(void);

/* ------------------------------------------------------------------------------------------
 * GEMM / implicit-GEMM 3x3 convolution on fp16 MFMA with fp32 accumulation.
 *   out[m][n] = sum_k A[m][k] * W[n][k]  (+ bias[n]) (+ row_add[m / rows_per_group][n])   [row pitch ld_row_add]
 *                                        (+ residual[m][n])
 * Replaces nn.Linear (seva/modules/transformer.py:11,30,52-57,187,200), nn.Conv2d 1x1/3x3
 * (seva/modules/layers.py:40,55,101,106-108,113,118; seva/model.py:57,173), the nearest-2x
 * upsample feeding a conv (layers.py:43-45), the GEGLU gate (transformer.py:13-15) and the
 * residual / timestep-embedding adds fused into them (layers.py:133-138, transformer.py:107-109).
 *
 * mode 0 (plain): A is [M][lda] f16.
 * mode 1 (conv3x3, pad 1): A is an NHWC f16 image [n][ih][iw][cin]; K = 9*cin ordered
 *   (ky, kx, ci); M = n*oh*ow.  `stride` is 1 or 2.  `upsample`=1 applies the conv to the
 *   nearest-neighbour 2x upsampling of A without materialising it (oh = 2*ih).
 * epilogue 0: linear.  epilogue 1 (GEGLU): W rows are interleaved in groups of 64 as
 *   [32 value rows | 32 gate rows] (same for bias); output has N/2 columns:
 *   out[m][f] = (v + bv) * gelu_erf(g + bg).
 * Requirements: K % 64 == 0, cin % 64 == 0 (mode 1), N % 4 == 0, all row pitches % 4 == 0,
 * pointers 16-byte aligned.
 */
typedef struct seva_gemm_desc {
  const void* a;
  const void* w;         /* f16 [N][K] */
  const float* bias;     /* [N] or NULL */
  const float* row_add;  /* [ceil(M / rows_per_group)][ld_row_add] or NULL */
  const float* residual; /* [M][ldr] or NULL */
  float* out_f32;        /* [M][ldo32] or NULL */
  void* out_f16;         /* [M][ldo16] or NULL */
  int64_t M, N, K;
  int64_t lda, ldr, ldo32, ldo16;
  int64_t rows_per_group;
  int64_t ld_row_add;    /* row pitch of row_add (>= N; 0 means N) */
  int32_t mode;
  int32_t epilogue;
  int32_t n, ih, iw, cin, oh, ow, stride, upsample;
  /* out[:, f] = (a @ w^T + bias)[:, f] * col_scale for f < col_scale_n (fp32, before the f16
   * rounding); 0 columns = off.  Plain epilogue without residual / row_add only.  Used to fold the
   * softmax scale * log2(e) into the q third of a fused QKV projection. */
  float col_scale;
  int32_t col_scale_n;
  /* conv mode: 1 = zero padding only at the bottom / right edge (taps start AT pixel (stride*oy, stride*ox)); the
   * stride-2 Downsample2D of the diffusers VAE encoder pads (0,1,0,1).  0 = symmetric pad 1 (every UNet conv). */
  int32_t pad_br_only;
  /* seva_gemm_fp8 only (NULL / 0 for seva_gemm_f16): */
  const void* w_exp;     /* uint8 [N]: E8M0 scale byte 127 + e[n]; weight row n holds e4m3(w[n] * 2^-e[n]) */
  void* out_f8;          /* GEGLU epilogue: e4m3 [M][ldo8] (saturating), the next fp8 GEMM's A operand; or NULL */
  int64_t ldo8;
  /* optional (NULL = off): GroupNorm statistics of out_f32, emitted by the epilogue while the values are in registers,
   * so that the GroupNorm consuming this tensor (seva_groupnorm_desc.stats1 / stats2) needs no statistics pass over it.
   * float [ceil(M / 64)][2][N]: for every block of 64 output rows and every output channel, the sum ([..][0][n]) and the
   * sum of squares ([..][1][n]) of the fp32 values stored (after bias / row_add / residual); rows >= M contribute
   * nothing.  A block is 64 consecutive rows aligned to multiples of 64 rows of the whole tensor, EXCEPT for 3x3
   * convolutions over images at least 144 pixels wide with N % 160 != 0 (the 2-D tiles of the window-staged kernel, round
   * 4), where the blocks [i * hw / 64, (i + 1) * hw / 64) partition the pixels of image i in tile order: consumers must
   * only rely on the blocks of an image adding up to that image (seva_groupnorm does).  Per channel, so any grouping or
   * channel concatenation can be formed by the consumer.  Plain epilogue with out_f32 and N >= 128 only; forces 128-row tiles. */
  float* ch_stats;
  /* optional (NULL = off) workspace that lets seva_gemm_f16 run a convolution over small images (<= 128 output pixels per
   * sample, K >= 1024: the 9x9 level of a 576x576 step) as split-K = 2 on 128-row tiles: two workgroups per tile, the
   * upper half of K exported raw, added by the partner before the epilogue (fixed association: deterministic, and chosen
   * from per-sample dimensions only).  Layout: 16384 int flags (zero before first use; every launch leaves them zero), then
   * one 128 x BN fp32 tile per output tile: splitk_ws_bytes >= 4 * (16384 + tiles * 128 * 160), tiles = ceil(M / 128) *
   * ceil(N / 160).  The hand-off
   * uses agent-scope (sc1) stores / loads, no cache-wide fence.  One launch at a time may
   * use a given workspace (launches on ONE stream are fine).  A workspace too small for a launch that qualifies for the split is
   * an ERROR (ABI 8; it was a silent fall-back to the unsplit kernel): whether a sample is split never depends on the batch. */
  float* splitk_ws;
  int64_t splitk_ws_bytes;
  /* optional second A operand of a 3x3 convolution (ABI 7; mode 1, no upsample): a2 [M][lda2] f16 whose K2 columns (K2 % 64 == 0)
   * FOLLOW the nine taps in the reduction, K = 9 * cin + K2 and w = [w_conv | w_2] per output row.  Folds the ResBlock's 1x1
   * skip convolution (seva/modules/layers.py:137-139: `skip_connection(x) + h`) into its second 3x3 conv: one accumulation, no
   * fp32 round trip of the skip result, one launch fewer. */
  const void* a2;
  int64_t lda2;
  int64_t K2;
  /* accounting only (ABI 8; seva_prof_collect): reduction length of the REFERENCE-equivalent operator when the executed one is
   * longer -- zero-padded input channels (the stem: 11 of 64), split-precision [hi | lo] operands (K doubled).  0 = K.  The
   * profiler's FLOP count of the launch is 2 M N alg_K, so that a kernel class's achieved TFLOP/s is never credited with padding. */
  int64_t alg_K;
} seva_gemm_desc;
int seva_gemm_f16(const seva_gemm_desc* d, seva_stream_t stream);
/* BASELINE config 5 ("fp8 weights, CDNA4 fp8 MFMA"): the same operator with BOTH operands in OCP e4m3 (a: [M][lda]
 * bytes or an NHWC e4m3 image, w: [N][K] bytes) on v_mfma_scale_f32_16x16x128_f8f6f4 with fp32 accumulation; the
 * per-output-channel power-of-two weight scale enters as the MFMA's E8M0 block scale, so the accumulator holds the
 * de-quantised product and bias / row_add / residual / GEGLU / col_scale behave exactly as in seva_gemm_f16.
 * K % 128 == 0 (conv: cin % 128 == 0), N % 16 == 0, no fused upsample.  The reference keeps bf16 weights
 * (seva/utils.py:50-53): this is a separate precision mode, reported separately from the f16 parity mode. */
int seva_gemm_fp8(const seva_gemm_desc* d, seva_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused GEGLU feed-forward (FeedForward of seva/modules/transformer.py:18-34) for narrow levels, C in {64,128,256,320}:
 *   out[m][:] = W2 . (v * gelu_erf(g)) + b2 (+ residual[m][:]),  [v ; g] = W1 . a[m][:] + b1
 * a: f16 [M][lda]; w1: f16 [8C][C] and b1: [8C] in the interleaved GEGLU layout of seva_gemm_f16 (groups of 64 rows =
 * [32 value | 32 gate]); w2: f16 [C][4C]; b2: [C].  The 4C-wide hidden activations stay in registers (rounded to f16
 * once, as the two-kernel form rounds the stored tensor); fp32 accumulation.  The preceding LayerNorm can be folded in.
 */
typedef struct seva_ff_desc {
  const void* a;
  const void* w1;
  const float* b1;
  const void* w2;
  const float* b2;
  const float* residual; /* [M][ldr] or NULL */
  float* out_f32;        /* [M][ldo32] or NULL */
  void* out_f16;         /* [M][ldo16] or NULL */
  int64_t M, lda, ldr, ldo32, ldo16;
  int32_t C;
  /* optional LayerNorm prologue (transformer.py:102-104,141-143: ff(norm(x))): when ln_x is set the A operand is
   * LayerNorm(ln_x) * ln_gamma + ln_beta computed in registers from fp32 rows [M][ldx] and `a` is ignored. */
  const float* ln_x;
  const float* ln_gamma;
  const float* ln_beta;
  int64_t ldx;
  float ln_eps;
} seva_ff_desc;
int seva_ff_fused_f16(const seva_ff_desc* d, seva_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Scaled-dot-product attention, head dim 64, no mask, fp16 in/out, fp32 softmax.
 * Replaces F.scaled_dot_product_attention under sdpa_kernel(FLASH_ATTENTION)
 * (seva/modules/transformer.py:66-72) for the three regimes of SURVEY.md §2.1: per-frame,
 * joint (view*h*w) and temporal (L = num_frames, batch = pixels).  The temporal regime reads
 * the (b t) s c layout in place through strides instead of the transposes at
 * transformer.py:149,154.
 * Element (b0, b1, token l, head h, d) of q lives at q + b0*q_sb0 + b1*q_sb1 + l*q_sl + h*64 + d
 * (strides in elements); k and v share strides; out likewise.
 */
typedef struct seva_attn_desc {
  const void* q;
  const void* k;
  const void* v;
  void* out;
  int64_t q_sb0, q_sb1, q_sl;
  int64_t k_sb0, k_sb1, k_sl;
  int64_t o_sb0, o_sb1, o_sl;
  int32_t nb0, nb1;
  int32_t heads;
  int32_t lq, lk;
  float scale;
  /* non-zero: q already holds q * scale * log2(e) (see seva_gemm_desc.col_scale); `scale` is ignored
   * and the kernel evaluates softmax as exp2(q' . k - max) */
  int32_t q_prescaled;
  /* optional workspace (ABI 7) for the K/V split of LONG sequences: with it, launches whose key length is >= 6144 (the joint
   * view x space attention at 36x36 / 18x18: L = 27216 / 6804) run every query block as TWO workgroups over the two halves of
   * the K/V tiles plus a small fp32 combine -- the split is a function of lk only, so a sample's result does not depend on
   * the batch.  It fills the last, partly empty round of workgroup slots (2140 workgroups = 4.18 rounds of the 512 slots ->
   * 8.36 half-length rounds).  Needs 2 * nb0 * nb1 * heads * lq * 66 floats; NULL = never split. */
  float* split_ws;
  int64_t split_ws_bytes;
} seva_attn_desc;
int seva_attention_f16(const seva_attn_desc* d, seva_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * GroupNorm (fp32 statistics) over a channels-last tensor that may be the channel
 * concatenation of two sources (the UNet skip concat, seva/model.py:206-207, is never
 * materialised), optionally followed by SiLU and the Pluecker scale/shift modulation
 *   y = silu(gn(x)) * (1 + scale) + shift,  [scale|shift] = dense_w @ dense[n][p][:] + dense_b
 * Replaces GroupNorm32 + SiLU + dense_emb_layers of seva/modules/layers.py:61-63,98-100,
 * 106-111,122-131, the output head norm (seva/model.py:171) and the transformer input norm
 * (seva/modules/transformer.py:186,231).
 * workspace: at least n * 1024 * groups * 2 floats (SEVA_GN_WORKSPACE_SLABS slab slots per sample).
 */
#define SEVA_GN_WORKSPACE_SLABS 1024
typedef struct seva_groupnorm_desc {
  const float* x1; /* [n][hw][c1] */
  const float* x2; /* [n][hw][c2] or NULL */
  const float* gamma;
  const float* beta;    /* [c1 + c2] */
  const float* dense;   /* [n][hw][dense_c] or NULL */
  const float* dense_w; /* [2*(c1+c2)][dense_c] */
  const float* dense_b; /* [2*(c1+c2)] */
  void* out_f16;        /* [n][hw][c1+c2] */
  float* workspace;
  int32_t n, hw, c1, c2, groups, dense_c, silu;
  float eps;
  /* optional second output: the un-normalised (concatenated) input cast to f16, [n][hw][c1+c2] -- the A operand of
   * the ResBlock's 1x1 skip convolution (seva/modules/layers.py:137), written in the same pass instead of by a
   * separate seva_cast_concat_f16 read of both sources.  NULL = off. */
  void* raw_f16;
  /* optional e4m3 output (saturating), same layout: the A operand of seva_gemm_fp8 / an fp8 conv.  When set, out_f16
   * may be NULL. */
  void* out_f8;
  int64_t ld_out_f8; /* pixel pitch of out_f8 in bytes (>= c1 + c2; 0 = c1 + c2): lets a 320-channel tensor feed an fp8 conv whose
                      * channel count is padded to a multiple of 128 (pad bytes are never written: keep them zero) */
  /* optional: per-channel partial statistics of x1 / x2 as written by the kernel that produced them
   * (seva_gemm_desc.ch_stats: [n * hw / 64][2][c]).  When given for every source, the statistics pass over the fp32
   * tensors is skipped (x1 / x2 are then read once, by the apply pass).  Requires hw % 64 == 0 (a 64-row block then
   * belongs to one sample and a sample's statistics stay bitwise independent of the batch). */
  const float* stats1;
  const float* stats2;
  /* split-precision outputs (ABI 7).  Non-zero: the output has pixel pitch 2 * (c1 + c2); channels [0, C) hold hi = f16(v) as
   * usual and channels [C, 2C) lo = f16(v - f32(hi)).  A consumer GEMM / conv whose weights are duplicated over the two
   * halves then sees the operand to ~22 bits (fp32 accumulation of hi * w + lo * w).  Used for the three operand roundings
   * that dominate the network's error (tests/test_f16_floor_cpu.py): the head conv's input (split_out_f16) and the 1x1 skip
   * convs' input (split_raw_f16).  Plain (non-modulated) and 6-component-modulated GroupNorms only. */
  int32_t split_out_f16;
  int32_t split_raw_f16;
} seva_groupnorm_desc;
int seva_groupnorm_f16(const seva_groupnorm_desc* d, seva_stream_t stream);

/* LayerNorm over the last dim, fp32 in, f16 out (nn.LayerNorm, transformer.py:102-104,124,141-143). */
int seva_layernorm_f16(const float* x, const float* gamma, const float* beta, void* out_f16,
                       int64_t rows, int32_t c, float eps, seva_stream_t stream);
/* Same normalisation with OCP e4m3 output (saturating at +-448): the A operand of seva_gemm_fp8. */
int seva_layernorm_fp8(const float* x, const float* gamma, const float* beta, void* out_f8, int64_t rows,
                       int32_t c, float eps, int64_t ld_out /* row pitch in bytes, >= c; 0 = c; pad bytes untouched */,
                       seva_stream_t stream);

/* Row softmax: out[r][c] = softmax_c(x[r][c] * scale) as f16 for c < cols; columns cols..cols_pad-1
 * of `out` are written as 0 (so `out` can be the K-padded A operand of the following P*V GEMM).
 * Used for the single-head, d=512 attention of the VAE mid block, which runs as
 * GEMM(QK^T) -> softmax -> GEMM(PV) (diffusers AutoencoderKL decoder, reference autoencoder.py:38). */
int seva_softmax_rows_f16(const float* x, int64_t ldx, void* out_f16, int64_t ldo, int64_t rows,
                          int32_t cols, int32_t cols_pad, float scale, seva_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Layout / elementwise helpers.
 */
/* [n][c][h][w] f32 (two sources concatenated on c; second may be NULL) -> [n][h*w][cpad] f16,
 * channels >= c1+c2 zero-filled.  Replaces torch.cat at seva/model.py:227 + the NCHW->NLC
 * rearranges (transformer.py:232). `scale` (per-n, may be NULL) multiplies source 1 (c_in of
 * the denoiser, sampling.py:149). */
int seva_nchw_to_nhwc_f16(const float* x1, int32_t c1, const float* x2, int32_t c2,
                          const float* scale, void* out_f16, int32_t n, int32_t hw, int32_t cpad,
                          seva_stream_t stream);
/* Same, split precision (ABI 7): channels [C, 2C), C = c1 + c2, receive lo = f16(v - f32(f16(v))) of the value whose f16(v)
 * sits in [0, C); cpad >= 2C.  The stem conv (model.py:103) has 11 real channels in one 64-channel K-tile, so with its weights
 * duplicated over [C, 2C) the network input enters at ~22 bits for free. */
int seva_nchw_to_nhwc_f16_split(const float* x1, int32_t c1, const float* x2, int32_t c2,
                                const float* scale, void* out_f16, int32_t n, int32_t hw, int32_t cpad,
                                seva_stream_t stream);
/* [rows][ld] f32 channels-last -> [n][c][hw] f32 (first c channels). */
int seva_nhwc_to_nchw_f32(const float* x, int64_t ld, float* out, int32_t n, int32_t c,
                          int32_t hw, seva_stream_t stream);
/* concat-cast: [rows][c1] f32 ‖ [rows][c2] f32 -> [rows][c1+c2] f16 (x2 may be NULL). */
int seva_cast_concat_f16(const float* x1, int32_t c1, const float* x2, int32_t c2, void* out_f16,
                         int64_t rows, seva_stream_t stream);
/* Bilinear resize, align_corners=True, [n][c][sh][sw] f32 -> channels-last [n][oh*ow][c] f32
 * (F.interpolate at seva/modules/layers.py:126-130; step-invariant). */
int seva_bilinear_to_nhwc_f32(const float* src, float* out, int32_t n, int32_t c, int32_t sh,
                              int32_t sw, int32_t oh, int32_t ow, seva_stream_t stream);
/* Sinusoidal timestep embedding cos(t*f)||sin(t*f) (layers.py:11-32), f16 out [n][dim]; t is
 * int64; freqs = exp(-ln(max_period) * arange(dim/2) / (dim/2)) as an f32 device table. */
int seva_timestep_embedding_f16(const int64_t* t, const float* freqs, void* out_f16, int32_t n,
                                int32_t dim, seva_stream_t stream);
/* out = silu(x) as f16 (nn.SiLU before emb_layers / inside time_embed). */
int seva_silu_f16(const float* x, void* out_f16, int64_t count, seva_stream_t stream);
/* out[i] = a[i] + b[i] (SkipConnect, transformer.py:158-165), fp32. */
int seva_add_f32(const float* a, const float* b, float* out, int64_t count, seva_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Sampler elementwise (seva/sampling.py).  x tensors are [n][c][h][w] f32, `chw` = c*h*w.
 */
/* DiscreteDenoiser input blend, sampling.py:146-148:  out = x*(1-mask) + lat*mask, where
 * replace = [n][c+1][hw] holds (lat, mask). */
int seva_replace_blend_f32(const float* x, const float* replace, float* out, int32_t n, int32_t c,
                           int32_t hw, seva_stream_t stream);
/* out = net * c_out[n] + x * c_skip[n]  (sampling.py:149-152). */
int seva_denoiser_combine_f32(const float* net, const float* x, const float* c_out,
                              const float* c_skip, float* out, int32_t n, int64_t chw,
                              seva_stream_t stream);
/* x += eps * noise_scale[n]  (sampler_step, sampling.py:359-362); in-place allowed. */
int seva_add_noise_f32(const float* x, const float* eps, const float* noise_scale, float* out,
                       int32_t n, int64_t chw, seva_stream_t stream);
/* CFG + Euler update in one pass (sampling.py:204-213,364-368):
 *   den = u + scale[n]*(c - u);  d = (x - den)/sigma_hat[n];  out = x + dt[n]*d
 * `den2` = [2n][chw] with the uncond half first. */
int seva_cfg_euler_f32(const float* x, const float* den2, const float* scale,
                       const float* sigma_hat, const float* dt, float* out, int32_t n, int64_t chw,
                       seva_stream_t stream);

/* Classifier-free guidance combine alone (ConstantGuidance, sampling.py:204-213):
 *   out = u + scale[n]*(c - u), den2 = [2n][chw] with the uncond half first. */
int seva_cfg_combine_f32(const float* den2, const float* scale, float* out, int32_t n, int64_t chw,
                         seva_stream_t stream);
/* Euler step alone (to_d + update, sampling.py:24-25,366-368): out = x + dt[n]*(x - den)/sigma[n]. */
int seva_euler_step_f32(const float* x, const float* den, const float* sigma_hat, const float* dt,
                        float* out, int32_t n, int64_t chw, seva_stream_t stream);
/* d = (x - den)/sigma[n]  (to_d, sampling.py:24-25). */
int seva_to_d_f32(const float* x, const float* den, const float* sigma, float* out, int32_t n,
                  int64_t chw, seva_stream_t stream);
/* out = x * s[n] (input * c_in, sampling.py:150); in-place allowed. */
int seva_scale_rows_f32(const float* x, const float* s, float* out, int32_t n, int64_t chw,
                        seva_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Conditioning geometry.
 * Pluecker ray maps, seva/geometry.py:119-165 (get_plucker_coordinates) with get_center_and_ray
 * (102-116) and get_image_grid (82-89): out[v][0:3][y][x] = unit ray direction of latent pixel
 * (x+0.5, y+0.5) of view v, out[v][3:6] = camera centre x direction, all in the source camera's frame.
 * kinv [views][9]: row-major inverse of view v's intrinsics in latent-pixel units;
 * pose_inv [views][12]: rows of inverse(to_hom(extrinsics_rel[v]))[:3, :4] (the two small matrix inverses
 * per view stay on the host, as in the reference).  out: [views][6][h][w] fp32.
 */
int seva_plucker_f32(const float* kinv, const float* pose_inv, float* out, int32_t views, int32_t h,
                     int32_t w, seva_stream_t stream);
/* Channel assembly of do_sample (seva/eval.py:1255-1270): c_concat[v] = [mask[v] broadcast | plucker[v]],
 * uc_concat[v] = [0 | plucker[v]]; both [views][7][h][w] fp32; mask: one byte per view. */
int seva_cond_concat_f32(const float* plucker, const uint8_t* mask, float* c_concat, float* uc_concat,
                         int32_t views, int32_t h, int32_t w, seva_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * CLIP ViT-H-14 image conditioner (seva/modules/conditioner.py:7-39).  Its GEMMs / LayerNorms are
 * seva_gemm_f16 / seva_layernorm_*; these three entry points are what those cannot express.
 */
/* LayerNorm with fp32 output (CLIP's ln_pre: the result IS the residual stream).  No overlap of x and out. */
int seva_layernorm_f32(const float* x, const float* gamma, const float* beta, float* out_f32, int64_t rows,
                       int32_t c, float eps, seva_stream_t stream);
/* conditioner.py:24-34: kornia.geometry.resize(x, (out,out), bicubic, align_corners=True, antialias) -> (x+1)/2 ->
 * normalize(mean, std); x: [n][3][H][W] fp32 in [-1,1].  The result is written as the f16 patch matrix of the
 * patch x patch / stride-patch embedding conv: row = image * (out/patch)^2 + py * (out/patch) + px, column =
 * (c * patch + ky) * patch + kx (= conv weight .reshape(width, 3*patch*patch)); columns beyond 3*patch^2 are untouched. */
int seva_clip_preprocess_f16(const float* x, void* patches_f16, int32_t n, int32_t H, int32_t W, int32_t out_size,
                             int32_t patch, int32_t ld_patches, const float* mean, const float* std,
                             int32_t antialias, seva_stream_t stream);
/* softmax(q k^T * scale) v for short sequences and any even head dim <= 128 (ViT-H-14: L = 257, d = 80), f16 in/out,
 * fp32 math.  Element (b, token l, head h, d) of q at q + b*q_sb + l*q_sl + h*head_dim + d; k, v share strides. */
int seva_attention_small_f16(const void* q, const void* k, const void* v, void* out, int64_t q_sb, int64_t q_sl,
                             int64_t k_sb, int64_t k_sl, int64_t o_sb, int64_t o_sl, int32_t batch, int32_t heads,
                             int32_t L, int32_t head_dim, float scale, seva_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Benchmark / debugging knobs.  The library reads its SEVA_* environment variables ONCE, when it is loaded (nothing
 * on the launch path calls getenv); a host changes a knob at run time with seva_set_knob (tests, tools).  Names:
 * gemm_chunks, gemm_dbg, gemm_stagger, gemm_bm, gemm_bn, gemm_astat, attn_dbg, attn_no_tr, attn_two, attn_split,
 * gn_min_iter, conv_win (0: per-tap conv gather everywhere; 1 / 2: force the 4-wave / 8-wave family of the window-staged conv kernel)
 * (environment: SEVA_ + upper case).  -1 = unset (default heuristics).  None is needed in production.
 */
int seva_set_knob(const char* name, int32_t value);
int seva_get_knob(const char* name, int32_t* value);

/* ------------------------------------------------------------------------------------------
 * hipGraph helpers: capture everything enqueued on `stream` between begin/end, replay later.
 */
int seva_graph_begin(seva_stream_t stream);
int seva_graph_end(seva_stream_t stream, void** graph_exec_out);
int seva_graph_launch(void* graph_exec, seva_stream_t stream);
int seva_graph_destroy(void* graph_exec);

/* ------------------------------------------------------------------------------------------
 * Per-kernel-class timing with HIP events on the launch stream (bench.py roofline leg).
 * Classes: 0 gemm, 1 conv, 2 attention, 3 norm, 4 elementwise.
 */
#define SEVA_PROF_CLASSES 5
int seva_prof_enable(int on);
/* Synchronises; fills ms[SEVA_PROF_CLASSES], launches[...], work[...] (algorithmic flop for classes 0-2, bytes for
 * 3-4) and bytes[...] (algorithmic HBM bytes: every operand read once, every result written once); resets. */
int seva_prof_collect(double* ms, int64_t* launches, double* work, double* bytes);

#ifdef __cplusplus
}
#endif
#endif /* SEVA_HIP_H */
