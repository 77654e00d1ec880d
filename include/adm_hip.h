/*
 * adm_hip.h -- C ABI of libadm_hip.so: the MI355X (gfx950) kernels behind
 * AutoDiffusion's candidate-evaluation hot path.
 *
 * The reference (lilijiangg/AutoDiffusion) is 100 % Python on stock PyTorch
 * ops: it has NO plugin / FFI interface for this path (SURVEY.md section 2.2,
 * 8b).  The entry points below are therefore the build's own boundary; each
 * one cites the reference Python call site it replaces.  A maintainer binds
 * them with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain pointers and sizes; every pointer is a DEVICE pointer unless the
 *     name ends in _host; no torch types, no allocation, no ownership transfer
 *     (workspaces are caller-owned), no hidden synchronisation;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - return value: 0 = ok, negative = invalid argument (see ADM_E_*),
 *     positive = hipError_t from the launch; adm_last_error() returns a
 *     thread-local description of the last failure;
 *   - activations between kernels are bf16 NHWC ("pixel-major": [N][H][W][C]);
 *     the UNet's external input / output stay fp32 NCHW as in the reference;
 *   - safe to call concurrently on different streams / devices.
 */
#ifndef ADM_HIP_H
#define ADM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADM_ABI_VERSION 9   /* 2: adm_conv_args gained in_up / res_up; 3: ksplit / ws; 4: the Inception layer entry points; 5: up_phase; 6: geglu;
                               7: adm_gn_finalize_add gained stats, adm_gn_bwd_finalize gained add / add_stride; the classifier's other heads;
                               8: adm_conv_args gained fold0 / fold1 / fc0 / fc1;
                               9: adm_conv_args gained out_scale; CU-partitioned streams (adm_stream_create_cumask / _set_cus / _destroy) */

#define ADM_E_ARG      (-1)  /* bad pointer / size / flag combination          */
#define ADM_E_SHAPE    (-2)  /* shape not supported by the gfx950 tiling       */
#define ADM_E_ALIGN    (-3)  /* pointer not 16-byte aligned                    */

typedef uint16_t adm_bf16;   /* raw bfloat16 bits */

int adm_abi_version(void);
const char* adm_last_error(void);

/* ---------------------------------------------------------------- CU-partitioned streams
 * The reference runs eps(x_t) and the guidance gradient one after the other on one CUDA stream
 * (gaussian_diffusion.py:381-393 condition_score after p_mean_variance); both depend on x_t only, so this build overlaps them
 * on two streams -- and may give each network its OWN compute units instead of time-sharing all of them:
 * adm_stream_create_cumask creates a HIP stream whose kernels run only on the CUs named by `mask` (`words` x 32 bits, bit i = CU i
 * in the driver's numbering: on an 8-XCD device consecutive bits go round-robin over the XCDs) and registers its CU count, which the
 * persistent kernels (adm_conv) read to size their grids.  adm_stream_set_cus(stream, n) registers / updates (n = 0: forgets) the
 * budget of a stream created elsewhere.  Results never depend on the partition.                                                  */
int adm_stream_create_cumask(const uint32_t* mask, int words, void** stream_out);
int adm_stream_set_cus(void* stream, int ncu);
int adm_stream_destroy(void* stream);
/* diagnostic: out_dev[b] = XCC_ID | HW_ID << 8 of block b of a grid of `nblocks` blocks launched on `stream` (which CUs a mask selects) */
int adm_stream_probe(uint32_t* out_dev, int nblocks, void* stream);

/* ---------------------------------------------------------------- sampler (K9, A7, A10)
 * Per-step scalars = the reference's float64 tables cast to float32 exactly as
 * _extract_into_tensor does (gaussian_diffusion.py:910-923).                           */
typedef struct adm_step_coefs {
  float sqrt_recip_ac;    /* sqrt(1/abar_i)            */
  float sqrt_recipm1_ac;  /* sqrt(1/abar_i - 1)        */
  float ac;               /* abar_i                    */
  float ac_prev;          /* abar_{i-1} (1.0 at i = 0) */
  float coef1, coef2;     /* posterior_mean_coef1/2    */
  float log_var_lo;       /* posterior_log_variance_clipped_i (LEARNED_RANGE min) or fixed log-variance */
  float log_var_hi;       /* log(beta_i) (LEARNED_RANGE max); unused for fixed variance   */
  float fixed_var;        /* fixed variance (FIXED_SMALL/LARGE); unused for learned range */
  float eta;              /* DDIM eta                  */
  int32_t nonzero;        /* 0 at i == 0, else 1       */
  int32_t learned_range;  /* model_out carries 2C channels (eps | v)      */
  int32_t predict_xstart; /* model output is x0 instead of eps            */
  int32_t clip_denoised;  /* clamp pred_xstart to [-1, 1]                 */
} adm_step_coefs;

/* ddim_sample: p_mean_variance + condition_score + eq. 12 update, one pass per pixel.
 * Replaces gaussian_diffusion.py:258-326, :381-393, :565-584.
 * x, noise, grad, x_prev, pred_xstart: fp32 [N,C,H,W]; model_out fp32 [N,C or 2C,H,W];
 * grad / noise / pred_xstart / u8_nhwc may be NULL.  If u8_nhwc != NULL the final image is
 * also packed as ((s+1)*127.5).clamp(0,255) truncated to uint8, NHWC
 * (search_imagenet64_classifier_guidance.py:352-354).                                   */
int adm_ddim_step(const float* x, const float* model_out, const float* grad, const float* noise,
                  float* x_prev, float* pred_xstart, uint8_t* u8_nhwc,
                  int n, int c, int h, int w, const adm_step_coefs* coefs_host, void* stream);

/* p_sample: p_mean_variance + condition_mean + ancestral update.
 * Replaces gaussian_diffusion.py:258-326, :365-369, :430-439.                           */
int adm_ddpm_step(const float* x, const float* model_out, const float* grad, const float* noise,
                  float* x_prev, float* pred_xstart, uint8_t* u8_nhwc,
                  int n, int c, int h, int w, const adm_step_coefs* coefs_host, void* stream);

/* fp32 NCHW in [-1,1] -> uint8 NHWC (truncation), search_...guidance.py:352-354.       */
int adm_pack_u8_nhwc(const float* x, uint8_t* out, int n, int c, int h, int w, void* stream);

/* ---------------------------------------------------------------- embeddings (K1, A4)
 * timestep_embedding (nn.py:103-121): t fp32 [N] -> [N, dim] = cos | sin (| 0 if dim odd). */
int adm_timestep_embedding(const float* t, float* out, int n, int dim, float max_period, void* stream);

/* out[n, o] = sum_k act(in[n, k]) * w[o, k] + bias[o] (+ table[idx[n], o]);  fp32.
 * act = SiLU if silu_in.  Replaces time_embed / label_emb / every ResBlock emb_layers
 * (unet.py:470-478, 199-205, 245, 648-654); all emb_layers are batched in one call by
 * concatenating their weights along o.                                                  */
int adm_linear_f32(const float* in, const float* w, const float* bias, const float* table,
                   const int64_t* idx, float* out, int n, int k, int o, int silu_in, void* stream);

/* ---------------------------------------------------------------- stem (A5, first conv)
 * input_blocks.0.0: conv3x3 pad 1 on the fp32 NCHW image -> bf16 NHWC (unet.py:480-483, 656-658). */
int adm_stem_conv3x3(const float* x_nchw, const float* w /*[Cout,Cin,3,3]*/, const float* bias,
                     adm_bf16* out_nhwc, int n, int cin, int h, int w_, int cout, void* stream);

/* The same layer on the MFMA path (what the engine uses): cast the image to bf16 NHWC with the channels
 * zero-padded to cpad = 32, then run adm_conv (3x3, raw prologue) with the stem weight zero-padded to 32
 * input channels.  The image is rounded to bf16 like every other activation (the reference rounds it to
 * fp16, unet.py:656).                                                                              */
int adm_nchw_to_nhwc_pad(const float* x_nchw, adm_bf16* out_nhwc, int n, int c, int h, int w, int cpad,
                         void* stream);

/* ---------------------------------------------------------------- GroupNorm (K2/K3 statistics)
 * GroupNorm32(32, C) statistics over a (virtually concatenated) bf16 NHWC tensor, in fp32 /
 * fp64 (nn.py:17-19).  Two launches:
 *   adm_gn_partial : per (image, pixel slab, channel) sum and sum of squares
 *                    -> partial fp32 [N][slabs][C][2]
 *   adm_gn_finalize: per (image, channel) affine  y = a*x + b  with
 *        a = rstd*gamma*(1+scale),  b = (beta - mean*rstd*gamma)*(1+scale) + shift
 *      (scale/shift = the ResBlock's FiLM pair, unet.py:248-252; NULL -> plain GN).
 *      film points at scale[0] of image 0; shift = film + C; film_stride = row stride.   */
int adm_gn_partial(const adm_bf16* in0, int c0, const adm_bf16* in1, int c1, float* partial,
                   int n, int hw, int slabs, void* stream);
int adm_gn_finalize(const float* partial, const float* gamma, const float* beta,
                    const float* film, int film_stride, float* aff_a, float* aff_b,
                    float* stats /* nullable: [N][32][2] = (mean, rstd), kept for adm_gn_bwd_* */,
                    int n, int c, int hw, int slabs, float eps, void* stream);
/* Same for a virtual concat whose two parts carry their own partial sums (e.g. from adm_conv's out_stats):
 * partial0 fp32 [N][slabs0][c0][2], partial1 fp32 [N][slabs1][c1][2] (NULL / 0 when there is one part). */
int adm_gn_finalize2(const float* partial0, int c0, int slabs0, const float* partial1, int c1, int slabs1,
                     const float* gamma, const float* beta, const float* film, int film_stride,
                     float* aff_a, float* aff_b, float* stats, int n, int hw, float eps, void* stream);

/* h_upd / x_upd of an up/down ResBlock (unet.py:190-195, 237-242):
 * out = resample(act(a*in + b)), act = SiLU when aff_a != NULL, identity copy otherwise.
 * mode 1 = AvgPool2d(2) (H,W -> H/2,W/2), mode 2 = nearest x2, mode 3 = every second pixel
 * (turns a stride-1 3x3 conv into the stride-2 Downsample of the latent UNet), mode 4 = zero-insert x2
 * (out[2y][2x] = in[y][x], 0 elsewhere, no affine: the backward-data conv of a stride-2 3x3 conv -- the classifier's
 * Downsample with classifier_resblock_updown = False, unet.py:115-140 -- is the stride-1 conv with transposed, flipped
 * weights over this tensor).                                                                */
int adm_resample(const adm_bf16* in, const float* aff_a, const float* aff_b, adm_bf16* out,
                 int n, int h, int w, int c, int mode, void* stream);

/* ---------------------------------------------------------------- fused conv / GEMM (K2,K3,K4,K7,K8)
 * Implicit-GEMM 3x3 (pad 1) or 1x1 convolution on MFMA, bf16 in / fp32 accumulate:
 *   out[p, o] = bias[o] + sum_{tap, c} act(a[n,c]*in[p+tap, c] + b[n,c]) * w[o, c, tap] (+ res[p, o])
 * with the input a virtual concat of (in0 | in1) along channels (th.cat, unet.py:662),
 * prologue: 0 = raw input, 1 = affine (GroupNorm, attention norm), 2 = affine + SiLU
 * (in_layers / out_layers / out, unet.py:182-186, 206-222, 612-616); zero padding is applied
 * AFTER the activation, as in the reference.  res (bf16 NHWC [.., cout]) may be NULL
 * (skip_connection(x) + h, unet.py:256; x + h, unet.py:305).
 * out_mode 0: bf16 NHWC; 1: fp32 NCHW (the UNet head, unet.py:664-665).
 * w_packed comes from adm_pack_conv_weight.  cin (= c0 + c1), c0, c1 % 32 == 0.          */
typedef struct adm_conv_args {
  const adm_bf16* in0; const adm_bf16* in1;
  const adm_bf16* w_packed; const float* bias;
  const float* aff_a; const float* aff_b;
  const adm_bf16* res; void* out;
  int32_t n, h, w, c0, c1, cout;
  int32_t taps;      /* 9 or 1 */
  int32_t prologue;  /* 0 raw, 1 affine, 2 affine + SiLU; 3 = raw input + GroupNorm-backward EPILOGUE (backward-data convs): with res = x
                        (the GroupNorm input in front, same shape as out; not added) and aff_a / aff_b its affine,
                        out = acc * SiLU'(a x + b) and out_stats = (sum out, sum out * x): adm_gn_bwd_partial is then not needed */
  int32_t out_mode;  /* 0,1    */
  int32_t variant;   /* tiling variant: 0 = auto (5 or 6 by least Cout padding, 3 for cout <= 16); 5 = 192-wide and
                        6 = 128-wide 8-wave tiles; 3 = 16-wide (output head / stem backward); 7 = 32x32x16 MFMA kernel;
                        10 = 1x1 with the activation tile resident in LDS (auto for 1x1, cin % 64 == 0, cout >= 256) */
  float* out_stats;  /* optional: fp32 [N][slabs][cout][2] = per-(image, slab, channel) sum and sum of squares of
                        the bf16 OUTPUT, accumulated in the epilogue (slabs = adm_conv_stat_slabs(args)); the
                        consumer's GroupNorm then needs no adm_gn_partial pass over the tensor */
  const adm_bf16* w_packed32; /* optional: the same weight in the 32x32x16 fragment order
                                 (adm_pack_conv_weight32); enables variant 7, the v_mfma_f32_32x32x16_bf16
                                 kernel for 3x3 convs on maps >= 16x16 */
  int32_t in_up;     /* 1: in0 is [n][h/2][w/2][c0] and is read through a virtual nearest-neighbour 2x upsample (the
                        prologue is applied on the way): ResBlock(up=True)'s h_upd(in_layers[:-1](x)) without the
                        upsampled tensor ever existing.  3x3, bf16 output, c1 == 0, variant 0/5/6 only */
  int32_t res_up;    /* 1: res is [n][h/2][w/2][cout], added through the same virtual upsample (x_upd(x)) */
  int32_t ksplit;    /* > 1: split-K for small batches (3x3 or 1x1, bf16 output, variant 5 / 6): the K loop of every output tile is
                        cut into `ksplit` runs of (c0 + c1) / 32 / ksplit chunks (an even count) that run as separate
                        tiles into `ws`, and a reduce pass adds them in index order with bias, residual and the output
                        statistics.  Deterministic; the result depends on ksplit (fp32 summation order), not on n */
  float* ws;         /* split-K workspace: fp32 [ksplit][n*h*w][cout], caller-owned */
  int32_t up_phase;  /* 1..4 = phase (py, px) = ((up_phase-1) >> 1, (up_phase-1) & 1) of conv3x3(nearest-upsample-2x(in0)): in0 is
                        the half-resolution [n][h][w][c0] source, `out` the [n][2h][2w][cout] tensor of which this launch writes
                        the pixels (2y + py, 2x + px).  Each phase is a 2x2-tap conv of the source: w_packed holds its pre-summed
                        weights embedded in a 3x3 window (adm_pack_conv_weight of the host-made 3x3 tensor: rows {w0, w1 + w2, 0}
                        for py = 0, {0, w0 + w1, w2} for py = 1, likewise for columns); the 5 zero taps are skipped: 4/9 of the
                        MACs of in_up = 1.  up_phase = 5 runs all four phases in ONE launch: w_packed then holds the four packed weights
                        back to back (phase-major) and the phase is part of the tile index (4 x as many tiles: small batches
                        still fill the chip).  out_stats has adm_conv_stat_slabs = 4 x (h / 16) x (w / 16) slabs, one per
                        (source tile, phase).  3x3, bf16 output, c1 == 0, no residual, source >= 16x16, variant 0/5/6            */
  int32_t geglu;     /* 1: GEGLU epilogue (Stable-Diffusion feed-forward, ldm/modules/attention.py:37-44: `x, gate = proj(x).chunk(2);
                        return x * gelu(gate)`): w_packed / bias hold the projection with its output rows INTERLEAVED (row 2m = value
                        m = reference row m, row 2m + 1 = gate m = reference row cout/2 + m), and `out` is [n][h][w][cout / 2] =
                        value * gelu(gate) (exact erf GELU on the fp32 accumulators): the [.., cout] tensor is never written.  1x1 on the
                        resident-tile kernel only: raw input, no res / out_stats, cout %% 16 == 0, cout > 192                      */
  const adm_bf16* fold0;  /* non-NULL: the ResBlock's `skip_connection(x) + h` (unet.py:216-222, 256) inside this out_layers conv: after the
                             nine-tap K loop, (fc0 + fc1) / 32 one-tap raw K-steps over the block's INPUT x = (fold0 | fold1) [n][h][w][fc]
                             with the skip_connection's 1x1 weights, which follow the 3x3 weights in w_packed (adm_pack_conv_weight of
                             each, concatenated); bias = conv bias + skip bias.  3x3, prologue 2, bf16 output, no res, map 8x8 or >= 16x16,
                             variant 0 / 5 / 6.  The 1x1 launch, its output tensor and the residual read of these tiles disappear.   */
  const adm_bf16* fold1;
  int32_t fc0, fc1;
  float out_scale;   /* fp32 NCHW output (out_mode 1) only: out = (acc + bias) * out_scale, applied in fp32 in the epilogue; 0 = 1.0.
                        The classifier's LAST backward-data conv (the stem's) undoes the static 2^10 scale of an fp16 backward
                        network here -- in fp32, not in its fp16 weights, where 2^-10 pushed a quarter of them into subnormals */
} adm_conv_args;
int adm_conv(const adm_conv_args* args_host, void* stream);
/* slabs of out_stats for these arguments (0 = fused statistics not offered for this shape / variant). */
int adm_conv_stat_slabs(const adm_conv_args* args_host);
/* the tiling variant adm_conv will run for these arguments (resolves variant 0; 10 = resident-tile 1x1 kernel) */
int adm_conv_pick_variant(const adm_conv_args* args_host);

/* fp32 [cout, cin, kh, kw] (kh*kw = taps) -> bf16 fragment-ordered image
 * [cin/32][taps][ceil(cout/16)][64 lanes][8]; out must hold adm_packed_weight_elems().   */
int64_t adm_packed_weight_elems(int cout, int cin, int taps);
/* 32x32x16 fragment order [cin/32][taps][ceil(cout/32)][2][64 lanes][8]. */
int64_t adm_packed_weight32_elems(int cout, int cin, int taps);
int adm_pack_conv_weight32(const float* w, adm_bf16* out, int cout, int cin, int taps, void* stream);
int adm_pack_conv_weight(const float* w, adm_bf16* out, int cout, int cin, int taps, void* stream);

/* ---------------------------------------------------------------- attention (K5)
 * QKVAttention / QKVAttentionLegacy (unet.py:361-393 / 328-358) as a flash-style MFMA kernel:
 * qkv bf16 [N][T][3*H*D] (token-major, the 1x1 qkv conv's NHWC output) -> out bf16 [N][T][H*D];
 * softmax in fp32 over all T keys, logits scaled by 1/sqrt(D) (= the reference's
 * ch^-1/4 on q and on k).  new_order 1: channels are [3][H][D]; 0 (legacy): [H][3][D].
 * D in {32, 48, 64, 80, 96, 128} (tuned path; 48 / 80 add one 16-deep MFMA step) or {160, 192, 256} (ADM-128 / SD wide heads: sized to fit, not tuned). */
int adm_attention(const adm_bf16* qkv, adm_bf16* out, int n, int t, int heads, int d,
                  int new_order, void* stream);

/* Same, also writing the log2-domain log-sum-exp of the scaled logits, fp32 [N][H][T] (nullable):
 * P = exp2(s * log2(e)/sqrt(D) - lse).  Needed by adm_attention_bwd.                          */
int adm_attention_lse(const adm_bf16* qkv, adm_bf16* out, float* lse, int n, int t, int heads, int d,
                      int new_order, void* stream);

/* Attention with separate query and key/value tensors, heads of any supported width laid out (h d), and an explicit
 * logit scale: the SpatialTransformer's CrossAttention ("Stable Diffusion"/ldm/modules/attention.py:152-194:
 * softmax(q k^T * dim_head^-0.5) v; self-attention when kv aliases the query tensor).
 *   q   bf16 [N][tq][q_stride]       head h = columns h*d .. h*d+d
 *   kv  bf16 [N][kv_rows][kv_stride] K head h = columns h*d .., V head h = columns heads*d + h*d ..; only the first
 *                                    tk rows of every image are attended to (kv_rows >= tk is the row pitch)
 *   out bf16 [N][tq][heads*d];  scale <= 0 selects 1/sqrt(d).  d in {32, 48, 64, 80, 96, 128, 160, 192, 256}: other reference
 *   head widths (SD's 40 channels) are zero-padded by the caller's projection weights, with scale = true_dim^-0.5.     */
int adm_attention_cross(const adm_bf16* q, int q_stride, const adm_bf16* kv, int kv_stride, int kv_rows,
                        adm_bf16* out, int n, int tq, int tk, int heads, int d, float scale, void* stream);

/* ---------------------------------------------------------------- Stable-Diffusion latent UNet
 * LayerNorm over the channels of every token (torch.nn.LayerNorm(C), attention.py:205-207): bf16 [rows][C] -> bf16.  */
int adm_layernorm(const adm_bf16* x, const float* gamma, const float* beta, adm_bf16* out, int64_t rows, int c,
                  float eps, void* stream);
/* GEGLU gate (attention.py:36-44): out[r][i] = u[r][i] * gelu(u[r][inner+i]), u bf16 [rows][2*inner], exact GELU. */
int adm_geglu(const adm_bf16* u, adm_bf16* out, int64_t rows, int inner, void* stream);
/* GroupNorm affine of (x + e[n, c]) applied to the STORED x (ResBlock without scale-shift norm, openaimodel.py:253-258:
 * h = h + emb_out; h = out_layers(h)): statistics of x + e are derived from the partial sums of x, and
 * a = rstd*gamma, b = beta + (e - mean)*rstd*gamma.  add fp32 [N][add_stride].                                   */
int adm_gn_finalize_add(const float* partial, const float* gamma, const float* beta, const float* add, int add_stride,
                        float* aff_a, float* aff_b, float* stats /* nullable: (mean, rstd) of x + e per (image, group), for the backward pass */,
                        int n, int c, int hw, int slabs, float eps, void* stream);

/* One DDIM / PLMS update of the latent samplers (ddim.py:165-203, plms.py:195-258), fp32 tensors of numel elements:
 *   e     = eps_uncond ? eps_uncond + cfg_scale*(eps_cond - eps_uncond) : eps_cond        (written to e_out if given)
 *   e'    = w[0]*e + w[1]*h1 + w[2]*h2 + w[3]*h3                        (h* nullable: PLMS's old_eps, newest first)
 *   x0    = (x - sqrt_one_minus_at*e') / sqrt_at
 *   x_prev = sqrt_a_prev*x0 + dir_coef*e' + sigma*noise                 (dir_coef = sqrt(1 - a_prev - sigma^2))  */
typedef struct adm_sd_step_coefs {
  float cfg_scale;
  float w[4];
  float sqrt_one_minus_at, sqrt_at, sqrt_a_prev, dir_coef, sigma;
} adm_sd_step_coefs;
int adm_sd_step(const float* x, const float* eps_uncond, const float* eps_cond, const float* h1, const float* h2,
                const float* h3, const float* noise, float* x_prev, float* pred_x0, float* e_out, int64_t numel,
                const adm_sd_step_coefs* coefs_host, void* stream);

/* One multistep DPM-Solver++ update in data-prediction form (dpm_solver/dpm_solver.py:700-735, 755-810 with
 * model_wrapper's classifier-free guidance :289-330): e = guided eps, m = (x - sigma_s*e)/alpha_s (written to m_out, the
 * next step's model_prev), x_next = a*x + b0*m + b1*m_prev (m_prev nullable: first-order step).  fp32, numel elements. */
int adm_dpm_step(const float* x, const float* eps_uncond, const float* eps_cond, const float* m_prev, float* x_next,
                 float* m_out, int64_t numel, float cfg_scale, float sigma_s, float alpha_s, float a, float b0, float b1,
                 void* stream);

/* ---------------------------------------------------------------- classifier guidance, backward-data (K10, A9)
 * The reference gets grad_x log p(y|x,t) from torch.autograd over EncoderUNetModel
 * (search_imagenet64_classifier_guidance.py:319-326, unet.py:685-896).  Here the backward network
 * is explicit; only data gradients exist (no weight gradients).
 *
 * Attention backward (flash-style recomputation; unet.py:297 re-runs the forward too):
 * qkv / dqkv bf16 [N][T][3*H*D], out / dout bf16 [N][T][H*D], lse from adm_attention_lse,
 * delta_ws fp32 [N][H][T] workspace.  D in {32, 64}.                                          */
int adm_attention_bwd(const adm_bf16* qkv, const adm_bf16* out, const adm_bf16* dout, const float* lse,
                      float* delta_ws, adm_bf16* dqkv, int n, int t, int heads, int d, int new_order,
                      void* stream);

/* GroupNorm(+FiLM)(+SiLU) backward for y = act(a*x + b) with (a, b, stats) from adm_gn_finalize:
 *   dx = a*dz + k1*x + k0 (+ add),  dz = dy * SiLU'(a*x+b)  (dz = dy when silu == 0).
 * partial: fp32 [N][slabs][C][2]; k1/k0: fp32 [N][C].  dy_half / add_half: that tensor lives at
 * (h/2, w/2) and is read as 0.25 * t[y/2, x/2] (AvgPool2d backward of down ResBlocks).          */
int adm_gn_bwd_partial(const adm_bf16* x, const adm_bf16* dy, const float* aff_a, const float* aff_b,
                       float* partial, int n, int h, int w, int c, int slabs, int silu, int dy_half,
                       void* stream);
/* add (nullable, fp32 [N][add_stride]): the layer normalised x + add[n, c] (adm_gn_finalize_add) while x is what was stored and
 * summed; the sums are corrected (sum dz (x + e) = sum dz x + e sum dz) and k0 absorbs k1 * e, so adm_gn_bwd_apply runs unchanged. */
int adm_gn_bwd_finalize(const float* partial, const float* aff_a, const float* stats, const float* add, int add_stride,
                        float* k1, float* k0, int n, int c, int hw, int slabs, void* stream);
int adm_gn_bwd_apply(const adm_bf16* x, const adm_bf16* dy, const float* aff_a, const float* aff_b,
                     const float* k1, const float* k0, const adm_bf16* add, adm_bf16* out,
                     int n, int h, int w, int c, int silu, int dy_half, int add_half, void* stream);
/* out = a + (b_half ? 0.25 * b[y/2, x/2] : b): gradient accumulation at a fan-out.             */
int adm_grad_add(const adm_bf16* a, const adm_bf16* b, adm_bf16* out, int n, int h, int w, int c,
                 int b_half, void* stream);

/* dlogits = scale * (onehot(y) - softmax(logits)) = scale * d/dlogits sum_n log_softmax[n, y_n];
 * logp_sel (nullable) receives log_softmax[n, y_n].  fp32 [N][K].                              */
int adm_logsoftmax_grad(const float* logits, const int64_t* y, float* dlogits, float* logp_sel,
                        int n, int k, float scale, void* stream);

/* AttentionPool2d (unet.py:22-51) around its single used query (token 0):
 *   adm_pool_prep     tok[n,0,:] = mean_p act + pos[:,0]; tok[n,1+p,:] = act[n,p,:] + pos[:,1+p],
 *                     act = SiLU(a*h + b); tok bf16 [N][tpad][C] (rows >= HW+1 zero); pos fp32 [C][HW+1]
 *   (qkv_proj = adm_conv 1x1 on tok viewed as 8x8 maps)
 *   adm_pool_attn_fwd a0[n,:] = attention output of token 0 (fp32 [N][C]); wts fp32 [N][H][tpad]
 *   (c_proj = adm_linear_f32)
 *   adm_pool_attn_bwd dqkv bf16 [N][tpad][3C] from da0 fp32 [N][C]
 *   adm_pool_prep_bwd d_act[n,p,:] = dtok[n,1+p,:] + dtok[n,0,:]/HW                               */
int adm_pool_prep(const adm_bf16* h, const float* aff_a, const float* aff_b, const float* pos, adm_bf16* tok,
                  int n, int hw, int c, int tpad, void* stream);
int adm_pool_attn_fwd(const adm_bf16* qkv, float* a0, float* wts, int n, int t, int tpad, int heads, int d,
                      void* stream);
int adm_pool_attn_bwd(const adm_bf16* qkv, const float* wts, const float* da0, adm_bf16* dqkv, int n, int t,
                      int tpad, int heads, int d, void* stream);
int adm_pool_prep_bwd(const adm_bf16* dtok, adm_bf16* dact, int n, int hw, int c, int tpad, void* stream);

/* Backward-data weight image for adm_conv: the conv with cin' = cout, cout' = cin and
 * w'[ci][co][t] = w[co][ci][taps-1-t].  out holds adm_packed_weight_elems(cin, cout, taps).     */
int adm_pack_conv_weight_bwd(const float* w, adm_bf16* out, int cout, int cin, int taps, void* stream);

/* ---------------------------------------------------------------- FID statistics (K11, A11)
 * Streaming float64 accumulation of s1[j] += sum_n a[n][j], s2[i][j] += sum_n a[n][i]*a[n][j] over a
 * batch of fp32 activations [n][d] (np.mean / np.cov, evaluations/evaluator_v1.py:218-221).
 * s1 fp64 [d], s2 fp64 [d][d], zero-initialised by the caller before the first batch.        */
int adm_fid_accumulate(const float* acts, double* s1, double* s2, int n, int d, void* stream);

/* ---------------------------------------------------------------- Inception-v3 pool3 features (K12, A11 compute_activations)
 * The layers of the FID feature extractor the reference runs through third-party code: a frozen TensorFlow graph
 * (evaluations/evaluator_v1.py:263-269, 665-679) / pytorch_fid.inception.InceptionV3 (Stable Diffusion
 * scripts/search_ea.py:95-127, 171-182).  BasicConv2d = conv(no bias) + BatchNorm(eps 1e-3) + ReLU: the caller folds the
 * BatchNorm scale into the packed weights and passes the shift as `bias`.
 *
 * adm_conv2d: general 2-D convolution, NHWC.  `in` = [n][h][w_in][in_stride] elements of which channels [0, cin_pad) are
 * read (cin_pad % 32 == 0; channels beyond the layer's real cin must hold zeros or meet zero weights); `w` from
 * adm_pack_conv2d_weight = [cout][kh*kw][cin_pad]; `out` = [n][oh][ow][out_stride] of which channels [0, cout) are written
 * -- point it at a channel slice of a concatenated tensor (cout % 4 == 0, 8-byte aligned).  oh = (h + 2 pad_h - kh)/stride + 1. */
typedef struct adm_conv2d_args {
  const adm_bf16* in; const adm_bf16* w; const float* bias; adm_bf16* out;
  int32_t n, h, w_in, cin_pad, in_stride, cout, out_stride, kh, kw, stride, pad_h, pad_w;
  int32_t relu;      /* 1: max(x, 0) after the bias */
} adm_conv2d_args;
int adm_conv2d(const adm_conv2d_args* args, void* stream);
/* fp32 [cout][cin][kh][kw] (torch Conv2d.weight), optionally times scale[cout], -> the 16-bit layout above */
int adm_pack_conv2d_weight(const float* w, const float* scale, adm_bf16* out, int cout, int cin, int kh, int kw, int cin_pad,
                           void* stream);
/* k x k pooling of a channel slice (c % 8 == 0).  mode 0: max (F.max_pool2d), 1: average over the taps inside the image
 * (F.avg_pool2d(count_include_pad=False), the FID variant of the Inception blocks)                                        */
int adm_pool2d(const adm_bf16* in, adm_bf16* out, int n, int h, int w, int c, int in_stride, int out_stride, int k, int stride,
               int pad, int mode, void* stream);
/* mean over the hw pixels: [n][hw][c] -> fp32 [n][c] (the pool3 features)                                                 */
int adm_global_avgpool_f32(const adm_bf16* in, float* out, int n, int hw, int c, void* stream);
/* 3-channel images -> [n][oh][ow][cpad] (channels 3.. zero), value * scale + shift.  kind 0: uint8 NHWC, 1: fp32 NCHW,
 * 2: fp32 NHWC.  half_pixel 1: torch bilinear, align_corners=False (pytorch_fid); 0: TensorFlow-1 ResizeBilinear.          */
int adm_resize_bilinear(const void* in, adm_bf16* out, int n, int h, int w, int oh, int ow, int cpad, int kind, int half_pixel,
                        float scale, float shift, void* stream);

/* ---------------------------------------------------------------- the classifier's other heads (A9 variants)
 * EncoderUNetModel pool = "adaptive" | "spatial" | "spatial_v2" (guided_diffusion/unet.py:826-856, 880-896); create_classifier's
 * default and every launch script use "attention" (above).  The Linear layers between these are adm_linear_f32.
 *   adm_channel_mean  out[n][ch] = mean_p act(h[n][p][ch]), act = SiLU(a h + b) with the affine (adaptive: GN -> SiLU -> AvgPool),
 *                     identity without (spatial: h.mean(dim=(2, 3)) of every block, written at a column offset of the feature row)
 *   adm_bcast_add     out[n][p][ch] = (add ? add[n][p][ch] : 0) + v[n][ch] * scale: the backward of a pixel mean (scale = 1 / HW)
 *                     accumulated into the gradient that flows through the same tensor
 *   adm_vec_act       mode 1 SiLU, 2 ReLU on fp32 vectors: out = act(x), or with dy: out = dy * act'(x)
 *   adm_vec_gn(_bwd)  GroupNorm32(32, C) of fp32 [N][C] rows (spatial_v2's normalization(2048): no spatial axis), stats fp32
 *                     [N][32][2] = (mean, rstd); backward-data dx from dz = d(loss)/d(y)                                    */
int adm_channel_mean(const adm_bf16* h, const float* aff_a, const float* aff_b, float* out, int out_stride, int n, int hw, int c,
                     void* stream);
int adm_bcast_add(const float* v, int v_stride, float scale, const adm_bf16* add, adm_bf16* out, int n, int hw, int c, void* stream);
int adm_vec_act(const float* x, const float* dy, float* out, int64_t items, int mode, void* stream);
int adm_vec_gn(const float* x, const float* gamma, const float* beta, float* y, float* stats, int n, int c, float eps, void* stream);
int adm_vec_gn_bwd(const float* x, const float* gamma, const float* stats, const float* dz, float* dx, int n, int c, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ADM_HIP_H */
