/*
 * condmdi.h — C-ABI of libcondmdi_hip.so, the MI355X (gfx950) engine for the CondMDI
 * diffusion-sampling hot path.
 *
 * The reference (setarehc/diffusion-motion-inbetweening) has no FFI: its hot path sits behind a
 * Python API.  This header is what a binding of that API to native code needs; every entry point
 * names the reference interface it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - every pointer named d_* is a DEVICE pointer (HIP), everything else is a host pointer;
 *   - all floating point is IEEE fp32, timesteps are int64, masks are uint8 (0/1);
 *   - motion tensors use the reference layout [B, J, 1, T] with T contiguous (J*1 = n_feats);
 *   - every call returns 0 on success or a negative CMDI_E_* code; cmdi_last_error() gives text;
 *   - calls only ENQUEUE work on `stream` (a hipStream_t); they never synchronise and never throw;
 *   - the caller owns every buffer it passes; the library owns only its workspace and its private
 *     copies of weights / schedule / condition tensors (copied at the set_* / load_* call).
 */
#ifndef CONDMDI_H
#define CONDMDI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cmdi_engine* cmdi_handle;
typedef void* cmdi_stream; /* hipStream_t */

enum {
    CMDI_OK = 0,
    CMDI_E_INVALID = -1,   /* bad argument / unsupported dimension           */
    CMDI_E_STATE = -2,     /* call order violated (weights/schedule missing) */
    CMDI_E_HIP = -3,       /* a HIP runtime call failed                      */
    CMDI_E_NOMEM = -4,     /* workspace allocation failed                    */
    CMDI_E_UNKNOWN_WEIGHT = -5,
    CMDI_E_RANGE = -6      /* a value left the f16 range of the split-f16 GEMM path      */
};

/* Arithmetic of the encoder-layer GEMMs.  Inputs, outputs and accumulation are IEEE fp32 in both
 * modes; they differ in how the products are formed:
 *   CMDI_PREC_F32    v_mfma_f32_32x32x2_f32, exact fp32 products (157 TFLOP/s peak);
 *   CMDI_PREC_F16X3  every fp32 operand is carried as two f16 (hi, lo*2^11; 22 significant bits) and
 *                    a product is three v_mfma_f32_32x32x16_f16 accumulated in fp32 — error below
 *                    the fp32 accumulation roundoff of the same dot product, valid for |x| < 65504
 *                    (checked: CMDI_E_RANGE at cmdi_finalize_weights, cmdi_range_status after a run).
 *   CMDI_PREC_BF16X6 every fp32 operand is carried EXACTLY as three bf16 (8 + 8 + 8 = 24 significant bits, fp32's
 *                    exponent range: no range limit) and a product is six v_mfma_f32_32x32x16_bf16 (the three dropped
 *                    partial products are < 2^-26 of the product), fp32 accumulate, leading and small terms apart —
 *                    fp32-class results without operand truncation at 2.65x the fp32-MFMA peak (csrc/gemm_x6.hpp).
 *                    Attention, LayerNorm and the I/O projections run the CMDI_PREC_F32 kernels.
 *   CMDI_PREC_DEFAULT  the environment variable CMDI_PRECISION = f32 | bf16x6 | f16x3 if set, else the library default
 *                    (cmdi_precision reports it). */
enum { CMDI_PREC_DEFAULT = 0, CMDI_PREC_F32 = 1, CMDI_PREC_F16X3 = 2, CMDI_PREC_BF16X6 = 3 };
enum { CMDI_ARCH_TRANS_ENC = 0, CMDI_ARCH_UNET = 1 };

/* Model geometry.  Replaces the keyword arguments of MDM.__init__ (model/mdm.py:11-36) that the
 * trans_enc / hml_vec path reads, as produced by get_model_args (utils/model_util.py:40-119). */
typedef struct {
    int32_t n_layers;    /* num_layers   (8)                                   */
    int32_t d_model;     /* latent_dim   (512; must be a multiple of 128)      */
    int32_t d_ff;        /* ff_size      (1024; multiple of 128)               */
    int32_t n_heads;     /* num_heads    (4); d_model / n_heads must be 128    */
    int32_t n_feats;     /* njoints*nfeats (263)                               */
    int32_t max_frames;  /* largest T this engine will see (<= 223)            */
    int32_t max_batch;   /* largest B (samples, before the CFG doubling)       */
    int32_t pe_rows;     /* rows of sequence_pos_encoder.pe (5000)             */
    int32_t text_cond;   /* 1 if cond_mode contains 'text' (embed_text exists) */
    int32_t want_grad;   /* 1: allocate the activation stash for cmdi_mdm_vjp (both architectures) */
    int32_t precision;   /* CMDI_PREC_*                                        */
    int32_t arch;        /* CMDI_ARCH_TRANS_ENC (MDM, model/mdm.py) or CMDI_ARCH_UNET (MDM_UNET, model/mdm_unet.py:561-849;
                            n_layers / d_ff / n_heads unused, d_model = latent_dim = 512)                        */
    int32_t unet_added;  /* UNET: extra input channels = n_feats if keyframe_conditioned (cat(x, obs_mask)) else 0 */
    int32_t unet_mults[4]; /* UNET: dim_mults (channels = d_model * mult; the built configuration has equal mults) */
    int32_t unet_attention; /* UNET: 1 = attention=True, Residual(PreNorm(LinearAttention)) sites (model/mdm_unet.py:102-156,262,273,298) */
} cmdi_model_desc;

/* Model-output → x0 conventions (diffusion/gaussian_diffusion.py:74-95). */
enum { CMDI_MEAN_START_X = 0, CMDI_MEAN_EPSILON = 1 };
enum { CMDI_SAMPLER_DDPM = 0, CMDI_SAMPLER_DDIM = 1 };

/* ---- lifetime ------------------------------------------------------------------------------ */

/* Replaces MDM(**get_model_args(args, data)) for arch='trans_enc' (utils/model_util.py:26-37,
 * model/mdm.py:105-114,136-165): allocates weight storage and the activation workspace. */
int cmdi_create(const cmdi_model_desc* desc, cmdi_handle* out);
int cmdi_destroy(cmdi_handle h);
const char* cmdi_last_error(void);
const char* cmdi_version(void);

/* Replaces load_model_wo_clip / nn.Module.load_state_dict (utils/model_util.py:19-23,168-182).
 * `name` is the reference state-dict key, e.g.
 * "seqTransEncoder.layers.3.self_attn.in_proj_weight"; `d_src` is a contiguous fp32 DEVICE tensor
 * of `numel` elements in the reference's own layout.  Copies on `stream`. */
int cmdi_load_weight(cmdi_handle h, const char* name, const float* d_src, int64_t numel,
                     cmdi_stream stream);
/* Call once after all weights: packs padded/transposed copies and precomputes
 * TimestepEmbedder.forward (model/mdm.py:351-353) for every original timestep < n_time_rows. */
int cmdi_finalize_weights(cmdi_handle h, int32_t n_time_rows, cmdi_stream stream);

/* ---- schedule -------------------------------------------------------------------------------
 * Replaces the float64 tables of GaussianDiffusion.__init__ (diffusion/gaussian_diffusion.py:
 * 184-217) after SpacedDiffusion's respacing (diffusion/respace.py:74-91) and the per-step
 * _extract_into_tensor(...).float() casts (:2215-2229).  All arrays have n_steps entries, already
 * cast to fp32 by the host; timestep_map is respace.py:78-88's list (respaced index → original t).
 */
typedef struct {
    int32_t n_steps;
    int32_t mean_type;               /* CMDI_MEAN_*                                          */
    const float* post_coef1;         /* posterior_mean_coef1                                 */
    const float* post_coef2;         /* posterior_mean_coef2                                 */
    const float* sigma;              /* exp(0.5 * model_log_variance) for the model_var_type */
    const float* sqrt_ab;            /* sqrt_alphas_cumprod                                  */
    const float* sqrt_1mab;          /* sqrt_one_minus_alphas_cumprod                        */
    const float* sqrt_recip_ab;      /* sqrt_recip_alphas_cumprod                            */
    const float* sqrt_recipm1_ab;    /* sqrt_recipm1_alphas_cumprod                          */
    const float* ab;                 /* alphas_cumprod                                       */
    const float* ab_prev;            /* alphas_cumprod_prev                                  */
    const int64_t* timestep_map;     /* respaced i → original t                              */
    float clip_x0;                   /* > 0 (CMDI_MEAN_EPSILON only): clamp the derived x0 to [-clip_x0, clip_x0] —
                                        process_xstart with clip_denoised for abs_3d trajectory models
                                        (diffusion/gaussian_diffusion.py:489-505); 0 = no clamp          */
} cmdi_schedule;
int cmdi_set_schedule(cmdi_handle h, const cmdi_schedule* s);

/* ---- per-call conditioning ------------------------------------------------------------------
 * Replaces model_kwargs['y'] as read on the path: text embedding (the output of
 * MDM.encode_text, model/mdm.py:211-237, [B,512]) and text_scale for ClassifierFreeSampleModel
 * (model/cfg_sampler.py:25-35); inpainting_mask & y['mask'] and inpainted_motion for imputation /
 * reconstruction guidance (diffusion/gaussian_diffusion.py:405-435); the host-evaluated gates of
 * utils/editing_util.py:325-346 and the per-step w_r = grad_ws[i] * reconstruction_weight
 * (:418-420, editing_util.py:299-322). */
typedef struct {
    int32_t batch;                   /* B                                                      */
    int32_t n_frames;                /* T                                                      */
    int32_t cfg;                     /* 1: ClassifierFreeSampleModel semantics (2 passes)      */
    const float* d_enc_text;         /* [B, clip_dim=512] or NULL (no_cond / all-uncond)       */
    const float* d_text_scale;       /* [B] or NULL (required when cfg=1)                      */
    const uint8_t* d_inpaint_mask;   /* [B,J,1,T] already AND-ed with y['mask'], or NULL       */
    const float* d_inpaint_motion;   /* [B,J,1,T] or NULL                                      */
    int32_t imputate;                /* y['imputate']: 0 off; 1 = replacement_distribution 'conditional' (impute at every
                                        step >= stop_imputation_at); 2 = 'marginal': a no-op in the reference's plain
                                        branch (:437-439), so impute only at steps where reconstruction guidance runs
                                        (its branch imputes whatever the distribution, :424)              */
    int32_t stop_imputation_at;      /* y['stop_imputation_at'] (respaced index)               */
    int32_t recon_guidance;          /* y['reconstruction_guidance']                           */
    int32_t stop_recguidance_at;     /* y['stop_recguidance_at']                               */
    const float* recon_w;            /* HOST [n_steps]: grad_ws[i]*reconstruction_weight, or NULL */
    const float* d_obs_x0;           /* UNET: model_kwargs['obs_x0'] [B,J,1,T] or NULL (mdm_unet.py:766-782)  */
    const uint8_t* d_obs_mask;       /* UNET: model_kwargs['obs_mask'] [B,J,1,T] (0/1) or NULL                */
} cmdi_condition;
int cmdi_set_condition(cmdi_handle h, const cmdi_condition* c, cmdi_stream stream);

/* ---- denoiser -------------------------------------------------------------------------------
 * Replaces MDM.forward (model/mdm.py:239-306), and ClassifierFreeSampleModel.forward
 * (model/cfg_sampler.py:25-35) when the condition was set with cfg=1.
 * d_x [B,J,1,T]; d_t int64[B] ORIGINAL timesteps (after _WrappedModel, respace.py:128-133);
 * d_out [B,J,1,T].  If d_out_uncond != NULL (cfg only) the two raw passes are returned instead of
 * the guided combination: d_out = conditional, d_out_uncond = unconditional. */
int cmdi_mdm_forward(cmdi_handle h, const float* d_x, const int64_t* d_t, float* d_out,
                     float* d_out_uncond, cmdi_stream stream);

/* Vector-Jacobian product of the (CFG-combined) denoiser output w.r.t. d_x, evaluated at the
 * inputs of the LAST cmdi_mdm_forward on this handle (needs want_grad=1).  Replaces
 * torch.autograd.grad(loss, z) in the reconstruction-guidance block
 * (diffusion/gaussian_diffusion.py:411-416).  d_gout, d_gx: [B,J,1,T].  CMDI_ARCH_UNET: the same through the
 * U-Net (entries of x replaced by obs_x0 get gradient 0, as x = obs*m + x*~m at model/mdm_unet.py:781). */
int cmdi_mdm_vjp(cmdi_handle h, const float* d_gout, float* d_gx, cmdi_stream stream);

/* ---- sampler --------------------------------------------------------------------------------
 * One denoising step at respaced index `step`: p_sample (diffusion/gaussian_diffusion.py:656-713)
 * or ddim_sample (:1300-1416) including p_mean_variance's imputation / reconstruction-guidance
 * branches (:405-445).  d_x is x_t on entry and x_{t-1} on exit.  d_noise [B,J,1,T] is the draw
 * th.randn_like(x) of :696; if NULL the engine draws it itself (Philox4x32-10 keyed by
 * (seed, first_sample + b, step, element), independent of how the batch is sharded).
 * d_pred_xstart (optional) receives out["pred_xstart"]. */
int cmdi_step(cmdi_handle h, int32_t sampler, int32_t step, float eta, float* d_x,
              float* d_pred_xstart, const float* d_noise, uint64_t seed, int64_t first_sample,
              cmdi_stream stream);

/* The loop of p_sample_loop_progressive / ddim_sample_loop_progressive
 * (diffusion/gaussian_diffusion.py:1270-1297,1564-1587): for step = first_step .. last_step
 * (descending, inclusive) call cmdi_step.  d_noise_stream, if not NULL, holds one [B,J,1,T] draw
 * per step in loop order (the injected-noise parity mode of SURVEY.md §8c). */
int cmdi_sample_loop(cmdi_handle h, int32_t sampler, int32_t first_step, int32_t last_step,
                     float eta, float* d_x, const float* d_noise_stream, uint64_t seed,
                     int64_t first_sample, cmdi_stream stream);

/* hipGraph replay for cmdi_sample_loop (also CMDI_GRAPH=1 in the environment): per-step scalars move to
 * device tables indexed by a device cursor, one step's launch sequence (all streams) is captured once
 * per kind (with / without reconstruction guidance) and replayed.  Same kernels, same values: results
 * are bitwise identical to the eager loop.  Ignored (eager launches) when a noise stream is injected. */
int cmdi_set_graph(cmdi_handle h, int32_t on);

/* Sampler arithmetic alone, for denoisers that are not the native MDM (any callable model):
 * given the model output (already CFG-combined) apply imputation, the x0/eps conversion and the
 * posterior / DDIM update.  Same semantics as cmdi_step minus the model call; the
 * reconstruction-guidance gradient, if any, is passed in d_recon_grad (already masked). */
int cmdi_sampler_update(cmdi_handle h, int32_t sampler, int32_t step, float eta,
                        const float* d_model_out, const float* d_recon_grad, float* d_x,
                        float* d_pred_xstart, const float* d_noise, uint64_t seed,
                        int64_t first_sample, cmdi_stream stream);

/* q_sample (diffusion/gaussian_diffusion.py:311-328): d_out = sqrt_ab[step]*x0 + sqrt_1mab[step]*noise. */
int cmdi_q_sample(cmdi_handle h, int32_t step, const float* d_x0, const float* d_noise,
                  float* d_out, int64_t numel, cmdi_stream stream);

/* Standard-normal fill with the engine's counter-based generator (same keying as cmdi_step with
 * step = -1); used for x_T when the caller passes noise=None (gaussian_diffusion.py:1248). */
int cmdi_randn(cmdi_handle h, float* d_out, int32_t batch, int64_t per_sample, uint64_t seed,
               int64_t first_sample, int32_t step, cmdi_stream stream);

/* ---- after the loop -------------------------------------------------------------------------
 * The step every caller runs right after p_sample_loop (sample/conditional_synthesis.py:229-235,
 * sample/edit.py, sample/synthesize.py): t2m_dataset.inv_transform (data * std + mean,
 * data_loaders/humanml/data/dataset.py:378-382) followed by recover_from_ric
 * (data_loaders/humanml/scripts/motion_process.py:402-441,474-491) and the final permute, on the device:
 * d_sample [B, n_feats, 1, T] -> d_xyz [B, n_joints, 3, T].  d_mean / d_std [n_feats] device pointers, or
 * both NULL if the sample is already un-normalised.  abs_3d as in the reference (absolute root yaw / XZ). */
int cmdi_recover_xyz(const float* d_sample, const float* d_mean, const float* d_std, float* d_xyz,
                     int32_t batch, int32_t n_feats, int32_t n_frames, int32_t n_joints, int32_t abs_3d,
                     cmdi_stream stream);

/* ---- before the loop: the CLIP text tower ---------------------------------------------------------
 * Replaces clip_model.encode_text(tokens).float() of MDM.encode_text (model/mdm.py:211-237; clip.load('ViT-B/32') at
 * :173-186): openai/CLIP's text transformer (token + positional embedding, `layers` pre-LN residual attention blocks with
 * a causal mask and QuickGELU MLPs, ln_final, the end-of-text row @ text_projection), fp32, on the device.  Weights go in
 * under openai/CLIP's own state-dict names ("token_embedding.weight", "positional_embedding",
 * "transformer.resblocks.<l>.attn.in_proj_weight", ..., "ln_final.weight", "text_projection") as fp32 device tensors;
 * d_tokens are the int32 ids of clip.tokenize [batch, context]; d_out receives [batch, embed_dim], which is what
 * cmdi_condition.d_enc_text takes. */
typedef struct cmdi_clip_text* cmdi_clip_handle;
typedef struct {
    int32_t vocab_size;   /* 49408 */
    int32_t width;        /* transformer_width 512 (= heads * 64) */
    int32_t heads;        /* 8 */
    int32_t layers;       /* 12 */
    int32_t context;      /* context_length 77 */
    int32_t embed_dim;    /* 512 */
    int32_t max_batch;
} cmdi_clip_desc;
int cmdi_clip_create(const cmdi_clip_desc* desc, cmdi_clip_handle* out);
int cmdi_clip_destroy(cmdi_clip_handle h);
int cmdi_clip_load_weight(cmdi_clip_handle h, const char* name, const float* d_src, int64_t numel, cmdi_stream stream);
int cmdi_clip_encode_text(cmdi_clip_handle h, const int32_t* d_tokens, int32_t batch, float* d_out, cmdi_stream stream);

/* ---- introspection for tests / bench --------------------------------------------------------- */
/* Raw NT GEMM used by every projection: C[M,N] = epi(A[M,K] · W[N,K]^T + bias[N]); fp32 MFMA.
 * epi: 0 = bias, 1 = bias + GELU(erf), 3 = bias + residual d_resid[M,N].  tile selects the block
 * shape / pipeline variant (0 = the engine's heuristic).  K % 32 == 0, N % 32 == 0. */
int cmdi_gemm_nt(const float* d_a, const float* d_w, const float* d_bias, const float* d_resid,
                 float* d_c, int32_t m, int32_t n, int32_t k, int32_t epi, int32_t tile,
                 cmdi_stream stream);
/* The precision the handle runs at (CMDI_PREC_F32, CMDI_PREC_F16X3 or CMDI_PREC_BF16X6; a CMDI_ARCH_UNET handle is
 * CMDI_PREC_F16X3 or, since round 5, CMDI_PREC_BF16X6 — the fp32-MFMA engine is not built for that architecture). */
int cmdi_precision(cmdi_handle h);
/* Status bits raised on the device since the last call (cleared by it):
 *   bit 0 (F16X3 only): an activation left the f16 range (|x| >= 65504 or non-finite) while being split; the results
 *         of that run are invalid and the caller should re-run on an engine created with CMDI_PREC_BF16X6 (exact operands,
 *         fp32's exponent range; or CMDI_PREC_F32).  CMDI_ARCH_UNET: CMDI_PREC_BF16X6 likewise (unet_attention = 0);
 *   bit 1: a timestep outside [0, n_time_rows) reached the time-embedding lookup (the reference raises IndexError
 *         at pe[timesteps], model/mdm.py:352); the row was clamped.
 * SYNCHRONISES `stream` (one 4-byte read-back); call it once per sampling chain, not per step. */
int cmdi_range_status(cmdi_handle h, int32_t* out_flag, cmdi_stream stream);
/* Forget whatever the status flag holds (events of earlier, unrelated calls), ordered on `stream`, without a read-back:
 * the sampling loops call it before their first step so that a chain reports its own events only. */
int cmdi_range_clear(cmdi_handle h, cmdi_stream stream);
/* Split-f16 GEMM family alone (test / bench hooks).  cmdi_split_f16: fp32 [rows, cols] -> split rows
 * [rows, 2*cols] f16; per 32-column chunk: 32 hi values f16(x), then 32 lo values
 * f16((x - hi) * 2^11); cols % 32 == 0.
 * cmdi_gemm_h3: C[M,N] = epi(A[M,K] · W[N,K]^T + bias[N]) with A, W in split rows; epi as in
 * cmdi_gemm_nt (1 = bias + GELU writes split rows to d_c_split; 0 writes fp32 to d_c, or split rows
 * if d_c_split != NULL; 3 adds the fp32 residual d_resid [M,N]; 4 adds a residual given as split rows [M,2N] in
 * d_resid's place); K % 32 == 0, N % 32 == 0. */
int cmdi_split_f16(const float* d_src, void* d_dst, int64_t rows, int32_t cols, cmdi_stream stream);
int cmdi_gemm_h3(const void* d_a_split, const void* d_w_split, const float* d_bias,
                 const float* d_resid, float* d_c, void* d_c_split, int32_t m, int32_t n, int32_t k,
                 int32_t epi, int32_t tile, cmdi_stream stream);
/* bf16x6 GEMM alone (test / bench hooks).  cmdi_pack_x6: fp32 W [rows, cols] -> three bf16 planes
 * [rows][cols/32][3][32] with W = p0 + p1 + p2 exactly (cols % 32 == 0; d_dst holds rows * cols * 6 bytes).
 * cmdi_gemm_x6: C[M,N] = epi(A[M,K] · W[N,K]^T + bias[N]) with A plain fp32 and W packed; epi as cmdi_gemm_nt
 * (0 bias, 1 bias + GELU, 3 bias + residual); variant 0 / 1 = the two K-loop schedules of gemm_x6.hpp. */
int cmdi_pack_x6(const float* d_src, void* d_dst, int64_t rows, int32_t cols, cmdi_stream stream);
int cmdi_gemm_x6(const float* d_a, const void* d_w_packed, const float* d_bias, const float* d_resid, float* d_c,
                 int32_t m, int32_t n, int32_t k, int32_t epi, int32_t variant, cmdi_stream stream);
/* Self-attention core alone (test / bench hook): d_qkv [n_seq*S, 3*H*128] -> d_out [n_seq*S, H*128]. */
int cmdi_attention_fwd(const float* d_qkv, float* d_out, int32_t n_seq, int32_t seq_len,
                       int32_t n_heads, cmdi_stream stream);
/* 1-D convolution over token rows as a split-f16 GEMM (building block of the UNET denoiser, test hook):
 * activations are rows [.., a_ld halves] in split format, sequences framed by zero halo rows (tp rows per
 * sequence, valid positions [t_lo, t_hi)); output row m * c_row_mul + c_row_add =
 * bias + sum_tap W_tap · A[a_row_mul * m - pad + tap]; weights [n, 2 * taps * cin] split rows with K in CHUNK-MAJOR
 * order (round 5): K column (chunk * taps + tap) * 32 + c holds tap `tap` of input channel chunk * 32 + c, so the taps of
 * one 32-channel chunk are consecutive K steps and their shifted re-reads of the same activation lines hit the L2.
 * Stride-2 convolution: a_row_mul = 2; transposed convolution: c_row_mul = 2.
 * tile: 0 = the library's choice, 50 = the persistent kernel (fp32 output only, n % 256 == 0), 51 = the persistent kernel
 * over the frames [t_lo, t_hi) of every sequence only (m = whole framed sequences; plain convolutions) — same bits all. */
int cmdi_conv_rows_h3(const void* d_a_split, int32_t a_ld, const void* d_w_split, const float* d_bias,
                      const float* d_resid, float* d_c, void* d_c_split, int32_t m, int32_t n, int32_t cin,
                      int32_t taps, int32_t pad, int32_t a_row_mul, int32_t c_row_mul, int32_t c_row_add,
                      int32_t tp, int32_t t_lo, int32_t t_hi, int32_t tile, cmdi_stream stream);
/* The same convolution over token rows with EXACT operands (bf16x6, round 5: the U-Net's CMDI_PREC_BF16X6 mode; replaces
 * torch Conv1d / ConvTranspose1d of the reference's plain-fp32 U-Net at any activation scale, model/mdm_unet.py:15-99,561-849):
 * activations are plain fp32 rows [.., a_ld floats], weights cmdi_pack_x6 of the [n, taps * cin] matrix in the chunk-major K
 * order of cmdi_conv_rows_h3; row / frame arguments as there.  The result (+ d_resid[row][n] when given) goes to d_c [.., n]
 * and / or d_c2 [.., ldc2] (fp32 both).  variant: 2 = the rotated K step of gemm_x6.hpp, 0 = the compiler's schedule. */
int cmdi_conv_rows_x6(const float* d_a, int32_t a_ld, const void* d_w_packed, const float* d_bias, const float* d_resid,
                      float* d_c, float* d_c2, int32_t ldc2, int32_t m, int32_t n, int32_t cin, int32_t taps, int32_t pad,
                      int32_t a_row_mul, int32_t c_row_mul, int32_t c_row_add, int32_t tp, int32_t t_lo, int32_t t_hi,
                      int32_t variant, cmdi_stream stream);
/* Y = LayerNorm((A · W^T + bias) + resid; gamma, beta, eps 1e-5) with the normalisation fused into the
 * GEMM epilogue (N must be 512 = d_model); d_y fp32 [M,N], d_y_split optional split rows [M,2N]. */
int cmdi_gemm_h3_ln(const void* d_a_split, const void* d_w_split, const float* d_bias,
                    const float* d_resid, const float* d_gamma, const float* d_beta, float* d_y,
                    void* d_y_split, int32_t m, int32_t n, int32_t k, cmdi_stream stream);
/* The same on the f16 matrix pipe (split-f16 products): d_qkv_split = cmdi_split_f16 of d_qkv. */
int cmdi_attention_fwd_h3(const void* d_qkv_split, float* d_out, int32_t n_seq, int32_t seq_len,
                          int32_t n_heads, cmdi_stream stream);
/* Input-VJP of the self-attention core alone on the f16 matrix pipe (test / bench hook; in the engine it runs inside
 * cmdi_mdm_vjp and replaces torch.autograd through torch MultiheadAttention at diffusion/gaussian_diffusion.py:411-416):
 * d_dqkv_split [n_seq*S, 6*H*128] split rows = (d out / d qkv)^T · d_dout, with d_dout fp32 [n_seq*S, H*128].  The forward
 * pass is re-run first for its row statistics.  d_work: n_seq*S*H*128*2 + n_seq*H*(2*S + 96*ceil(S/32)) + 4 floats of scratch
 * (16-B aligned; the + 4 lets the tile-statistics block start on 16 bytes when n_seq*H*S is odd). */
int cmdi_attention_vjp_h3(const void* d_qkv_split, const float* d_dout, void* d_dqkv_split, float* d_work,
                          int32_t n_seq, int32_t seq_len, int32_t n_heads, cmdi_stream stream);
/* Philox4x32-10 raw block (host, for known-answer tests): out[4] = philox(counter[4], key[2]). */
void cmdi_philox4x32_10(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]);
/* Live timing of the dominant kernel (the self-attention in_proj GEMM, one launch per layer):
 * while enabled, every such launch is bracketed by a pair of HIP events on its own stream.
 * cmdi_profile_read waits for the recorded events and returns their summed duration and count,
 * then clears them.  Used by bench.py's roofline leg; off by default (no events, no overhead). */
int cmdi_profile_enable(cmdi_handle h, int32_t on);
/* Which kernel the events bracket: 0 = the in_proj GEMM (default; cmdi_profile_read's m, n, k = its shape), 1 = the
 * self-attention kernel (m, n, k = sequences, tokens per sequence, heads).  One kind per instrumented pass. */
int cmdi_profile_select(cmdi_handle h, int32_t which);
int cmdi_profile_read(cmdi_handle h, double* total_ms, int64_t* launches, int32_t* m, int32_t* n,
                      int32_t* k);
/* Kernel family the most recently bracketed launches DISPATCHED to ("gemm_h3_kernel", "gemm_h3p_kernel" (persistent),
 * "gemm_x6_kernel", "gemm_nt_kernel", "attention_h3_kernel", "attention_fwd_kernel"): recorded at launch time, so
 * bench.py's roofline names the kernel that ran rather than the one it expects.  "" before the first profiled launch. */
const char* cmdi_profile_kernel(cmdi_handle h);
/* Number of independent batch pipelines cmdi_sample_loop cuts the CURRENT condition's batch into (1 = none; 2 from
 * 8192 token rows up, CMDI_GROUPS overrides): each part runs its whole chain on its own stream. */
int cmdi_pipeline_parts(cmdi_handle h);
/* Bytes of device memory held by the handle. */
int64_t cmdi_workspace_bytes(cmdi_handle h);

#ifdef __cplusplus
}
#endif
#endif /* CONDMDI_H */
