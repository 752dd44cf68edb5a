// Host-side launchers of the gfx950 kernels (one translation unit per kernel family).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_params.hpp"

namespace cmdi {

// ---- gemm_f32.hip -------------------------------------------------------------------------
enum GemmKind {
    GK_PLAIN = 0,      // A plain, W plain, C = v + bias
    GK_GELU,           // ... C = gelu(v + bias), optional pre-activation stash in aux
    GK_SILU,
    GK_RESID,          // C = v + bias + R
    GK_ACCUM,          // C = v + R
    GK_GELUGRAD,       // C = v * gelu'(aux)
    GK_INPROJ,         // A = motion tensor (transposed read), EPI_INPROJ
    GK_OUTPROJ,        // A = W_out rows (features), W = token rows, EPI_MOTION (transposed store)
    GK_OUTPROJ_BWD,    // A = d_out motion tensor, W = W_outᵀ, EPI_TOKOUT
};
// tile: 0 auto, 1 = 128x128, 2 = 64x128, 3 = 128x64, 4 = 64x64, 5 = 256x128 (8 waves)
hipError_t launch_gemm(GemmKind kind, const GemmParams& p, int tile, hipStream_t stream);
int gemm_auto_tile(int M, int N);

// ---- gemm_x6.hip (exact fp32 operands as three bf16 planes, six bf16 MFMA products per fp32 product) -----------
// p.Wx = launch_pack_x6(W): [N][K/32][3][32] bf16; A stays plain fp32.  Kinds: PLAIN, GELU, RESID, ACCUM, GELUGRAD.
bool gemm_x6_supports(GemmKind kind, const GemmParams& p);
hipError_t launch_gemm_x6(GemmKind kind, const GemmParams& p, hipStream_t stream, int variant = 0);
hipError_t launch_pack_x6(const float* src, void* dst, int64_t rows, int cols, int64_t ld, hipStream_t stream);
// 1-D convolution over tap-shifted fp32 rows with exact operands (GemmParams taps / a_row_mul / c_row_* / tp / t_* / C2 / r_ld;
// the U-Net's bf16x6 mode, round 5).  Kinds: PLAIN, RESID.  variant 2 = the rotated K step, 0 = the compiler's schedule.
bool gemm_x6_conv_supports(GemmKind kind, const GemmParams& p);
hipError_t launch_gemm_x6_conv(GemmKind kind, const GemmParams& p, hipStream_t stream, int variant = 2);

// ---- gemm_h3.hip (split-f16, fp32-equivalent) ---------------------------------------------------
// epi: H3Epi; tile: 0 auto, 1 = 128x128 (2 stages), 2 = 256x128 8 waves (3 stages), 3 = same (2 stages),
// 4 = 128x64 (2), 5 = 128x64 (3), 6 = 64x128 (2), 7 = 128x128 8 waves (3), 8 = same (2), 9 = 128x256 8 waves (2),
// 10 / 11 = 256x128 / 128x256 16 waves (3), 20 = mixed grid: 128x128 8 waves for whole rounds + 64x128 tail,
// 21 = 64x128 8 waves (2)
hipError_t launch_gemm_h3(int epi, const H3Params& p, int tile, hipStream_t stream);
int gemm_h3_auto_tile(int M, int N);
// gemm_h3p.hip: the persistent, phase-alternating kernel (tile id 50 of launch_gemm_h3); same bits as the tiles above
bool gemm_h3p_supports(int epi, const H3Params& p);
const char* gemm_h3_last_route();   // kernel family of this thread's most recent launch_gemm_h3
bool gemm_h3_persistent_for(int M);   // policy (CMDI_H3_PERSIST), gemm_h3.hip
hipError_t launch_gemm_h3p(int epi, const H3Params& p, hipStream_t stream, int ablation = 0);   // ablation: probes library only
// gemm_h3w.hip: the weight-stationary kernel for K = 512 (tile id 60 of launch_gemm_h3; W fragments resident in the accumulation
// registers, A streamed through LDS); same bits as the tiles above
bool gemm_h3w_supports(int epi, const H3Params& p);
hipError_t launch_gemm_h3w(int epi, const H3Params& p, hipStream_t stream);
hipError_t launch_pack_w_h3w(const _Float16* w_split, _Float16* w_packed, int n, hipStream_t stream);   // [n][1024] -> fragment order
// fp32 [rows][cols] (row stride ld_src) -> split rows [rows][2*cols] halves (format: gemm_h3.hpp)
hipError_t launch_split_f16(const float* src, _Float16* dst, int64_t rows, int cols, int64_t ld_src,
                            int* range_flag, hipStream_t stream);

// ---- attention_f32.hip ----------------------------------------------------------------------
// out (fp32 [M, d]) and out_split (split rows [M, 2d] for the f16-pipe out_proj GEMM) may each be
// null; row_stats != null stashes (max, 1/sum) per query row for the backward pass.
hipError_t launch_attention_fwd(const float* qkv, float* out, _Float16* out_split, int* range_flag,
                                float* row_stats, int n_seq, int S, int H, hipStream_t stream);
// ---- attention_h3.hip (split-f16 products, fp32-equivalent) -------------------------------------
// qkv_split: split rows [M, 2*3d] (gemm_h3.hpp format); outputs as launch_attention_fwd
// head_major: qkv_split was written by a GEMM with cs_head_major (gemm_params.hpp) over exactly these n_seq * S rows
hipError_t launch_attention_h3(const _Float16* qkv_split, float* out, _Float16* out_split,
                               int* range_flag, float* row_stats, int n_seq, int S, int H,
                               hipStream_t stream, bool head_major = false);
// ---- attention_bwd_h3.hip: backward on the f16 pipe: everything in split rows; the forward output O (for
// D = rowsum(dO*O)) as fp32 rows (o_fwd) or as split rows (o_fwd_split: the stash of the folded forward schedule), one of
// the two; d_scratch holds attention_bwd_scratch_floats(n_seq, S, H) floats (per-tile row statistics)
size_t attention_bwd_scratch_floats(int n_seq, int S, int H);
hipError_t launch_attention_bwd_h3(const _Float16* qkv_split, const float* o_fwd, const _Float16* o_fwd_split,
                                   const float* row_stats,
                                   const _Float16* d_out_split, _Float16* d_qkv_split,
                                   float* d_scratch, int n_seq, int S, int H, hipStream_t stream);
#ifdef CMDI_PROBES
hipError_t read_bwd_stamps(void* host_dst);
#endif
// ---- attention_bwd_f32.hip ------------------------------------------------------------------
// d_qkv[M,3d] from d_out[M,d]; P is recomputed from the forward's row statistics; d_rowdot is a
// [n_seq*H*S] scratch (D = rowsum(dO*O)) written by the dQ kernel and read by the dK/dV kernel.
hipError_t launch_attention_bwd(const float* qkv, const float* o_fwd, const float* row_stats,
                                const float* d_out, float* d_qkv, float* d_rowdot, int n_seq, int S,
                                int H, hipStream_t stream);

// ---- elementwise.hip ------------------------------------------------------------------------
// y_split (optional): the same output as split rows [rows][2d] for the f16-pipe GEMMs
hipError_t launch_layernorm(const float* x, const float* gamma, const float* beta, float* y,
                            _Float16* y_split, int* range_flag,
                            float* stats /* [rows][2] mean,rstd or null */, int rows, int d,
                            hipStream_t stream, const _Float16* x_split_in = nullptr /* read split rows instead of x */);
// W' = W diag(gamma), c1 = row sums of W', c2 = W beta + bias: LayerNorm folded into the GEMM that consumes it
hipError_t launch_fold_ln(const float* W, const float* gamma, const float* beta, const float* bias, float* Wf, float* c1,
                          float* c2, int N, int K, hipStream_t stream);
// dx = LN backward of dy (optionally + extra residual gradient dres added to the result)
// (dx optional when dx_split is given; x_split: x as split rows instead of the fp32 x)
hipError_t launch_layernorm_bwd(const float* x, const float* stats, const float* gamma,
                                const float* dy, float* dx /* optional */, _Float16* dx_split /* optional */,
                                int rows, int d, hipStream_t stream, const _Float16* x_split = nullptr);
// tok[b*S + 0][:] = time_table[t_b] + text_term[b] + pe[0]
// tmap_dev / cursor (graph replay): t = tmap_dev[*cursor] for every sequence
hipError_t launch_token0(float* tok, const float* time_table, const float* text_term,
                         const float* pe, const int64_t* t_dev, int64_t t_scalar, int n_seq,
                         int n_per_pass, int S, int d, int n_time_rows, hipStream_t stream,
                         const int64_t* tmap_dev = nullptr, const int* cursor = nullptr,
                         _Float16* tok_split = nullptr /* write split rows instead of fp32 */, int* range_flag = nullptr);
// x [nb][C][T] -> split frame rows [nb*T][2*Kp] (Kp = C rounded up to 32, zero padded): A operand of the input projection
// (gs_bits: multiply by the power-of-two gradient scale first — the output gradient on its way into the input-VJP)
hipError_t launch_pose_rows_split(const float* x, _Float16* xs, int nb, int C, int T, int Kp, int* range_flag,
                                  hipStream_t stream, const unsigned* gs_bits = nullptr);
// text_term[b'] rows: conditional rows get proj[b] (already W·c+b), unconditional rows get bias
hipError_t launch_fill_rows(float* dst, const float* row, int rows, int d, hipStream_t stream);
hipError_t launch_add2(float* dst, const float* a, const float* b, int64_t n, hipStream_t stream);
// atomicMax of the bit pattern of max|x| into *out (zeroed by the caller)
hipError_t launch_absmax_bits(const float* x, int64_t n, unsigned* out, hipStream_t stream);
// dst[r][c] = c < cols ? src[r][c] : 0   (dst row stride ldd >= cols)
hipError_t launch_pad_copy(float* dst, const float* src, int rows, int cols, int ldd,
                           hipStream_t stream);
// dst[c][r] = r < rows ? src[r][c] : 0 for r < ldd  (src [rows][cols] -> dst [cols][ldd])
hipError_t launch_transpose_pad(float* dst, const float* src, int rows, int cols, int ldd,
                                hipStream_t stream);

// ---- unet.hip: MDM_UNET denoiser ------------------------------------------------------------------
struct UnetModel;
// x6: CMDI_PREC_BF16X6 — fp32 activation rows, every convolution / gradient GEMM on gemm_x6's convolution form (exact operands)
UnetModel* unet_new(int n_feats, int added, int dim, const int mults[4], int max_seq, bool text, bool want_grad,
                    bool attention, bool x6);
const char* unet_error(const UnetModel* u);
const char* unet_probe_route(const UnetModel* u);   // kernel family of the bench-probed convolution GEMM
int64_t unet_bytes(const UnetModel* u);
void unet_free(UnetModel* u);
int unet_load_weight(UnetModel* u, const char* name, const float* d_src, int64_t numel, hipStream_t s);
int unet_finalize(UnetModel* u, hipStream_t s);
// ev0 / ev1 (bench): recorded around the second k=5 convolution GEMM of downs.0.1; its M, N, K land in probe_mnk
int unet_forward(UnetModel* u, const float* x, const float* obs, const uint8_t* mask, const float* emb, int B, int nseq,
                 int T, float* out, hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr,
                 int* probe_mnk = nullptr, bool keep = false /* stash activations for unet_backward */);
// input-VJP of the last stashing forward pass: gx [nseq, J, T] from gout [nseq, J, T]
int unet_backward(UnetModel* u, const float* gout, const uint8_t* mask, const unsigned* gs_bits, int B, int nseq, int T,
                  float* gx, hipStream_t s);
int unet_range_flag(UnetModel* u, int* flag, hipStream_t s);
int unet_range_clear(UnetModel* u, hipStream_t s);
hipError_t launch_unet_emb(float* emb, const float* time_table, const float* text_term, const int64_t* t_dev,
                           int64_t t_scalar, int n_seq, int n_per_pass, int d, int n_time_rows, hipStream_t stream,
                           const int64_t* tmap_dev = nullptr, const int* cursor = nullptr);

// ---- clip_text.hip: CLIP ViT-B/32 text tower (the step before the loop) ---------------------------------------
struct ClipText;
ClipText* clip_new(int vocab, int width, int heads, int layers, int ctx, int embed, int max_batch);
const char* clip_error(const ClipText* c);
void clip_free(ClipText* c);
int clip_load_weight(ClipText* c, const char* name, const float* src, int64_t numel, hipStream_t s);
int clip_encode_text(ClipText* c, const int32_t* tokens, int B, float* out, hipStream_t s);
int clip_status(ClipText* c, int* flag, hipStream_t s);

// ---- postprocess.hip ------------------------------------------------------------------------------
// x [B, n_feats, 1, T] (z-scored if mean/std given) -> joint positions out [B, n_joints, 3, T]
hipError_t launch_recover_xyz(const float* x, const float* mean, const float* std, float* out, int batch,
                              int n_feats, int n_frames, int n_joints, int abs_3d, hipStream_t stream);

// ---- sampler.hip ----------------------------------------------------------------------------
struct StepCoef {
    float c1, c2;          // posterior_mean_coef1/2[i]
    float sig_nz;          // [i != 0] * sigma
    float sra, srm1a;      // sqrt_recip_alphas_cumprod[i], sqrt_recipm1_alphas_cumprod[i]
    float sqrt_abp, dir;   // DDIM: sqrt(ab_prev), sqrt(1 - ab_prev - sigma^2)
    float gcoef;           // reconstruction guidance: (w_r * sqrt_ab) / 2
    float clip;            // > 0: clamp the x0 derived from an eps-prediction to [-clip, clip]
    int ddim, mean_eps, impute, recon;
};
struct SamplerIO {
    float* x;                 // in: x_t, out: x_{t-1}
    const float* out_c;       // conditional (or only) model output
    const float* out_u;       // unconditional output or null
    const float* text_scale;  // [B] or null
    const uint8_t* mask;      // inpainting mask or null
    const float* inpaint;     // inpainted motion or null
    const float* grad_c;      // reconstruction-guidance gradient halves (null if none)
    const float* grad_u;
    const float* noise;       // injected draw or null (engine RNG)
    float* pred_xstart;       // optional
};
// ktab / cursor (both or neither): read the step index from *cursor and the coefficients from
// ktab[*cursor] instead of the by-value arguments (hipGraph replay of a whole step)
hipError_t launch_sampler_step(const SamplerIO& io, const StepCoef& k, int batch, int64_t per_sample,
                               uint64_t seed, int64_t first_sample, int step, hipStream_t stream,
                               const StepCoef* ktab = nullptr, const int* cursor = nullptr);
hipError_t launch_cursor_add(int* cursor, int delta, hipStream_t stream);
// gout_c = s*g, gout_u = (1-s)*g with g = 2*(hat - inpaint)*mask, hat = CFG(out_c, out_u)
hipError_t launch_recon_gout(const float* out_c, const float* out_u, const float* text_scale,
                             const uint8_t* mask, const float* inpaint, float* gout_c, float* gout_u,
                             int batch, int64_t per_sample, hipStream_t stream);
hipError_t launch_cfg_combine(const float* out_c, const float* out_u, const float* text_scale,
                              float* out, int batch, int64_t per_sample, hipStream_t stream);
hipError_t launch_cfg_split(const float* g, const float* text_scale, float* gc, float* gu, int batch,
                            int64_t per_sample, hipStream_t stream);
hipError_t launch_q_sample(const float* x0, const float* noise, float* out, float a, float b,
                           int64_t n, hipStream_t stream);
hipError_t launch_randn(float* out, int batch, int64_t per_sample, uint64_t seed,
                        int64_t first_sample, int step, hipStream_t stream);
void philox4x32_10_host(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

}  // namespace cmdi
