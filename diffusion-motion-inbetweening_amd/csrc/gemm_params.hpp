// Operand / epilogue descriptors of the fp32 NT GEMM family (host + device).
#pragma once
#include <stddef.h>

namespace cmdi {

// How a logical operand row maps to memory.
enum RowMode {
    ROWS_PLAIN = 0,   // row r at ptr + r*ld, K contiguous
    ROWS_TOK = 1,     // logical row (b,t) -> token row b*S + 1 + t   (drops the conditioning token)
    ROWS_MOTION = 2,  // logical row (b,t), element k=c at x[b][c][t] (motion tensor, T contiguous)
};

enum EpiMode {
    EPI_PLAIN = 0,     // C = v + bias[n]
    EPI_GELU = 1,      // aux = v + bias (optional stash); C = gelu_erf(v + bias)
    EPI_SILU = 2,      // C = silu(v + bias)
    EPI_RESID = 3,     // C = (v + bias[n]) + R[m][n]
    EPI_INPROJ = 4,    // m=(b,t): tok[b*S+1+t][n] = v + bias[n] + pe[1+t][n]  (and the uncond copy)
    EPI_MOTION = 5,    // m=c (feature), n=(b,t): out[b][c][t] = v + bias[m]   (transposed store)
    EPI_TOKOUT = 6,    // m=(b,t): C[b*S+1+t][n] = v            (backward of the output projection)
    EPI_GELUGRAD = 7,  // C = v * gelu'(aux[m][n])              (backward through linear1's GELU)
    EPI_ACCUM = 8,     // C = v + R[m][n] (no bias)              (gradient accumulation)
};

struct GemmParams {
    const float* A;
    const float* W;
    const float* bias;
    float* C;
    const float* R;
    float* aux;
    const float* pe;
    int M, N, K;        // logical sizes (K a multiple of BK after padding)
    int lda, ldw, ldc;  // leading dimensions of PLAIN/TOK operands and of C
    int T, S, Cf;       // frames, tokens per sequence (T+1), motion features (263)
    int Bdup;           // EPI_INPROJ: B if an unconditional copy is also written, else 0
    float out_scale;    // EPI_MOTION: multiplies v (1 for the forward pass)
    const unsigned* gs_bits;  // optional gradient scale (common.hpp grad_scale_from_bits):
                              // EPI_TOKOUT multiplies v by it, EPI_MOTION divides v by it
    const void* Wx;           // bf16x6 path (gemm_x6.hpp): W pre-split into three bf16 planes [N][K/32][3][32], or null
    // ---- 1-D convolution over tap-shifted fp32 rows (gemm_x6_kernel<.., CONV = true>: the U-Net on exact operands, round 5).
    // Same meaning as the H3Params fields of these names: A row of GEMM row m is row (a_row_mul * m) of a [rows, lda] fp32
    // matrix whose pointer the caller has moved back by the padding; K = taps * cin_p in chunk-major order (K step kt = chunk
    // kt / taps of the row kt % taps frames further on); the result goes to row mo = m * c_row_mul + c_row_add, and only if
    // that row's position inside its tp-row frame lies in [t_lo, t_hi).  Outputs: C[mo][ldc] and / or C2[mo][ldc2] (both fp32:
    // the "split rows" of the f16x3 U-Net are plain fp32 rows here); R[mo][r_ld] is added first when given.
    int taps, a_row_mul, c_row_mul, c_row_add, tp, t_lo, t_hi;
    float* C2;
    int ldc2, r_ld;
};

// ---- split-f16 (fp32-equivalent) GEMM family, gemm_h3.hpp ----------------------------------------
enum H3Epi {
    H3_PLAIN = 0,       // C = v + bias[n]                                   (fp32)
    H3_GELU_SPLIT = 1,  // aux = v + bias (optional); Cs = split(gelu_erf(v + bias))
    H3_RESID = 2,       // C = (v + bias[n]) + R[m][n]  (fp32; C may be null) and, if Cs != null, Cs = split(C);
                        // R fp32, or Rs: the split rows a LayerNorm already wrote for the next GEMM
    H3_PLAIN_SPLIT = 3, // aux = v + bias (optional fp32 copy); Cs = split(v + bias[n])
    H3_GELUGRAD_SPLIT = 4, // Cs = split(v * gelu'(aux[m][n]))   (backward through linear1's GELU)
    H3_RESID_LN = 5,    // x = (v + bias) + R; aux = x (optional); y = LayerNorm(x) -> C (fp32) and Cs (split,
                        // optional); needs a tile that spans the whole row (N == BN)
    H3_TOKENS = 6,      // input projection: GEMM row m = (b, t) -> token row b*S + 1 + t of Cs:
                        // Cs = split((v + bias[n]) + pe[1 + t][n]), also written for sequence b + tok_dup (CFG); pe optional; with C:
                        // the same rows as fp32 (the output projection's backward: dTok rows, no pe, no Cs)
    H3_CONV_GN = 8,     // convolution + GroupNorm [+ AdaGN] + Mish [+ R] in one kernel: the tile owns whole (sequence, group)
                        // blocks (BM == tp rows = one framed sequence, BN = 128 columns = 1 or 2 groups of gn_cg channels);
                        // mean / variance from the accumulators (two block reductions), then as H3_RESID: C, Cs optional
    H3_MOTION = 7,      // output projection, roles swapped: A = weight rows (m = feature c < M), W = token rows
                        // (n = b*S + s): out[b][c][s - 1] = v + bias[c] for s >= 1  (T contiguous, fp32)
};

struct H3Params {
    const _Float16* A;  // split rows [M][2K]
    const _Float16* W;  // split rows [N][2K]
    const _Float16* Wp; // gemm_h3w (weight-stationary, K = 512) only: W in fragment order (pack_w_h3w_kernel), same bytes as W
    const float* bias;  // [N] or null
    float* C;           // fp32 output [M][ldc]
    _Float16* Cs;       // split output [M][2N]
    const float* R;     // residual [M][r_ld ? r_ld : ldc]
    int r_ld;
    int m_fast;         // 1: consecutive block ids walk down a column of tiles (same W panel) instead of along a row
    int ksplit;         // H3_PLAIN only: > 1 = that many blocks per tile, each over a slice of K; slice s writes its
    long slice_stride;  // partial sums to C + s * slice_stride (the consumer adds the slices); bias joins slice 0
    const float* gn_ss; // H3_CONV_GN: (scale | shift) rows [seq][gn_ss_ld] of the AdaGN block, or null; ln_g / ln_b = gamma / beta
    int gn_ss_ld, gn_cg;
    const float* pe;    // H3_TOKENS: positional table [.][N]
    int tok_T, tok_S, tok_dup;   // H3_TOKENS / H3_MOTION: frames, tokens (= frames + 1) per sequence; CFG copy offset
    const _Float16* Rs; // H3_RESID: the residual as split rows [M][2N] instead (R = hi + lo * 2^-11)
    float* aux;         // optional pre-activation stash [M][ldc]
    const float* ln_g;  // H3_RESID_LN: LayerNorm weight / bias [N], optional (mean, rstd) [M][2]
    const float* ln_b;
    float* ln_stats;    // H3_RESID_LN: (mean, rstd) of the rows normalised; with ln_part: (mean, rstd) derived from the partials,
                        // written by the blocks of the first column panel (stash of a forward pass that keeps activations)
    int* range_flag;    // set to 1 if a split output leaves the f16 range
    int M, N, K, ldc;
    int n_big, m_split; // mixed-granularity launch (set by launch_gemm_h3)
    // ---- 1-D convolution as a GEMM over tap-shifted rows (UNET denoiser, unet.hip); all 0 = plain GEMM ----
    // A row of output row m is row (a_row_mul * m) of a [rows, a_ld halves] split matrix whose pointer the
    // caller has already moved back by the padding; K = taps * cpt * 32 and K step kt reads 32-channel chunk kt / taps
    // of the row kt % taps frames further on (chunk-major, round 5; rounds 1-4: tap kt / cpt, chunk kt % cpt).  The result goes to row m * c_row_mul + c_row_add, and only
    // if that row's position inside its tp-row sequence frame lies in [t_lo, t_hi) (halo rows stay zero).
    int a_ld, a_row_mul, taps, cpt;
    int c_row_mul, c_row_add, tp, t_lo, t_hi;
    int rc_tv;          // > 0 (persistent kernel, convolutions with a_row_mul <= 1 and no c_row_mul): GEMM row g is only a
                        // LOGICAL row — frame g % rc_tv of sequence g / rc_tv, i.e. physical row (g / rc_tv) * tp + t_lo +
                        // g % rc_tv of both the A rows and the output; M counts rc_tv rows per sequence, so the halo rows of
                        // the framed layout (12.5 % of a 256-row frame) are neither multiplied nor masked
    int cs_ld;          // halves per row of the split output (0 = 2N); Cs may point at a column block of a wider matrix
    int cs_head_major;  // H3_PLAIN_SPLIT: write the output head-major — column block n / 128 (= q|k|v x head) is its own
                        // [M][128] split matrix (512 contiguous bytes per token): what attention_h3 streams per (sequence, head)
    // ---- LayerNorm folded into the GEMMs around it (no LayerNorm pass, api_denoiser.hip run_layers) -------------------------------
    // The residual stream travels as its PRE-LayerNorm value P (split rows) plus per-row partial statistics: for every
    // row 16 x (sum, sum of squared deviations) over its 32-column blocks, written by the producing GEMM's epilogue
    // (out_part) and combined (Chan) by the consumer into (mean, rstd).  A consumer whose A operand is LN(P) multiplies
    // the RAW P by weights with gamma folded in (W' = W diag(gamma)) and corrects in the epilogue:
    //     LN(P) W^T + b = rstd_m (P W'^T - mean_m c1[n]) + c2[n],   c1[n] = sum_k W'[n,k],  c2 = W beta + b  (-> bias)
    // and a consumer whose RESIDUAL is LN(P) normalises the residual rows it reads anyway.
    const float* ln_part;   // partial statistics [M][16][2] of this GEMM's row tensor (A operand or residual), or null
    const float* ln_c1;     // [N]: the A operand is LN(P) folded as above (bias must hold c2); null = A is used as is
    const float* ln_rg;     // [N] gamma / beta: the residual rows Rs are P and LN(P) is what gets added; null = Rs as is
    const float* ln_rb;
    float* out_part;        // H3_RESID: partial statistics [M][16][2] of the value written (N == 512)
    const unsigned* gs_bits;  // H3_MOTION: divide the result by the power-of-two gradient scale (common.hpp grad_scale_from_bits): the
                              // last GEMM of the input-VJP undoes the scale its first one applied; null = 1
    int dbg;            // bench-only ablations: 1 = no in-loop loads, 2 = no epilogue stores, 16 = timestamps
    long long* dbg_buf; // dbg & 16: per block {start, loop start, loop end, end} (s_memtime)
};

}  // namespace cmdi
