// C-ABI of libcondmdi_hip.so (include/condmdi.h): engine handle, weight packing, the MDM
// trans_enc forward / dX-backward schedule of kernel launches, and the sampling loop.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/condmdi.h"
#include "kernels.hpp"

using namespace cmdi;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIPCHK(expr)                                                                      \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess)                                                             \
            return fail(CMDI_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));   \
    } while (0)

// Library default of cmdi_model_desc.precision = CMDI_PREC_DEFAULT.  Rule: the default must not drift more
// than the exact-fp32 engine over the full 1000-step chain.  Measured on MI355X against the reference's float64 chain
// (tests/test_gpu_parity.py::test_long_chain_drift_vs_reference): f16x3 1.0e-6, bf16x6 1.4e-6,
// fp32-MFMA 1.9e-6, the reference's own fp32 CPU chain 1.3e-6 — so the fastest mode is the default; bf16x6 (no operand
// truncation, no range limit) is what the f16-range guard falls back to.
constexpr int kDefaultPrecision = CMDI_PREC_F16X3;

struct LayerW {
    float *in_w = nullptr, *in_b = nullptr, *out_w = nullptr, *out_b = nullptr;
    float *l1_w = nullptr, *l1_b = nullptr, *l2_w = nullptr, *l2_b = nullptr;
    float *n1_g = nullptr, *n1_b = nullptr, *n2_g = nullptr, *n2_b = nullptr;
    // transposed copies for the dX backward GEMMs (want_grad only)
    float *in_wT = nullptr, *out_wT = nullptr, *l1_wT = nullptr, *l2_wT = nullptr;
    // split-f16 copies (hi | lo*2^11 rows, gemm_h3.hpp) for the f16-pipe forward GEMMs
    _Float16 *in_ws = nullptr, *out_ws = nullptr, *l1_ws = nullptr, *l2_ws = nullptr;
    _Float16 *in_wTs = nullptr, *out_wTs = nullptr, *l1_wTs = nullptr, *l2_wTs = nullptr;  // want_grad
    // f16x3 with the LayerNorms folded into their consumers (gemm_params.hpp): in_proj weights with the PREVIOUS layer's
    // norm2 gamma folded in (layers >= 1), linear1 weights with this layer's norm1 gamma; c1 = row sums, c2 = W beta + b
    _Float16 *in_wsf = nullptr, *l1_wsf = nullptr;
    float *in_c1 = nullptr, *in_c2 = nullptr, *l1_c1 = nullptr, *l1_c2 = nullptr;
    // bf16x6: three-plane bf16 copies (gemm_x6.hpp) of the forward weights and, want_grad, of their transposes
    void *in_wx = nullptr, *out_wx = nullptr, *l1_wx = nullptr, *l2_wx = nullptr;
    void *in_wTx = nullptr, *out_wTx = nullptr, *l1_wTx = nullptr, *l2_wTx = nullptr;
};
struct LayerStash {
    float *qkv = nullptr, *attn = nullptr, *row_stats = nullptr;
    _Float16* qkvS = nullptr;   // f16x3: the split qkv replaces the fp32 copy (attention fwd and bwd read it)
    float *pre1 = nullptr, *stats1 = nullptr, *aux = nullptr, *pre2 = nullptr, *stats2 = nullptr;
};

}  // namespace

struct cmdi_engine {
    cmdi_model_desc desc{};
    int L = 0, d = 0, f = 0, H = 0, C = 0, Cpad = 0, Tmax = 0, Bmax = 0, clip_dim = 512;
    std::vector<void*> allocs;
    int64_t bytes = 0;

    // weights
    float *w_in = nullptr, *b_in = nullptr, *w_in_pad = nullptr, *w_inT = nullptr;
    float *pe = nullptr;
    float *t1_w = nullptr, *t1_b = nullptr, *t2_w = nullptr, *t2_b = nullptr;
    float *txt_w = nullptr, *txt_b = nullptr;
    float *w_out = nullptr, *b_out = nullptr, *w_outT_pad = nullptr;
    std::vector<LayerW> layers;
    float* time_table = nullptr;
    int n_time_rows = 0;
    bool finalized = false;

    // schedule (host)
    bool have_schedule = false;
    int n_steps = 0, mean_type = 0;
    float clip_x0 = 0.f;
    std::vector<float> c1, c2, sigma, sqrt_ab, sqrt_1mab, sra, srm1a, ab, ab_prev;
    std::vector<int64_t> tmap;

    // condition
    bool have_cond = false;
    int B = 0, T = 0, cfg = 0;
    int imputate = 0, stop_imp = 0, recon = 0, stop_rec = 0;
    bool have_mask = false, have_text = false;
    std::vector<float> recon_w;
    float *text_term = nullptr, *text_scale = nullptr, *inpaint = nullptr, *enc_text = nullptr;
    uint8_t* mask = nullptr;

    // workspace
    float *tokA = nullptr, *tokB = nullptr, *bufH = nullptr, *qkv = nullptr, *attn = nullptr,
          *ffn = nullptr, *out_raw = nullptr;
    std::vector<LayerStash> stash;
    bool stash_valid = false;
    float *dA = nullptr, *dB = nullptr, *dH = nullptr, *dqkv = nullptr, *dffn = nullptr,
          *drowdot = nullptr, *gout = nullptr, *gx = nullptr;
    // precision of the encoder-layer GEMMs: CMDI_PREC_F32 (exact fp32 MFMA) or CMDI_PREC_F16X3
    // (fp32-equivalent split-f16 products on the f16 matrix pipe, gemm_h3.hpp)
    int precision = CMDI_PREC_F16X3;
    _Float16 *tokS = nullptr, *bufHS = nullptr, *attnS = nullptr, *ffnS = nullptr, *qkvS = nullptr;
    _Float16 *dBS = nullptr, *dffnS = nullptr, *dqkvS = nullptr, *dOS = nullptr;  // backward operands (want_grad)
    UnetModel* unet = nullptr;     // arch = CMDI_ARCH_UNET: the MDM_UNET denoiser (unet.hip)
    float* uemb = nullptr;          // [2 Bmax, d] time (+ text) embedding per sequence
    float* obs_x0 = nullptr;        // keyframe conditioning of the UNET (model_kwargs obs_x0 / obs_mask)
    uint8_t* obs_mask = nullptr;
    bool have_obs = false;
    int* range_flag = nullptr;
    unsigned* gs_bits = nullptr;   // max|gout| bits -> power-of-two gradient scale (f16x3 backward); [1 + parts]
    float* text_term_p = nullptr;  // text_term in the slot order of the independent pipelines (cmdi_sample_loop)
    int pipelines = 1;             // CMDI_PIPELINES=0: keep the fork/join-per-step schedule in cmdi_sample_loop
    int h3_tile_qkv = 0, h3_tile_proj = 0, h3_tile_ffn1 = 0, h3_tile_ffn2 = 0;
    // f16x3: LayerNorm inside the out_proj / linear2 GEMM epilogue (d_model = 512).  Off by default:
    // the full-row 64x512 tile it needs (197 blocks, 8 waves per CU) loses more in the GEMM than the
    // saved LayerNorm pass returns (B=32 CFG: 2.60 vs 2.29 ms per step on MI355X).
    int ln_fuse = 0;
    // f16x3, forward without a stash: NO LayerNorm pass — the residual stream travels as its pre-LayerNorm value plus
    // per-row partial statistics and every LayerNorm is folded into the GEMM that consumes it (gemm_params.hpp);
    // CMDI_LN_FOLD=0 keeps the separate LayerNorm kernels
    int ln_fold = 0;
    int qkv_head_major = 0;   // folded path: in_proj writes q | k | v head-major for the attention kernel (CMDI_QKV_HEAD_MAJOR=1;
                              // measured: no gain — attention 36.0 vs 35.1 us, step 2.14 vs 2.07 ms — so off)
    float *partA = nullptr, *partB = nullptr;   // [M][16][2] partial statistics of pre1 / pre2
    int io_pipe = 0;   // 1: software-pipelined input / output projection GEMMs
    // f16x3: input / output projections on the f16 pipe too (frame rows split by pose_rows_split_kernel, token and
    // motion-layout epilogues in gemm_h3.hpp); CMDI_IO_H3=0 keeps them on the fp32 MFMA kernels
    int io_h3 = 0;
    _Float16 *w_in_s = nullptr, *w_out_s = nullptr, *xS = nullptr;
    int x6_variant = 2;   // K-loop schedule of the bf16x6 GEMM (CMDI_X6_VAR; 2 = rotated barrier, the fastest measured)
    int gemm_tile = 0;
    int tile_inproj = 0, tile_proj = 0, tile_ffn1 = 0, tile_ffn2 = 0;  // per-GEMM overrides (0 = auto)

    // Independent sequence groups run on their own HIP streams (fork after the input projection,
    // join before the output projection): a kernel boundary is then a barrier for ONE group only,
    // so the tail of one group's GEMM overlaps the body of another's.
    int n_groups = 0;
    std::vector<hipStream_t> gstreams;
    std::vector<hipEvent_t> gevents;  // [0] = fork, [1 + g] = join of group g

    // hipGraph replay of whole denoising steps (cmdi_sample_loop): per-step scalars live in device
    // tables indexed by a device cursor, so ONE captured launch sequence serves every step.
    int use_graph = 0;                 // CMDI_GRAPH=1 / cmdi_set_graph
    StepCoef* coef_dev = nullptr;      // [n_steps] for the (sampler, eta) of the running chain
    int64_t* tmap_dev = nullptr;       // [n_steps] timestep_map
    int* cursor_dev = nullptr;         // current respaced step index
    int table_cap = 0;
    hipGraphExec_t graph_exec[2] = {nullptr, nullptr};   // [reconstruction guidance active?]
    bool graph_warm[2] = {false, false};
    hipStream_t graph_stream = nullptr;
    hipStream_t own_stream = nullptr;   // capture cannot start on the legacy default stream
    hipEvent_t own_ev[2] = {nullptr, nullptr};
    uint64_t graph_seed = 0;
    int64_t graph_first = 0;
    float* graph_x = nullptr;

    // optional live timing of the in_proj GEMM (bench.py roofline leg)
    bool profile = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    int prof_m = 0, prof_n = 0, prof_k = 0;
};

namespace {

int dalloc(cmdi_engine* e, void** p, size_t nbytes) {
    if (nbytes == 0) nbytes = 16;
    hipError_t err = hipMalloc(p, nbytes);
    if (err != hipSuccess)
        return fail(CMDI_E_NOMEM, std::string("hipMalloc(") + std::to_string(nbytes) +
                                      "): " + hipGetErrorString(err));
    e->allocs.push_back(*p);
    e->bytes += (int64_t)nbytes;
    return CMDI_OK;
}
template <class Tp>
int falloc(cmdi_engine* e, Tp** p, size_t count) {
    return dalloc(e, reinterpret_cast<void**>(p), count * sizeof(Tp));
}
#define ALLOC(ptr, count)                       \
    do {                                        \
        int _rc = falloc(e, &(ptr), (count));   \
        if (_rc != CMDI_OK) return _rc;         \
    } while (0)

void drop_graphs(cmdi_engine* e) {
    for (int i = 0; i < 2; ++i) {
        if (e->graph_exec[i]) (void)hipGraphExecDestroy(e->graph_exec[i]);
        e->graph_exec[i] = nullptr;
        e->graph_warm[i] = false;
    }
}

GemmParams gp(const float* A, const float* W, const float* bias, float* C, int M, int N, int K,
              int lda, int ldw, int ldc) {
    GemmParams p{};
    p.A = A; p.W = W; p.bias = bias; p.C = C;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = ldc;
    p.out_scale = 1.0f;
    return p;
}

// fp32-engine GEMM: on the bf16 pipe with exact three-plane operands when the packed weight is given (CMDI_PREC_BF16X6),
// else the fp32 MFMA kernel
hipError_t gemm_any(const cmdi_engine* e, GemmKind kind, GemmParams p, const void* wx, int tile, hipStream_t s) {
    if (wx) {
        p.Wx = wx;
        if (gemm_x6_supports(kind, p)) return launch_gemm_x6(kind, p, s, e->x6_variant);
        p.Wx = nullptr;
    }
    return launch_gemm(kind, p, tile, s);
}

// ---- encoder layers over sequences [seq0, seq0 + nseq) on stream s -----------------------------
int run_layers(cmdi_engine* e, int seq0, int nseq, bool keep, bool prof, hipStream_t s) {
    const int S = e->T + 1, d = e->d, f = e->f;
    const int M = nseq * S;
    const size_t r0 = (size_t)seq0 * S;
    float* tokA = e->tokA + r0 * d;
    float* tokB = e->tokB + r0 * d;
    float* bufH = e->bufH + r0 * d;
    float* ffn = e->ffn + r0 * f;
    const bool h3 = e->precision == CMDI_PREC_F16X3;
    _Float16* tokS = h3 ? e->tokS + r0 * 2 * d : nullptr;
    _Float16* bufHS = h3 ? e->bufHS + r0 * 2 * d : nullptr;
    _Float16* attnS = h3 ? e->attnS + r0 * 2 * d : nullptr;
    _Float16* ffnS = h3 ? e->ffnS + r0 * 2 * f : nullptr;
    _Float16* qkvS = h3 ? e->qkvS + r0 * 6 * d : nullptr;
    auto hp = [&](const _Float16* A, const _Float16* W, const float* bias, float* C, _Float16* Cs,
                  int N, int K) {
        H3Params p{};
        p.A = A; p.W = W; p.bias = bias; p.C = C; p.Cs = Cs; p.range_flag = e->range_flag;
        p.M = M; p.N = N; p.K = K; p.ldc = N;
        return p;
    };
    if (h3 && !e->io_h3)  // layer 0 reads the tokens assembled by token0 + the input projection (fp32)
        HIPCHK(launch_split_f16(tokA, tokS, M, d, d, e->range_flag, s));
    if (h3 && e->ln_fold && !keep && e->io_h3) {
        // ---- no LayerNorm pass: P (tokS) is the layer input BEFORE its LayerNorm (layer 0: the tokens themselves) ----
        float* partA = e->partA + r0 * 32;
        float* partB = e->partB + r0 * 32;
        for (int l = 0; l < e->L; ++l) {
            const LayerW& w = e->layers[l];
            const LayerW* prev = l > 0 ? &e->layers[l - 1] : nullptr;
            if (prof) {
                if (e->ev_used + 2 > e->ev_pool.size()) {
                    hipEvent_t a, b;
                    HIPCHK(hipEventCreate(&a));
                    HIPCHK(hipEventCreate(&b));
                    e->ev_pool.push_back(a);
                    e->ev_pool.push_back(b);
                }
                HIPCHK(hipEventRecord(e->ev_pool[e->ev_used], s));
            }
            {   // qkv = in_proj(LN2_prev(P))
                H3Params p = hp(tokS, prev ? w.in_wsf : w.in_ws, prev ? w.in_c2 : w.in_b, nullptr, qkvS, 3 * d, d);
                if (prev) { p.ln_part = partB; p.ln_c1 = w.in_c1; }
                p.cs_head_major = e->qkv_head_major;
                HIPCHK(launch_gemm_h3(H3_PLAIN_SPLIT, p, e->h3_tile_qkv, s));
            }
            if (prof) {
                HIPCHK(hipEventRecord(e->ev_pool[e->ev_used + 1], s));
                e->ev_used += 2;
                e->prof_m = M; e->prof_n = 3 * d; e->prof_k = d;
            }
            HIPCHK(launch_attention_h3(qkvS, nullptr, attnS, e->range_flag, nullptr, nseq, S, e->H, s,
                                       e->qkv_head_major != 0));
            {   // pre1 = LN2_prev(P) + out_proj(attn)   -> bufHS (+ partial statistics A)
                H3Params p = hp(attnS, w.out_ws, w.out_b, nullptr, bufHS, d, d);
                p.Rs = tokS;
                if (prev) { p.ln_part = partB; p.ln_rg = prev->n2_g; p.ln_rb = prev->n2_b; }
                p.out_part = partA;
                HIPCHK(launch_gemm_h3(H3_RESID, p, e->h3_tile_proj, s));
            }
            {   // ffn = gelu(linear1(LN1(pre1)))
                H3Params p = hp(bufHS, w.l1_wsf, w.l1_c2, nullptr, ffnS, f, d);
                p.ln_part = partA; p.ln_c1 = w.l1_c1;
                HIPCHK(launch_gemm_h3(H3_GELU_SPLIT, p, e->h3_tile_ffn1, s));
            }
            {   // pre2 = LN1(pre1) + linear2(ffn)   -> tokS (+ partial statistics B): the next layer's P
                H3Params p = hp(ffnS, w.l2_ws, w.l2_b, nullptr, tokS, d, f);
                p.Rs = bufHS; p.ln_part = partA; p.ln_rg = w.n1_g; p.ln_rb = w.n1_b;
                p.out_part = partB;
                HIPCHK(launch_gemm_h3(H3_RESID, p, e->h3_tile_ffn2, s));
            }
        }
        // the encoder output is LN2 of the last layer: the one LayerNorm launch that remains (split rows in, in place)
        const LayerW& last = e->layers[e->L - 1];
        HIPCHK(launch_layernorm(nullptr, last.n2_g, last.n2_b, nullptr, tokS, e->range_flag, nullptr, M, d, s, tokS));
        return CMDI_OK;
    }
    for (int l = 0; l < e->L; ++l) {
        const LayerW& w = e->layers[l];
        const LayerStash* st = keep ? &e->stash[l] : nullptr;
        float* qkv = (keep && !h3 ? st->qkv : e->qkv) + r0 * 3 * d;
        float* attn = (keep ? st->attn : e->attn) + r0 * d;
        float* pre1 = keep ? st->pre1 + r0 * d : tokB;
        float* pre2 = keep ? st->pre2 + r0 * d : tokB;
        float* row_stats = keep ? st->row_stats + (size_t)seq0 * e->H * S * 2 : nullptr;
        if (h3) {
            // same layer on the f16 matrix pipe; every A operand arrives as split rows written by
            // its producer (LayerNorm, attention, the GELU epilogue)
            if (prof) {
                if (e->ev_used + 2 > e->ev_pool.size()) {
                    hipEvent_t a, b;
                    HIPCHK(hipEventCreate(&a));
                    HIPCHK(hipEventCreate(&b));
                    e->ev_pool.push_back(a);
                    e->ev_pool.push_back(b);
                }
                HIPCHK(hipEventRecord(e->ev_pool[e->ev_used], s));
            }
            // qkv leaves as split rows for the f16-pipe attention (stashed per layer for the backward)
            _Float16* qkvL = keep ? st->qkvS + r0 * 6 * d : qkvS;
            HIPCHK(launch_gemm_h3(H3_PLAIN_SPLIT, hp(tokS, w.in_ws, w.in_b, nullptr, qkvL, 3 * d, d),
                                  e->h3_tile_qkv, s));
            if (prof) {
                HIPCHK(hipEventRecord(e->ev_pool[e->ev_used + 1], s));
                e->ev_used += 2;
                e->prof_m = M; e->prof_n = 3 * d; e->prof_k = d;
            }
            HIPCHK(launch_attention_h3(qkvL, keep ? attn : nullptr, attnS, e->range_flag, row_stats,
                                       nseq, S, e->H, s));
            if (e->ln_fuse) {   // x = norm1(x + out_proj(attn)) in one kernel
                H3Params p = hp(attnS, w.out_ws, w.out_b, bufH, bufHS, d, d);
                p.R = tokA; p.ln_g = w.n1_g; p.ln_b = w.n1_b;
                p.aux = keep ? pre1 : nullptr;
                p.ln_stats = keep ? st->stats1 + r0 * 2 : nullptr;
                HIPCHK(launch_gemm_h3(H3_RESID_LN, p, 0, s));
            } else {
                // the residual stream lives in split rows only: LayerNorm writes them for the next GEMM and
                // the residual epilogue reads the same rows back (hi + lo * 2^-11, 22 bits) — 8 instead of
                // 12 bytes per element through each LayerNorm
                H3Params p = hp(attnS, w.out_ws, w.out_b, pre1, nullptr, d, d);
                p.Rs = tokS;
                HIPCHK(launch_gemm_h3(H3_RESID, p, e->h3_tile_proj, s));
                HIPCHK(launch_layernorm(pre1, w.n1_g, w.n1_b, nullptr, bufHS, e->range_flag,
                                        keep ? st->stats1 + r0 * 2 : nullptr, M, d, s));
            }
            {
                H3Params p = hp(bufHS, w.l1_ws, w.l1_b, nullptr, ffnS, f, d);
                p.aux = keep ? st->aux + r0 * f : nullptr;
                HIPCHK(launch_gemm_h3(H3_GELU_SPLIT, p, e->h3_tile_ffn1, s));
            }
            if (e->ln_fuse) {   // x = norm2(x + linear2(gelu(linear1(x))))
                H3Params p = hp(ffnS, w.l2_ws, w.l2_b, tokA, l + 1 < e->L ? tokS : nullptr, d, f);
                p.R = bufH; p.ln_g = w.n2_g; p.ln_b = w.n2_b;
                p.aux = keep ? pre2 : nullptr;
                p.ln_stats = keep ? st->stats2 + r0 * 2 : nullptr;
                HIPCHK(launch_gemm_h3(H3_RESID_LN, p, 0, s));
            } else {
                H3Params p = hp(ffnS, w.l2_ws, w.l2_b, pre2, nullptr, d, f);
                p.Rs = bufHS;
                HIPCHK(launch_gemm_h3(H3_RESID, p, e->h3_tile_ffn2, s));
                const bool last = l + 1 == e->L && !e->io_h3;   // an fp32 output projection reads fp32 rows
                HIPCHK(launch_layernorm(pre2, w.n2_g, w.n2_b, last ? tokA : nullptr, last ? nullptr : tokS,
                                        e->range_flag, keep ? st->stats2 + r0 * 2 : nullptr, M, d, s));
            }
            continue;
        }
        // self-attention block: x = norm1(x + out_proj(MHA(x)))
        if (prof) {
            if (e->ev_used + 2 > e->ev_pool.size()) {
                hipEvent_t a, b;
                HIPCHK(hipEventCreate(&a));
                HIPCHK(hipEventCreate(&b));
                e->ev_pool.push_back(a);
                e->ev_pool.push_back(b);
            }
            HIPCHK(hipEventRecord(e->ev_pool[e->ev_used], s));
        }
        HIPCHK(gemm_any(e, GK_PLAIN, gp(tokA, w.in_w, w.in_b, qkv, M, 3 * d, d, d, d, 3 * d), w.in_wx,
                        e->tile_inproj, s));
        if (prof) {
            HIPCHK(hipEventRecord(e->ev_pool[e->ev_used + 1], s));
            e->ev_used += 2;
            e->prof_m = M; e->prof_n = 3 * d; e->prof_k = d;
        }
        HIPCHK(launch_attention_fwd(qkv, attn, nullptr, nullptr, row_stats, nseq, S, e->H, s));
        {
            GemmParams p = gp(attn, w.out_w, w.out_b, pre1, M, d, d, d, d, d);
            p.R = tokA;
            HIPCHK(gemm_any(e, GK_RESID, p, w.out_wx, e->tile_proj, s));
        }
        HIPCHK(launch_layernorm(pre1, w.n1_g, w.n1_b, bufH, nullptr, nullptr, keep ? st->stats1 + r0 * 2 : nullptr, M, d, s));
        // feed-forward block: x = norm2(x + linear2(gelu(linear1(x))))
        {
            GemmParams p = gp(bufH, w.l1_w, w.l1_b, ffn, M, f, d, d, d, f);
            p.aux = keep ? st->aux + r0 * f : nullptr;
            HIPCHK(gemm_any(e, GK_GELU, p, w.l1_wx, e->tile_ffn1, s));
        }
        {
            GemmParams p = gp(ffn, w.l2_w, w.l2_b, pre2, M, d, f, f, f, d);
            p.R = bufH;
            HIPCHK(gemm_any(e, GK_RESID, p, w.l2_wx, e->tile_ffn2, s));
        }
        HIPCHK(launch_layernorm(pre2, w.n2_g, w.n2_b, tokA, nullptr, nullptr, keep ? st->stats2 + r0 * 2 : nullptr, M, d, s));
    }
    return CMDI_OK;
}

// Run `fn(seq0, nseq, stream)` over the sequence groups: on the caller's stream if there is one
// group, else fork to the engine's streams and join back.
template <class Fn>
int for_groups(cmdi_engine* e, int n_seq, hipStream_t s, Fn fn) {
    // default: two groups once there is enough work per group to fill the chip (measured: B=32 CFG
    // 4.995 -> 4.621 ms/step with 2 groups, worse with 4); CMDI_GROUPS overrides
    int G = e->n_groups > 0 ? e->n_groups : ((long)n_seq * (e->T + 1) >= 8192 ? 2 : 1);
    if (G > n_seq) G = n_seq;
    if (G <= 1 || e->profile) return fn(0, n_seq, s);
    while ((int)e->gstreams.size() < G) {
        hipStream_t st;
        HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        e->gstreams.push_back(st);
    }
    while ((int)e->gevents.size() < G + 1) {
        hipEvent_t ev;
        HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        e->gevents.push_back(ev);
    }
    HIPCHK(hipEventRecord(e->gevents[0], s));
    for (int g = 0; g < G; ++g) {
        const int lo = (int)((long)n_seq * g / G), hi = (int)((long)n_seq * (g + 1) / G);
        HIPCHK(hipStreamWaitEvent(e->gstreams[g], e->gevents[0], 0));
        int rc = fn(lo, hi - lo, e->gstreams[g]);
        if (rc != CMDI_OK) return rc;
        HIPCHK(hipEventRecord(e->gevents[1 + g], e->gstreams[g]));
        HIPCHK(hipStreamWaitEvent(s, e->gevents[1 + g], 0));
    }
    return CMDI_OK;
}

// Input / output projections on the f16 pipe.  x [nb][C][T] -> frame rows (split, K padded to Cpad) -> token rows
// 1..T of tok_split (+ the same rows for sequence b + dup: the unconditional half sees the same frames).
static int input_projection_h3(cmdi_engine* e, const float* x, _Float16* xs, _Float16* tok_split, int nb, int dup,
                               hipStream_t s) {
    const int T = e->T;
    HIPCHK(launch_pose_rows_split(x, xs, nb, e->C, T, e->Cpad, e->range_flag, s));
    H3Params p{};
    p.A = xs; p.W = e->w_in_s; p.bias = e->b_in; p.Cs = tok_split; p.range_flag = e->range_flag;
    p.M = nb * T; p.N = e->d; p.K = e->Cpad; p.ldc = e->d;
    p.pe = e->pe; p.tok_T = T; p.tok_S = T + 1; p.tok_dup = dup;
    HIPCHK(launch_gemm_h3(H3_TOKENS, p, 0, s));
    return CMDI_OK;
}

// out[n][c][t] = W_out[c] · tok[n*S + 1 + t] + b_out[c]: weight rows take the A role so that the stores run along T
static int output_projection_h3(cmdi_engine* e, const _Float16* tok_split, float* out, int nseq, hipStream_t s) {
    const int T = e->T;
    H3Params p{};
    p.A = e->w_out_s; p.W = tok_split; p.bias = e->b_out; p.C = out;
    p.M = e->C; p.N = nseq * (T + 1); p.K = e->d; p.ldc = 0;
    p.tok_T = T; p.tok_S = T + 1;
    HIPCHK(launch_gemm_h3(H3_MOTION, p, 0, s));
    return CMDI_OK;
}

// ---- forward pass of MDM trans_enc (model/mdm.py:239-306) over n_seq = B or 2B sequences --------
int mdm_forward(cmdi_engine* e, const float* x, const int64_t* t_dev, int64_t t_scalar,
                float* out_buf, bool keep, hipStream_t s, bool tables = false) {
    const int B = e->B, T = e->T, S = T + 1, d = e->d, C = e->C;
    const int n_seq = e->cfg ? 2 * B : B;
    if (e->unet) {   // MDM_UNET.forward (model/mdm_unet.py:766-849)
        HIPCHK(launch_unet_emb(e->uemb, e->time_table, e->have_text ? e->text_term : nullptr, t_dev, t_scalar,
                               n_seq, B, d, e->n_time_rows, s));
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        int mnk[3] = {0, 0, 0};
        if (e->profile) {   // bench: HIP events around one level-0 convolution GEMM per evaluation
            if (e->ev_used + 2 > e->ev_pool.size()) {
                hipEvent_t a, b;
                HIPCHK(hipEventCreate(&a));
                HIPCHK(hipEventCreate(&b));
                e->ev_pool.push_back(a);
                e->ev_pool.push_back(b);
            }
            ev0 = e->ev_pool[e->ev_used]; ev1 = e->ev_pool[e->ev_used + 1];
        }
        const int urc = unet_forward(e->unet, x, e->have_obs ? e->obs_x0 : nullptr, e->have_obs ? e->obs_mask : nullptr,
                                     e->uemb, B, n_seq, T, out_buf, s, ev0, ev1, mnk, keep);
        if (e->profile && mnk[0]) { e->ev_used += 2; e->prof_m = mnk[0]; e->prof_n = mnk[1]; e->prof_k = mnk[2]; }
        if (urc != 0)
            return fail(CMDI_E_HIP, std::string("UNET: ") + unet_error(e->unet));
        e->stash_valid = keep;
        return CMDI_OK;
    }

    HIPCHK(launch_token0(e->tokA, e->time_table, e->have_text ? e->text_term : nullptr, e->pe, t_dev,
                         t_scalar, n_seq, B, S, d, e->n_time_rows, s, tables ? e->tmap_dev : nullptr,
                         tables ? e->cursor_dev : nullptr, e->io_h3 ? e->tokS : nullptr, e->range_flag));
    // InputProcess + sequence_pos_encoder for the frame tokens (mdm.py:271,279-280)
    if (e->io_h3) {
        int rc = input_projection_h3(e, x, e->xS, e->tokS, B, e->cfg ? B : 0, s);
        if (rc != CMDI_OK) return rc;
    } else {
        GemmParams p = gp(x, e->w_in_pad, e->b_in, e->tokA, B * T, d, e->Cpad, 0, e->Cpad, d);
        p.pe = e->pe; p.T = T; p.S = S; p.Cf = C; p.Bdup = e->cfg ? B : 0;
        HIPCHK(launch_gemm(GK_INPROJ, p, e->io_pipe, s));
    }
    int rc = for_groups(e, n_seq, s, [&](int seq0, int nseq, hipStream_t gs) {
        return run_layers(e, seq0, nseq, keep, e->profile, gs);
    });
    if (rc != CMDI_OK) return rc;
    // OutputProcess on tokens 1..T, stored straight into [n_seq, C, 1, T] (mdm.py:284,304,412-422)
    if (e->io_h3) {
        int rc2 = output_projection_h3(e, e->tokS, out_buf, n_seq, s);
        if (rc2 != CMDI_OK) return rc2;
    } else {
        GemmParams p = gp(e->w_out, e->tokA, e->b_out, out_buf, C, n_seq * T, d, d, d, 0);
        p.T = T; p.S = S; p.Cf = C;
        HIPCHK(launch_gemm(GK_OUTPROJ, p, e->io_pipe, s));
    }
    e->stash_valid = keep;
    return CMDI_OK;
}

// ---- dX backward of the encoder layers over sequences [seq0, seq0 + nseq) -----------------------
int run_layers_bwd(cmdi_engine* e, int seq0, int nseq, hipStream_t s) {
    const int S = e->T + 1, d = e->d, f = e->f;
    const int M = nseq * S;
    const size_t r0 = (size_t)seq0 * S;
    float* dA = e->dA + r0 * d;
    float* dB = e->dB + r0 * d;
    float* dH = e->dH + r0 * d;
    float* dqkv = e->dqkv + r0 * 3 * d;
    float* dffn = e->dffn + r0 * f;
    const int tile = e->gemm_tile;
    const bool h3 = e->precision == CMDI_PREC_F16X3;
    _Float16* dBS = h3 ? e->dBS + r0 * 2 * d : nullptr;
    _Float16* dffnS = h3 ? e->dffnS + r0 * 2 * f : nullptr;
    _Float16* dqkvS = h3 ? e->dqkvS + r0 * 6 * d : nullptr;
    _Float16* dOS = h3 ? e->dOS + r0 * 2 * d : nullptr;
    auto hp = [&](const _Float16* A, const _Float16* W, float* C, _Float16* Cs, int N, int K) {
        H3Params p{};
        // gradients carry no range flag: a gradient beyond the f16 range would already have shown
        // up as a non-finite sample (bench / callers check), and the forward pass guards the rest
        p.A = A; p.W = W; p.C = C; p.Cs = Cs;
        p.M = M; p.N = N; p.K = K; p.ldc = N;
        return p;
    };
    for (int l = e->L - 1; l >= 0; --l) {
        const LayerW& w = e->layers[l];
        const LayerStash& st = e->stash[l];
        if (h3) {
            // same chain with the four dX GEMMs on the f16 matrix pipe (weights: split transposes)
            HIPCHK(launch_layernorm_bwd(st.pre2 + r0 * d, st.stats2 + r0 * 2, w.n2_g, dA, dB, dBS, M, d, s));
            {   // dffn = (dB · W2) * gelu'(aux)
                H3Params p = hp(dBS, w.l2_wTs, nullptr, dffnS, f, d);
                p.aux = st.aux + r0 * f;
                HIPCHK(launch_gemm_h3(H3_GELUGRAD_SPLIT, p, e->h3_tile_ffn1, s));
            }
            {   // dH = dffn · W1 + dB
                H3Params p = hp(dffnS, w.l1_wTs, dH, nullptr, d, f);
                p.R = dB;
                HIPCHK(launch_gemm_h3(H3_RESID, p, e->h3_tile_ffn2, s));
            }
            HIPCHK(launch_layernorm_bwd(st.pre1 + r0 * d, st.stats1 + r0 * 2, w.n1_g, dH, dB, dBS, M, d, s));
            {   // d attn = dB · Wo -> dH (fp32, for D = rowsum(dO*O)) and dOS (split, MFMA operand)
                H3Params p = hp(dBS, w.out_wTs, nullptr, dOS, d, d);
                p.aux = dH;
                HIPCHK(launch_gemm_h3(H3_PLAIN_SPLIT, p, e->h3_tile_proj, s));
            }
            HIPCHK(launch_attention_bwd_h3(st.qkvS + r0 * 6 * d, st.attn + r0 * d,
                                           st.row_stats + (size_t)seq0 * e->H * S * 2, dH, dOS, dqkvS,
                                           e->drowdot + (size_t)seq0 * e->H * S, nseq, S, e->H, s));
            {   // dA = dqkv · Wqkv + dB
                H3Params p = hp(dqkvS, w.in_wTs, dA, nullptr, d, 3 * d);
                p.R = dB;
                HIPCHK(launch_gemm_h3(H3_RESID, p, e->h3_tile_proj, s));
            }
            continue;
        }
        // norm2
        HIPCHK(launch_layernorm_bwd(st.pre2 + r0 * d, st.stats2 + r0 * 2, w.n2_g, dA, dB, nullptr, M, d, s));
        // linear2 + GELU: dffn = (dB · W2) * gelu'(aux)
        {
            GemmParams p = gp(dB, w.l2_wT, nullptr, dffn, M, f, d, d, d, f);
            p.aux = st.aux + r0 * f;
            HIPCHK(gemm_any(e, GK_GELUGRAD, p, w.l2_wTx, tile, s));
        }
        // linear1 + residual: dH = dffn · W1 + dB
        {
            GemmParams p = gp(dffn, w.l1_wT, nullptr, dH, M, d, f, f, f, d);
            p.R = dB;
            HIPCHK(gemm_any(e, GK_ACCUM, p, w.l1_wTx, tile, s));
        }
        // norm1
        HIPCHK(launch_layernorm_bwd(st.pre1 + r0 * d, st.stats1 + r0 * 2, w.n1_g, dH, dB, nullptr, M, d, s));
        // out_proj: d attn = dB · Wo   (no bias, no residual) -> reuse dH as d attn
        HIPCHK(gemm_any(e, GK_PLAIN, gp(dB, w.out_wT, nullptr, dH, M, d, d, d, d, d), w.out_wTx, tile, s));
        // attention core
        HIPCHK(launch_attention_bwd(st.qkv + r0 * 3 * d, st.attn + r0 * d,
                                    st.row_stats + (size_t)seq0 * e->H * S * 2, dH, dqkv,
                                    e->drowdot + (size_t)seq0 * e->H * S, nseq, S, e->H, s));
        // in_proj + residual: dA = dqkv · Wqkv + dB
        {
            GemmParams p = gp(dqkv, w.in_wT, nullptr, dA, M, d, 3 * d, 3 * d, 3 * d, d);
            p.R = dB;
            HIPCHK(gemm_any(e, GK_ACCUM, p, w.in_wTx, tile, s));
        }
    }
    return CMDI_OK;
}

// ---- dX backward: gx[n_seq,C,T] = (d out / d x)ᵀ · gout[n_seq,C,T], per sequence ---------------
int mdm_backward(cmdi_engine* e, const float* gout, float* gx, hipStream_t s) {
    const int B = e->B, T = e->T, S = T + 1, d = e->d, C = e->C;
    const int n_seq = e->cfg ? 2 * B : B;
    const int M = n_seq * S;
    if (!e->stash_valid) return fail(CMDI_E_STATE, "cmdi_mdm_vjp: no stashed forward pass");
    if (e->unet) {   // the U-Net's own input-VJP (unet.hip), under the same power-of-two gradient scale
        HIPCHK(hipMemsetAsync(e->gs_bits, 0, sizeof(unsigned), s));
        HIPCHK(launch_absmax_bits(gout, (int64_t)n_seq * C * T, e->gs_bits, s));
        if (unet_backward(e->unet, gout, e->have_obs ? e->obs_mask : nullptr, e->gs_bits, B, n_seq, T, gx, s) != 0)
            return fail(CMDI_E_HIP, std::string("UNET backward: ") + unet_error(e->unet));
        return CMDI_OK;
    }

    // output projection: d tok[b*S+1+t][k] = sum_c gout[b][c][t] W_out[c][k]; token 0 rows get 0
    HIPCHK(hipMemsetAsync(e->dA, 0, (size_t)M * d * sizeof(float), s));
    const bool h3 = e->precision == CMDI_PREC_F16X3;
    if (h3) {   // power-of-two gradient scale: applied here, undone by the last GEMM below
        HIPCHK(hipMemsetAsync(e->gs_bits, 0, sizeof(unsigned), s));
        HIPCHK(launch_absmax_bits(gout, (int64_t)n_seq * C * T, e->gs_bits, s));
    }
    {
        GemmParams p = gp(gout, e->w_outT_pad, nullptr, e->dA, n_seq * T, d, e->Cpad, 0, e->Cpad, d);
        p.T = T; p.S = S; p.Cf = C;
        p.gs_bits = h3 ? e->gs_bits : nullptr;
        HIPCHK(launch_gemm(GK_OUTPROJ_BWD, p, 0, s));
    }
    int rc = for_groups(e, n_seq, s, [&](int seq0, int nseq, hipStream_t gs) {
        return run_layers_bwd(e, seq0, nseq, gs);
    });
    if (rc != CMDI_OK) return rc;
    {   // input projection: gx[b][c][t] = sum_n dA[b*S+1+t][n] W_in[n][c]
        GemmParams p = gp(e->w_inT, e->dA, nullptr, gx, C, n_seq * T, d, d, d, 0);
        p.T = T; p.S = S; p.Cf = C;
        p.gs_bits = h3 ? e->gs_bits : nullptr;
        HIPCHK(launch_gemm(GK_OUTPROJ, p, 0, s));
    }
    return CMDI_OK;
}

int check_ready(cmdi_engine* e, bool need_schedule, bool need_model = true) {
    if (!e) return fail(CMDI_E_INVALID, "null handle");
    if (need_model && e->L == 0 && !e->unet) return fail(CMDI_E_STATE, "sampler-only engine has no denoiser");
    if (!e->finalized) return fail(CMDI_E_STATE, "weights not finalized (cmdi_finalize_weights)");
    if (!e->have_cond) return fail(CMDI_E_STATE, "condition not set (cmdi_set_condition)");
    if (need_schedule && !e->have_schedule) return fail(CMDI_E_STATE, "schedule not set (cmdi_set_schedule)");
    return CMDI_OK;
}

// Gates of utils/editing_util.py:325-346 as host integers.  imputate == 2 ('marginal'): only inside the
// reconstruction-guidance branch (gaussian_diffusion.py:424 vs :437-439).
inline bool recon_at(const cmdi_engine* e, int step) { return e->recon && step >= e->stop_rec; }
inline bool impute_at(const cmdi_engine* e, int step, bool recon) {
    return e->imputate && step >= e->stop_imp && (e->imputate != 2 || recon);
}

int build_coef(cmdi_engine* e, int sampler, int step, float eta, bool impute, bool recon,
               StepCoef* k) {
    if (step < 0 || step >= e->n_steps) return fail(CMDI_E_INVALID, "step out of range");
    std::memset(k, 0, sizeof(*k));
    k->mean_eps = e->mean_type == CMDI_MEAN_EPSILON;
    k->clip = k->mean_eps ? e->clip_x0 : 0.f;
    k->impute = impute;
    k->recon = recon;
    k->sra = e->sra[step];
    k->srm1a = e->srm1a[step];
    const float nz = step != 0 ? 1.0f : 0.0f;
    if (sampler == CMDI_SAMPLER_DDPM) {
        k->ddim = 0;
        k->c1 = e->c1[step];
        k->c2 = e->c2[step];
        k->sig_nz = nz * e->sigma[step];
    } else {
        // ddim_sample (gaussian_diffusion.py:1339-1351), every operation in fp32 like the tensors
        // produced by _extract_into_tensor(...).float()
        k->ddim = 1;
        const float ab = e->ab[step], abp = e->ab_prev[step];
        const volatile float r1 = sqrtf((1.0f - abp) / (1.0f - ab));
        const volatile float r2 = sqrtf(1.0f - ab / abp);
        const volatile float sig = (eta * r1) * r2;
        k->sqrt_abp = sqrtf(abp);
        const volatile float sig2 = sig * sig;
        const volatile float inner = (1.0f - abp) - sig2;
        k->dir = sqrtf(inner);
        k->sig_nz = nz * sig;
    }
    if (recon) {
        if ((int)e->recon_w.size() != e->n_steps)
            return fail(CMDI_E_STATE, "reconstruction guidance needs recon_w[n_steps]");
        const volatile float ws = e->recon_w[step] * e->sqrt_ab[step];
        k->gcoef = ws / 2.0f;
    }
    return CMDI_OK;
}

}  // namespace

extern "C" {

const char* cmdi_last_error(void) { return g_err.c_str(); }
const char* cmdi_version(void) { return "condmdi-hip 0.3 (gfx950: fp32 MFMA | exact bf16x6 MFMA | split-f16 MFMA)"; }

int cmdi_create(const cmdi_model_desc* desc, cmdi_handle* out) {
    if (!desc || !out) return fail(CMDI_E_INVALID, "null argument");
    if (desc->n_feats < 1 || desc->max_batch < 1 || desc->max_frames < 1)
        return fail(CMDI_E_INVALID, "bad model geometry");
    if (desc->arch == CMDI_ARCH_UNET) {
        // MDM_UNET denoiser: the embedding front end (pe, time_embed, embed_text) of the transformer engine +
        // the temporal U-Net of unet.hip; the sampler / condition machinery is shared
        if (desc->d_model != 512 || desc->max_frames > 224 || desc->pe_rows < 1)
            return fail(CMDI_E_INVALID, "UNET engine: latent_dim must be 512, max_frames <= 224");
        if (desc->precision == CMDI_PREC_F32) return fail(CMDI_E_INVALID, "UNET engine: only the f16x3 precision is built");
        cmdi_engine* e = new cmdi_engine();
        e->desc = *desc;
        e->d = desc->d_model; e->C = desc->n_feats; e->Tmax = desc->max_frames; e->Bmax = desc->max_batch;
        e->precision = CMDI_PREC_F16X3;
        *out = e;
        const int d = e->d;
        const size_t nseq = 2 * (size_t)e->Bmax;
        ALLOC(e->pe, (size_t)desc->pe_rows * d);
        ALLOC(e->t1_w, (size_t)d * d); ALLOC(e->t1_b, d); ALLOC(e->t2_w, (size_t)d * d); ALLOC(e->t2_b, d);
        if (desc->text_cond) { ALLOC(e->txt_w, (size_t)d * e->clip_dim); ALLOC(e->txt_b, d); }
        ALLOC(e->text_term, nseq * d); ALLOC(e->text_term_p, nseq * d); ALLOC(e->text_scale, e->Bmax);
        ALLOC(e->enc_text, (size_t)e->Bmax * e->clip_dim);
        ALLOC(e->inpaint, (size_t)e->Bmax * e->C * e->Tmax);
        ALLOC(e->obs_x0, (size_t)e->Bmax * e->C * e->Tmax);
        {
            int rc = dalloc(e, reinterpret_cast<void**>(&e->mask), (size_t)e->Bmax * e->C * e->Tmax);
            if (rc != CMDI_OK) return rc;
            rc = dalloc(e, reinterpret_cast<void**>(&e->obs_mask), (size_t)e->Bmax * e->C * e->Tmax);
            if (rc != CMDI_OK) return rc;
        }
        ALLOC(e->uemb, nseq * d);
        ALLOC(e->out_raw, nseq * e->C * e->Tmax);
        ALLOC(e->range_flag, 1);
        HIPCHK(hipMemset(e->range_flag, 0, sizeof(int)));
        ALLOC(e->gs_bits, 16);
        if (desc->want_grad) { ALLOC(e->gout, nseq * e->C * e->Tmax); ALLOC(e->gx, nseq * e->C * e->Tmax); }
        e->unet = unet_new(desc->n_feats, desc->unet_added, desc->d_model, desc->unet_mults, (int)nseq,
                           desc->text_cond != 0, desc->want_grad != 0);
        if (unet_error(e->unet)[0]) return fail(CMDI_E_INVALID, std::string("UNET: ") + unet_error(e->unet));
        e->bytes += unet_bytes(e->unet);
        e->pipelines = 0;
        return CMDI_OK;
    }
    if (desc->n_layers == 0) {
        // sampler-only engine: schedule + condition + cmdi_sampler_update / q_sample / randn, for
        // denoisers that are not the native MDM
        cmdi_engine* e = new cmdi_engine();
        e->desc = *desc;
        e->desc.text_cond = 0;
        e->desc.want_grad = 0;
        e->C = desc->n_feats; e->Tmax = desc->max_frames; e->Bmax = desc->max_batch;
        *out = e;
        ALLOC(e->text_scale, e->Bmax);
        ALLOC(e->inpaint, (size_t)e->Bmax * e->C * e->Tmax);
        int rc = dalloc(e, reinterpret_cast<void**>(&e->mask), (size_t)e->Bmax * e->C * e->Tmax);
        if (rc != CMDI_OK) return rc;
        e->finalized = true;
        return CMDI_OK;
    }
    if (desc->n_heads <= 0 || desc->d_model != desc->n_heads * 128)
        return fail(CMDI_E_INVALID, "d_model / n_heads must be 128");
    if (desc->d_model % 256 != 0 || desc->d_model > 1024)
        return fail(CMDI_E_INVALID, "d_model must be 256, 512, 768 or 1024");
    if (desc->d_ff % 128 != 0) return fail(CMDI_E_INVALID, "d_ff must be a multiple of 128");
    if (desc->max_frames > 223) return fail(CMDI_E_INVALID, "max_frames must be in [1, 223]");
    if (desc->n_layers < 1 || desc->pe_rows < desc->max_frames + 1)
        return fail(CMDI_E_INVALID, "bad model geometry");
    cmdi_engine* e = new cmdi_engine();
    e->desc = *desc;
    e->L = desc->n_layers; e->d = desc->d_model; e->f = desc->d_ff; e->H = desc->n_heads;
    e->C = desc->n_feats; e->Cpad = (desc->n_feats + 31) / 32 * 32;
    e->Tmax = desc->max_frames; e->Bmax = desc->max_batch;
    // Runtime configuration read from the environment: CMDI_PRECISION, CMDI_GROUPS, CMDI_PIPELINES, CMDI_GRAPH (all
    // select between complete, parity-tested schedules).  Tile / fusion tuning knobs exist in the probes build only.
    auto env_int = [](const char* name, int dflt) {
        const char* v = std::getenv(name);
        return v ? std::atoi(v) : dflt;
    };
#ifdef CMDI_PROBES
    auto env_probe = env_int;
#else
    auto env_probe = [](const char*, int dflt) { return dflt; };
#endif
    e->gemm_tile = env_probe("CMDI_GEMM_TILE", 0);
    e->tile_inproj = env_probe("CMDI_TILE_INPROJ", e->gemm_tile);
    e->tile_proj = env_probe("CMDI_TILE_PROJ", e->gemm_tile);
    e->tile_ffn1 = env_probe("CMDI_TILE_FFN1", e->gemm_tile);
    e->tile_ffn2 = env_probe("CMDI_TILE_FFN2", e->gemm_tile);
    e->n_groups = env_int("CMDI_GROUPS", 0);  // 0 = automatic
    e->io_pipe = env_probe("CMDI_IO_PIPE", 0);
    e->use_graph = env_int("CMDI_GRAPH", 0);
    e->pipelines = env_int("CMDI_PIPELINES", 1);
    {
        int prec = desc->precision;
        if (prec == CMDI_PREC_DEFAULT) {
            const char* v = std::getenv("CMDI_PRECISION");
            const std::string name = v ? v : "";
            prec = name == "f32" ? CMDI_PREC_F32 : name == "f16x3" ? CMDI_PREC_F16X3 : name == "bf16x6" ? CMDI_PREC_BF16X6
                                                                                                        : kDefaultPrecision;
        }
        if (prec != CMDI_PREC_F32 && prec != CMDI_PREC_F16X3 && prec != CMDI_PREC_BF16X6)
            return fail(CMDI_E_INVALID, "precision must be CMDI_PREC_DEFAULT, _F32, _F16X3 or _BF16X6");
        e->x6_variant = env_int("CMDI_X6_VAR", 2);
        if (prec == CMDI_PREC_F16X3 && (desc->d_model % 32 != 0 || desc->d_ff % 32 != 0))
            return fail(CMDI_E_INVALID, "f16x3 precision needs d_model and d_ff multiples of 32");
        e->precision = prec;
    }
    e->h3_tile_qkv = env_probe("CMDI_H3_TILE_QKV", env_probe("CMDI_H3_TILE", 0));
    e->h3_tile_proj = env_probe("CMDI_H3_TILE_PROJ", env_probe("CMDI_H3_TILE", 0));
    e->h3_tile_ffn1 = env_probe("CMDI_H3_TILE_FFN1", env_probe("CMDI_H3_TILE", 0));
    e->h3_tile_ffn2 = env_probe("CMDI_H3_TILE_FFN2", env_probe("CMDI_H3_TILE", 0));
    e->ln_fuse = env_probe("CMDI_LN_FUSE", 0) && desc->d_model == 512;
    e->io_h3 = e->precision == CMDI_PREC_F16X3 && !e->ln_fuse && env_probe("CMDI_IO_H3", 1);
    e->ln_fold = e->io_h3 && desc->d_model == 512 && env_int("CMDI_LN_FOLD", 1);
    e->qkv_head_major = env_int("CMDI_QKV_HEAD_MAJOR", 0);
    const int d = e->d, f = e->f, C = e->C;
    const size_t nseq = 2 * (size_t)e->Bmax, Smax = e->Tmax + 1, Mmax = nseq * Smax;
    *out = e;  // so that cmdi_destroy can free a half-built engine

    ALLOC(e->w_in, (size_t)d * C); ALLOC(e->b_in, d); ALLOC(e->w_in_pad, (size_t)d * e->Cpad);
    ALLOC(e->pe, (size_t)desc->pe_rows * d);
    ALLOC(e->t1_w, (size_t)d * d); ALLOC(e->t1_b, d); ALLOC(e->t2_w, (size_t)d * d); ALLOC(e->t2_b, d);
    if (desc->text_cond) { ALLOC(e->txt_w, (size_t)d * e->clip_dim); ALLOC(e->txt_b, d); }
    ALLOC(e->w_out, (size_t)C * d); ALLOC(e->b_out, C);
    e->layers.resize(e->L);
    for (LayerW& w : e->layers) {
        ALLOC(w.in_w, (size_t)3 * d * d); ALLOC(w.in_b, 3 * d);
        ALLOC(w.out_w, (size_t)d * d); ALLOC(w.out_b, d);
        ALLOC(w.l1_w, (size_t)f * d); ALLOC(w.l1_b, f);
        ALLOC(w.l2_w, (size_t)d * f); ALLOC(w.l2_b, d);
        ALLOC(w.n1_g, d); ALLOC(w.n1_b, d); ALLOC(w.n2_g, d); ALLOC(w.n2_b, d);
        if (desc->want_grad) {
            ALLOC(w.in_wT, (size_t)3 * d * d); ALLOC(w.out_wT, (size_t)d * d);
            ALLOC(w.l1_wT, (size_t)f * d); ALLOC(w.l2_wT, (size_t)d * f);
        }
    }
    ALLOC(e->text_term, nseq * d); ALLOC(e->text_term_p, nseq * d); ALLOC(e->text_scale, e->Bmax);
    ALLOC(e->enc_text, (size_t)e->Bmax * e->clip_dim);
    ALLOC(e->inpaint, (size_t)e->Bmax * C * e->Tmax);
    {
        int rc = dalloc(e, reinterpret_cast<void**>(&e->mask), (size_t)e->Bmax * C * e->Tmax);
        if (rc != CMDI_OK) return rc;
    }
    ALLOC(e->tokA, Mmax * d); ALLOC(e->tokB, Mmax * d); ALLOC(e->bufH, Mmax * d);
    ALLOC(e->qkv, Mmax * 3 * d); ALLOC(e->attn, Mmax * d); ALLOC(e->ffn, Mmax * f);
    ALLOC(e->out_raw, nseq * C * e->Tmax);
    ALLOC(e->range_flag, 1);
    ALLOC(e->gs_bits, 16);
    HIPCHK(hipMemset(e->range_flag, 0, sizeof(int)));
    if (e->precision == CMDI_PREC_BF16X6) {
        auto xalloc = [&](void** ptr, size_t elems) { return dalloc(e, ptr, elems * 6); };
        for (LayerW& w : e->layers) {
            int rc = xalloc(&w.in_wx, (size_t)3 * d * d); if (rc != CMDI_OK) return rc;
            rc = xalloc(&w.out_wx, (size_t)d * d); if (rc != CMDI_OK) return rc;
            rc = xalloc(&w.l1_wx, (size_t)f * d); if (rc != CMDI_OK) return rc;
            rc = xalloc(&w.l2_wx, (size_t)d * f); if (rc != CMDI_OK) return rc;
            if (desc->want_grad) {
                rc = xalloc(&w.in_wTx, (size_t)3 * d * d); if (rc != CMDI_OK) return rc;
                rc = xalloc(&w.out_wTx, (size_t)d * d); if (rc != CMDI_OK) return rc;
                rc = xalloc(&w.l1_wTx, (size_t)f * d); if (rc != CMDI_OK) return rc;
                rc = xalloc(&w.l2_wTx, (size_t)d * f); if (rc != CMDI_OK) return rc;
            }
        }
    }
    if (e->precision == CMDI_PREC_F16X3) {
        for (LayerW& w : e->layers) {
            ALLOC(w.in_ws, (size_t)3 * d * d * 2); ALLOC(w.out_ws, (size_t)d * d * 2);
            ALLOC(w.l1_ws, (size_t)f * d * 2); ALLOC(w.l2_ws, (size_t)d * f * 2);
        }
        if (e->ln_fold) {
            for (LayerW& w : e->layers) {
                ALLOC(w.in_wsf, (size_t)3 * d * d * 2); ALLOC(w.l1_wsf, (size_t)f * d * 2);
                ALLOC(w.in_c1, 3 * d); ALLOC(w.in_c2, 3 * d); ALLOC(w.l1_c1, f); ALLOC(w.l1_c2, f);
            }
            ALLOC(e->partA, Mmax * 32); ALLOC(e->partB, Mmax * 32);
        }
        ALLOC(e->w_in_s, (size_t)d * e->Cpad * 2); ALLOC(e->w_out_s, (size_t)C * d * 2);
        ALLOC(e->xS, (size_t)e->Bmax * e->Tmax * e->Cpad * 2);
        ALLOC(e->tokS, Mmax * d * 2); ALLOC(e->bufHS, Mmax * d * 2);
        ALLOC(e->attnS, Mmax * d * 2); ALLOC(e->ffnS, Mmax * f * 2); ALLOC(e->qkvS, Mmax * 3 * d * 2);
        if (desc->want_grad) {
            for (LayerW& w : e->layers) {
                ALLOC(w.in_wTs, (size_t)3 * d * d * 2); ALLOC(w.out_wTs, (size_t)d * d * 2);
                ALLOC(w.l1_wTs, (size_t)f * d * 2); ALLOC(w.l2_wTs, (size_t)d * f * 2);
            }
            ALLOC(e->dBS, Mmax * d * 2); ALLOC(e->dffnS, Mmax * f * 2); ALLOC(e->dqkvS, Mmax * 3 * d * 2);
            ALLOC(e->dOS, Mmax * d * 2);
        }
    }
    if (desc->want_grad) {
        ALLOC(e->w_inT, (size_t)C * d); ALLOC(e->w_outT_pad, (size_t)d * e->Cpad);
        e->stash.resize(e->L);
        for (LayerStash& st : e->stash) {
            if (e->precision == CMDI_PREC_F16X3) ALLOC(st.qkvS, Mmax * 3 * d * 2);
            else ALLOC(st.qkv, Mmax * 3 * d);
            ALLOC(st.attn, Mmax * d);
            ALLOC(st.row_stats, nseq * e->H * Smax * 2);
            ALLOC(st.pre1, Mmax * d); ALLOC(st.stats1, Mmax * 2); ALLOC(st.aux, Mmax * f);
            ALLOC(st.pre2, Mmax * d); ALLOC(st.stats2, Mmax * 2);
        }
        ALLOC(e->dA, Mmax * d); ALLOC(e->dB, Mmax * d); ALLOC(e->dH, Mmax * d);
        ALLOC(e->dqkv, Mmax * 3 * d); ALLOC(e->dffn, Mmax * f);
        ALLOC(e->drowdot, nseq * e->H * Smax);
        ALLOC(e->gout, nseq * C * e->Tmax); ALLOC(e->gx, nseq * C * e->Tmax);
    }
    return CMDI_OK;
}

int cmdi_profile_enable(cmdi_handle e, int32_t on) {
    if (!e) return fail(CMDI_E_INVALID, "null handle");
    e->profile = on != 0;
    e->ev_used = 0;
    return CMDI_OK;
}

int cmdi_profile_read(cmdi_handle e, double* total_ms, int64_t* launches, int32_t* m, int32_t* n,
                      int32_t* k) {
    if (!e || !total_ms || !launches) return fail(CMDI_E_INVALID, "null argument");
    double sum = 0.0;
    for (size_t i = 0; i + 1 < e->ev_used; i += 2) {
        HIPCHK(hipEventSynchronize(e->ev_pool[i + 1]));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e->ev_pool[i], e->ev_pool[i + 1]));
        sum += ms;
    }
    *total_ms = sum;
    *launches = (int64_t)(e->ev_used / 2);
    if (m) *m = e->prof_m;
    if (n) *n = e->prof_n;
    if (k) *k = e->prof_k;
    e->ev_used = 0;
    return CMDI_OK;
}

int cmdi_destroy(cmdi_handle h) {
    if (!h) return CMDI_OK;
    drop_graphs(h);
    if (h->unet) unet_free(h->unet);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    for (hipEvent_t ev : h->own_ev) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : h->ev_pool) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : h->gevents) (void)hipEventDestroy(ev);
    for (hipStream_t st : h->gstreams) (void)hipStreamDestroy(st);
    for (void* p : h->allocs) (void)hipFree(p);
    delete h;
    return CMDI_OK;
}

int64_t cmdi_workspace_bytes(cmdi_handle h) { return h ? h->bytes : 0; }

int cmdi_load_weight(cmdi_handle e, const char* name, const float* d_src, int64_t numel,
                     cmdi_stream stream) {
    if (!e || !name || !d_src) return fail(CMDI_E_INVALID, "null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int d = e->d, f = e->f, C = e->C;
    if (e->unet) {
        const int ur = unet_load_weight(e->unet, name, d_src, numel, s);
        if (ur == 0) { e->finalized = false; return CMDI_OK; }
        if (ur < 0) return fail(CMDI_E_INVALID, std::string("UNET: ") + unet_error(e->unet));
    }
    float* dst = nullptr;
    int64_t want = -1;
    const std::string n(name);
    int l = -1;
    char rest[96] = {0};
    if (n == "input_process.poseEmbedding.weight") { dst = e->w_in; want = (int64_t)d * C; }
    else if (n == "input_process.poseEmbedding.bias") { dst = e->b_in; want = d; }
    else if (n == "sequence_pos_encoder.pe") { dst = e->pe; want = (int64_t)e->desc.pe_rows * d; }
    else if (n == "embed_timestep.sequence_pos_encoder.pe") { return CMDI_OK; /* alias of the above */ }
    else if (n == "embed_timestep.time_embed.0.weight") { dst = e->t1_w; want = (int64_t)d * d; }
    else if (n == "embed_timestep.time_embed.0.bias") { dst = e->t1_b; want = d; }
    else if (n == "embed_timestep.time_embed.2.weight") { dst = e->t2_w; want = (int64_t)d * d; }
    else if (n == "embed_timestep.time_embed.2.bias") { dst = e->t2_b; want = d; }
    else if (n == "embed_text.weight" && e->txt_w) { dst = e->txt_w; want = (int64_t)d * e->clip_dim; }
    else if (n == "embed_text.bias" && e->txt_b) { dst = e->txt_b; want = d; }
    else if (n == "output_process.poseFinal.weight") { dst = e->w_out; want = (int64_t)C * d; }
    else if (n == "output_process.poseFinal.bias") { dst = e->b_out; want = C; }
    else if (std::sscanf(name, "seqTransEncoder.layers.%d.%95s", &l, rest) == 2 && l >= 0 && l < e->L) {
        LayerW& w = e->layers[l];
        const std::string r(rest);
        if (r == "self_attn.in_proj_weight") { dst = w.in_w; want = (int64_t)3 * d * d; }
        else if (r == "self_attn.in_proj_bias") { dst = w.in_b; want = 3 * d; }
        else if (r == "self_attn.out_proj.weight") { dst = w.out_w; want = (int64_t)d * d; }
        else if (r == "self_attn.out_proj.bias") { dst = w.out_b; want = d; }
        else if (r == "linear1.weight") { dst = w.l1_w; want = (int64_t)f * d; }
        else if (r == "linear1.bias") { dst = w.l1_b; want = f; }
        else if (r == "linear2.weight") { dst = w.l2_w; want = (int64_t)d * f; }
        else if (r == "linear2.bias") { dst = w.l2_b; want = d; }
        else if (r == "norm1.weight") { dst = w.n1_g; want = d; }
        else if (r == "norm1.bias") { dst = w.n1_b; want = d; }
        else if (r == "norm2.weight") { dst = w.n2_g; want = d; }
        else if (r == "norm2.bias") { dst = w.n2_b; want = d; }
    }
    if (!dst) return fail(CMDI_E_UNKNOWN_WEIGHT, std::string("unknown weight: ") + name);
    if (numel != want)
        return fail(CMDI_E_INVALID, std::string(name) + ": expected " + std::to_string(want) +
                                        " elements, got " + std::to_string(numel));
    HIPCHK(hipMemcpyAsync(dst, d_src, (size_t)numel * sizeof(float), hipMemcpyDeviceToDevice, s));
    e->finalized = false;
    return CMDI_OK;
}

int cmdi_finalize_weights(cmdi_handle e, int32_t n_time_rows, cmdi_stream stream) {
    if (!e) return fail(CMDI_E_INVALID, "null handle");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int d = e->d, f = e->f, C = e->C;
    if (n_time_rows < 1 || n_time_rows > e->desc.pe_rows)
        return fail(CMDI_E_INVALID, "n_time_rows must be in [1, pe_rows]");
    if (e->unet) {
        const int ur = unet_finalize(e->unet, s);
        if (ur == -2) return fail(CMDI_E_RANGE, std::string("UNET: ") + unet_error(e->unet));
        if (ur != 0) return fail(CMDI_E_HIP, std::string("UNET: ") + unet_error(e->unet));
    } else
    HIPCHK(launch_pad_copy(e->w_in_pad, e->w_in, d, C, e->Cpad, s));
    if (e->desc.want_grad && !e->unet) {
        HIPCHK(launch_transpose_pad(e->w_inT, e->w_in, d, C, d, s));          // [C][d]
        HIPCHK(launch_transpose_pad(e->w_outT_pad, e->w_out, C, d, e->Cpad, s));  // [d][Cpad]
        for (LayerW& w : e->layers) {
            HIPCHK(launch_transpose_pad(w.in_wT, w.in_w, 3 * d, d, 3 * d, s));  // [d][3d]
            HIPCHK(launch_transpose_pad(w.out_wT, w.out_w, d, d, d, s));
            HIPCHK(launch_transpose_pad(w.l1_wT, w.l1_w, f, d, f, s));          // [d][f]
            HIPCHK(launch_transpose_pad(w.l2_wT, w.l2_w, d, f, d, s));          // [f][d]
        }
    }
    if (e->precision == CMDI_PREC_F16X3) {
        HIPCHK(hipMemsetAsync(e->range_flag, 0, sizeof(int), s));
        if (!e->unet) {
            HIPCHK(launch_split_f16(e->w_in_pad, e->w_in_s, d, e->Cpad, e->Cpad, e->range_flag, s));
            HIPCHK(launch_split_f16(e->w_out, e->w_out_s, C, d, d, e->range_flag, s));
        }
        for (LayerW& w : e->layers) {
            HIPCHK(launch_split_f16(w.in_w, w.in_ws, 3 * d, d, d, e->range_flag, s));
            HIPCHK(launch_split_f16(w.out_w, w.out_ws, d, d, d, e->range_flag, s));
            HIPCHK(launch_split_f16(w.l1_w, w.l1_ws, f, d, d, e->range_flag, s));
            HIPCHK(launch_split_f16(w.l2_w, w.l2_ws, d, f, f, e->range_flag, s));
            if (e->desc.want_grad) {
                HIPCHK(launch_split_f16(w.in_wT, w.in_wTs, d, 3 * d, 3 * d, e->range_flag, s));
                HIPCHK(launch_split_f16(w.out_wT, w.out_wTs, d, d, d, e->range_flag, s));
                HIPCHK(launch_split_f16(w.l1_wT, w.l1_wTs, d, f, f, e->range_flag, s));
                HIPCHK(launch_split_f16(w.l2_wT, w.l2_wTs, f, d, d, e->range_flag, s));
            }
        }
    }
    if (e->precision == CMDI_PREC_F16X3 && e->ln_fold && !e->unet) {
        // LayerNorm folded into its consumers: gamma into the weights, (row sums, W beta + b) for the epilogue
        float* tmp = nullptr;
        const size_t tmp_n = (size_t)(3 * d > f ? 3 * d : f) * d;
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&tmp), tmp_n * sizeof(float)));
        hipError_t fe = hipSuccess;
        for (int l = 0; l < e->L && fe == hipSuccess; ++l) {
            LayerW& w = e->layers[l];
            fe = launch_fold_ln(w.l1_w, w.n1_g, w.n1_b, w.l1_b, tmp, w.l1_c1, w.l1_c2, f, d, s);
            if (fe == hipSuccess) fe = launch_split_f16(tmp, w.l1_wsf, f, d, d, e->range_flag, s);
            if (l > 0 && fe == hipSuccess) {
                const LayerW& pv = e->layers[l - 1];
                fe = launch_fold_ln(w.in_w, pv.n2_g, pv.n2_b, w.in_b, tmp, w.in_c1, w.in_c2, 3 * d, d, s);
                if (fe == hipSuccess) fe = launch_split_f16(tmp, w.in_wsf, 3 * d, d, d, e->range_flag, s);
            }
        }
        hipError_t se = hipStreamSynchronize(s);   // one-time setup: tmp must outlive the kernels
        (void)hipFree(tmp);
        HIPCHK(fe); HIPCHK(se);
    }
    if (e->precision == CMDI_PREC_BF16X6) {
        for (LayerW& w : e->layers) {
            HIPCHK(launch_pack_x6(w.in_w, w.in_wx, 3 * d, d, d, s));
            HIPCHK(launch_pack_x6(w.out_w, w.out_wx, d, d, d, s));
            HIPCHK(launch_pack_x6(w.l1_w, w.l1_wx, f, d, d, s));
            HIPCHK(launch_pack_x6(w.l2_w, w.l2_wx, d, f, f, s));
            if (e->desc.want_grad) {
                HIPCHK(launch_pack_x6(w.in_wT, w.in_wTx, d, 3 * d, 3 * d, s));
                HIPCHK(launch_pack_x6(w.out_wT, w.out_wTx, d, d, d, s));
                HIPCHK(launch_pack_x6(w.l1_wT, w.l1_wTx, d, f, f, s));
                HIPCHK(launch_pack_x6(w.l2_wT, w.l2_wTx, f, d, d, s));
            }
        }
    }
    // TimestepEmbedder (mdm.py:351-353) for every original timestep: Linear -> SiLU -> Linear on pe[t]
    if (!e->time_table || e->n_time_rows < n_time_rows) {
        int rc = falloc(e, &e->time_table, (size_t)n_time_rows * d);
        if (rc != CMDI_OK) return rc;
    }
    e->n_time_rows = n_time_rows;
    float* tmp = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&tmp), (size_t)n_time_rows * d * sizeof(float)));
    hipError_t e1 = launch_gemm(GK_SILU, gp(e->pe, e->t1_w, e->t1_b, tmp, n_time_rows, d, d, d, d, d), 0, s);
    hipError_t e2 = launch_gemm(GK_PLAIN, gp(tmp, e->t2_w, e->t2_b, e->time_table, n_time_rows, d, d, d, d, d), 4, s);
    hipError_t e3 = hipStreamSynchronize(s);  // one-time setup: tmp must outlive the kernels
    (void)hipFree(tmp);
    HIPCHK(e1); HIPCHK(e2); HIPCHK(e3);
    if (e->precision == CMDI_PREC_F16X3) {
        int flag = 0;
        HIPCHK(hipMemcpy(&flag, e->range_flag, sizeof(int), hipMemcpyDeviceToHost));
        if (flag)
            return fail(CMDI_E_RANGE, "a weight is not finite or exceeds the f16 range (|w| >= 65504): "
                                      "create the engine with precision = CMDI_PREC_BF16X6 (or CMDI_PREC_F32)");
    }
    e->finalized = true;
    return CMDI_OK;
}

int cmdi_set_schedule(cmdi_handle e, const cmdi_schedule* sc) {
    if (!e || !sc) return fail(CMDI_E_INVALID, "null argument");
    if (sc->n_steps < 1) return fail(CMDI_E_INVALID, "n_steps < 1");
    if (!sc->post_coef1 || !sc->post_coef2 || !sc->sigma || !sc->sqrt_ab || !sc->sqrt_1mab ||
        !sc->sqrt_recip_ab || !sc->sqrt_recipm1_ab || !sc->ab || !sc->ab_prev || !sc->timestep_map)
        return fail(CMDI_E_INVALID, "schedule table missing");
    const int n = sc->n_steps;
    e->n_steps = n;
    e->mean_type = sc->mean_type;
    e->clip_x0 = sc->clip_x0 > 0.f ? sc->clip_x0 : 0.f;
    e->c1.assign(sc->post_coef1, sc->post_coef1 + n);
    e->c2.assign(sc->post_coef2, sc->post_coef2 + n);
    e->sigma.assign(sc->sigma, sc->sigma + n);
    e->sqrt_ab.assign(sc->sqrt_ab, sc->sqrt_ab + n);
    e->sqrt_1mab.assign(sc->sqrt_1mab, sc->sqrt_1mab + n);
    e->sra.assign(sc->sqrt_recip_ab, sc->sqrt_recip_ab + n);
    e->srm1a.assign(sc->sqrt_recipm1_ab, sc->sqrt_recipm1_ab + n);
    e->ab.assign(sc->ab, sc->ab + n);
    e->ab_prev.assign(sc->ab_prev, sc->ab_prev + n);
    e->tmap.assign(sc->timestep_map, sc->timestep_map + n);
    for (int i = 0; i < n; ++i)
        if (e->tmap[i] < 0) return fail(CMDI_E_INVALID, "negative timestep in timestep_map");
    e->have_schedule = true;
    drop_graphs(e);
    return CMDI_OK;
}

int cmdi_set_condition(cmdi_handle e, const cmdi_condition* c, cmdi_stream stream) {
    if (!e || !c) return fail(CMDI_E_INVALID, "null argument");
    if (!e->finalized) return fail(CMDI_E_STATE, "weights not finalized");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (c->batch < 1 || c->batch > e->Bmax) return fail(CMDI_E_INVALID, "batch exceeds max_batch");
    if (c->n_frames < 1 || c->n_frames > e->Tmax) return fail(CMDI_E_INVALID, "n_frames exceeds max_frames");
    if (c->cfg && !c->d_text_scale) return fail(CMDI_E_INVALID, "cfg needs text_scale");
    if ((c->imputate || c->recon_guidance) && (!c->d_inpaint_mask || !c->d_inpaint_motion))
        return fail(CMDI_E_INVALID, "imputation / reconstruction guidance need inpainting_mask and inpainted_motion");
    if (c->recon_guidance && (e->L > 0 || e->unet) && !e->desc.want_grad)
        return fail(CMDI_E_STATE, "reconstruction guidance needs an engine created with want_grad=1");
    if (c->recon_guidance && !c->recon_w) return fail(CMDI_E_INVALID, "reconstruction guidance needs recon_w");
    const int B = c->batch, T = c->n_frames, d = e->d;
    const size_t n = (size_t)B * e->C * T;
    e->B = B; e->T = T; e->cfg = c->cfg ? 1 : 0;
    e->imputate = c->imputate; e->stop_imp = c->stop_imputation_at;
    e->recon = c->recon_guidance; e->stop_rec = c->stop_recguidance_at;
    e->have_mask = c->d_inpaint_mask != nullptr;
    if (c->d_text_scale)
        HIPCHK(hipMemcpyAsync(e->text_scale, c->d_text_scale, B * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (c->d_inpaint_mask)
        HIPCHK(hipMemcpyAsync(e->mask, c->d_inpaint_mask, n, hipMemcpyDeviceToDevice, s));
    if (c->d_inpaint_motion)
        HIPCHK(hipMemcpyAsync(e->inpaint, c->d_inpaint_motion, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    e->have_obs = false;
    if (e->unet) {
        if ((c->d_obs_x0 == nullptr) != (c->d_obs_mask == nullptr))
            return fail(CMDI_E_INVALID, "with spatial-conditioning, both obs_x0 and obs_mask must be provided");
        if (e->desc.unet_added && !c->d_obs_x0)
            return fail(CMDI_E_INVALID, "a keyframe-conditioned UNET needs obs_x0 and obs_mask");
        if (c->d_obs_x0) {
            HIPCHK(hipMemcpyAsync(e->obs_x0, c->d_obs_x0, n * sizeof(float), hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemcpyAsync(e->obs_mask, c->d_obs_mask, n, hipMemcpyDeviceToDevice, s));
            e->have_obs = true;
        }
    }
    e->recon_w.clear();
    if (c->recon_w) {
        if (!e->have_schedule) return fail(CMDI_E_STATE, "set the schedule before a condition with recon_w");
        e->recon_w.assign(c->recon_w, c->recon_w + e->n_steps);
    }
    // embed_text(mask_cond(enc_text)) (mdm.py:248-251): conditional rows W·c + b, unconditional rows
    // (force_mask -> zeros) collapse to the bias.
    e->have_text = e->desc.text_cond != 0;
    if (e->have_text) {
        if (c->d_enc_text) {
            HIPCHK(hipMemcpyAsync(e->enc_text, c->d_enc_text, (size_t)B * e->clip_dim * sizeof(float),
                                  hipMemcpyDeviceToDevice, s));
            HIPCHK(launch_gemm(GK_PLAIN, gp(e->enc_text, e->txt_w, e->txt_b, e->text_term, B, d,
                                            e->clip_dim, e->clip_dim, e->clip_dim, d), 4, s));
        } else {
            HIPCHK(launch_fill_rows(e->text_term, e->txt_b, B, d, s));
        }
        if (e->cfg) HIPCHK(launch_fill_rows(e->text_term + (size_t)B * d, e->txt_b, B, d, s));
    }
    e->have_cond = true;
    e->stash_valid = false;
    drop_graphs(e);
    return CMDI_OK;
}

int cmdi_mdm_forward(cmdi_handle e, const float* d_x, const int64_t* d_t, float* d_out,
                     float* d_out_uncond, cmdi_stream stream) {
    int rc = check_ready(e, false);
    if (rc != CMDI_OK) return rc;
    if (!d_x || !d_t || !d_out) return fail(CMDI_E_INVALID, "null tensor");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t n = (size_t)e->B * e->C * e->T;
    const bool keep = e->desc.want_grad != 0;
    if (!e->cfg) {
        if (d_out_uncond) return fail(CMDI_E_INVALID, "d_out_uncond needs a cfg condition");
        return mdm_forward(e, d_x, d_t, 0, d_out, keep, s);
    }
    rc = mdm_forward(e, d_x, d_t, 0, e->out_raw, keep, s);
    if (rc != CMDI_OK) return rc;
    if (d_out_uncond) {
        HIPCHK(hipMemcpyAsync(d_out, e->out_raw, n * sizeof(float), hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(d_out_uncond, e->out_raw + n, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    } else {
        HIPCHK(launch_cfg_combine(e->out_raw, e->out_raw + n, e->text_scale, d_out, e->B,
                                  (int64_t)e->C * e->T, s));
    }
    return CMDI_OK;
}

int cmdi_mdm_vjp(cmdi_handle e, const float* d_gout, float* d_gx, cmdi_stream stream) {
    int rc = check_ready(e, false);
    if (rc != CMDI_OK) return rc;
    if (!e->desc.want_grad) return fail(CMDI_E_STATE, "engine created without want_grad");
    if (!d_gout || !d_gx) return fail(CMDI_E_INVALID, "null tensor");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t per = (int64_t)e->C * e->T;
    const size_t n = (size_t)e->B * per;
    if (!e->cfg) return mdm_backward(e, d_gout, d_gx, s);
    // hat = out_u + s*(out_c - out_u)  =>  d out_c = s*g, d out_u = g - s*g; x feeds both passes
    HIPCHK(launch_cfg_split(d_gout, e->text_scale, e->gout, e->gout + n, e->B, per, s));
    rc = mdm_backward(e, e->gout, e->gx, s);
    if (rc != CMDI_OK) return rc;
    HIPCHK(launch_add2(d_gx, e->gx, e->gx + n, (int64_t)n, s));
    return CMDI_OK;
}

int cmdi_sampler_update(cmdi_handle e, int32_t sampler, int32_t step, float eta,
                        const float* d_model_out, const float* d_recon_grad, float* d_x,
                        float* d_pred_xstart, const float* d_noise, uint64_t seed,
                        int64_t first_sample, cmdi_stream stream) {
    int rc = check_ready(e, true, false);
    if (rc != CMDI_OK) return rc;
    if (!d_model_out || !d_x) return fail(CMDI_E_INVALID, "null tensor");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool recon = recon_at(e, step) && d_recon_grad != nullptr;
    const bool impute = impute_at(e, step, recon);
    if (e->mean_type == CMDI_MEAN_EPSILON && (impute || recon))
        return fail(CMDI_E_INVALID, "This feature supports only X_start pred for now!");
    StepCoef k;
    rc = build_coef(e, sampler, step, eta, impute, recon, &k);
    if (rc != CMDI_OK) return rc;
    SamplerIO io{};
    io.x = d_x; io.out_c = d_model_out; io.out_u = nullptr; io.text_scale = nullptr;
    io.mask = e->mask; io.inpaint = e->inpaint; io.grad_c = d_recon_grad; io.grad_u = nullptr;
    io.noise = d_noise; io.pred_xstart = d_pred_xstart;
    HIPCHK(launch_sampler_step(io, k, e->B, (int64_t)e->C * e->T, seed, first_sample, step, s));
    return CMDI_OK;
}

// ---- independent pipelines -------------------------------------------------------------------------
// Samples never interact inside a chain, so cmdi_sample_loop can cut the batch into G contiguous parts
// and run each part's WHOLE chain (input projection, layers, output projection, sampler update, next
// step ...) on its own stream with no cross-stream dependency until the end: nothing is serialised by
// the narrow stages before / after the encoder layers, and the parts drift out of phase, so one part's
// memory-bound stretches (epilogue bursts, LayerNorm) overlap the other's MFMA-bound ones.
// Sequence slots are laid out per part: [cond rows of the part | uncond rows of the part].
struct Part {
    int b0, nb;          // samples [b0, b0 + nb)
    int slot0, nslot;    // sequence slots [slot0, slot0 + nslot): nslot = nb or 2 nb (CFG)
    int idx;
    hipStream_t s;
};

static int n_parts(const cmdi_engine* e) {
    const int n_seq = e->cfg ? 2 * e->B : e->B;
    int G = e->n_groups > 0 ? e->n_groups : ((long)n_seq * (e->T + 1) >= 8192 ? 2 : 1);
    if (G > e->B) G = e->B;
    if (G > 15) G = 15;
    return G < 1 ? 1 : G;
}

static Part make_part(const cmdi_engine* e, int g, int G) {
    Part p{};
    p.b0 = (int)((long)e->B * g / G);
    p.nb = (int)((long)e->B * (g + 1) / G) - p.b0;
    p.slot0 = e->cfg ? 2 * p.b0 : p.b0;
    p.nslot = e->cfg ? 2 * p.nb : p.nb;
    p.idx = g;
    return p;
}

static int part_forward(cmdi_engine* e, const Part& pt, const float* x_part, int64_t t_scalar, bool keep) {
    const int T = e->T, S = T + 1, d = e->d, C = e->C;
    hipStream_t s = pt.s;
    float* tok = e->tokA + (size_t)pt.slot0 * S * d;
    _Float16* tok_s = e->io_h3 ? e->tokS + (size_t)pt.slot0 * S * 2 * d : nullptr;
    HIPCHK(launch_token0(tok, e->time_table, e->have_text ? e->text_term_p + (size_t)pt.slot0 * d : nullptr,
                         e->pe, nullptr, t_scalar, pt.nslot, pt.nb, S, d, e->n_time_rows, s, nullptr, nullptr,
                         tok_s, e->range_flag));
    if (e->io_h3) {
        int rc0 = input_projection_h3(e, x_part, e->xS + (size_t)pt.b0 * T * 2 * e->Cpad, tok_s, pt.nb,
                                      e->cfg ? pt.nb : 0, s);
        if (rc0 != CMDI_OK) return rc0;
    } else {
        GemmParams p = gp(x_part, e->w_in_pad, e->b_in, tok, pt.nb * T, d, e->Cpad, 0, e->Cpad, d);
        p.pe = e->pe; p.T = T; p.S = S; p.Cf = C; p.Bdup = e->cfg ? pt.nb : 0;
        HIPCHK(launch_gemm(GK_INPROJ, p, e->io_pipe, s));
    }
    int rc = run_layers(e, pt.slot0, pt.nslot, keep, false, s);
    if (rc != CMDI_OK) return rc;
    float* out = e->out_raw + (size_t)pt.slot0 * C * T;
    if (e->io_h3) {
        rc = output_projection_h3(e, tok_s, out, pt.nslot, s);
        if (rc != CMDI_OK) return rc;
    } else {
        GemmParams p = gp(e->w_out, tok, e->b_out, out, C, pt.nslot * T, d, d, d, 0);
        p.T = T; p.S = S; p.Cf = C;
        HIPCHK(launch_gemm(GK_OUTPROJ, p, e->io_pipe, s));
    }
    return CMDI_OK;
}

static int part_backward(cmdi_engine* e, const Part& pt) {
    const int T = e->T, S = T + 1, d = e->d, C = e->C;
    hipStream_t s = pt.s;
    const size_t per = (size_t)C * T;
    const float* gout = e->gout + (size_t)pt.slot0 * per;
    float* gx = e->gx + (size_t)pt.slot0 * per;
    float* dA = e->dA + (size_t)pt.slot0 * S * d;
    unsigned* gs = e->gs_bits + 1 + pt.idx;
    const bool h3 = e->precision == CMDI_PREC_F16X3;
    HIPCHK(hipMemsetAsync(dA, 0, (size_t)pt.nslot * S * d * sizeof(float), s));
    if (h3) {
        HIPCHK(hipMemsetAsync(gs, 0, sizeof(unsigned), s));
        HIPCHK(launch_absmax_bits(gout, (int64_t)pt.nslot * per, gs, s));
    }
    {
        GemmParams p = gp(gout, e->w_outT_pad, nullptr, dA, pt.nslot * T, d, e->Cpad, 0, e->Cpad, d);
        p.T = T; p.S = S; p.Cf = C;
        p.gs_bits = h3 ? gs : nullptr;
        HIPCHK(launch_gemm(GK_OUTPROJ_BWD, p, 0, s));
    }
    int rc = run_layers_bwd(e, pt.slot0, pt.nslot, s);
    if (rc != CMDI_OK) return rc;
    {
        GemmParams p = gp(e->w_inT, dA, nullptr, gx, C, pt.nslot * T, d, d, d, 0);
        p.T = T; p.S = S; p.Cf = C;
        p.gs_bits = h3 ? gs : nullptr;
        HIPCHK(launch_gemm(GK_OUTPROJ, p, 0, s));
    }
    return CMDI_OK;
}

static int part_step(cmdi_engine* e, const Part& pt, int32_t sampler, int32_t step, float eta, float* d_x,
                     const float* d_noise, uint64_t seed, int64_t first_sample) {
    const int64_t per = (int64_t)e->C * e->T;
    const bool recon = recon_at(e, step);
    const bool impute = impute_at(e, step, recon);
    float* x = d_x + (size_t)pt.b0 * per;
    int rc = part_forward(e, pt, x, e->tmap[step], recon);
    if (rc != CMDI_OK) return rc;
    const float* out_c = e->out_raw + (size_t)pt.slot0 * per;
    const float* out_u = e->cfg ? out_c + (size_t)pt.nb * per : nullptr;
    const float *grad_c = nullptr, *grad_u = nullptr;
    if (recon) {
        float* gc = e->gout + (size_t)pt.slot0 * per;
        HIPCHK(launch_recon_gout(out_c, out_u, e->text_scale + pt.b0, e->mask + (size_t)pt.b0 * per,
                                 e->inpaint + (size_t)pt.b0 * per, gc, gc + (size_t)pt.nb * per, pt.nb, per, pt.s));
        rc = part_backward(e, pt);
        if (rc != CMDI_OK) return rc;
        grad_c = e->gx + (size_t)pt.slot0 * per;
        grad_u = e->cfg ? grad_c + (size_t)pt.nb * per : nullptr;
    }
    StepCoef k;
    rc = build_coef(e, sampler, step, eta, impute, recon, &k);
    if (rc != CMDI_OK) return rc;
    SamplerIO io{};
    io.x = x; io.out_c = out_c; io.out_u = out_u; io.text_scale = e->text_scale + pt.b0;
    io.mask = e->mask + (size_t)pt.b0 * per; io.inpaint = e->inpaint + (size_t)pt.b0 * per;
    io.grad_c = grad_c; io.grad_u = grad_u;
    io.noise = d_noise ? d_noise + (size_t)pt.b0 * per : nullptr; io.pred_xstart = nullptr;
    HIPCHK(launch_sampler_step(io, k, pt.nb, per, seed, first_sample + pt.b0, step, pt.s));
    return CMDI_OK;
}

static int check_step(cmdi_engine* e, int32_t step);

static int sample_loop_pipelines(cmdi_engine* e, int32_t sampler, int32_t first_step, int32_t last_step,
                                 float eta, float* d_x, const float* d_noise_stream, uint64_t seed,
                                 int64_t first_sample, hipStream_t s) {
    const int G = n_parts(e);
    const int d = e->d;
    for (int step = last_step; step <= first_step; ++step) {
        int rc = check_step(e, step);
        if (rc != CMDI_OK) return rc;
    }
    while ((int)e->gstreams.size() < G) {
        hipStream_t st;
        HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        e->gstreams.push_back(st);
    }
    while ((int)e->gevents.size() < G + 1) {
        hipEvent_t ev;
        HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        e->gevents.push_back(ev);
    }
    std::vector<Part> parts;
    for (int g = 0; g < G; ++g) {
        Part pt = make_part(e, g, G);
        pt.s = G == 1 ? s : e->gstreams[g];
        parts.push_back(pt);
    }
    // text terms in slot order (cond rows of the part, then its uncond rows)
    if (e->have_text) {
        for (const Part& pt : parts) {
            HIPCHK(hipMemcpyAsync(e->text_term_p + (size_t)pt.slot0 * d, e->text_term + (size_t)pt.b0 * d,
                                  (size_t)pt.nb * d * sizeof(float), hipMemcpyDeviceToDevice, s));
            if (e->cfg)
                HIPCHK(hipMemcpyAsync(e->text_term_p + (size_t)(pt.slot0 + pt.nb) * d,
                                      e->text_term + (size_t)(e->B + pt.b0) * d,
                                      (size_t)pt.nb * d * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
    }
    if (G > 1) {
        HIPCHK(hipEventRecord(e->gevents[0], s));
        for (const Part& pt : parts) HIPCHK(hipStreamWaitEvent(pt.s, e->gevents[0], 0));
    }
    const size_t n = (size_t)e->B * e->C * e->T;
    for (int step = first_step, i = 0; step >= last_step; --step, ++i) {
        const float* nz = d_noise_stream ? d_noise_stream + (size_t)i * n : nullptr;
        for (const Part& pt : parts) {
            int rc = part_step(e, pt, sampler, step, eta, d_x, nz, seed, first_sample);
            if (rc != CMDI_OK) return rc;
        }
    }
    if (G > 1) {
        for (const Part& pt : parts) {
            HIPCHK(hipEventRecord(e->gevents[1 + pt.idx], pt.s));
            HIPCHK(hipStreamWaitEvent(s, e->gevents[1 + pt.idx], 0));
        }
    }
    e->stash_valid = false;   // the stash holds slot-ordered rows of the last step: not a cmdi_mdm_forward stash
    return CMDI_OK;
}

static int step_impl(cmdi_engine* e, int32_t sampler, int32_t step, float eta, float* d_x,
                     float* d_pred_xstart, const float* d_noise, uint64_t seed, int64_t first_sample,
                     hipStream_t s, bool tables) {
    const int64_t per = (int64_t)e->C * e->T;
    const size_t n = (size_t)e->B * per;
    const bool recon = recon_at(e, step);
    const bool impute = impute_at(e, step, recon);
    int rc = mdm_forward(e, d_x, nullptr, e->tmap[step], e->out_raw, recon, s, tables);
    if (rc != CMDI_OK) return rc;
    const float* out_c = e->out_raw;
    const float* out_u = e->cfg ? e->out_raw + n : nullptr;
    const float *grad_c = nullptr, *grad_u = nullptr;
    if (recon) {
        HIPCHK(launch_recon_gout(out_c, out_u, e->text_scale, e->mask, e->inpaint, e->gout,
                                 e->gout + n, e->B, per, s));
        rc = mdm_backward(e, e->gout, e->gx, s);
        if (rc != CMDI_OK) return rc;
        grad_c = e->gx;
        grad_u = e->cfg ? e->gx + n : nullptr;
    }
    StepCoef k;
    rc = build_coef(e, sampler, step, eta, impute, recon, &k);
    if (rc != CMDI_OK) return rc;
    SamplerIO io{};
    io.x = d_x; io.out_c = out_c; io.out_u = out_u; io.text_scale = e->text_scale;
    io.mask = e->mask; io.inpaint = e->inpaint; io.grad_c = grad_c; io.grad_u = grad_u;
    io.noise = d_noise; io.pred_xstart = d_pred_xstart;
    HIPCHK(launch_sampler_step(io, k, e->B, per, seed, first_sample, step, s,
                               tables ? e->coef_dev : nullptr, tables ? e->cursor_dev : nullptr));
    if (tables) HIPCHK(launch_cursor_add(e->cursor_dev, -1, s));
    return CMDI_OK;
}

static int check_step(cmdi_engine* e, int32_t step) {
    if (step < 0 || step >= e->n_steps) return fail(CMDI_E_INVALID, "step out of range");
    const bool recon = recon_at(e, step);
    const bool impute = impute_at(e, step, recon);
    if (e->mean_type == CMDI_MEAN_EPSILON && (impute || recon))
        return fail(CMDI_E_INVALID, "This feature supports only X_start pred for now!");
    if (e->tmap[step] >= e->n_time_rows)
        return fail(CMDI_E_STATE, "timestep_map exceeds the finalized time-embedding table");
    return CMDI_OK;
}

int cmdi_step(cmdi_handle e, int32_t sampler, int32_t step, float eta, float* d_x,
              float* d_pred_xstart, const float* d_noise, uint64_t seed, int64_t first_sample,
              cmdi_stream stream) {
    int rc = check_ready(e, true);
    if (rc != CMDI_OK) return rc;
    if (!d_x) return fail(CMDI_E_INVALID, "null tensor");
    rc = check_step(e, step);
    if (rc != CMDI_OK) return rc;
    return step_impl(e, sampler, step, eta, d_x, d_pred_xstart, d_noise, seed, first_sample,
                     static_cast<hipStream_t>(stream), false);
}

int cmdi_set_graph(cmdi_handle e, int32_t on) {
    if (!e) return fail(CMDI_E_INVALID, "null handle");
    e->use_graph = on != 0;
    drop_graphs(e);
    return CMDI_OK;
}

// The chain [first_step .. last_step] as hipGraph replays: the first step of each kind (with / without
// reconstruction guidance) runs eagerly (one-time function attributes, stream creation), the second is
// captured, the rest are replayed.  Every variant launches the same kernels on the same device tables,
// so eager, captured and replayed steps are bitwise identical.
static int sample_loop_graph_on(cmdi_engine* e, int32_t sampler, int32_t first_step, int32_t last_step,
                                float eta, float* d_x, uint64_t seed, int64_t first_sample, hipStream_t s);

static int sample_loop_graph(cmdi_engine* e, int32_t sampler, int32_t first_step, int32_t last_step,
                             float eta, float* d_x, uint64_t seed, int64_t first_sample, hipStream_t s) {
    if (s != nullptr)
        return sample_loop_graph_on(e, sampler, first_step, last_step, eta, d_x, seed, first_sample, s);
    // the caller is on the legacy default stream, which cannot be captured: run the chain on an
    // engine-owned stream, ordered after / before the caller's stream with events
    if (!e->own_stream) {
        HIPCHK(hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&e->own_ev[0], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&e->own_ev[1], hipEventDisableTiming));
    }
    HIPCHK(hipEventRecord(e->own_ev[0], nullptr));
    HIPCHK(hipStreamWaitEvent(e->own_stream, e->own_ev[0], 0));
    int rc = sample_loop_graph_on(e, sampler, first_step, last_step, eta, d_x, seed, first_sample, e->own_stream);
    HIPCHK(hipEventRecord(e->own_ev[1], e->own_stream));
    HIPCHK(hipStreamWaitEvent(nullptr, e->own_ev[1], 0));
    return rc;
}

static int sample_loop_graph_on(cmdi_engine* e, int32_t sampler, int32_t first_step, int32_t last_step,
                                float eta, float* d_x, uint64_t seed, int64_t first_sample, hipStream_t s) {
    const int n = e->n_steps;
    if (e->table_cap < n) {
        int rc = falloc(e, &e->coef_dev, (size_t)n);
        if (rc != CMDI_OK) return rc;
        rc = falloc(e, &e->tmap_dev, (size_t)n);
        if (rc != CMDI_OK) return rc;
        if (!e->cursor_dev) { rc = falloc(e, &e->cursor_dev, 1); if (rc != CMDI_OK) return rc; }
        e->table_cap = n;
    }
    std::vector<StepCoef> tab((size_t)n);
    for (int step = last_step; step <= first_step; ++step) {
        int rc = check_step(e, step);
        if (rc != CMDI_OK) return rc;
        const bool recon = recon_at(e, step);
        const bool impute = impute_at(e, step, recon);
        rc = build_coef(e, sampler, step, eta, impute, recon, &tab[(size_t)step]);
        if (rc != CMDI_OK) return rc;
    }
    // one-off uploads per chain (pageable host memory: these copies are synchronous with the host)
    HIPCHK(hipMemcpyAsync(e->coef_dev, tab.data(), (size_t)n * sizeof(StepCoef), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(e->tmap_dev, e->tmap.data(), (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, s));
    const int first = first_step;
    HIPCHK(hipMemcpyAsync(e->cursor_dev, &first, sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));   // the host vectors above go out of scope
    if (e->graph_stream != s || e->graph_seed != seed || e->graph_first != first_sample || e->graph_x != d_x) {
        drop_graphs(e);   // captured pointers / scalars changed
        e->graph_stream = s; e->graph_seed = seed; e->graph_first = first_sample; e->graph_x = d_x;
    }
    for (int step = first_step; step >= last_step; --step) {
        const int kind = (e->recon && step >= e->stop_rec) ? 1 : 0;
        if (e->graph_exec[kind]) {
            HIPCHK(hipGraphLaunch(e->graph_exec[kind], s));
            continue;
        }
        if (!e->graph_warm[kind]) {
            int rc = step_impl(e, sampler, step, eta, d_x, nullptr, nullptr, seed, first_sample, s, true);
            if (rc != CMDI_OK) return rc;
            e->graph_warm[kind] = true;
            continue;
        }
        hipGraph_t graph = nullptr;
        HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        int rc = step_impl(e, sampler, step, eta, d_x, nullptr, nullptr, seed, first_sample, s, true);
        hipError_t ce = hipStreamEndCapture(s, &graph);
        if (rc != CMDI_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (ce != hipSuccess) return fail(CMDI_E_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
        hipError_t ie = hipGraphInstantiate(&e->graph_exec[kind], graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ie != hipSuccess) return fail(CMDI_E_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ie));
        HIPCHK(hipGraphLaunch(e->graph_exec[kind], s));
    }
    return CMDI_OK;
}

int cmdi_sample_loop(cmdi_handle e, int32_t sampler, int32_t first_step, int32_t last_step,
                     float eta, float* d_x, const float* d_noise_stream, uint64_t seed,
                     int64_t first_sample, cmdi_stream stream) {
    int rc = check_ready(e, true);
    if (rc != CMDI_OK) return rc;
    if (first_step < last_step || last_step < 0 || first_step >= e->n_steps)
        return fail(CMDI_E_INVALID, "need n_steps > first_step >= last_step >= 0");
    if (!d_x) return fail(CMDI_E_INVALID, "null tensor");
    if (e->use_graph && !d_noise_stream && !e->profile && !e->unet)
        return sample_loop_graph(e, sampler, first_step, last_step, eta, d_x, seed, first_sample,
                                 static_cast<hipStream_t>(stream));
    if (e->pipelines && !e->profile && !e->unet)
        return sample_loop_pipelines(e, sampler, first_step, last_step, eta, d_x, d_noise_stream, seed,
                                     first_sample, static_cast<hipStream_t>(stream));
    const size_t n = (size_t)e->B * e->C * e->T;
    for (int step = first_step, i = 0; step >= last_step; --step, ++i) {
        const float* nz = d_noise_stream ? d_noise_stream + (size_t)i * n : nullptr;
        rc = cmdi_step(e, sampler, step, eta, d_x, nullptr, nz, seed, first_sample, stream);
        if (rc != CMDI_OK) return rc;
    }
    return CMDI_OK;
}

int cmdi_pipeline_parts(cmdi_handle e) {
    if (!e || !e->have_cond) return 0;
    return (e->pipelines && !e->unet && !e->use_graph && e->L > 0) ? n_parts(e) : 1;
}

int cmdi_q_sample(cmdi_handle e, int32_t step, const float* d_x0, const float* d_noise, float* d_out,
                  int64_t numel, cmdi_stream stream) {
    if (!e || !e->have_schedule) return fail(CMDI_E_STATE, "schedule not set");
    if (step < 0 || step >= e->n_steps) return fail(CMDI_E_INVALID, "step out of range");
    HIPCHK(launch_q_sample(d_x0, d_noise, d_out, e->sqrt_ab[step], e->sqrt_1mab[step], numel,
                           static_cast<hipStream_t>(stream)));
    return CMDI_OK;
}

int cmdi_randn(cmdi_handle, float* d_out, int32_t batch, int64_t per_sample, uint64_t seed,
               int64_t first_sample, int32_t step, cmdi_stream stream) {
    if (!d_out || batch < 1 || per_sample < 1) return fail(CMDI_E_INVALID, "bad argument");
    HIPCHK(launch_randn(d_out, batch, per_sample, seed, first_sample, step,
                        static_cast<hipStream_t>(stream)));
    return CMDI_OK;
}

int cmdi_recover_xyz(const float* d_sample, const float* d_mean, const float* d_std, float* d_xyz,
                     int32_t batch, int32_t n_feats, int32_t n_frames, int32_t n_joints, int32_t abs_3d,
                     cmdi_stream stream) {
    if (!d_sample || !d_xyz) return fail(CMDI_E_INVALID, "null tensor");
    hipError_t err = launch_recover_xyz(d_sample, d_mean, d_std, d_xyz, batch, n_feats, n_frames, n_joints,
                                        abs_3d, static_cast<hipStream_t>(stream));
    if (err != hipSuccess)
        return fail(err == hipErrorInvalidValue ? CMDI_E_INVALID : CMDI_E_HIP,
                    std::string("cmdi_recover_xyz: ") + hipGetErrorString(err));
    return CMDI_OK;
}

int cmdi_gemm_nt(const float* d_a, const float* d_w, const float* d_bias, const float* d_resid,
                 float* d_c, int32_t m, int32_t n, int32_t k, int32_t epi, int32_t tile,
                 cmdi_stream stream) {
    if (!d_a || !d_w || !d_c) return fail(CMDI_E_INVALID, "null tensor");
    if (k % 32 != 0 || n % 32 != 0) return fail(CMDI_E_INVALID, "K and N must be multiples of 32");
    GemmKind kind;
    switch (epi) {
        case 0: kind = GK_PLAIN; break;
        case 1: kind = GK_GELU; break;
        case 3: kind = GK_RESID; break;
        default: return fail(CMDI_E_INVALID, "epi must be 0 (bias), 1 (bias+gelu) or 3 (bias+residual)");
    }
    if (kind == GK_RESID && !d_resid) return fail(CMDI_E_INVALID, "residual epilogue needs d_resid");
    GemmParams p = gp(d_a, d_w, d_bias, d_c, m, n, k, k, k, n);
    p.R = d_resid;
    hipError_t err = launch_gemm(kind, p, tile, static_cast<hipStream_t>(stream));
    if (err != hipSuccess) return fail(CMDI_E_HIP, std::string("launch_gemm: ") + hipGetErrorString(err));
    return CMDI_OK;
}

struct cmdi_clip_text { ClipText* t; };

int cmdi_clip_create(const cmdi_clip_desc* desc, cmdi_clip_handle* out) {
    if (!desc || !out) return fail(CMDI_E_INVALID, "null argument");
    ClipText* t = clip_new(desc->vocab_size, desc->width, desc->heads, desc->layers, desc->context, desc->embed_dim,
                           desc->max_batch);
    if (clip_error(t)[0]) {
        const std::string msg = clip_error(t);
        clip_free(t);
        return fail(msg.find("hipMalloc") != std::string::npos ? CMDI_E_NOMEM : CMDI_E_INVALID, msg);
    }
    *out = new cmdi_clip_text{t};
    return CMDI_OK;
}

int cmdi_clip_destroy(cmdi_clip_handle h) {
    if (!h) return CMDI_OK;
    clip_free(h->t);
    delete h;
    return CMDI_OK;
}

int cmdi_clip_load_weight(cmdi_clip_handle h, const char* name, const float* d_src, int64_t numel, cmdi_stream stream) {
    if (!h || !name || !d_src) return fail(CMDI_E_INVALID, "null argument");
    const int rc = clip_load_weight(h->t, name, d_src, numel, static_cast<hipStream_t>(stream));
    if (rc != 0) return fail(rc == -5 ? CMDI_E_UNKNOWN_WEIGHT : rc == -3 ? CMDI_E_HIP : CMDI_E_INVALID, clip_error(h->t));
    return CMDI_OK;
}

int cmdi_clip_encode_text(cmdi_clip_handle h, const int32_t* d_tokens, int32_t batch, float* d_out, cmdi_stream stream) {
    if (!h || !d_tokens || !d_out) return fail(CMDI_E_INVALID, "null argument");
    const int rc = clip_encode_text(h->t, d_tokens, batch, d_out, static_cast<hipStream_t>(stream));
    if (rc != 0) return fail(rc == -3 ? CMDI_E_HIP : CMDI_E_INVALID, clip_error(h->t));
    return CMDI_OK;
}

int cmdi_pack_x6(const float* d_src, void* d_dst, int64_t rows, int32_t cols, cmdi_stream stream) {
    if (!d_src || !d_dst || rows < 1 || cols < 32 || cols % 32 != 0)
        return fail(CMDI_E_INVALID, "bad argument (cols must be a positive multiple of 32)");
    HIPCHK(launch_pack_x6(d_src, d_dst, rows, cols, cols, static_cast<hipStream_t>(stream)));
    return CMDI_OK;
}

int cmdi_gemm_x6(const float* d_a, const void* d_w_packed, const float* d_bias, const float* d_resid, float* d_c,
                 int32_t m, int32_t n, int32_t k, int32_t epi, int32_t variant, cmdi_stream stream) {
    if (!d_a || !d_w_packed || !d_c) return fail(CMDI_E_INVALID, "null tensor");
    GemmKind kind;
    switch (epi) {
        case 0: kind = GK_PLAIN; break;
        case 1: kind = GK_GELU; break;
        case 3: kind = GK_RESID; break;
        default: return fail(CMDI_E_INVALID, "epi must be 0 (bias), 1 (bias+gelu) or 3 (bias+residual)");
    }
    if (kind == GK_RESID && !d_resid) return fail(CMDI_E_INVALID, "residual epilogue needs d_resid");
    GemmParams p = gp(d_a, nullptr, d_bias, d_c, m, n, k, k, k, n);
    p.R = d_resid;
    p.Wx = d_w_packed;
    if (!gemm_x6_supports(kind, p)) return fail(CMDI_E_INVALID, "bf16x6 GEMM needs K % 32 == 0 and N % 4 == 0");
    hipError_t err = launch_gemm_x6(kind, p, static_cast<hipStream_t>(stream), variant);
    if (err != hipSuccess) return fail(CMDI_E_HIP, std::string("launch_gemm_x6: ") + hipGetErrorString(err));
    return CMDI_OK;
}

int cmdi_attention_fwd(const float* d_qkv, float* d_out, int32_t n_seq, int32_t seq_len,
                       int32_t n_heads, cmdi_stream stream) {
    if (!d_qkv || !d_out || n_seq < 1 || seq_len < 1 || seq_len > 224 || n_heads < 1)
        return fail(CMDI_E_INVALID, "bad argument");
    HIPCHK(launch_attention_fwd(d_qkv, d_out, nullptr, nullptr, nullptr, n_seq, seq_len, n_heads,
                                static_cast<hipStream_t>(stream)));
    return CMDI_OK;
}

int cmdi_precision(cmdi_handle e) { return e ? e->precision : CMDI_E_INVALID; }

int cmdi_range_status(cmdi_handle e, int32_t* out_flag, cmdi_stream stream) {
    if (!e || !out_flag) return fail(CMDI_E_INVALID, "null argument");
    *out_flag = 0;
    if (!e->range_flag) return CMDI_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int flag = 0;
    HIPCHK(hipMemcpyAsync(&flag, e->range_flag, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (flag) HIPCHK(hipMemsetAsync(e->range_flag, 0, sizeof(int), s));
    if (e->unet) {
        int uf = 0;
        if (unet_range_flag(e->unet, &uf, s) != 0) return fail(CMDI_E_HIP, "UNET: range flag read-back failed");
        flag |= uf;
    }
    *out_flag = flag;
    return CMDI_OK;
}

int cmdi_split_f16(const float* d_src, void* d_dst, int64_t rows, int32_t cols, cmdi_stream stream) {
    if (!d_src || !d_dst || rows < 1 || cols < 32 || cols % 32 != 0)
        return fail(CMDI_E_INVALID, "bad argument (cols must be a positive multiple of 32)");
    HIPCHK(launch_split_f16(d_src, static_cast<_Float16*>(d_dst), rows, cols, cols, nullptr,
                            static_cast<hipStream_t>(stream)));
    return CMDI_OK;
}

int cmdi_gemm_h3(const void* d_a_split, const void* d_w_split, const float* d_bias,
                 const float* d_resid, float* d_c, void* d_c_split, int32_t m, int32_t n, int32_t k,
                 int32_t epi, int32_t tile, cmdi_stream stream) {
    if (!d_a_split || !d_w_split) return fail(CMDI_E_INVALID, "null tensor");
    if (k % 32 != 0 || n % 32 != 0) return fail(CMDI_E_INVALID, "K and N must be multiples of 32");
    H3Params p{};
    p.A = static_cast<const _Float16*>(d_a_split);
    p.W = static_cast<const _Float16*>(d_w_split);
    p.bias = d_bias; p.C = d_c; p.Cs = static_cast<_Float16*>(d_c_split); p.R = d_resid;
    p.M = m; p.N = n; p.K = k; p.ldc = n;
#ifdef CMDI_PROBES
    { const char* v = std::getenv("CMDI_H3_DBG"); p.dbg = v ? std::atoi(v) : 0; }
    if (p.dbg & 16) {   // bench-only: the timestamp buffer rides in d_resid's place when epi != 3
        p.dbg_buf = (epi != 3) ? reinterpret_cast<long long*>(const_cast<float*>(d_resid)) : nullptr;
        if (epi != 3) p.R = nullptr;
    }
#endif
    int kind;
    switch (epi) {
        case 0: kind = d_c_split ? H3_PLAIN_SPLIT : H3_PLAIN; break;
        case 1: kind = H3_GELU_SPLIT; break;
        case 3: kind = H3_RESID; break;
        case 4: kind = H3_RESID; p.R = nullptr; p.Rs = reinterpret_cast<const _Float16*>(d_resid); break;
        default: return fail(CMDI_E_INVALID, "epi must be 0 (bias), 1 (bias+gelu, split output), 3 (bias+residual) or 4 (bias + split-rows residual)");
    }
    if ((kind == H3_PLAIN || kind == H3_RESID) && !d_c) return fail(CMDI_E_INVALID, "fp32 output needs d_c");
    if ((kind == H3_GELU_SPLIT || kind == H3_PLAIN_SPLIT) && !d_c_split)
        return fail(CMDI_E_INVALID, "split output needs d_c_split");
    if (kind == H3_RESID && !d_resid) return fail(CMDI_E_INVALID, "residual epilogue needs d_resid");
    hipError_t err = launch_gemm_h3(kind, p, tile, static_cast<hipStream_t>(stream));
    if (err != hipSuccess) return fail(CMDI_E_HIP, std::string("launch_gemm_h3: ") + hipGetErrorString(err));
    return CMDI_OK;
}

int cmdi_conv_rows_h3(const void* d_a_split, int32_t a_ld, const void* d_w_split, const float* d_bias,
                      const float* d_resid, float* d_c, void* d_c_split, int32_t m, int32_t n, int32_t cin,
                      int32_t taps, int32_t pad, int32_t a_row_mul, int32_t c_row_mul, int32_t c_row_add,
                      int32_t tp, int32_t t_lo, int32_t t_hi, int32_t tile, cmdi_stream stream) {
    if (!d_a_split || !d_w_split || (!d_c && !d_c_split)) return fail(CMDI_E_INVALID, "null tensor");
    if (cin % 32 != 0 || n % 32 != 0 || taps < 1 || a_ld < 2 * cin)
        return fail(CMDI_E_INVALID, "cin and n must be multiples of 32, a_ld >= 2 * cin");
    H3Params p{};
    p.A = static_cast<const _Float16*>(d_a_split) - (ptrdiff_t)pad * a_ld;
    p.W = static_cast<const _Float16*>(d_w_split);
    p.bias = d_bias; p.C = d_c; p.Cs = static_cast<_Float16*>(d_c_split); p.R = d_resid;
    p.M = m; p.N = n; p.K = taps * cin; p.ldc = n;
    p.a_ld = a_ld; p.a_row_mul = a_row_mul; p.taps = taps; p.cpt = cin / 32;
    p.c_row_mul = c_row_mul; p.c_row_add = c_row_add; p.tp = tp; p.t_lo = t_lo; p.t_hi = t_hi;
    const int kind = d_c_split ? H3_PLAIN_SPLIT : (d_resid ? H3_RESID : H3_PLAIN);
    hipError_t err = launch_gemm_h3(kind, p, tile, static_cast<hipStream_t>(stream));
    if (err != hipSuccess) return fail(CMDI_E_HIP, std::string("launch_gemm_h3: ") + hipGetErrorString(err));
    return CMDI_OK;
}

int cmdi_gemm_h3_ln(const void* d_a_split, const void* d_w_split, const float* d_bias,
                    const float* d_resid, const float* d_gamma, const float* d_beta, float* d_y,
                    void* d_y_split, int32_t m, int32_t n, int32_t k, cmdi_stream stream) {
    if (!d_a_split || !d_w_split || !d_resid || !d_gamma || !d_beta || !d_y)
        return fail(CMDI_E_INVALID, "null tensor");
    if (n != 512 || k % 32 != 0) return fail(CMDI_E_INVALID, "the fused LayerNorm epilogue needs N = 512, K % 32 == 0");
    H3Params p{};
    p.A = static_cast<const _Float16*>(d_a_split);
    p.W = static_cast<const _Float16*>(d_w_split);
    p.bias = d_bias; p.R = d_resid; p.ln_g = d_gamma; p.ln_b = d_beta;
    p.C = d_y; p.Cs = static_cast<_Float16*>(d_y_split);
    p.M = m; p.N = n; p.K = k; p.ldc = n;
    hipError_t err = launch_gemm_h3(H3_RESID_LN, p, 0, static_cast<hipStream_t>(stream));
    if (err != hipSuccess) return fail(CMDI_E_HIP, std::string("launch_gemm_h3: ") + hipGetErrorString(err));
    return CMDI_OK;
}

int cmdi_attention_fwd_h3(const void* d_qkv_split, float* d_out, int32_t n_seq, int32_t seq_len,
                          int32_t n_heads, cmdi_stream stream) {
    if (!d_qkv_split || !d_out || n_seq < 1 || seq_len < 1 || seq_len > 224 || n_heads < 1)
        return fail(CMDI_E_INVALID, "bad argument");
    // (CMDI_ATTN_DBG & 16, bench only: 32 B of cycle stamps per block are written BEHIND the output,
    // the caller allocates n_seq * n_heads * 32 extra bytes)
#ifdef CMDI_PROBES
    static const bool stamps = std::getenv("CMDI_ATTN_DBG") && (std::atoi(std::getenv("CMDI_ATTN_DBG")) & 16);
#else
    constexpr bool stamps = false;
#endif
    HIPCHK(launch_attention_h3(static_cast<const _Float16*>(d_qkv_split), d_out, nullptr, nullptr,
                               stamps ? d_out + (size_t)n_seq * seq_len * n_heads * 128 : nullptr, n_seq, seq_len, n_heads,
                               static_cast<hipStream_t>(stream)));
    return CMDI_OK;
}

void cmdi_philox4x32_10(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]) {
    philox4x32_10_host(counter, key, out);
}

}  // extern "C"
