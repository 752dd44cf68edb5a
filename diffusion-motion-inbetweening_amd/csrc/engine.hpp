// Engine state behind the cmdi_handle of include/condmdi.h, shared by the translation units of the C-ABI:
//   api_engine.hip   create / destroy, weight packing, schedule and condition
//   api_denoiser.hip the MDM trans_enc forward / dX-backward schedule of kernel launches
//   api_sampler.hip  model entry points, sampler update, the sampling loop (pipelines, hipGraph replay)
//   api_hooks.hip    single-kernel entry points the parity tests call
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/condmdi.h"
#include "kernels.hpp"

namespace cmdi {
namespace host {

// sets the thread's cmdi_last_error() text and returns `code` (api_engine.hip)
int fail(int code, const std::string& msg);

}  // namespace host
}  // namespace cmdi

#define HIPCHK(expr)                                                                      \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess)                                                             \
            return fail(CMDI_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));   \
    } while (0)

namespace cmdi {
namespace host {

// Library default of cmdi_model_desc.precision = CMDI_PREC_DEFAULT.  Rule: the default must not drift more
// than the exact-fp32 engine over the full 1000-step chain.  Measured on MI355X against the reference's float64 chain
// (tests/test_gpu_parity.py::test_long_chain_drift_vs_reference): f16x3 1.0e-6, bf16x6 1.4e-6,
// fp32-MFMA 1.9e-6, the reference's own fp32 CPU chain 1.3e-6 — so the fastest mode is the default; bf16x6 (no operand
// truncation, no range limit) is what the f16-range guard falls back to.
constexpr int kDefaultPrecision = CMDI_PREC_F16X3;
// fp32 copies the folded stashing forward keeps beside its split rows (cmdi_engine::stash_f32, CMDI_STASH_F32 overrides):
// bits 1 = attention output, 2 = pre1, 4 = pre2.  Round 5: set from the measured attribution of the guided-chain error.
constexpr int kDefaultStashF32 = 0;

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
    // fp32 copies beside the split rows of the folded schedule's stash (cmdi_engine::stash_f32 bits 1 / 2 / 4), or null
    float *attn_f = nullptr, *pre1_f = nullptr, *pre2_f = nullptr;
};

}  // namespace host
}  // namespace cmdi

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
    std::vector<cmdi::host::LayerW> layers;
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
    std::vector<cmdi::host::LayerStash> stash;
    bool stash_valid = false;
    bool stash_split = false;   // the layer stash (attn / pre1 / pre2) holds split rows [M][2d] halves instead of fp32 rows (run_layers)
    float *dA = nullptr, *dB = nullptr, *dH = nullptr, *dqkv = nullptr, *dffn = nullptr,
          *drowdot = nullptr, *gout = nullptr, *gx = nullptr;
    // precision of the encoder-layer GEMMs: CMDI_PREC_F32 (exact fp32 MFMA) or CMDI_PREC_F16X3
    // (fp32-equivalent split-f16 products on the f16 matrix pipe, gemm_h3.hpp)
    int precision = CMDI_PREC_F16X3;
    _Float16 *tokS = nullptr, *bufHS = nullptr, *attnS = nullptr, *ffnS = nullptr, *qkvS = nullptr;
    _Float16 *dBS = nullptr, *dffnS = nullptr, *dqkvS = nullptr, *dOS = nullptr;  // backward operands (want_grad)
    cmdi::UnetModel* unet = nullptr;     // arch = CMDI_ARCH_UNET: the MDM_UNET denoiser (unet.hip)
    float* uemb = nullptr;          // [2 Bmax, d] time (+ text) embedding per sequence
    float* obs_x0 = nullptr;        // keyframe conditioning of the UNET (model_kwargs obs_x0 / obs_mask)
    uint8_t* obs_mask = nullptr;
    bool have_obs = false;
    int* range_flag = nullptr;
    unsigned* gs_bits = nullptr;   // max|gout| bits -> power-of-two gradient scale (f16x3 backward); [1 + parts]
    float* text_term_p = nullptr;  // text_term in the slot order of the independent pipelines (cmdi_sample_loop)
    int pipelines = 1;             // CMDI_PIPELINES=0: keep the fork/join-per-step schedule in cmdi_sample_loop
    int h3_tile_qkv = 0, h3_tile_proj = 0, h3_tile_ffn1 = 0, h3_tile_ffn2 = 0;
    // weight-stationary GEMM (gemm_h3w.hpp, K = d = 512): fragment-ordered copies of the split weights it can take, keyed by the
    // split copy's address (filled at create, packed at cmdi_finalize_weights); h3w = CMDI_H3W (0: never route to it)
    int h3w = 0, h3w_min_m = 8192;                 // CMDI_H3W, CMDI_H3W_MIN_M: route launches of at least that many rows
    std::unordered_map<const void*, _Float16*> h3w_packed;
    const _Float16* packed(const _Float16* w_split, int m) const {
        if (!h3w || m < h3w_min_m) return nullptr;
        auto it = h3w_packed.find(w_split);
        return it == h3w_packed.end() ? nullptr : it->second;
    }
    // f16x3: LayerNorm inside the out_proj / linear2 GEMM epilogue (d_model = 512).  Off by default:
    // the full-row 64x512 tile it needs (197 blocks, 8 waves per CU) loses more in the GEMM than the
    // saved LayerNorm pass returns (B=32 CFG: 2.60 vs 2.29 ms per step on MI355X).
    int ln_fuse = 0;
    // f16x3, forward without a stash: NO LayerNorm pass — the residual stream travels as its pre-LayerNorm value plus
    // per-row partial statistics and every LayerNorm is folded into the GEMM that consumes it (gemm_params.hpp);
    // CMDI_LN_FOLD=0 keeps the separate LayerNorm kernels
    int ln_fold = 0;
    int ln_fold_keep = 1;     // the folded schedule also for forward passes that stash activations (CMDI_LN_FOLD_KEEP)
    int qkv_head_major = 0;   // folded path: in_proj writes q | k | v head-major for the attention kernel (CMDI_QKV_HEAD_MAJOR=1;
                              // measured: no gain — attention 36.0 vs 35.1 us, step 2.14 vs 2.07 ms — so off)
    float *partA = nullptr, *partB = nullptr;   // [M][16][2] partial statistics of pre1 / pre2
    int io_pipe = 0;   // 1: software-pipelined input / output projection GEMMs
    // f16x3: input / output projections on the f16 pipe too (frame rows split by pose_rows_split_kernel, token and
    // motion-layout epilogues in gemm_h3.hpp); CMDI_IO_H3=0 keeps them on the fp32 MFMA kernels
    int io_h3 = 0;
    _Float16 *w_in_s = nullptr, *w_out_s = nullptr, *xS = nullptr;
    _Float16 *w_inT_s = nullptr, *w_outT_s = nullptr, *gS = nullptr;   // input-VJP boundary GEMMs on the f16 pipe: W_in^T [C][2d], W_out^T [d][2 Cpad], scaled output-gradient rows [2B T][2 Cpad]
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
    cmdi::StepCoef* coef_dev = nullptr;      // [n_steps] for the (sampler, eta) of the running chain
    int64_t* tmap_dev = nullptr;       // [n_steps] timestep_map
    int* cursor_dev = nullptr;         // current respaced step index
    int table_cap = 0;
    hipGraphExec_t graph_exec[2] = {nullptr, nullptr};   // [reconstruction guidance active?]
    // the pipelined schedule (round 4): one graph per (part, kind), captured on and replayed into the part's own stream;
    // the parts' cursors are cursor_dev[1 + part]
    std::vector<hipGraphExec_t> part_graph;              // [part * 2 + kind]
    std::vector<char> part_warm;
    int part_graph_parts = 0;
    int cursor_cap = 0;
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
    const char* prof_kernel = "";   // kernel family the bracketed launches dispatched to (cmdi_profile_kernel)
    int stash_f32 = 0;           // folded stashing forward: fp32 copy of 1 = the attention output, 2 = pre1, 4 = pre2 for the backward
    int prof_which = 0;          // kernel kind the events bracket: 0 = in_proj GEMM, 1 = attention
};

namespace cmdi {
namespace host {

inline int dalloc(cmdi_engine* e, void** p, size_t nbytes) {
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

inline void drop_graphs(cmdi_engine* e) {
    for (int i = 0; i < 2; ++i) {
        if (e->graph_exec[i]) (void)hipGraphExecDestroy(e->graph_exec[i]);
        e->graph_exec[i] = nullptr;
        e->graph_warm[i] = false;
    }
    for (hipGraphExec_t& g : e->part_graph) {
        if (g) (void)hipGraphExecDestroy(g);
        g = nullptr;
    }
    e->part_graph.clear();
    e->part_warm.clear();
}

inline GemmParams gp(const float* A, const float* W, const float* bias, float* C, int M, int N, int K,
              int lda, int ldw, int ldc) {
    GemmParams p{};
    p.A = A; p.W = W; p.bias = bias; p.C = C;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = ldc;
    p.out_scale = 1.0f;
    return p;
}

// fp32-engine GEMM: on the bf16 pipe with exact three-plane operands when the packed weight is given (CMDI_PREC_BF16X6),
// else the fp32 MFMA kernel
inline hipError_t gemm_any(const cmdi_engine* e, GemmKind kind, GemmParams p, const void* wx, int tile, hipStream_t s) {
    if (wx) {
        p.Wx = wx;
        if (gemm_x6_supports(kind, p)) return launch_gemm_x6(kind, p, s, e->x6_variant);
        p.Wx = nullptr;
    }
    return launch_gemm(kind, p, tile, s);
}

// ---- api_denoiser.hip ------------------------------------------------------------------------------
int run_layers(cmdi_engine* e, int seq0, int nseq, bool keep, bool prof, hipStream_t s);
int run_layers_bwd(cmdi_engine* e, int seq0, int nseq, hipStream_t s);
// boundary GEMMs of the input-VJP over the sequences [slot0, slot0 + nslot) (api_denoiser.hip)
int vjp_output_projection(cmdi_engine* e, const float* gout, float* dA, int slot0, int nslot, const unsigned* gs, hipStream_t s);
int vjp_input_projection(cmdi_engine* e, const float* dA, float* gx, int slot0, int nslot, const unsigned* gs, hipStream_t s);
int input_projection_h3(cmdi_engine* e, const float* x, _Float16* xs, _Float16* tok_split, int nb, int dup,
                        hipStream_t s);
int output_projection_h3(cmdi_engine* e, const _Float16* tok_split, float* out, int nseq, hipStream_t s);
int mdm_forward(cmdi_engine* e, const float* x, const int64_t* t_dev, int64_t t_scalar,
                float* out, bool keep, hipStream_t s, bool tables = false);
int mdm_backward(cmdi_engine* e, const float* gout, float* gx, hipStream_t s);

// ---- api_sampler.hip -------------------------------------------------------------------------------
int check_ready(cmdi_engine* e, bool need_schedule, bool need_model = true);

}  // namespace host
}  // namespace cmdi
