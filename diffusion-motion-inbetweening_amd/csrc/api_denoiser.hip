// C-ABI of libcondmdi_hip.so, denoiser part: the MDM trans_enc forward / dX-backward schedule of kernel launches.
#include "engine.hpp"

using namespace cmdi;
using namespace cmdi::host;

namespace cmdi {
namespace host {

// the input-VJP's boundary GEMMs run on the f16 pipe (split rows in, power-of-two gradient scale) when the forward's do
static inline bool vjp_boundary_h3(const cmdi_engine* e) {
    return e->precision == CMDI_PREC_F16X3 && e->io_h3 && e->gS && e->w_inT_s && !e->unet;
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
        p.Wp = e->packed(W, M);
        p.M = M; p.N = N; p.K = K; p.ldc = N;
        return p;
    };
    // live timing (cmdi_profile_enable): HIP events around ONE kernel kind per pass — the in_proj GEMM (prof_which 0) or the
    // attention kernel (1)
    auto prof_begin = [&](int which) -> int {
        if (!prof || e->prof_which != which) return CMDI_OK;
        if (e->ev_used + 2 > e->ev_pool.size()) {
            hipEvent_t a, b;
            HIPCHK(hipEventCreate(&a));
            HIPCHK(hipEventCreate(&b));
            e->ev_pool.push_back(a);
            e->ev_pool.push_back(b);
        }
        HIPCHK(hipEventRecord(e->ev_pool[e->ev_used], s));
        return CMDI_OK;
    };
    auto prof_end = [&](int which) -> int {
        if (!prof || e->prof_which != which) return CMDI_OK;
        HIPCHK(hipEventRecord(e->ev_pool[e->ev_used + 1], s));
        e->ev_used += 2;
        if (which == 0) {
            e->prof_m = M; e->prof_n = 3 * d; e->prof_k = d;
            e->prof_kernel = h3 ? gemm_h3_last_route() : (e->precision == CMDI_PREC_BF16X6 ? "gemm_x6_kernel" : "gemm_nt_kernel");
        } else {      // (sequences, tokens, heads)
            e->prof_m = nseq; e->prof_n = S; e->prof_k = e->H;
            e->prof_kernel = h3 ? "attention_h3_kernel" : "attention_fwd_kernel";
        }
        return CMDI_OK;
    };
    if (h3 && !e->io_h3)  // layer 0 reads the tokens assembled by token0 + the input projection (fp32)
        HIPCHK(launch_split_f16(tokA, tokS, M, d, d, e->range_flag, s));
    if (h3 && e->ln_fold && e->io_h3 && (!keep || e->ln_fold_keep)) {
        // ---- no LayerNorm pass: P (tokS) is the layer input BEFORE its LayerNorm (layer 0: the tokens themselves) ----
        float* partA = e->partA + r0 * 32;
        float* partB = e->partB + r0 * 32;
        // keep (round 3): the folded schedule also serves the forward pass that stashes activations for the input-VJP
        // (reconstruction guidance, torch.autograd through the module): the GEMM epilogues write what the backward reads —
        // split qkv per layer, the FFN pre-activation — and the GEMM that CONSUMES a LayerNorm's partial statistics writes
        // its (mean, rstd) out (p.ln_stats).  16 LayerNorm launches per evaluation less than the schedule below, which it
        // replaces unless CMDI_LN_FOLD_KEEP=0.  Round 4: the stash IS the stream — the attention output and the two
        // pre-LayerNorm sums of a layer are written ONCE, as the split rows the next GEMM multiplies, straight into the
        // layer's stash (st->attn / pre1 / pre2 hold [M][2d] halves = the bytes of their fp32 form; e->stash_split tells the
        // backward); rounds 1-3 wrote an fp32 copy beside every one of them (3 x 26 MB per layer at C3).
        if (keep) e->stash_split = true;
        const _Float16* Pin = tokS;                    // layer input BEFORE its LayerNorm
        for (int l = 0; l < e->L; ++l) {
            const LayerW& w = e->layers[l];
            const LayerW* prev = l > 0 ? &e->layers[l - 1] : nullptr;
            const LayerStash* st = keep ? &e->stash[l] : nullptr;
            _Float16* qkvL = keep ? st->qkvS + r0 * 6 * d : qkvS;
            _Float16* attnL = keep ? reinterpret_cast<_Float16*>(st->attn + r0 * d) : attnS;
            _Float16* pre1L = keep ? reinterpret_cast<_Float16*>(st->pre1 + r0 * d) : bufHS;
            _Float16* pre2L = keep ? reinterpret_cast<_Float16*>(st->pre2 + r0 * d) : tokS;
            const bool head_major = e->qkv_head_major != 0 && !keep;   // (the attention backward reads token-major rows)
            { int prc = prof_begin(0); if (prc != CMDI_OK) return prc; }
            {   // qkv = in_proj(LN2_prev(P))
                H3Params p = hp(Pin, prev ? w.in_wsf : w.in_ws, prev ? w.in_c2 : w.in_b, nullptr, qkvL, 3 * d, d);
                if (prev) { p.ln_part = partB; p.ln_c1 = w.in_c1; }
                if (prev && keep) p.ln_stats = e->stash[l - 1].stats2 + r0 * 2;
                p.cs_head_major = head_major;
                HIPCHK(launch_gemm_h3(H3_PLAIN_SPLIT, p, e->h3_tile_qkv, s));
            }
            { int prc = prof_end(0); if (prc != CMDI_OK) return prc; }
            { int prc = prof_begin(1); if (prc != CMDI_OK) return prc; }
            HIPCHK(launch_attention_h3(qkvL, keep && st->attn_f ? st->attn_f + r0 * d : nullptr, attnL, e->range_flag,
                                       keep ? st->row_stats + (size_t)seq0 * e->H * S * 2 : nullptr, nseq, S, e->H, s, head_major));
            { int prc = prof_end(1); if (prc != CMDI_OK) return prc; }
            {   // pre1 = LN2_prev(P) + out_proj(attn)   (+ partial statistics A)
                H3Params p = hp(attnL, w.out_ws, w.out_b, nullptr, pre1L, d, d);
                p.Rs = Pin;
                if (prev) { p.ln_part = partB; p.ln_rg = prev->n2_g; p.ln_rb = prev->n2_b; }
                p.out_part = partA;
                if (keep && st->pre1_f) p.C = st->pre1_f + r0 * d;
                HIPCHK(launch_gemm_h3(H3_RESID, p, e->h3_tile_proj, s));
            }
            {   // ffn = gelu(linear1(LN1(pre1)))
                H3Params p = hp(pre1L, w.l1_wsf, w.l1_c2, nullptr, ffnS, f, d);
                p.ln_part = partA; p.ln_c1 = w.l1_c1;
                if (keep) { p.aux = st->aux + r0 * f; p.ln_stats = st->stats1 + r0 * 2; }
                HIPCHK(launch_gemm_h3(H3_GELU_SPLIT, p, e->h3_tile_ffn1, s));
            }
            {   // pre2 = LN1(pre1) + linear2(ffn)   (+ partial statistics B): the next layer's P
                H3Params p = hp(ffnS, w.l2_ws, w.l2_b, nullptr, pre2L, d, f);
                p.Rs = pre1L; p.ln_part = partA; p.ln_rg = w.n1_g; p.ln_rb = w.n1_b;
                p.out_part = partB;
                if (keep && st->pre2_f) p.C = st->pre2_f + r0 * d;
                HIPCHK(launch_gemm_h3(H3_RESID, p, e->h3_tile_ffn2, s));
            }
            Pin = pre2L;
        }
        // the encoder output is LN2 of the last layer: the one LayerNorm launch that remains (split rows in -> tokS; in
        // place unless the rows are a stash)
        const LayerW& last = e->layers[e->L - 1];
        HIPCHK(launch_layernorm(nullptr, last.n2_g, last.n2_b, nullptr, tokS, e->range_flag,
                                keep ? e->stash[e->L - 1].stats2 + r0 * 2 : nullptr, M, d, s, Pin));
        return CMDI_OK;
    }
    if (keep) e->stash_split = false;   // this schedule stashes fp32 rows
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
            { int prc = prof_begin(0); if (prc != CMDI_OK) return prc; }
            // qkv leaves as split rows for the f16-pipe attention (stashed per layer for the backward)
            _Float16* qkvL = keep ? st->qkvS + r0 * 6 * d : qkvS;
            HIPCHK(launch_gemm_h3(H3_PLAIN_SPLIT, hp(tokS, w.in_ws, w.in_b, nullptr, qkvL, 3 * d, d),
                                  e->h3_tile_qkv, s));
            { int prc = prof_end(0); if (prc != CMDI_OK) return prc; }
            { int prc = prof_begin(1); if (prc != CMDI_OK) return prc; }
            HIPCHK(launch_attention_h3(qkvL, keep ? attn : nullptr, attnS, e->range_flag, row_stats,
                                       nseq, S, e->H, s));
            { int prc = prof_end(1); if (prc != CMDI_OK) return prc; }
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
        { int prc = prof_begin(0); if (prc != CMDI_OK) return prc; }
        HIPCHK(gemm_any(e, GK_PLAIN, gp(tokA, w.in_w, w.in_b, qkv, M, 3 * d, d, d, d, 3 * d), w.in_wx,
                        e->tile_inproj, s));
        { int prc = prof_end(0); if (prc != CMDI_OK) return prc; }
        { int prc = prof_begin(1); if (prc != CMDI_OK) return prc; }
        HIPCHK(launch_attention_fwd(qkv, attn, nullptr, nullptr, row_stats, nseq, S, e->H, s));
        { int prc = prof_end(1); if (prc != CMDI_OK) return prc; }
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
int input_projection_h3(cmdi_engine* e, const float* x, _Float16* xs, _Float16* tok_split, int nb, int dup,
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
int output_projection_h3(cmdi_engine* e, const _Float16* tok_split, float* out, int nseq, hipStream_t s) {
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
                float* out_buf, bool keep, hipStream_t s, bool tables) {
    const int B = e->B, T = e->T, S = T + 1, d = e->d, C = e->C;
    const int n_seq = e->cfg ? 2 * B : B;
    if (e->unet) {   // MDM_UNET.forward (model/mdm_unet.py:766-849)
        HIPCHK(launch_unet_emb(e->uemb, e->time_table, e->have_text ? e->text_term : nullptr, t_dev, t_scalar,
                               n_seq, B, d, e->n_time_rows, s, tables ? e->tmap_dev : nullptr, tables ? e->cursor_dev : nullptr));
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
        if (e->profile && mnk[0]) {
            e->ev_used += 2; e->prof_m = mnk[0]; e->prof_n = mnk[1]; e->prof_k = mnk[2];
            e->prof_kernel = unet_probe_route(e->unet);
        }
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

// ---- the two boundary GEMMs of the input-VJP (round 4: on the f16 pipe when the I/O projections are) ------------------------
// d tok[b*S+1+t][k] = scale * sum_c gout[b][c][t] W_out[c][k] for the sequences [slot0, slot0 + nslot); token-0 rows of dA
// must already be zero.  f16x3: the output gradient becomes scaled split frame rows (pose_rows_split), one H3_TOKENS GEMM
// writes the fp32 token rows the LayerNorm backward reads (rounds 1-3: an fp32-MFMA GEMM with a transposing A load, 63 us).
int vjp_output_projection(cmdi_engine* e, const float* gout, float* dA, int slot0, int nslot, const unsigned* gs, hipStream_t s) {
    const int T = e->T, S = T + 1, d = e->d, C = e->C;
    if (vjp_boundary_h3(e)) {
        _Float16* gs_rows = e->gS + (size_t)slot0 * T * 2 * e->Cpad;
        HIPCHK(launch_pose_rows_split(gout, gs_rows, nslot, C, T, e->Cpad, nullptr, s, gs));
        H3Params p{};
        p.A = gs_rows; p.W = e->w_outT_s; p.C = dA;
        p.M = nslot * T; p.N = d; p.K = e->Cpad; p.ldc = d;
        p.tok_T = T; p.tok_S = S;
        HIPCHK(launch_gemm_h3(H3_TOKENS, p, 0, s));
        return CMDI_OK;
    }
    GemmParams p = gp(gout, e->w_outT_pad, nullptr, dA, nslot * T, d, e->Cpad, 0, e->Cpad, d);
    p.T = T; p.S = S; p.Cf = C;
    p.gs_bits = gs;
    HIPCHK(launch_gemm(GK_OUTPROJ_BWD, p, 0, s));
    return CMDI_OK;
}
// gx[b][c][t] = (1 / scale) * sum_n dA[b*S+1+t][n] W_in[n][c].  f16x3: layer 0's last GEMM left dA as split rows (dOS) and the
// forward's output-projection kernel (H3_MOTION, weight rows as the A operand: stores run along T) takes W_in^T instead.
int vjp_input_projection(cmdi_engine* e, const float* dA, float* gx, int slot0, int nslot, const unsigned* gs, hipStream_t s) {
    const int T = e->T, S = T + 1, d = e->d, C = e->C;
    if (vjp_boundary_h3(e)) {
        H3Params p{};
        p.A = e->w_inT_s; p.W = e->dOS + (size_t)slot0 * S * 2 * d; p.C = gx;
        p.M = C; p.N = nslot * S; p.K = d; p.ldc = 0;
        p.tok_T = T; p.tok_S = S; p.gs_bits = gs;
        HIPCHK(launch_gemm_h3(H3_MOTION, p, 0, s));
        return CMDI_OK;
    }
    GemmParams p = gp(e->w_inT, dA, nullptr, gx, C, nslot * T, d, d, d, 0);
    p.T = T; p.S = S; p.Cf = C;
    p.gs_bits = gs;
    HIPCHK(launch_gemm(GK_OUTPROJ, p, 0, s));
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
        p.Wp = e->packed(W, M);
        p.M = M; p.N = N; p.K = K; p.ldc = N;
        return p;
    };
    for (int l = e->L - 1; l >= 0; --l) {
        const LayerW& w = e->layers[l];
        const LayerStash& st = e->stash[l];
        if (h3) {
            // same chain with the four dX GEMMs on the f16 matrix pipe (weights: split transposes).  Round 4: every gradient
            // between two kernels travels ONCE — the LayerNorm backward writes only split rows (the GEMM operand), the
            // residual epilogues add those same rows (Rs), and D = rowsum(dO * O) is taken from the split dO the attention
            // kernels multiply: three 26-MB fp32 tensors per layer are no longer written
            const bool ss = e->stash_split;   // the stash holds split rows (folded forward schedule) or fp32 rows
            const _Float16* pre2S = ss ? reinterpret_cast<const _Float16*>(st.pre2 + r0 * d) : nullptr;
            const _Float16* pre1S = ss ? reinterpret_cast<const _Float16*>(st.pre1 + r0 * d) : nullptr;
            const _Float16* attnS_ = ss ? reinterpret_cast<const _Float16*>(st.attn + r0 * d) : nullptr;
            // (stash_f32: fp32 copies of single tensors beside the folded schedule's split rows — which of them the backward
            //  takes is the engine's accuracy / traffic trade, DESIGN.md section 4 "guided-chain error")
            const float* pre2F = ss ? (st.pre2_f ? st.pre2_f + r0 * d : nullptr) : st.pre2 + r0 * d;
            const float* pre1F = ss ? (st.pre1_f ? st.pre1_f + r0 * d : nullptr) : st.pre1 + r0 * d;
            const float* attnF = ss ? (st.attn_f ? st.attn_f + r0 * d : nullptr) : st.attn + r0 * d;
            HIPCHK(launch_layernorm_bwd(pre2F, st.stats2 + r0 * 2, w.n2_g, dA, nullptr, dBS, M, d, s, pre2F ? nullptr : pre2S));
            {   // dffn = (dB · W2) * gelu'(aux)
                H3Params p = hp(dBS, w.l2_wTs, nullptr, dffnS, f, d);
                p.aux = st.aux + r0 * f;
                HIPCHK(launch_gemm_h3(H3_GELUGRAD_SPLIT, p, e->h3_tile_ffn1, s));
            }
            {   // dH = dffn · W1 + dB
                H3Params p = hp(dffnS, w.l1_wTs, dH, nullptr, d, f);
                p.Rs = dBS;
                HIPCHK(launch_gemm_h3(H3_RESID, p, e->h3_tile_ffn2, s));
            }
            HIPCHK(launch_layernorm_bwd(pre1F, st.stats1 + r0 * 2, w.n1_g, dH, nullptr, dBS, M, d, s, pre1F ? nullptr : pre1S));
            {   // d attn = dB · Wo -> dOS (split: MFMA operand of the attention backward and the dO of its D = rowsum(dO * O))
                H3Params p = hp(dBS, w.out_wTs, nullptr, dOS, d, d);
                HIPCHK(launch_gemm_h3(H3_PLAIN_SPLIT, p, e->h3_tile_proj, s));
            }
            HIPCHK(launch_attention_bwd_h3(st.qkvS + r0 * 6 * d, attnF, attnF ? nullptr : attnS_,
                                           st.row_stats + (size_t)seq0 * e->H * S * 2, dOS, dqkvS,
                                           e->drowdot + attention_bwd_scratch_floats(seq0, S, e->H), nseq, S, e->H, s));
            {   // dA = dqkv · Wqkv + dB   (layer 0 with the boundary GEMM on the f16 pipe: as split rows, its W operand)
                const bool to_split = l == 0 && vjp_boundary_h3(e);
                H3Params p = hp(dqkvS, w.in_wTs, to_split ? nullptr : dA, to_split ? dOS : nullptr, d, 3 * d);
                p.Rs = dBS;
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
    int rc = vjp_output_projection(e, gout, e->dA, 0, n_seq, h3 ? e->gs_bits : nullptr, s);
    if (rc != CMDI_OK) return rc;
    rc = for_groups(e, n_seq, s, [&](int seq0, int nseq, hipStream_t gs) {
        return run_layers_bwd(e, seq0, nseq, gs);
    });
    if (rc != CMDI_OK) return rc;
    // input projection: gx[b][c][t] = sum_n dA[b*S+1+t][n] W_in[n][c]
    return vjp_input_projection(e, e->dA, gx, 0, n_seq, h3 ? e->gs_bits : nullptr, s);
}

}  // namespace host
}  // namespace cmdi
