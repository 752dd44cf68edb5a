// C-ABI of libcondmdi_hip.so (include/condmdi.h), engine part: handle life cycle, weight packing, schedule, condition.
#include <utility>
#include "engine.hpp"

using namespace cmdi;
using namespace cmdi::host;

namespace {
thread_local std::string g_err;
}

namespace cmdi {
namespace host {
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
}  // namespace host
}  // namespace cmdi

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
        // precisions of this arch: f16x3 (default) and, round 5, bf16x6 — exact three-plane operands on every convolution, no
        // f16 range limit: what a checkpoint whose activations pass 65,504 falls back to (the reference U-Net is plain fp32
        // at any scale, model/mdm_unet.py:561-849).  The fp32-MFMA engine is not built for the U-Net.
        int uprec = desc->precision;
        if (uprec == CMDI_PREC_DEFAULT) {
            const char* v = std::getenv("CMDI_PRECISION");
            const std::string name = v ? v : "";
            if (name == "f32")
                return fail(CMDI_E_INVALID, "UNET engine: CMDI_PRECISION names a precision that is not built for this arch (f16x3, bf16x6)");
            uprec = name == "bf16x6" ? CMDI_PREC_BF16X6 : CMDI_PREC_F16X3;
        }
        if (uprec != CMDI_PREC_F16X3 && uprec != CMDI_PREC_BF16X6)
            return fail(CMDI_E_INVALID, "UNET engine: the precisions built are CMDI_PREC_F16X3 (default) and CMDI_PREC_BF16X6");
        cmdi_engine* e = new cmdi_engine();
        e->desc = *desc;
        e->d = desc->d_model; e->C = desc->n_feats; e->Tmax = desc->max_frames; e->Bmax = desc->max_batch;
        e->precision = uprec;
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
                           desc->text_cond != 0, desc->want_grad != 0, desc->unet_attention != 0, uprec == CMDI_PREC_BF16X6);
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
    e->h3w_min_m = env_int("CMDI_H3W_MIN_M", 8192);
    e->h3w = env_int("CMDI_H3W", 0);    // (round 6: the weight-stationary kernel wins no shape of C2 — opt-in; profiles/r06_h3w_stall_table.md)
    e->ln_fuse = env_probe("CMDI_LN_FUSE", 0) && desc->d_model == 512;
    e->io_h3 = e->precision == CMDI_PREC_F16X3 && !e->ln_fuse && env_probe("CMDI_IO_H3", 1);
    e->ln_fold = e->io_h3 && desc->d_model == 512 && env_int("CMDI_LN_FOLD", 1);
    e->ln_fold_keep = env_int("CMDI_LN_FOLD_KEEP", 1);
    e->qkv_head_major = env_int("CMDI_QKV_HEAD_MAJOR", 0);
    e->stash_f32 = env_int("CMDI_STASH_F32", kDefaultStashF32) & 7;
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
        if (e->h3w && d == 512) {     // fragment-ordered copies for the weight-stationary GEMM (K = 512, N % 128 == 0)
            auto pk = [&](const _Float16* ws, size_t n) -> int {
                if (!ws || n % 128 != 0) return CMDI_OK;
                _Float16* wp = nullptr;
                int rc = dalloc(e, reinterpret_cast<void**>(&wp), n * 2048);
                if (rc == CMDI_OK) e->h3w_packed[ws] = wp;
                return rc;
            };
            for (LayerW& w : e->layers) {
                int rc = pk(w.in_ws, 3 * d); if (rc != CMDI_OK) return rc;
                rc = pk(w.in_wsf, 3 * d); if (rc != CMDI_OK) return rc;
                rc = pk(w.out_ws, d); if (rc != CMDI_OK) return rc;
                rc = pk(w.l1_ws, f); if (rc != CMDI_OK) return rc;
                rc = pk(w.l1_wsf, f); if (rc != CMDI_OK) return rc;
            }
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
            if (e->h3w && d == 512) {     // the two dX GEMMs with K = 512: out_proj^T [d][d], linear2^T [f][d]
                for (LayerW& w : e->layers)
                    for (auto pr : {std::make_pair(w.out_wTs, (size_t)d), std::make_pair(w.l2_wTs, (size_t)f)}) {
                        if (pr.second % 128 != 0) continue;
                        _Float16* wp = nullptr;
                        int rc = dalloc(e, reinterpret_cast<void**>(&wp), pr.second * 2048);
                        if (rc != CMDI_OK) return rc;
                        e->h3w_packed[pr.first] = wp;
                    }
            }
            ALLOC(e->dBS, Mmax * d * 2); ALLOC(e->dffnS, Mmax * f * 2); ALLOC(e->dqkvS, Mmax * 3 * d * 2);
            ALLOC(e->dOS, Mmax * d * 2);
            ALLOC(e->w_inT_s, (size_t)C * d * 2); ALLOC(e->w_outT_s, (size_t)d * e->Cpad * 2);
            ALLOC(e->gS, (size_t)nseq * e->Tmax * e->Cpad * 2);
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
            if (e->precision == CMDI_PREC_F16X3 && e->ln_fold && e->ln_fold_keep) {
                if (e->stash_f32 & 1) ALLOC(st.attn_f, Mmax * d);
                if (e->stash_f32 & 2) ALLOC(st.pre1_f, Mmax * d);
                if (e->stash_f32 & 4) ALLOC(st.pre2_f, Mmax * d);
            }
        }
        ALLOC(e->dA, Mmax * d); ALLOC(e->dB, Mmax * d); ALLOC(e->dH, Mmax * d);
        ALLOC(e->dqkv, Mmax * 3 * d); ALLOC(e->dffn, Mmax * f);
        {   // f32 backward: D per (head, query); f16x3 backward: per-tile row statistics (attention_bwd_h3.hip)
            const size_t plain = (size_t)nseq * e->H * Smax, tiles = attention_bwd_scratch_floats((int)nseq, (int)Smax, e->H);
            ALLOC(e->drowdot, plain > tiles ? plain : tiles);
        }
        ALLOC(e->gout, nseq * C * e->Tmax); ALLOC(e->gx, nseq * C * e->Tmax);
    }
    return CMDI_OK;
}

int cmdi_profile_select(cmdi_handle e, int32_t which) {
    if (!e || which < 0 || which > 1) return fail(CMDI_E_INVALID, "cmdi_profile_select: 0 = in_proj GEMM, 1 = attention kernel");
    e->prof_which = which;
    e->ev_used = 0;
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

const char* cmdi_profile_kernel(cmdi_handle e) { return e ? e->prof_kernel : ""; }

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
            if (e->desc.want_grad) {   // boundary GEMMs of the input-VJP on the f16 pipe (round 4)
                HIPCHK(launch_split_f16(e->w_inT, e->w_inT_s, C, d, d, e->range_flag, s));
                HIPCHK(launch_split_f16(e->w_outT_pad, e->w_outT_s, d, e->Cpad, e->Cpad, e->range_flag, s));
            }
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
    for (const auto& kv : e->h3w_packed) {   // every split weight the weight-stationary GEMM can take, in fragment order
        int n = 0;
        for (const LayerW& w : e->layers) {
            if (kv.first == w.in_ws || kv.first == w.in_wsf) n = 3 * d;
            else if (kv.first == w.out_ws || kv.first == w.out_wTs) n = d;
            else if (kv.first == w.l1_ws || kv.first == w.l1_wsf || kv.first == w.l2_wTs) n = f;
            if (n) break;
        }
        if (!n) return fail(CMDI_E_INVALID, "h3w_packed holds an unknown weight");
        HIPCHK(launch_pack_w_h3w(static_cast<const _Float16*>(kv.first), kv.second, n, s));
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

}  // extern "C"
