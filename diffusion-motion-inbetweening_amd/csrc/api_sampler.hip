// C-ABI of libcondmdi_hip.so, sampler part: model entry points, the sampler update, and cmdi_sample_loop (independent
// pipelines, fork/join groups, hipGraph replay).
#include "engine.hpp"

using namespace cmdi;
using namespace cmdi::host;

namespace cmdi {
namespace host {

int check_ready(cmdi_engine* e, bool need_schedule, bool need_model) {
    if (!e) return fail(CMDI_E_INVALID, "null handle");
    if (need_model && e->L == 0 && !e->unet) return fail(CMDI_E_STATE, "sampler-only engine has no denoiser");
    if (!e->finalized) return fail(CMDI_E_STATE, "weights not finalized (cmdi_finalize_weights)");
    if (!e->have_cond) return fail(CMDI_E_STATE, "condition not set (cmdi_set_condition)");
    if (need_schedule && !e->have_schedule) return fail(CMDI_E_STATE, "schedule not set (cmdi_set_schedule)");
    return CMDI_OK;
}

}  // namespace host
}  // namespace cmdi

namespace {

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

// cursor != null (graph replay): the timestep comes from tmap_dev[*cursor] instead of the scalar
static int part_forward(cmdi_engine* e, const Part& pt, const float* x_part, int64_t t_scalar, bool keep,
                        const int* cursor = nullptr) {
    const int T = e->T, S = T + 1, d = e->d, C = e->C;
    hipStream_t s = pt.s;
    float* tok = e->tokA + (size_t)pt.slot0 * S * d;
    _Float16* tok_s = e->io_h3 ? e->tokS + (size_t)pt.slot0 * S * 2 * d : nullptr;
    HIPCHK(launch_token0(tok, e->time_table, e->have_text ? e->text_term_p + (size_t)pt.slot0 * d : nullptr,
                         e->pe, nullptr, t_scalar, pt.nslot, pt.nb, S, d, e->n_time_rows, s, cursor ? e->tmap_dev : nullptr,
                         cursor, tok_s, e->range_flag));
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
    int rc = vjp_output_projection(e, gout, dA, pt.slot0, pt.nslot, h3 ? gs : nullptr, s);
    if (rc != CMDI_OK) return rc;
    rc = run_layers_bwd(e, pt.slot0, pt.nslot, s);
    if (rc != CMDI_OK) return rc;
    return vjp_input_projection(e, dA, gx, pt.slot0, pt.nslot, h3 ? gs : nullptr, s);
}

// cursor != null (graph replay): per-step scalars come from the device tables at *cursor, which the step decrements at its end
static int part_step(cmdi_engine* e, const Part& pt, int32_t sampler, int32_t step, float eta, float* d_x,
                     const float* d_noise, uint64_t seed, int64_t first_sample, int* cursor = nullptr) {
    const int64_t per = (int64_t)e->C * e->T;
    const bool recon = recon_at(e, step);
    const bool impute = impute_at(e, step, recon);
    float* x = d_x + (size_t)pt.b0 * per;
    int rc = part_forward(e, pt, x, e->tmap[step], recon, cursor);
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
    HIPCHK(launch_sampler_step(io, k, pt.nb, per, seed, first_sample + pt.b0, step, pt.s, cursor ? e->coef_dev : nullptr, cursor));
    if (cursor) HIPCHK(launch_cursor_add(cursor, -1, pt.s));
    return CMDI_OK;
}

static int check_step(cmdi_engine* e, int32_t step);
static int upload_step_tables(cmdi_engine* e, int32_t sampler, int32_t first_step, int32_t last_step, float eta, int n_cursors,
                              hipStream_t s);

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
    int rc = CMDI_OK;
    // hipGraph replay of the pipelined schedule (round 4, VERDICT r3 task 6): every part's step is captured ON ITS OWN STREAM
    // (per-step scalars from device tables at the part's cursor) and replayed into it — the parts stay independent chains,
    // nothing is serialised by the capture.  Same kernels, same tables: bitwise the eager chain.
    const bool graph = e->use_graph && !d_noise_stream && G > 1;
    if (graph) {
        rc = upload_step_tables(e, sampler, first_step, last_step, eta, G, s);   // (before the fork: the parts wait for s)
        if (rc != CMDI_OK) return rc;
        if (e->graph_stream != s || e->graph_seed != seed || e->graph_first != first_sample || e->graph_x != d_x ||
            e->part_graph_parts != G) {
            drop_graphs(e);   // captured pointers / scalars changed
            e->graph_stream = s; e->graph_seed = seed; e->graph_first = first_sample; e->graph_x = d_x;
            e->part_graph_parts = G;
        }
        e->part_graph.resize((size_t)2 * G, nullptr);
        e->part_warm.resize((size_t)2 * G, 0);
        HIPCHK(hipEventRecord(e->gevents[0], s));     // the tables are uploaded on s
        for (const Part& pt : parts) HIPCHK(hipStreamWaitEvent(pt.s, e->gevents[0], 0));
    }
    for (int step = first_step, i = 0; step >= last_step && rc == CMDI_OK; --step, ++i) {
        const float* nz = d_noise_stream ? d_noise_stream + (size_t)i * n : nullptr;
        for (const Part& pt : parts) {
            if (!graph) {
                rc = part_step(e, pt, sampler, step, eta, d_x, nz, seed, first_sample);
                if (rc != CMDI_OK) break;
                continue;
            }
            const size_t slot = (size_t)2 * pt.idx + (recon_at(e, step) ? 1 : 0);
            int* cursor = e->cursor_dev + 1 + pt.idx;
            if (e->part_graph[slot]) {
                if (hipGraphLaunch(e->part_graph[slot], pt.s) != hipSuccess) { rc = fail(CMDI_E_HIP, "hipGraphLaunch (part)"); break; }
                continue;
            }
            if (!e->part_warm[slot]) {       // first step of a kind: eager (one-time function attributes)
                rc = part_step(e, pt, sampler, step, eta, d_x, nullptr, seed, first_sample, cursor);
                if (rc != CMDI_OK) break;
                e->part_warm[slot] = 1;
                continue;
            }
            hipGraph_t g = nullptr;
            if (hipStreamBeginCapture(pt.s, hipStreamCaptureModeThreadLocal) != hipSuccess) { rc = fail(CMDI_E_HIP, "hipStreamBeginCapture (part)"); break; }
            rc = part_step(e, pt, sampler, step, eta, d_x, nullptr, seed, first_sample, cursor);
            const hipError_t ce = hipStreamEndCapture(pt.s, &g);
            if (rc != CMDI_OK) { if (g) (void)hipGraphDestroy(g); break; }
            if (ce != hipSuccess) { rc = fail(CMDI_E_HIP, std::string("hipStreamEndCapture (part): ") + hipGetErrorString(ce)); break; }
            const hipError_t ie = hipGraphInstantiate(&e->part_graph[slot], g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (ie != hipSuccess) { rc = fail(CMDI_E_HIP, std::string("hipGraphInstantiate (part): ") + hipGetErrorString(ie)); break; }
            if (hipGraphLaunch(e->part_graph[slot], pt.s) != hipSuccess) { rc = fail(CMDI_E_HIP, "hipGraphLaunch (part)"); break; }
        }
    }
    // join — ALSO on the error path: the part streams may still be writing d_x, out_raw and the stash; the caller's
    // stream (and whatever the caller does next on it, engine teardown by the Python retry path included) must be ordered
    // after them before the error is returned
    if (G > 1) {
        for (const Part& pt : parts) {
            const hipError_t e1 = hipEventRecord(e->gevents[1 + pt.idx], pt.s);
            const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(s, e->gevents[1 + pt.idx], 0) : e1;
            if (e2 != hipSuccess) {
                (void)hipStreamSynchronize(pt.s);      // last resort: order by blocking
                if (rc == CMDI_OK) rc = fail(CMDI_E_HIP, std::string("pipeline join: ") + hipGetErrorString(e2));
            }
        }
    }
    e->stash_valid = false;   // the stash holds slot-ordered rows of the last step: not a cmdi_mdm_forward stash
    return rc;
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

// per-chain device tables of the graph replays: StepCoef and timestep per respaced step, and 1 + n_cursors cursors (slot 0:
// the single-stream chain; 1 + g: pipeline part g), all set to first_step
static int upload_step_tables(cmdi_engine* e, int32_t sampler, int32_t first_step, int32_t last_step, float eta, int n_cursors,
                              hipStream_t s) {
    const int n = e->n_steps;
    if (e->table_cap < n) {
        int rc = falloc(e, &e->coef_dev, (size_t)n);
        if (rc != CMDI_OK) return rc;
        rc = falloc(e, &e->tmap_dev, (size_t)n);
        if (rc != CMDI_OK) return rc;
        e->table_cap = n;
    }
    if (e->cursor_cap < 1 + n_cursors) {
        int rc = falloc(e, &e->cursor_dev, (size_t)16);     // (n_parts() <= 15)
        if (rc != CMDI_OK) return rc;
        e->cursor_cap = 16;
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
    int first[16];
    for (int i = 0; i < 16; ++i) first[i] = first_step;
    HIPCHK(hipMemcpyAsync(e->cursor_dev, first, sizeof(first), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));   // the host buffers above go out of scope
    return CMDI_OK;
}

static int sample_loop_graph_on(cmdi_engine* e, int32_t sampler, int32_t first_step, int32_t last_step,
                                float eta, float* d_x, uint64_t seed, int64_t first_sample, hipStream_t s) {
    {
        int rc = upload_step_tables(e, sampler, first_step, last_step, eta, 0, s);
        if (rc != CMDI_OK) return rc;
    }
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
    // graph replay: of the pipelined schedule where the batch is cut into parts (each part's steps on its own stream),
    // else of the single-stream step (round 6: MDM_UNET too — its embedding kernel reads the timestep from the chain's device
    // table like token0_kernel; ~300 launches per evaluation, ~900 per guided step)
    if (e->use_graph && !d_noise_stream && !e->profile && (e->unet || !(e->pipelines && n_parts(e) > 1)))
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
    return (e->pipelines && !e->unet && e->L > 0) ? n_parts(e) : 1;
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

}  // extern "C"
