// CLIP ViT-B/32 TEXT tower on the device: the step in front of the sampling loop (reference model/mdm.py:173-186,211-237:
// clip.load('ViT-B/32') -> clip_model.encode_text(tokens).float()).  The architecture is openai/CLIP's (clip/model.py,
// an un-pinned pip dependency of the reference, README.md:54): token embedding + learned positions -> 12 pre-LN residual
// attention blocks (width 512, 8 heads x 64, CAUSAL mask, MLP 512 -> 2048 -> QuickGELU -> 512) -> ln_final -> the row of
// the end-of-text token (argmax of the ids) -> @ text_projection.  All arithmetic fp32 (the reference runs the tower in
// fp16): the four projections of a block on the fp32 MFMA GEMM family (gemm_f32.hpp), LayerNorm with layernorm_kernel,
// and the two kernels below for what the MDM engine does not have (64-wide causal heads, QuickGELU).
// 6 GFLOP per prompt, once per sampling CALL — 0.04 % of a 1000-step chain; built for completeness, not for speed.
#include <string>
#include <vector>

#include "common.hpp"
#include "kernels.hpp"

namespace cmdi {

// x[b*S + s][:] = token_embedding[tokens[b][s]] + positional_embedding[s]
__global__ void clip_embed_kernel(const int32_t* __restrict__ tokens, const float* __restrict__ tok_emb,
                                  const float* __restrict__ pos, float* __restrict__ x, int S, int d, int vocab,
                                  int* __restrict__ status) {
    const int row = blockIdx.x, s = row % S;
    int id = tokens[row];
    if (id < 0 || id >= vocab) {             // torch's Embedding raises; report and clamp
        if (threadIdx.x == 0 && status) atomicOr(status, 2);
        id = id < 0 ? 0 : vocab - 1;
    }
    for (int n = threadIdx.x * 4; n < d; n += blockDim.x * 4) {
        const float4 e = *reinterpret_cast<const float4*>(tok_emb + (size_t)id * d + n);
        const float4 p = *reinterpret_cast<const float4*>(pos + (size_t)s * d + n);
        *reinterpret_cast<float4*>(x + (size_t)row * d + n) = make_float4(e.x + p.x, e.y + p.y, e.z + p.z, e.w + p.w);
    }
}

// Causal self-attention of one (sequence, head): qkv [B*S][3*H*64] -> out [B*S][H*64].  S <= 96, head width 64.
// K and V of the head sit in LDS; a wave owns query rows wave, wave + 4, ...: lanes run over keys for the scores
// (softmax by wave reductions), then over the 64 output features for P·V.
__global__ __launch_bounds__(256) void clip_attention_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                             int S, int H) {
    constexpr int DH = 64, SMAX = 96;   // 52 KiB of static LDS
    __shared__ float ks[SMAX][DH + 1], vs[SMAX][DH + 1], ps[4][SMAX], qs[4][DH];
    const int b = blockIdx.x / H, h = blockIdx.x % H, d = H * DH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* base = qkv + (size_t)b * S * 3 * d + h * DH;
    for (int i = threadIdx.x; i < S * DH; i += 256) {
        const int s = i / DH, c = i % DH;
        ks[s][c] = base[(size_t)s * 3 * d + d + c];
        vs[s][c] = base[(size_t)s * 3 * d + 2 * d + c];
    }
    __syncthreads();
    for (int i = wave; i < S; i += 4) {
        qs[wave][lane] = base[(size_t)i * 3 * d + lane] * 0.125f;          // q / sqrt(64), as torch MultiheadAttention
        __builtin_amdgcn_wave_barrier();
        float sc[2], mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int j = lane + 64 * r;
            float a = -INFINITY;
            if (j <= i) {                                                  // causal: keys 0..i
                a = 0.f;
#pragma unroll 16
                for (int c = 0; c < DH; ++c) a += qs[wave][c] * ks[j][c];
            }
            sc[r] = a;
            mx = fmaxf(mx, a);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int j = lane + 64 * r;
            const float e = j <= i ? expf(sc[r] - mx) : 0.f;
            if (j < SMAX) ps[wave][j] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        __builtin_amdgcn_wave_barrier();
        float o = 0.f;
        for (int j = 0; j <= i; ++j) o += ps[wave][j] * vs[j][lane];
        out[((size_t)b * S + i) * d + h * DH + lane] = o / sum;
        __builtin_amdgcn_wave_barrier();
    }
}

// QuickGELU in place: x * sigmoid(1.702 x) (clip/model.py)
__global__ void quick_gelu_kernel(float* __restrict__ x, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<float4*>(x)[i];
        v.x = v.x / (1.0f + expf(-1.702f * v.x)); v.y = v.y / (1.0f + expf(-1.702f * v.y));
        v.z = v.z / (1.0f + expf(-1.702f * v.z)); v.w = v.w / (1.0f + expf(-1.702f * v.w));
        reinterpret_cast<float4*>(x)[i] = v;
    }
}

// rows[b] = x[b*S + argmax_s tokens[b][s]]   (the end-of-text token has the largest id; first maximum, as torch.argmax)
__global__ void clip_eot_gather_kernel(const int32_t* __restrict__ tokens, const float* __restrict__ x,
                                       float* __restrict__ rows, int S, int d) {
    const int b = blockIdx.x;
    int best = 0, bid = tokens[(size_t)b * S];
    for (int s = 1; s < S; ++s) {
        const int id = tokens[(size_t)b * S + s];
        if (id > bid) { bid = id; best = s; }
    }
    for (int n = threadIdx.x; n < d; n += blockDim.x) rows[(size_t)b * d + n] = x[((size_t)b * S + best) * d + n];
}

struct ClipBlock {
    float *ln1_g, *ln1_b, *in_w, *in_b, *out_w, *out_b, *ln2_g, *ln2_b, *fc_w, *fc_b, *proj_w, *proj_b;
};

struct ClipText {
    int vocab = 0, width = 0, heads = 0, layers = 0, ctx = 0, embed = 0, max_batch = 0;
    std::vector<void*> allocs;
    std::string err;
    float *tok_emb = nullptr, *pos = nullptr, *lnf_g = nullptr, *lnf_b = nullptr, *text_proj = nullptr, *text_projT = nullptr;
    std::vector<ClipBlock> blocks;
    float *x = nullptr, *x2 = nullptr, *h = nullptr, *qkv = nullptr, *attn = nullptr, *ff = nullptr, *rows = nullptr;
    int* status = nullptr;
    bool transposed = false;
};

static bool clip_alloc(ClipText* c, float** p, size_t n) {
    if (hipMalloc(reinterpret_cast<void**>(p), (n ? n : 4) * sizeof(float)) != hipSuccess) { c->err = "hipMalloc failed"; return false; }
    c->allocs.push_back(*p);
    return true;
}

ClipText* clip_new(int vocab, int width, int heads, int layers, int ctx, int embed, int max_batch) {
    ClipText* c = new ClipText();
    c->vocab = vocab; c->width = width; c->heads = heads; c->layers = layers; c->ctx = ctx; c->embed = embed; c->max_batch = max_batch;
    if (width != heads * 64 || width % 256 != 0 || width > 1024 || ctx < 1 || ctx > 96 || embed % 32 != 0 || vocab < 1 ||
        layers < 1 || max_batch < 1) {
        c->err = "CLIP text tower: width must be heads * 64 and a multiple of 256 (<= 1024), context <= 96, embed_dim % 32 == 0";
        return c;
    }
    const size_t d = width, M = (size_t)max_batch * ctx;
    bool ok = clip_alloc(c, &c->tok_emb, (size_t)vocab * d) && clip_alloc(c, &c->pos, (size_t)ctx * d) &&
              clip_alloc(c, &c->lnf_g, d) && clip_alloc(c, &c->lnf_b, d) && clip_alloc(c, &c->text_proj, d * embed) &&
              clip_alloc(c, &c->text_projT, d * embed);
    c->blocks.resize(layers);
    for (ClipBlock& b : c->blocks)
        ok = ok && clip_alloc(c, &b.ln1_g, d) && clip_alloc(c, &b.ln1_b, d) && clip_alloc(c, &b.in_w, 3 * d * d) &&
             clip_alloc(c, &b.in_b, 3 * d) && clip_alloc(c, &b.out_w, d * d) && clip_alloc(c, &b.out_b, d) &&
             clip_alloc(c, &b.ln2_g, d) && clip_alloc(c, &b.ln2_b, d) && clip_alloc(c, &b.fc_w, 4 * d * d) &&
             clip_alloc(c, &b.fc_b, 4 * d) && clip_alloc(c, &b.proj_w, 4 * d * d) && clip_alloc(c, &b.proj_b, d);
    ok = ok && clip_alloc(c, &c->x, M * d) && clip_alloc(c, &c->x2, M * d) && clip_alloc(c, &c->h, M * d) &&
         clip_alloc(c, &c->qkv, M * 3 * d) && clip_alloc(c, &c->attn, M * d) && clip_alloc(c, &c->ff, M * 4 * d) &&
         clip_alloc(c, &c->rows, (size_t)max_batch * d);
    float* st = nullptr;
    ok = ok && clip_alloc(c, &st, 4);
    c->status = reinterpret_cast<int*>(st);
    if (ok) (void)hipMemset(c->status, 0, sizeof(int));
    return c;
}

const char* clip_error(const ClipText* c) { return c->err.c_str(); }

void clip_free(ClipText* c) {
    if (!c) return;
    for (void* p : c->allocs) (void)hipFree(p);
    delete c;
}

// openai/CLIP state-dict names of the text tower (clip/model.py: CLIP.__init__ / Transformer / ResidualAttentionBlock)
int clip_load_weight(ClipText* c, const char* name, const float* src, int64_t numel, hipStream_t s) {
    const size_t d = c->width;
    float* dst = nullptr;
    int64_t want = -1;
    const std::string n(name);
    int l = -1;
    char rest[96] = {0};
    if (n == "token_embedding.weight") { dst = c->tok_emb; want = (int64_t)c->vocab * d; }
    else if (n == "positional_embedding") { dst = c->pos; want = (int64_t)c->ctx * d; }
    else if (n == "ln_final.weight") { dst = c->lnf_g; want = d; }
    else if (n == "ln_final.bias") { dst = c->lnf_b; want = d; }
    else if (n == "text_projection") { dst = c->text_proj; want = (int64_t)d * c->embed; c->transposed = false; }
    else if (std::sscanf(name, "transformer.resblocks.%d.%95s", &l, rest) == 2 && l >= 0 && l < c->layers) {
        ClipBlock& b = c->blocks[l];
        const std::string r(rest);
        if (r == "ln_1.weight") { dst = b.ln1_g; want = d; }
        else if (r == "ln_1.bias") { dst = b.ln1_b; want = d; }
        else if (r == "attn.in_proj_weight") { dst = b.in_w; want = 3 * d * d; }
        else if (r == "attn.in_proj_bias") { dst = b.in_b; want = 3 * d; }
        else if (r == "attn.out_proj.weight") { dst = b.out_w; want = d * d; }
        else if (r == "attn.out_proj.bias") { dst = b.out_b; want = d; }
        else if (r == "ln_2.weight") { dst = b.ln2_g; want = d; }
        else if (r == "ln_2.bias") { dst = b.ln2_b; want = d; }
        else if (r == "mlp.c_fc.weight") { dst = b.fc_w; want = 4 * d * d; }
        else if (r == "mlp.c_fc.bias") { dst = b.fc_b; want = 4 * d; }
        else if (r == "mlp.c_proj.weight") { dst = b.proj_w; want = 4 * d * d; }
        else if (r == "mlp.c_proj.bias") { dst = b.proj_b; want = d; }
    }
    if (!dst) { c->err = std::string("unknown CLIP text weight: ") + name; return -5; }
    if (numel != want) { c->err = std::string(name) + ": expected " + std::to_string(want) + " elements"; return -1; }
    if (hipMemcpyAsync(dst, src, (size_t)numel * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) { c->err = "copy failed"; return -3; }
    return 0;
}

static GemmParams cgp(const float* A, const float* W, const float* bias, float* C, int M, int N, int K) {
    GemmParams p{};
    p.A = A; p.W = W; p.bias = bias; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldw = K; p.ldc = N; p.out_scale = 1.0f;
    return p;
}

#define CLIPCHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { c->err = std::string(#expr) + ": " + hipGetErrorString(_e); return -3; } } while (0)

int clip_encode_text(ClipText* c, const int32_t* tokens, int B, float* out, hipStream_t s) {
    if (B < 1 || B > c->max_batch) { c->err = "batch exceeds max_batch"; return -1; }
    const int S = c->ctx, d = c->width, M = B * S;
    if (!c->transposed) {   // x @ text_projection = NT GEMM against text_projection^T
        CLIPCHK(launch_transpose_pad(c->text_projT, c->text_proj, d, c->embed, d, s));
        c->transposed = true;
    }
    hipLaunchKernelGGL(clip_embed_kernel, dim3(M), dim3(128), 0, s, tokens, c->tok_emb, c->pos, c->x, S, d, c->vocab, c->status);
    float *x = c->x, *x2 = c->x2;
    for (const ClipBlock& b : c->blocks) {
        CLIPCHK(launch_layernorm(x, b.ln1_g, b.ln1_b, c->h, nullptr, nullptr, nullptr, M, d, s));
        CLIPCHK(launch_gemm(GK_PLAIN, cgp(c->h, b.in_w, b.in_b, c->qkv, M, 3 * d, d), 0, s));
        hipLaunchKernelGGL(clip_attention_kernel, dim3(B * c->heads), dim3(256), 0, s, c->qkv, c->attn, S, c->heads);
        {   // x = x + out_proj(attn)
            GemmParams p = cgp(c->attn, b.out_w, b.out_b, x2, M, d, d);
            p.R = x;
            CLIPCHK(launch_gemm(GK_RESID, p, 0, s));
        }
        CLIPCHK(launch_layernorm(x2, b.ln2_g, b.ln2_b, c->h, nullptr, nullptr, nullptr, M, d, s));
        CLIPCHK(launch_gemm(GK_PLAIN, cgp(c->h, b.fc_w, b.fc_b, c->ff, M, 4 * d, d), 0, s));
        hipLaunchKernelGGL(quick_gelu_kernel, dim3(1024), dim3(256), 0, s, c->ff, (int64_t)M * d);   // M * 4d / 4 float4s
        {   // x = x + c_proj(gelu(c_fc(ln_2(x))))
            GemmParams p = cgp(c->ff, b.proj_w, b.proj_b, x, M, d, 4 * d);
            p.R = x2;
            CLIPCHK(launch_gemm(GK_RESID, p, 0, s));
        }
    }
    CLIPCHK(launch_layernorm(x, c->lnf_g, c->lnf_b, c->h, nullptr, nullptr, nullptr, M, d, s));
    hipLaunchKernelGGL(clip_eot_gather_kernel, dim3(B), dim3(128), 0, s, tokens, c->h, c->rows, S, d);
    CLIPCHK(launch_gemm(GK_PLAIN, cgp(c->rows, c->text_projT, nullptr, out, B, c->embed, d), 0, s));
    CLIPCHK(hipGetLastError());
    return 0;
}

int clip_status(ClipText* c, int* flag, hipStream_t s) {
    if (hipMemcpyAsync(flag, c->status, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return -3;
    if (*flag) (void)hipMemsetAsync(c->status, 0, sizeof(int), s);
    return 0;
}

}  // namespace cmdi
