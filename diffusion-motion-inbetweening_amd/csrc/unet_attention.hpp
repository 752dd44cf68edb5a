// LinearAttention blocks of the temporal U-Net (attention=True): Residual(PreNorm(LayerNorm over channels,
// LinearAttention(dim, heads=4, dim_head=32))), reference model/mdm_unet.py:102-156.
//
//   xn  = (x - mean_c) / sqrt(var_c + 1e-5) * g + b            per frame, over the C channels          :111-121
//   qkv = to_qkv(xn)   (1x1 convolution, no bias)  -> q | k | v, each [4 heads x 32]                   :141-148
//   q *= 32^-0.5 ;  k = softmax over the FRAMES (dim=-1)                                               :149-151
//   context[d][e] = sum_n k[d][n] v[e][n] ;  out[e][n] = sum_d context[d][e] q[d][n]                   :152-155
//   y = to_out(out) + x                                                                                 :108,156
//
// The two 1x1 convolutions are GEMMs of gemm_h3 over the frame rows (unet.hip); this header holds what sits between
// them — the channel LayerNorm and the per-(sequence, head) core, forward and input-VJP — as fp32 VALU kernels: the
// core is 3 x [224 x 32] operands per block, far too small for the matrix pipe to matter.
// Rows are the U-Net's framed token rows: sequence-major, Tp rows per sequence, frames at [h, h + Tv).
#pragma once
#include "common.hpp"
#include "gemm_h3.hpp"

namespace cmdi {

constexpr int LA_HEADS = 4, LA_DH = 32, LA_HID = LA_HEADS * LA_DH;   // to_qkv: C -> 3 * 128, to_out: 128 -> C
constexpr int LA_MAXF = 224;                                         // frames per sequence at level 0

// ---- channel LayerNorm, one wave per frame row (C % 256 == 0, C <= 1024) ----------------------------------------------
__global__ __launch_bounds__(256) void chan_ln_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                      const float* __restrict__ b, _Float16* __restrict__ ys,
                                                      float* __restrict__ stats, int* __restrict__ range_flag, int C, int Tp,
                                                      int h, int Tv) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= Tv) return;
    const int lane = threadIdx.x & 63, nj = C >> 8;
    const size_t row = (size_t)blockIdx.y * Tp + h + t;
    const float* xr = x + row * C;
    float4 v[4];
    float sm = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < nj) {
            v[j] = *reinterpret_cast<const float4*>(xr + j * 256 + lane * 4);
            sm += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
    const float mean = wave_sum(sm) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < nj) {
            const float a = v[j].x - mean, bq = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
            q += (a * a + bq * bq) + (c * c + d * d);
        }
    const float var = wave_sum(q) / (float)C;          // torch.var(unbiased=False)
    const float sd = sqrtf(var + 1e-5f);
    if (lane == 0 && stats) { stats[2 * row] = mean; stats[2 * row + 1] = 1.0f / sd; }
    bool overflow = false;
    _Float16* yr = ys + row * (2 * (size_t)C);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < nj) {
            const int c0 = j * 256 + lane * 4;
            const float4 g4 = *reinterpret_cast<const float4*>(g + c0), b4 = *reinterpret_cast<const float4*>(b + c0);
            const float y[4] = {(v[j].x - mean) / sd * g4.x + b4.x, (v[j].y - mean) / sd * g4.y + b4.y,
                                (v[j].z - mean) / sd * g4.z + b4.z, (v[j].w - mean) / sd * g4.w + b4.w};
            h4 oh, ol;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                _Float16 a, l;
                split_f16(y[e], a, l);
                oh[e] = a; ol[e] = l;
                overflow |= !(fabsf(y[e]) < 65504.0f);
            }
            *reinterpret_cast<h4*>(yr + split_pos(c0)) = oh;
            *reinterpret_cast<h4*>(yr + split_pos(c0) + 32) = ol;
        }
    if (overflow && range_flag) atomicOr(range_flag, 1);
}

// d x (+= into dy, in place): LayerNorm backward of one frame row.  dyn = d xn (fp32 rows of C), x / stats from the
// forward pass;  dx = rstd (g dyn - mean(g dyn) - xhat mean(g dyn xhat)),  xhat = (x - mean) rstd.
__global__ __launch_bounds__(256) void chan_ln_bwd_kernel(const float* __restrict__ dyn, const float* __restrict__ x,
                                                          const float* __restrict__ stats, const float* __restrict__ g,
                                                          float* __restrict__ dy, int ld_dy, int C, int Tp, int h, int Tv) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= Tv) return;
    const int lane = threadIdx.x & 63, nj = C >> 8;
    const size_t row = (size_t)blockIdx.y * Tp + h + t;
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    float4 gd[4], xh[4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < nj) {
            const int c0 = j * 256 + lane * 4;
            const float4 d4 = *reinterpret_cast<const float4*>(dyn + row * C + c0);
            const float4 x4 = *reinterpret_cast<const float4*>(x + row * C + c0);
            const float4 g4 = *reinterpret_cast<const float4*>(g + c0);
            gd[j] = make_float4(d4.x * g4.x, d4.y * g4.y, d4.z * g4.z, d4.w * g4.w);
            xh[j] = make_float4((x4.x - mean) * rstd, (x4.y - mean) * rstd, (x4.z - mean) * rstd, (x4.w - mean) * rstd);
            s1 += (gd[j].x + gd[j].y) + (gd[j].z + gd[j].w);
            s2 += (gd[j].x * xh[j].x + gd[j].y * xh[j].y) + (gd[j].z * xh[j].z + gd[j].w * xh[j].w);
        }
    const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < nj) {
            float* o = dy + row * ld_dy + j * 256 + lane * 4;
            float4 a = *reinterpret_cast<float4*>(o);
            a.x += rstd * (gd[j].x - m1 - xh[j].x * m2);
            a.y += rstd * (gd[j].y - m1 - xh[j].y * m2);
            a.z += rstd * (gd[j].z - m1 - xh[j].z * m2);
            a.w += rstd * (gd[j].w - m1 - xh[j].w * m2);
            *reinterpret_cast<float4*>(o) = a;
        }
}

// ---- the core, one block per (head, sequence) ---------------------------------------------------------------------------
// Shared helper: P[n][d] = softmax over the Tv frames of k[n][d], in place in ks ([LA_MAXF][33] floats, padded rows).
// 256 threads = 8 frame groups x 32 columns; red = [2][8][32] floats of scratch.
__device__ __forceinline__ void la_softmax_frames(float* ks, float* red, int Tv) {
    const int d = threadIdx.x & 31, part = threadIdx.x >> 5;
    float mx = -INFINITY;
    for (int n = part; n < Tv; n += 8) mx = fmaxf(mx, ks[n * 33 + d]);
    red[part * 32 + d] = mx;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) mx = fmaxf(mx, red[q * 32 + d]);
    float sm = 0.f;
    for (int n = part; n < Tv; n += 8) {
        const float e = expf(ks[n * 33 + d] - mx);
        ks[n * 33 + d] = e;
        sm += e;
    }
    red[256 + part * 32 + d] = sm;
    __syncthreads();
    sm = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) sm += red[256 + q * 32 + d];
    for (int n = part; n < Tv; n += 8) ks[n * 33 + d] = ks[n * 33 + d] / sm;
    __syncthreads();
}

// qkv fp32 rows [.][3 * 128] -> attention output as split rows [.][2 * 128] (the A operand of to_out)
__global__ __launch_bounds__(256) void linattn_core_kernel(const float* __restrict__ qkv, _Float16* __restrict__ outs,
                                                           int* __restrict__ range_flag, int Tp, int h, int Tv) {
    __shared__ float ks[LA_MAXF * 33];      // k, then P, then q * scale
    __shared__ float vs[LA_MAXF * 32];
    __shared__ float ctx[32 * 33];
    __shared__ float red[512];
    const int head = blockIdx.x, tid = threadIdx.x;
    const size_t row0 = (size_t)blockIdx.y * Tp + h;
    const float* base = qkv + row0 * (3 * LA_HID) + head * LA_DH;
    for (int i = tid; i < Tv * 32; i += 256) {
        const int n = i >> 5, d = i & 31;
        ks[n * 33 + d] = base[(size_t)n * (3 * LA_HID) + LA_HID + d];
        vs[n * 32 + d] = base[(size_t)n * (3 * LA_HID) + 2 * LA_HID + d];
    }
    __syncthreads();
    la_softmax_frames(ks, red, Tv);
    {   // context[d][e] = sum_n P[n][d] v[n][e]
        const int d = tid >> 3, e0 = (tid & 7) * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int n = 0; n < Tv; ++n) {
            const float p = ks[n * 33 + d];
            const float4 v4 = *reinterpret_cast<const float4*>(vs + n * 32 + e0);
            acc.x += p * v4.x; acc.y += p * v4.y; acc.z += p * v4.z; acc.w += p * v4.w;
        }
        ctx[d * 33 + e0] = acc.x; ctx[d * 33 + e0 + 1] = acc.y; ctx[d * 33 + e0 + 2] = acc.z; ctx[d * 33 + e0 + 3] = acc.w;
    }
    __syncthreads();
    const float scale = 0.17677669529663687f;   // 32 ** -0.5
    for (int i = tid; i < Tv * 32; i += 256) {
        const int n = i >> 5, d = i & 31;
        ks[n * 33 + d] = base[(size_t)n * (3 * LA_HID) + d] * scale;
    }
    __syncthreads();
    bool overflow = false;
    const int e0 = (tid & 7) * 4;
    for (int n = tid >> 3; n < Tv; n += 32) {   // out[n][e] = sum_d context[d][e] q[n][d]
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int d = 0; d < 32; ++d) {
            const float qd = ks[n * 33 + d];
            o[0] += ctx[d * 33 + e0] * qd; o[1] += ctx[d * 33 + e0 + 1] * qd;
            o[2] += ctx[d * 33 + e0 + 2] * qd; o[3] += ctx[d * 33 + e0 + 3] * qd;
        }
        h4 oh, ol;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            _Float16 a, l;
            split_f16(o[e], a, l);
            oh[e] = a; ol[e] = l;
            overflow |= !(fabsf(o[e]) < 65504.0f);
        }
        _Float16* dst = outs + (row0 + n) * (2 * LA_HID) + head * 64 + e0;   // split_pos(head * 32 + e0)
        *reinterpret_cast<h4*>(dst) = oh;
        *reinterpret_cast<h4*>(dst + 32) = ol;
    }
    if (overflow && range_flag) atomicOr(range_flag, 1);
}

// input-VJP of the core: dO fp32 rows [.][128] (= d out) -> d(q | k | v) as split rows [.][2 * 384], the A operand of the
// to_qkv^T GEMM.  Everything is linear in dO, so the power-of-two gradient scale of the chain passes through.
//   dctx[d][e] = sum_n qs[n][d] dO[n][e]            dq[n][d] = scale sum_e dO[n][e] ctx[d][e]
//   dv[n][e]   = sum_d P[n][d] dctx[d][e]           dP[n][d] = sum_e dctx[d][e] v[n][e]
//   dk[n][d]   = P[n][d] (dP[n][d] - sum_n' P[n'][d] dP[n'][d])
constexpr size_t LA_BWD_LDS = (size_t)(2 * LA_MAXF * 33 + 2 * LA_MAXF * 32 + 2 * 32 * 33 + 512 + 32) * sizeof(float);
__global__ __launch_bounds__(256) void linattn_core_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dO,
                                                               _Float16* __restrict__ dqkv, int Tp, int h, int Tv) {
    extern __shared__ __attribute__((aligned(16))) char la_lds[];
    float* ks = reinterpret_cast<float*>(la_lds);        // [LA_MAXF][33]: k -> P
    float* qs = ks + LA_MAXF * 33;                        // [LA_MAXF][33]: q * scale
    float* vs = qs + LA_MAXF * 33;                        // [LA_MAXF][32]
    float* gs = vs + LA_MAXF * 32;                        // [LA_MAXF][32]: dO
    float* ctx = gs + LA_MAXF * 32;                       // [32][33]
    float* dctx = ctx + 32 * 33;                          // [32][33]
    float* red = dctx + 32 * 33;                          // [512]
    float* sdot = red + 512;                              // [32]
    const int head = blockIdx.x, tid = threadIdx.x;
    const size_t row0 = (size_t)blockIdx.y * Tp + h;
    const float* base = qkv + row0 * (3 * LA_HID) + head * LA_DH;
    const float scale = 0.17677669529663687f;
    for (int i = tid; i < Tv * 32; i += 256) {
        const int n = i >> 5, d = i & 31;
        qs[n * 33 + d] = base[(size_t)n * (3 * LA_HID) + d] * scale;
        ks[n * 33 + d] = base[(size_t)n * (3 * LA_HID) + LA_HID + d];
        vs[n * 32 + d] = base[(size_t)n * (3 * LA_HID) + 2 * LA_HID + d];
        gs[n * 32 + d] = dO[(row0 + n) * LA_HID + head * LA_DH + d];
    }
    __syncthreads();
    la_softmax_frames(ks, red, Tv);
    {   // ctx and dctx: thread = (d, 4 columns e)
        const int d = tid >> 3, e0 = (tid & 7) * 4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        for (int n = 0; n < Tv; ++n) {
            const float p = ks[n * 33 + d], q = qs[n * 33 + d];
            const float4 v4 = *reinterpret_cast<const float4*>(vs + n * 32 + e0);
            const float4 g4 = *reinterpret_cast<const float4*>(gs + n * 32 + e0);
            a.x += p * v4.x; a.y += p * v4.y; a.z += p * v4.z; a.w += p * v4.w;
            b.x += q * g4.x; b.y += q * g4.y; b.z += q * g4.z; b.w += q * g4.w;
        }
        ctx[d * 33 + e0] = a.x; ctx[d * 33 + e0 + 1] = a.y; ctx[d * 33 + e0 + 2] = a.z; ctx[d * 33 + e0 + 3] = a.w;
        dctx[d * 33 + e0] = b.x; dctx[d * 33 + e0 + 1] = b.y; dctx[d * 33 + e0 + 2] = b.z; dctx[d * 33 + e0 + 3] = b.w;
    }
    __syncthreads();
    // per (frame n, 4 columns c0..c0+3): dq, dv, and dP (kept in qs: q is no longer needed once dq of the row is out)
    const int c0 = (tid & 7) * 4;
    auto store4 = [&](size_t row, int col, const float* v) {
        h4 oh, ol;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            _Float16 a, l;
            split_f16(v[e], a, l);
            oh[e] = a; ol[e] = l;
        }
        _Float16* dst = dqkv + row * (2 * 3 * LA_HID) + split_pos(col);
        *reinterpret_cast<h4*>(dst) = oh;
        *reinterpret_cast<h4*>(dst + 32) = ol;
    };
    for (int n = tid >> 3; n < Tv; n += 32) {
        float dq[4] = {0.f, 0.f, 0.f, 0.f}, dv[4] = {0.f, 0.f, 0.f, 0.f}, dp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int j = 0; j < 32; ++j) {
            const float go = gs[n * 32 + j], vj = vs[n * 32 + j], pj = ks[n * 33 + j];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dq[e] += go * ctx[(c0 + e) * 33 + j];        // sum over e' = j of dO[n][j] ctx[d = c0+e][j]
                dp[e] += dctx[(c0 + e) * 33 + j] * vj;       // sum over e' = j of dctx[d = c0+e][j] v[n][j]
                dv[e] += pj * dctx[j * 33 + c0 + e];         // sum over d = j of P[n][j] dctx[j][e = c0+e]
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) dq[e] *= scale;
        store4(row0 + n, head * LA_DH + c0, dq);
        store4(row0 + n, 2 * LA_HID + head * LA_DH + c0, dv);
        // all 8 threads of frame n have read q-row n? they never read qs in this loop -> safe to overwrite
#pragma unroll
        for (int e = 0; e < 4; ++e) qs[n * 33 + c0 + e] = dp[e];
    }
    __syncthreads();
    {   // sdot[d] = sum_n P[n][d] dP[n][d]
        const int d = tid & 31, part = tid >> 5;
        float s = 0.f;
        for (int n = part; n < Tv; n += 8) s += ks[n * 33 + d] * qs[n * 33 + d];
        red[part * 32 + d] = s;
        __syncthreads();
        if (tid < 32) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) t += red[q * 32 + tid];
            sdot[tid] = t;
        }
        __syncthreads();
    }
    for (int n = tid >> 3; n < Tv; n += 32) {
        float dk[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) dk[e] = ks[n * 33 + c0 + e] * (qs[n * 33 + c0 + e] - sdot[c0 + e]);
        store4(row0 + n, LA_HID + head * LA_DH + c0, dk);
    }
}

}  // namespace cmdi
