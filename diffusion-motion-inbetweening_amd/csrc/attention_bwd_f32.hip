// dX-only backward of the self-attention block, used by the reconstruction-guidance VJP
// (replaces torch.autograd.grad through torch MultiheadAttention, reference call site
// diffusion/gaussian_diffusion.py:411-416).  With P = softmax(S), S = scale·Q Kᵀ, O = P V:
//   dV = Pᵀ dO,  dP = dO Vᵀ,  D[q] = sum_k P dP = sum_d dO·O,  dS = P ∘ (dP − D),
//   dQ = scale · dS K,  dK = scale · dSᵀ Q.
// P is recomputed from the forward's row statistics (max, 1/sum), nothing S×S is ever stored.
//
// Two kernels, each wave owning a 32-row block of the axis it reduces INTO:
//   attn_bwd_q : wave = 32 queries, loops over key tiles   -> dQ  (and writes D)
//   attn_bwd_kv: wave = 32 keys,    loops over query tiles -> dK, dV
// In both, the tile product is oriented so that the 16 accumulator registers of a lane already ARE
// the MFMA A-operand of the follow-up product (no transposes, no atomics).
#include "common.hpp"
#include "kernels.hpp"

namespace cmdi {

namespace {
constexpr int DH = 128, KT = 32, KLD = DH + 4;

__device__ __forceinline__ void stage(float* lds, const float* __restrict__ src, int row_ld,
                                      int row0, int S, int tid, float mul) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + i * 256;
        const int r = idx >> 5, c4 = idx & 31;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + r < S)
            v = *reinterpret_cast<const float4*>(src + (size_t)(row0 + r) * row_ld + c4 * 4);
        *reinterpret_cast<float4*>(&lds[r * KLD + c4 * 4]) =
            make_float4(v.x * mul, v.y * mul, v.z * mul, v.w * mul);
    }
}

// 32x128 block as MFMA B operand held in registers: lane (row=l31, hi) keeps
// M[row][c*8 + 4*hi + j], j = 0..3, c = 0..15.
__device__ __forceinline__ void load_bfrag(float4 f[16], const float* __restrict__ base, int row_ld,
                                           int row, bool ok, int hi, float mul) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) v = *reinterpret_cast<const float4*>(base + (size_t)row * row_ld + c * 8 + hi * 4);
        f[c] = make_float4(v.x * mul, v.y * mul, v.z * mul, v.w * mul);
    }
}

// acc += A_tile · Bfragᵀ where A_tile rows live in LDS (lane row l31) and Bfrag is in registers.
__device__ __forceinline__ f32x16 tile_dot(const float* lds_rows, const float4 f[16], int l31,
                                           int hi) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* ab = lds_rows + l31 * KLD + hi * 4;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const float4 a = *reinterpret_cast<const float4*>(ab + c * 8);
        acc = mfma32(a.x, f[c].x, acc);
        acc = mfma32(a.y, f[c].y, acc);
        acc = mfma32(a.z, f[c].z, acc);
        acc = mfma32(a.w, f[c].w, acc);
    }
    return acc;
}

// out[d] += Xᵀ-as-A · rows: step r contracts the row index mfma32_row(r, lane) of the LDS tile.
__device__ __forceinline__ void acc_rows(f32x16 out[4], const f32x16& x, const float* lds_rows,
                                         int lane) {
    const float* vb = lds_rows + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float* vr = vb + mfma32_row(r, lane) * KLD;
        const float a = x[r];
#pragma unroll
        for (int d = 0; d < 4; ++d) out[d] = mfma32(a, vr[d * 32], out[d]);
    }
}
}  // namespace

// ---- dQ ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(const float* __restrict__ qkv,
                                                         const float* __restrict__ o_fwd,
                                                         const float* __restrict__ row_stats,
                                                         const float* __restrict__ d_o,
                                                         float* __restrict__ d_qkv,
                                                         float* __restrict__ d_rowdot, int S, int H,
                                                         float scale) {
    __shared__ __attribute__((aligned(16))) float kt[KT * KLD];
    __shared__ __attribute__((aligned(16))) float vt[KT * KLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int d_model = H * DH, ld = 3 * d_model;
    const int q0 = blockIdx.y * 128 + wave * 32;
    const bool active = q0 < S;
    const int nkt = (S + KT - 1) / KT;
    const int q = q0 + l31;
    const bool qok = active && q < S;

    const float* qbase = qkv + (size_t)b * S * ld + h * DH;
    const float* kbase = qbase + d_model;
    const float* vbase = qbase + 2 * d_model;
    const float* dobase = d_o + (size_t)b * S * d_model + h * DH;
    const float* obase = o_fwd + (size_t)b * S * d_model + h * DH;

    float4 qf[16], dof[16];
    load_bfrag(qf, qbase, ld, q, qok, hi, scale);
    load_bfrag(dof, dobase, d_model, q, qok, hi, 1.0f);
    // D[q] = sum_d dO[q][d] * O[q][d]; this lane covers the dims of its hi-half
    float dsum = 0.f;
    if (qok) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float4 ov = *reinterpret_cast<const float4*>(obase + (size_t)q * d_model + c * 8 + hi * 4);
            dsum += (dof[c].x * ov.x + dof[c].y * ov.y) + (dof[c].z * ov.z + dof[c].w * ov.w);
        }
    }
    dsum += __shfl_xor(dsum, 32, 64);
    float mx = 0.f, inv = 0.f;
    if (qok) {
        mx = row_stats[((size_t)bh * S + q) * 2];
        inv = row_stats[((size_t)bh * S + q) * 2 + 1];
        if (hi == 0) d_rowdot[(size_t)bh * S + q] = dsum;
    }

    f32x16 dq[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;

    for (int t = 0; t < nkt; ++t) {
        __syncthreads();
        stage(kt, kbase, ld, t * KT, S, tid, 1.0f);
        stage(vt, vbase, ld, t * KT, S, tid, 1.0f);
        __syncthreads();
        if (active) {
            f32x16 st = tile_dot(kt, qf, l31, hi);    // Sᵀ tile: lane = query, regs = keys
            f32x16 dpt = tile_dot(vt, dof, l31, hi);  // dPᵀ tile
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = t * KT + mfma32_row(r, lane);
                const float p = (key < S && qok) ? expf(st[r] - mx) * inv : 0.f;
                st[r] = p * (dpt[r] - dsum) * scale;  // dS (scaled)
            }
            acc_rows(dq, st, kt, lane);  // dQ += dS · K_tile
        }
    }

    if (active) {
        float* ob = d_qkv + (size_t)b * S * ld + h * DH + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = q0 + mfma32_row(r, lane);
            if (qq < S) {
#pragma unroll
                for (int d = 0; d < 4; ++d) ob[(size_t)qq * ld + d * 32] = dq[d][r];
            }
        }
    }
}

// ---- dK, dV --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(const float* __restrict__ qkv,
                                                          const float* __restrict__ row_stats,
                                                          const float* __restrict__ d_rowdot,
                                                          const float* __restrict__ d_o,
                                                          float* __restrict__ d_qkv, int S, int H,
                                                          float scale) {
    __shared__ __attribute__((aligned(16))) float qt[KT * KLD];
    __shared__ __attribute__((aligned(16))) float dot[KT * KLD];
    __shared__ float st_m[KT], st_inv[KT], st_d[KT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int d_model = H * DH, ld = 3 * d_model;
    const int k0 = blockIdx.y * 128 + wave * 32;
    const bool active = k0 < S;
    const int nqt = (S + KT - 1) / KT;
    const int key = k0 + l31;
    const bool kok = active && key < S;

    const float* qbase = qkv + (size_t)b * S * ld + h * DH;
    const float* kbase = qbase + d_model;
    const float* vbase = qbase + 2 * d_model;
    const float* dobase = d_o + (size_t)b * S * d_model + h * DH;

    float4 kf[16], vf[16];
    load_bfrag(kf, kbase, ld, key, kok, hi, 1.0f);
    load_bfrag(vf, vbase, ld, key, kok, hi, 1.0f);

    f32x16 dk[4], dv[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dk[d][r] = 0.f;
            dv[d][r] = 0.f;
        }

    for (int t = 0; t < nqt; ++t) {
        __syncthreads();
        stage(qt, qbase, ld, t * KT, S, tid, scale);  // Q pre-scaled: S = (scale·Q) Kᵀ, dK = dSᵀ (scale·Q)
        stage(dot, dobase, d_model, t * KT, S, tid, 1.0f);
        if (tid < KT) {
            const int qq = t * KT + tid;
            const bool ok = qq < S;
            st_m[tid] = ok ? row_stats[((size_t)bh * S + qq) * 2] : 0.f;
            st_inv[tid] = ok ? row_stats[((size_t)bh * S + qq) * 2 + 1] : 0.f;  // 0 masks q >= S
            st_d[tid] = ok ? d_rowdot[(size_t)bh * S + qq] : 0.f;
        }
        __syncthreads();
        if (active) {
            f32x16 s = tile_dot(qt, kf, l31, hi);    // S tile: lane = key, regs = queries
            f32x16 dp = tile_dot(dot, vf, l31, hi);  // dP tile
            f32x16 ds;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qi = mfma32_row(r, lane);
                const float p = expf(s[r] - st_m[qi]) * st_inv[qi];
                s[r] = p;
                ds[r] = p * (dp[r] - st_d[qi]);
            }
            acc_rows(dv, s, dot, lane);  // dV += Pᵀ · dO_tile
            acc_rows(dk, ds, qt, lane);  // dK += dSᵀ · (scale·Q)_tile
        }
    }

    if (active) {
        float* kb = d_qkv + (size_t)b * S * ld + d_model + h * DH + l31;
        float* vb = kb + d_model;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = k0 + mfma32_row(r, lane);
            if (kk < S) {
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    kb[(size_t)kk * ld + d * 32] = dk[d][r];
                    vb[(size_t)kk * ld + d * 32] = dv[d][r];
                }
            }
        }
    }
}

hipError_t launch_attention_bwd(const float* qkv, const float* o_fwd, const float* row_stats,
                                const float* d_out, float* d_qkv, float* d_rowdot, int n_seq, int S,
                                int H, hipStream_t stream) {
    const dim3 grid(n_seq * H, (S + 127) / 128), block(256);
    const float scale = 1.0f / sqrtf((float)DH);
    hipLaunchKernelGGL(attn_bwd_q_kernel, grid, block, 0, stream, qkv, o_fwd, row_stats, d_out, d_qkv,
                       d_rowdot, S, H, scale);
    hipLaunchKernelGGL(attn_bwd_kv_kernel, grid, block, 0, stream, qkv, row_stats, d_rowdot, d_out,
                       d_qkv, S, H, scale);
    return hipGetLastError();
}

}  // namespace cmdi
