// Self-attention of one TransformerEncoderLayer (torch MultiheadAttention called from
// model/mdm.py:284 via nn.TransformerEncoder; math in SURVEY.md Appendix A.2):
//   P = softmax_row(Q_h K_hᵀ / sqrt(128)) over all S = T+1 tokens (no mask), A = concat_h(P V_h)
// fp32 on v_mfma_f32_32x32x2_f32, d_head = 128, S <= 224.
//
// Work split: grid = (B'·H, ceil(S/128)); 4 waves per block, each wave owns 32 queries.  All S
// keys fit one pass, so the softmax is the plain two-pass form (no online rescale):
//   phase 1: Sᵀ tile = K_tile · Qᵀ (swapped operands, so a lane's 16 accumulators of each tile are
//            16 KEYS of ONE query -> the row max / row sum are in-lane plus one lane^32 exchange);
//   phase 2: p = exp(s - max) / sum, kept in the same registers;
//   phase 3: O += P · V_tile with P's accumulator registers used directly as the MFMA A operand
//            (A[i=query][k=hi] is exactly what lane (query, hi) holds in register r).
// K / V tiles (32 keys x 128 dims) are staged through double-buffered LDS by all 4 waves.
#include "common.hpp"
#include "kernels.hpp"

namespace cmdi {

constexpr int DH = 128;          // head dim
constexpr int KT = 32;           // keys per tile
constexpr int KLD = DH + 4;      // LDS row stride (floats): 528 B = 33 slots of 16 B (odd) -> the
                                 // 16 rows of a ds_read_b128 lane group hit 16 distinct slots
constexpr int MAX_KT = 7;        // up to 224 keys

__device__ __forceinline__ void stage_tile(float* lds, const float* __restrict__ src, int row_ld,
                                           int key0, int S, int tid) {
    // 32 keys x 128 dims = 1024 float4, 4 per thread; rows past S are zero-filled
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + i * 256;
        const int r = idx >> 5, c4 = idx & 31;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (key0 + r < S) v = *reinterpret_cast<const float4*>(src + (size_t)(key0 + r) * row_ld + c4 * 4);
        *reinterpret_cast<float4*>(&lds[r * KLD + c4 * 4]) = v;
    }
}

template <bool STASH>
__global__ __launch_bounds__(256) void attention_fwd_kernel(const float* __restrict__ qkv,
                                                            float* __restrict__ out,
                                                            float* __restrict__ row_stats, int S,
                                                            int H, float scale) {
    __shared__ __attribute__((aligned(16))) float kv[2][KT * KLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int d_model = H * DH, ld = 3 * d_model;
    const int q0 = blockIdx.y * 128 + wave * 32;
    const bool active = q0 < S;  // wave-uniform
    const int nkt = (S + KT - 1) / KT;

    const float* qbase = qkv + (size_t)b * S * ld + h * DH;
    const float* kbase = qbase + d_model;
    const float* vbase = qbase + 2 * d_model;

    // Q fragment as MFMA B operand: lane (query=l31, hi) holds Q[q][c*8 + 4*hi + j], pre-scaled.
    float4 qf[16];
    {
        const int q = q0 + l31;
        const bool ok = active && q < S;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) v = *reinterpret_cast<const float4*>(qbase + (size_t)q * ld + c * 8 + hi * 4);
            qf[c] = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
        }
    }

    // ---- phase 1: scores -------------------------------------------------------------------
    f32x16 s[MAX_KT];
#pragma unroll
    for (int t = 0; t < MAX_KT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[t][r] = 0.f;

    stage_tile(kv[0], kbase, ld, 0, S, tid);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < MAX_KT; ++t) {
        if (t < nkt) {
            if (t + 1 < nkt) stage_tile(kv[(t + 1) & 1], kbase, ld, (t + 1) * KT, S, tid);
            if (active) {
                const float* kb = kv[t & 1] + l31 * KLD + hi * 4;
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const float4 a = *reinterpret_cast<const float4*>(kb + c * 8);
                    s[t] = mfma32(a.x, qf[c].x, s[t]);
                    s[t] = mfma32(a.y, qf[c].y, s[t]);
                    s[t] = mfma32(a.z, qf[c].z, s[t]);
                    s[t] = mfma32(a.w, qf[c].w, s[t]);
                }
            }
            __syncthreads();
        }
    }

    // ---- phase 2: softmax over keys (per query = per lane column) -------------------------
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < MAX_KT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = t * KT + mfma32_row(r, lane);
            if (key >= S) s[t][r] = -INFINITY;
            mx = fmaxf(mx, s[t][r]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < MAX_KT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = expf(s[t][r] - mx);
            s[t][r] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int t = 0; t < MAX_KT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[t][r] *= inv;

    if constexpr (STASH) {
        // Row statistics for the backward pass (P is recomputed there): [B'·H][S][2] = (max, 1/sum)
        const int q = q0 + l31;
        if (active && q < S && hi == 0) {
            row_stats[((size_t)bh * S + q) * 2] = mx;
            row_stats[((size_t)bh * S + q) * 2 + 1] = inv;
        }
    }

    // ---- phase 3: O = P · V -----------------------------------------------------------------
    f32x16 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;

    stage_tile(kv[0], vbase, ld, 0, S, tid);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < MAX_KT; ++t) {
        if (t < nkt) {
            if (t + 1 < nkt) stage_tile(kv[(t + 1) & 1], vbase, ld, (t + 1) * KT, S, tid);
            if (active) {
                const float* vb = kv[t & 1] + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // MFMA step r contracts key (r&3)+8(r>>2) [hi=0 lanes] and that key + 4 [hi=1]
                    const float* vr = vb + mfma32_row(r, lane) * KLD;
                    const float a = s[t][r];
#pragma unroll
                    for (int d = 0; d < 4; ++d) o[d] = mfma32(a, vr[d * 32], o[d]);
                }
            }
            __syncthreads();
        }
    }

    if (active) {
        float* ob = out + (size_t)b * S * d_model + h * DH + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = q0 + mfma32_row(r, lane);
            if (q < S) {
#pragma unroll
                for (int d = 0; d < 4; ++d) ob[(size_t)q * d_model + d * 32] = o[d][r];
            }
        }
    }
}

hipError_t launch_attention_fwd(const float* qkv, float* out, float* row_stats, int n_seq, int S,
                                int H, hipStream_t stream) {
    dim3 grid(n_seq * H, (S + 127) / 128);
    const float scale = 1.0f / sqrtf((float)DH);
    if (row_stats)
        hipLaunchKernelGGL(attention_fwd_kernel<true>, grid, dim3(256), 0, stream, qkv, out, row_stats,
                           S, H, scale);
    else
        hipLaunchKernelGGL(attention_fwd_kernel<false>, grid, dim3(256), 0, stream, qkv, out,
                           row_stats, S, H, scale);
    return hipGetLastError();
}

}  // namespace cmdi
