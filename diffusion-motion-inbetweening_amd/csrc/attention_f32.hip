// Self-attention of one TransformerEncoderLayer (torch MultiheadAttention called from
// model/mdm.py:284 via nn.TransformerEncoder; math in SURVEY.md Appendix A.2):
//   P = softmax_row(Q_h K_hᵀ / sqrt(128)) over all S = T+1 tokens (no mask), A = concat_h(P V_h)
// fp32 on v_mfma_f32_16x16x4_f32, d_head = 128, S <= 224.
//
// Work split: grid = (B'·H, ceil(ceil(S/16)/8)); 8 waves per block, each wave owns 16 queries (S =
// 197 -> 13 query blocks, 13 key sub-tiles: 5 % padding instead of the 29 % a 32-row MFMA costs).
// K / V stream through LDS in 32-key stages (LDS-DMA, double buffered); per 16-key sub-tile a wave
// computes
//   Sᵀ = K_sub · Qᵀ      operands swapped, so a lane's accumulators are 4 KEYS of ONE query:
//                        row max / sum are in-lane + two lane-xor exchanges (lanes l, l^16, l^32)
//   online softmax       running max m and partial sum l per query = per lane (scalars)
//   Oᵀ += V_subᵀ · Pᵀ    Pᵀ's accumulator registers ARE the MFMA B operand (B[k=g][n=q] is what
//                        lane (q, g) holds in register r), V comes from LDS as the A operand; Oᵀ
//                        keeps queries on lanes, so the online rescale is a lane-wise multiply.
#include "common.hpp"
#include "gemm_h3.hpp"
#include "kernels.hpp"

namespace cmdi {

namespace {
constexpr int DH = 128;     // head dim
constexpr int STG = 32;     // keys per LDS stage
constexpr int NWAVE = 8;

typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4v mfma16(float a, float b, f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// One 32-key x 128-dim tile of K and one of V per stage, by LDS-DMA (global_load_lds_dwordx4).  A
// wave-instruction fills 1 KiB = 2 rows and the image is lane-linear, so bank conflicts are
// removed differently for the two tiles:
//   K (read as ds_read_b128, 16 key rows x one 16-B chunk per lane group): XOR swizzle on the
//     SOURCE address — chunk c of row r lands at chunk position c ^ (r & 15);
//   V (read as ds_read_b32, rows k and k+4 per 32-lane group, 16 consecutive dims each): the row
//     PAIRS are placed 1024 + 32 B apart, which shifts row k+4 by 16 banks against row k and keeps
//     every address linear in the dim (immediate offsets, no per-read address arithmetic).
// Rows past S re-read row S-1 (their scores are masked, their P is exactly 0; values stay finite).
constexpr int VPAIR = 264;                  // floats between consecutive V row pairs (1056 B)
constexpr int KTILE = STG * DH;             // 4096 floats
constexpr int VTILE = (STG / 2) * VPAIR;    // 4224 floats
constexpr int STAGE_FLOATS = KTILE + VTILE;

__device__ __forceinline__ void stage_kv(float* kbuf, float* vbuf, const float* __restrict__ kbase,
                                         const float* __restrict__ vbase, int row_ld, int key0, int S,
                                         int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g = (i & 1) * NWAVE + wave;       // 0..15: pair of rows inside the tile
        const int r = g * 2 + (lane >> 5);
        const int c = i < 2 ? ((lane & 31) ^ (r & 15)) : (lane & 31);
        int key = key0 + r;
        key = key < S ? key : S - 1;
        const float* gp = (i < 2 ? kbase : vbase) + (size_t)key * row_ld + c * 4;
        float* dst = i < 2 ? kbuf + g * 256 : vbuf + g * VPAIR;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}
}  // namespace

template <bool STASH>
__global__ __launch_bounds__(512, 4) void attention_fwd_kernel(const float* __restrict__ qkv,
                                                            float* __restrict__ out,
                                                            _Float16* __restrict__ out_s,
                                                            int* __restrict__ range_flag,
                                                            float* __restrict__ row_stats, int S,
                                                            int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) float kv[];  // [buffer][K tile | V tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int d_model = H * DH, ld = 3 * d_model;
    const int q0 = (blockIdx.y * NWAVE + wave) * 16;
    const bool active = q0 < S;  // wave-uniform
    const int nstage = (S + STG - 1) / STG;

    const float* qbase = qkv + (size_t)b * S * ld + h * DH;
    const float* kbase = qbase + d_model;
    const float* vbase = qbase + 2 * d_model;

    // Q as MFMA B operand: lane (query = l15, g) holds Q[q][16c + 4g + j], j = 0..3 — MFMA number
    // 4c + j of a sub-tile contracts dims {16c + 4g' + j : g' = 0..3} (a permutation of k, shared
    // with the K reads below).
    const int q = q0 + l15;
    const bool qok = active && q < S;
    float4 qf[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (qok) v = *reinterpret_cast<const float4*>(qbase + (size_t)q * ld + c * 16 + g * 4);
        qf[c] = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
    }

    f32x4v o[8];  // Oᵀ: o[db][reg] = O[q][16 db + 4 g + reg]
#pragma unroll
    for (int db = 0; db < 8; ++db) o[db] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    stage_kv(kv, kv + KTILE, kbase, vbase, ld, 0, S, wave, lane);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's DMA pieces have landed
    __syncthreads();

    for (int st = 0; st < nstage; ++st) {
        const int cur = st & 1;
        if (st + 1 < nstage)
            stage_kv(kv + (cur ^ 1) * STAGE_FLOATS, kv + (cur ^ 1) * STAGE_FLOATS + KTILE, kbase, vbase,
                     ld, (st + 1) * STG, S, wave, lane);
        if (active) {
            const float* kt = kv + cur * STAGE_FLOATS;
            const float* vt = kt + KTILE;
            const int nsub = (S - st * STG) > 16 ? 2 : 1;  // sub-tiles with at least one valid key
            // one 16-key sub-tile at a time (not unrolled: keeps the live set near
            // qf + o + one score fragment, so two blocks = 4 waves per SIMD fit the register file)
#pragma unroll 1
            for (int j = 0; j < nsub; ++j) {
                // ---- scores --------------------------------------------------------------
                const int krow = j * 16 + l15;  // this lane's key row (A operand row)
                const float* kr = kt + krow * DH;
                // two accumulator chains: a dependent 16x16x4 MFMA has 40-cycle latency
                f32x4v a0 = (f32x4v){0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
                for (int c = 0; c < 8; c += 2) {
                    const float4 k0 =
                        *reinterpret_cast<const float4*>(kr + (((4 * c + g) ^ (krow & 15)) << 2));
                    const float4 k1 =
                        *reinterpret_cast<const float4*>(kr + (((4 * c + 4 + g) ^ (krow & 15)) << 2));
                    a0 = mfma16(k0.x, qf[c].x, a0);
                    a1 = mfma16(k1.x, qf[c + 1].x, a1);
                    a0 = mfma16(k0.y, qf[c].y, a0);
                    a1 = mfma16(k1.y, qf[c + 1].y, a1);
                    a0 = mfma16(k0.z, qf[c].z, a0);
                    a1 = mfma16(k1.z, qf[c + 1].z, a1);
                    a0 = mfma16(k0.w, qf[c].w, a0);
                    a1 = mfma16(k1.w, qf[c + 1].w, a1);
                }
                f32x4v s;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = st * STG + j * 16 + 4 * g + r;  // accumulator row = key
                    s[r] = key < S ? a0[r] + a1[r] : -INFINITY;
                }
                // ---- online softmax (per query = per lane column) ------------------------
                float mloc = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
                mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
                mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
                const float m_new = fmaxf(m_run, mloc);   // finite: every sub-tile holds a valid key
                const float alpha = expf(m_run - m_new);  // exp(-inf) = 0 on the first sub-tile
                float psum = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = expf(s[r] - m_new);  // masked keys: exp(-inf) = 0
                    s[r] = pv;
                    psum += pv;
                }
                l_run = l_run * alpha + psum;  // partial over this lane's keys; summed at the end
                m_run = m_new;
#pragma unroll
                for (int db = 0; db < 8; ++db) {
                    o[db][0] *= alpha; o[db][1] *= alpha; o[db][2] *= alpha; o[db][3] *= alpha;
                }
                // ---- Oᵀ += Vᵀ · Pᵀ -------------------------------------------------------
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // MFMA step r contracts keys {16 j + 4 g' + r : g' = 0..3}
                    const int vrow = j * 16 + 4 * g + r;
                    const float* vr = vt + (vrow >> 1) * VPAIR + (vrow & 1) * DH + l15;
                    const float pb = s[r];
#pragma unroll
                    for (int db = 0; db < 8; ++db) o[db] = mfma16(vr[db * 16], pb, o[db]);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
    }

    if (active) {
        float lsum = l_run;
        lsum += __shfl_xor(lsum, 16, 64);
        lsum += __shfl_xor(lsum, 32, 64);
        const float inv = 1.0f / lsum;
        if (qok) {
            if constexpr (STASH) {
                // row statistics for the backward pass (P is recomputed there): (max, 1/sum)
                if (g == 0) {
                    row_stats[((size_t)bh * S + q) * 2] = m_run;
                    row_stats[((size_t)bh * S + q) * 2 + 1] = inv;
                }
            }
            if (out) {
                float* ob = out + ((size_t)b * S + q) * d_model + h * DH + 4 * g;
#pragma unroll
                for (int db = 0; db < 8; ++db)
                    *reinterpret_cast<float4*>(ob + db * 16) =
                        make_float4(o[db][0] * inv, o[db][1] * inv, o[db][2] * inv, o[db][3] * inv);
            }
            if (out_s) {
                // split rows for the out_proj GEMM on the f16 pipe (format: gemm_h3.hpp)
                _Float16* ob = out_s + ((size_t)b * S + q) * (2 * d_model) + split_pos(h * DH + 4 * g);
                bool overflow = false;
#pragma unroll
                for (int db = 0; db < 8; ++db) {
                    h4 oh, ol;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = o[db][e] * inv;
                        _Float16 a, c;
                        split_f16(v, a, c);
                        oh[e] = a; ol[e] = c;
                        overflow |= !(fabsf(v) < 65504.0f);
                    }
                    // dims advance by 16 per db: two db per 32-column chunk (64 halves)
                    *reinterpret_cast<h4*>(ob + (db >> 1) * 64 + (db & 1) * 16) = oh;
                    *reinterpret_cast<h4*>(ob + (db >> 1) * 64 + (db & 1) * 16 + 32) = ol;
                }
                if (overflow && range_flag) atomicOr(range_flag, 1);
            }
        }
    }
}

hipError_t launch_attention_fwd(const float* qkv, float* out, _Float16* out_split, int* range_flag,
                                float* row_stats, int n_seq, int S, int H, hipStream_t stream) {
    const int qblocks = (S + 15) / 16;
    dim3 grid(n_seq * H, (qblocks + NWAVE - 1) / NWAVE);
    const float scale = 1.0f / sqrtf((float)DH);
    constexpr size_t lds = 2ull * STAGE_FLOATS * sizeof(float);  // 66,560 B > the 64 KiB default
    static PerDevice<bool> attr_done_dev;
    bool& attr_done = attr_done_dev.get();
    if (!attr_done) {
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_fwd_kernel<true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_fwd_kernel<false>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e1 != hipSuccess) return e1;
        if (e2 != hipSuccess) return e2;
        attr_done = true;
    }
    if (row_stats)
        hipLaunchKernelGGL(attention_fwd_kernel<true>, grid, dim3(64 * NWAVE), lds, stream, qkv, out,
                           out_split, range_flag, row_stats, S, H, scale);
    else
        hipLaunchKernelGGL(attention_fwd_kernel<false>, grid, dim3(64 * NWAVE), lds, stream, qkv, out,
                           out_split, range_flag, row_stats, S, H, scale);
    return hipGetLastError();
}

}  // namespace cmdi
