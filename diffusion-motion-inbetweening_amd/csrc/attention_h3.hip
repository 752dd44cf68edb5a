// Self-attention core on the f16 matrix pipe with fp32-equivalent split-f16 products (same math as
// attention_f32.hip: torch MultiheadAttention called from model/mdm.py:284, SURVEY.md Appendix A.2;
// operand format and error analysis in gemm_h3.hpp).
//
// Input: qkv as SPLIT ROWS [B'·S][2·3d] (written by the in_proj GEMM's epilogue): per 32-column chunk
// 32 hi halves then 32 lo halves (lo = (x - hi)·2^11), so one head of Q, K or V of one token is 512
// contiguous bytes.  d_head = 128, S <= 224.
//
// Work split: grid = (B'·H, ceil(S/128)); 4 waves per block, 2 blocks per CU; a wave owns 32 queries.
// K / V stream through LDS in 32-key stages (LDS-DMA, double buffered, 32 KiB per stage).
// Per 32-key block a wave computes, with v_mfma_f32_32x32x16_f16,
//   Sᵀ = K·Qᵀ        operands swapped so that a lane owns ONE query (column) and 16 keys: row max /
//                    sum are in-lane plus one lane^32 exchange; three accumulators (hi·hi, hi·lo,
//                    lo·hi) keep dependent MFMAs apart;  s = (a0 + (a1 + a2)·2^-11)·scale
//   online softmax   running max m and partial sum l per query = per lane
//   Oᵀ += Vᵀ·Pᵀ      Pᵀ's registers ARE the B operand (register r = key (r&3)+8(r>>2)+4·hi).  P lies
//                    in [0, 1] with max 1, so its low part needs no scaling: p = p_hi + p_lo exactly
//                    to 2^-25 absolute, and the three products V_hi·p_hi + V_hi·p_lo +
//                    V_lo'·(p_hi·2^-11) share ONE accumulator.  Vᵀ fragments come from the row-major
//                    V tile through ds_read_b64_tr_b16 (hardware 4x4 transpose).
#include "common.hpp"
#include "gemm_h3.hpp"
#include "kernels.hpp"

namespace cmdi {

namespace {
constexpr int DH = 128;
constexpr int KBLK = 32;                 // keys per stage
constexpr int NWAVE = 4;
constexpr int ROWB = 512;                // bytes of one head of one token: 4 chunks x (64 B hi + 64 B lo)
constexpr int TILE = KBLK * ROWB;        // 16 KiB per K or V tile
constexpr int STAGE = 2 * TILE;

typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));

// 16-B slot swizzle of a tile row (32 slots per key): slot t of key k is stored at t ^ kswz(k).
// kswz is a bijection of k & 15 (K: the 16 keys of a ds_read_b128 lane group hit 16 different slots
// of the 256-B bank row) whose bits 2-3 are k & 3 (V: the 4 keys of a transpose-read group land in
// 4 different 64-B quarters).
__device__ __forceinline__ int kswz(int k) { return ((k & 3) << 2) | ((k >> 2) & 3); }

__device__ __forceinline__ void stage_kv(char* stage, const _Float16* __restrict__ base, size_t ld,
                                         int koff, int voff, int key0, int S, int wave, int lane) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int pc = it * NWAVE + wave;        // 0..15 K pieces, 16..31 V pieces (2 keys each)
        const int mat = pc >> 4, g = pc & 15;
        const int kl = 2 * g + (lane >> 5);
        const int t = (lane & 31) ^ kswz(kl);
        int key = key0 + kl;
        key = key < S ? key : S - 1;             // rows past S: masked scores, P == 0, finite values
        const _Float16* src = base + (size_t)key * ld + (mat ? voff : koff) + t * 8;
        char* dst = stage + mat * TILE + g * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}

__device__ __forceinline__ h8 tr_pair(const char* p0, const char* p1) {
    const s4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)p0);
    const s4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)p1);
    return __builtin_bit_cast(h8, (s8v)__builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}
}  // namespace

template <bool STASH>
__global__ __launch_bounds__(256, 2) void attention_h3_kernel(const _Float16* __restrict__ qkv,
                                                              float* __restrict__ out,
                                                              _Float16* __restrict__ out_s,
                                                              int* __restrict__ range_flag,
                                                              float* __restrict__ row_stats, int S,
                                                              int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) char lds[];  // [2][K tile | V tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int d_model = H * DH;
    const size_t ld = 6 * (size_t)d_model;           // halves per token row of the split qkv
    const int qoff = 2 * h * DH, koff = 2 * (d_model + h * DH), voff = 2 * (2 * d_model + h * DH);
    const _Float16* base = qkv + (size_t)b * S * ld;
    const int q0 = (blockIdx.y * NWAVE + wave) * 32;
    const bool active = q0 < S;                      // wave-uniform
    const int nkb = (S + KBLK - 1) / KBLK;

    // Q as the B operand of Sᵀ = K·Qᵀ: lane (query l31, k-group hi) holds dims 16 ks + 8 hi .. + 7
    const int q = q0 + l31;
    const bool qok = active && q < S;
    h8 qh[8], ql[8];
    {
        const _Float16* qp = base + (size_t)(q < S ? q : S - 1) * ld + qoff + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const _Float16* pch = qp + (ks >> 1) * 64 + (ks & 1) * 16;
            qh[ks] = *reinterpret_cast<const h8*>(pch);
            ql[ks] = *reinterpret_cast<const h8*>(pch + 32);
        }
    }

    f32x16 o[4];  // Oᵀ: o[db][r] = O[q][32 db + (r&3) + 8 (r>>2) + 4 hi]
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // K fragment addressing: row l31, slot (ks>>1)*8 + plane*4 + (ks&1)*2 + hi, swizzled
    const int fk = kswz(l31);
    const int k_row = l31 * ROWB;
    // V transpose-read addressing (16-lane group G, lane L): source lane L points at
    // V[k0 + (L >> 2)][d0 + 4 (L & 3) ..+3], d0 = 16 (G & 1); the group's lane i receives
    // V[k0 .. k0+3][d0 + i]: the A-operand rows (dims) of lanes 16 G' .. and k-group G >> 1.
    const int G = lane >> 4, L = lane & 15;
    const int v_kl = 4 * (G >> 1) + (L >> 2);        // + 16 kk + 8 hr
    const int v_slot = 2 * (G & 1) + ((L & 3) >> 1); // + db * 8 + plane * 4
    const int v_half = (L & 1) * 8;

    stage_kv(lds, base, ld, koff, voff, 0, S, wave, lane);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();

    for (int kb = 0; kb < nkb; ++kb) {
        const int cur = kb & 1;
        if (kb + 1 < nkb)
            stage_kv(lds + (cur ^ 1) * STAGE, base, ld, koff, voff, (kb + 1) * KBLK, S, wave, lane);
        if (active) {
            const char* kt = lds + cur * STAGE;
            const char* vt = kt + TILE;
            // ---- scores -------------------------------------------------------------------------
            f32x16 a0, a1, a2;
#pragma unroll
            for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; a2[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int t = (ks >> 1) * 8 + (ks & 1) * 2 + hi;
                const h8 kh = *reinterpret_cast<const h8*>(kt + k_row + ((t ^ fk) << 4));
                const h8 kl = *reinterpret_cast<const h8*>(kt + k_row + (((t + 4) ^ fk) << 4));
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], a2, 0, 0, 0);
            }
            float s[16];
            float mloc = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * KBLK + mfma32_row(r, lane);
                const float v = (a0[r] + (a1[r] + a2[r]) * kLoInv) * scale;
                s[r] = key < S ? v : -INFINITY;
                mloc = fmaxf(mloc, s[r]);
            }
            // ---- online softmax (per query = per lane column) -----------------------------------
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            const float m_new = fmaxf(m_run, mloc);   // finite: every block holds a valid key
            const float alpha = expf(m_run - m_new);  // exp(-inf) = 0 on the first block
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = expf(s[r] - m_new);            // masked keys: exp(-inf) = 0
                psum += s[r];
            }
            l_run = l_run * alpha + psum;             // partial over this lane's keys
            m_run = m_new;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
            // ---- Oᵀ += Vᵀ · Pᵀ -------------------------------------------------------------------
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                h8 ph, pl, ps;   // p_hi, p - p_hi (unscaled), p_hi * 2^-11
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float pv = s[8 * kk + e];
                    const _Float16 a = (_Float16)pv;
                    ph[e] = a;
                    pl[e] = (_Float16)(pv - (float)a);
                    ps[e] = (_Float16)((float)a * kLoInv);
                }
                h8 vh[4], vl[4];
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    const int k0 = 16 * kk + v_kl, k1 = k0 + 8;
                    const int t = db * 8 + v_slot;
                    vh[db] = tr_pair(vt + k0 * ROWB + ((t ^ kswz(k0)) << 4) + v_half,
                                     vt + k1 * ROWB + ((t ^ kswz(k1)) << 4) + v_half);
                    vl[db] = tr_pair(vt + k0 * ROWB + (((t + 4) ^ kswz(k0)) << 4) + v_half,
                                     vt + k1 * ROWB + (((t + 4) ^ kswz(k1)) << 4) + v_half);
                }
#pragma unroll
                for (int db = 0; db < 4; ++db)
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[db], ph, o[db], 0, 0, 0);
#pragma unroll
                for (int db = 0; db < 4; ++db)
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[db], pl, o[db], 0, 0, 0);
#pragma unroll
                for (int db = 0; db < 4; ++db)
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[db], ps, o[db], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
    }

    if (active) {
        const float lsum = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / lsum;
        if (qok) {
            if constexpr (STASH) {
                if (hi == 0) {
                    row_stats[((size_t)bh * S + q) * 2] = m_run;
                    row_stats[((size_t)bh * S + q) * 2 + 1] = inv;
                }
            }
            if (out) {
                float* ob = out + ((size_t)b * S + q) * d_model + h * DH + 4 * hi;
#pragma unroll
                for (int db = 0; db < 4; ++db)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
                        *reinterpret_cast<float4*>(ob + db * 32 + g4 * 8) =
                            make_float4(o[db][4 * g4] * inv, o[db][4 * g4 + 1] * inv,
                                        o[db][4 * g4 + 2] * inv, o[db][4 * g4 + 3] * inv);
            }
            if (out_s) {
                // split rows for the out_proj GEMM: chunk (h*4 + db), columns 8 g4 + 4 hi .. + 3
                _Float16* ob = out_s + ((size_t)b * S + q) * (2 * d_model) + split_pos(h * DH) + 4 * hi;
                bool overflow = false;
#pragma unroll
                for (int db = 0; db < 4; ++db)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        h4 oh, ol;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = o[db][4 * g4 + e] * inv;
                            _Float16 a, c;
                            split_f16(v, a, c);
                            oh[e] = a; ol[e] = c;
                            overflow |= !(fabsf(v) < 65504.0f);
                        }
                        *reinterpret_cast<h4*>(ob + db * 64 + g4 * 8) = oh;
                        *reinterpret_cast<h4*>(ob + db * 64 + g4 * 8 + 32) = ol;
                    }
                if (overflow && range_flag) atomicOr(range_flag, 1);
            }
        }
    }
}

hipError_t launch_attention_h3(const _Float16* qkv_split, float* out, _Float16* out_split,
                               int* range_flag, float* row_stats, int n_seq, int S, int H,
                               hipStream_t stream) {
    if (S < 1 || S > 224) return hipErrorInvalidValue;
    dim3 grid(n_seq * H, (S + 32 * NWAVE - 1) / (32 * NWAVE));
    const float scale = 1.0f / sqrtf((float)DH);
    constexpr size_t lds = 2ull * STAGE;  // 64 KiB
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_h3_kernel<true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_h3_kernel<false>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e1 != hipSuccess) return e1;
        if (e2 != hipSuccess) return e2;
        attr_done = true;
    }
    if (row_stats)
        hipLaunchKernelGGL(attention_h3_kernel<true>, grid, dim3(64 * NWAVE), lds, stream, qkv_split,
                           out, out_split, range_flag, row_stats, S, H, scale);
    else
        hipLaunchKernelGGL(attention_h3_kernel<false>, grid, dim3(64 * NWAVE), lds, stream, qkv_split,
                           out, out_split, range_flag, row_stats, S, H, scale);
    return hipGetLastError();
}

}  // namespace cmdi
