// fp32-equivalent "NT" GEMM on the f16 matrix pipe:  C = epi(A[M,K] · W[N,K]^T), fp32 accumulate.
//
// Why: gfx950 has no TF32-class path; its exact fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of
// the f16 rate.  Every fp32 operand x is therefore carried as TWO halves
//     hi = f16(x),   lo = f16((x - hi) * 2^11)            (x = hi + lo * 2^-11 to 22 significant bits,
//                                                          |x| < 65504; checked, see split_f16_kernel)
// and a product of two such numbers is evaluated with THREE v_mfma_f32_32x32x16_f16:
//     acc0 += a_hi·w_hi          acc1 += a_hi·w_lo + a_lo·w_hi          C = acc0 + acc1 * 2^-11
// (f16 x f16 products are exact in the fp32 accumulator; the dropped lo·lo term is 2^-22 relative).
// The representation error (<= 2^-22 per operand) sits below the fp32 accumulation roundoff of a
// K >= 64 dot product, so results are fp32-class: tests/test_gpu_parity.py holds this path to the
// same tolerances as the exact-fp32 kernels, and tools/split_accuracy.py shows the error against
// float64 (f16x3: 2.6e-7 rel-L2 on a full denoiser evaluation; numpy fp32: 4.9e-7).
//
// Operand format ("split rows"): row r of a [rows, K] matrix is 2K halves — K hi values followed by
// K lo values — so a split matrix occupies exactly the bytes of its fp32 original.
//
// Roles are swapped inside the MFMA (W rows feed the A operand, activation rows the B operand): a
// lane's 16 accumulators are then 4 runs of 4 CONSECUTIVE output columns n of ONE row m, and the
// epilogue loads bias / residual and stores C as float4 (split outputs as 8-byte half4).
#pragma once
#include "common.hpp"
#include "gemm_params.hpp"

namespace cmdi {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr float kLoScale = 2048.0f;          // 2^11
constexpr float kLoInv = 1.0f / 2048.0f;

__device__ __forceinline__ void split_f16(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;                         // round to nearest even
    lo = (_Float16)((x - (float)hi) * kLoScale);
}

template <int BM_, int BN_, int BK_, int WM_, int WN_, int MINW_>
struct H3Tile {
    static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_, MINW = MINW_;
    static constexpr int NW = WM * WN, NT = 64 * NW;
    static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static constexpr int RB = BK * 2;              // bytes per tile row per plane
    static constexpr int SL = RB / 16;             // 16-B slots per row
    static constexpr int RPB = 16 / SL;            // rows per 256-B bank row
    static constexpr int PROWS = 1024 / RB;        // rows per LDS-DMA piece (one wave-instruction)
    static constexpr int STAGE = 2 * (BM + BN) * RB;   // bytes: A_hi, A_lo, W_hi, W_lo
    static constexpr size_t LDS_BYTES = 2ull * STAGE;
    static_assert(BK == 32 || BK == 64, "BK");
    static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "wave tile must be 32-aligned");
    static_assert((BM / PROWS) % NW == 0 && (BN / PROWS) % NW == 0, "DMA pieces per wave");
};

template <class TC, int EPI>
__global__ __launch_bounds__(TC::NT, TC::MINW) void gemm_h3_kernel(const H3Params p) {
    constexpr int BM = TC::BM, BN = TC::BN, BK = TC::BK, TM = TC::TM, TN = TC::TN, NW = TC::NW;
    constexpr int RB = TC::RB, SL = TC::SL, RPB = TC::RPB, PROWS = TC::PROWS, STAGE = TC::STAGE;
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave / TC::WN, wn = wave % TC::WN;

    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;

    f32x16 acc0[TN][TM], acc1[TN][TM];   // [W fragment (rows n)][A fragment (cols m)]
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[j][i][r] = 0.f; acc1[j][i][r] = 0.f; }

    // ---- LDS-DMA staging: HBM/L2 -> LDS, lane-linear 1-KiB pieces, swizzle on the SOURCE address:
    // the 16-B slot c of tile row r is stored at slot position c ^ ((r / RPB) % SL), so that the 16
    // rows of a ds_read_b128 lane group cover all 16 slots of a 256-B bank row.
    const int prow = lane / SL, pslot = lane % SL;
    const int K = p.K;
    auto issue = [&](int kt, int buf) {
        char* stage = lds + buf * STAGE;
#pragma unroll
        for (int plane = 0; plane < 2; ++plane) {
#pragma unroll
            for (int g0 = 0; g0 < BM / PROWS; g0 += NW) {
                const int g = g0 + wave;
                const int row = g * PROWS + prow;
                const int c = pslot ^ ((row / RPB) % SL);
                int grow = m0 + row;
                grow = grow < p.M ? grow : p.M - 1;
                const _Float16* src = p.A + (size_t)grow * (2 * K) + plane * K + kt * BK + c * 8;
                char* dst = stage + plane * BM * RB + g * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
        }
#pragma unroll
        for (int plane = 0; plane < 2; ++plane) {
#pragma unroll
            for (int g0 = 0; g0 < BN / PROWS; g0 += NW) {
                const int g = g0 + wave;
                const int row = g * PROWS + prow;
                const int c = pslot ^ ((row / RPB) % SL);
                int grow = n0 + row;
                grow = grow < p.N ? grow : p.N - 1;
                const _Float16* src = p.W + (size_t)grow * (2 * K) + plane * K + kt * BK + c * 8;
                char* dst = stage + 2 * BM * RB + plane * BN * RB + g * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
        }
    };

    const int swz = (l31 / RPB) % SL;
    int poff[BK / 16];   // byte offset inside a tile row of this lane's slot for k-substep ks
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) poff[ks] = ((2 * ks + hi) ^ swz) * 16;
    const int a_row = (wm * TM * 32 + l31) * RB;    // + i * 32 * RB
    const int w_row = (wn * TN * 32 + l31) * RB;    // + j * 32 * RB

    const int nk = K / BK;
    issue(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) issue(kt + 1, cur ^ 1);
        const char* st = lds + cur * STAGE;
        const char* a_hi = st + a_row;
        const char* a_lo = a_hi + BM * RB;
        const char* w_hi = st + 2 * BM * RB + w_row;
        const char* w_lo = w_hi + BN * RB;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            h8 ah[TM], al[TM], wh[TN], wl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *reinterpret_cast<const h8*>(a_hi + i * 32 * RB + poff[ks]);
                al[i] = *reinterpret_cast<const h8*>(a_lo + i * 32 * RB + poff[ks]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                wh[j] = *reinterpret_cast<const h8*>(w_hi + j * 32 * RB + poff[ks]);
                wl[j] = *reinterpret_cast<const h8*>(w_lo + j * 32 * RB + poff[ks]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc0[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], ah[i], acc0[j][i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc1[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[j], ah[i], acc1[j][i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc1[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], al[i], acc1[j][i], 0, 0, 0);
        }
        __syncthreads();  // with an LDS-DMA in flight hipcc puts s_waitcnt vmcnt(0) in front
    }

    // ---- epilogue: lane owns row m = .. + l31 and, per fragment, 4 runs of 4 consecutive n ------
    bool overflow = false;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + (wm * TM + i) * 32 + l31;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + (wn * TN + j) * 32 + 8 * q + 4 * hi;
                if (n >= p.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc0[j][i][4 * q + e] + acc1[j][i][4 * q + e] * kLoInv;
                if (p.bias) {
                    const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
                    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                }
                const size_t off = (size_t)m * p.ldc + n;
                if constexpr (EPI == H3_PLAIN) {
                    *reinterpret_cast<float4*>(p.C + off) = make_float4(v[0], v[1], v[2], v[3]);
                } else if constexpr (EPI == H3_RESID) {
                    const float4 rr = *reinterpret_cast<const float4*>(p.R + off);
                    *reinterpret_cast<float4*>(p.C + off) =
                        make_float4(v[0] + rr.x, v[1] + rr.y, v[2] + rr.z, v[3] + rr.w);
                } else {
                    if constexpr (EPI == H3_GELU_SPLIT) {
                        if (p.aux) *reinterpret_cast<float4*>(p.aux + off) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                    }
                    h4 oh, ol;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        _Float16 a, b;
                        split_f16(v[e], a, b);
                        oh[e] = a; ol[e] = b;
                        overflow |= !(fabsf(v[e]) < 65504.0f);
                    }
                    _Float16* dst = p.Cs + (size_t)m * (2 * p.N) + n;
                    *reinterpret_cast<h4*>(dst) = oh;
                    *reinterpret_cast<h4*>(dst + p.N) = ol;
                }
            }
        }
    }
    if constexpr (EPI == H3_GELU_SPLIT || EPI == H3_PLAIN_SPLIT) {
        if (overflow && p.range_flag) atomicOr(p.range_flag, 1);
    }
}

}  // namespace cmdi
