// fp32 "NT" GEMM on v_mfma_f32_32x32x2_f32 with fused epilogues:  C = epi(A[M,K] · W[N,K]^T).
//
// Every projection of the MDM denoiser (model/mdm.py:362,405; torch MultiheadAttention in_proj /
// out_proj; TransformerEncoderLayer.linear1/linear2; TimestepEmbedder.time_embed; embed_text) is a
// y = x·Wᵀ + b with W stored [out, in] (K contiguous), so one kernel family serves all of them, and
// — with the transposed weight copies made at cmdi_finalize_weights — the dX backward GEMMs too.
#pragma once
#include "common.hpp"
#include "gemm_params.hpp"

namespace cmdi {

template <int BM_, int BN_, int BK_, int WM_, int WN_>
struct Tile {
    static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_;
    static constexpr int NT = 64 * WM * WN;
    static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static constexpr int LDSK = BK + 4;  // row stride in floats: (BK+4)*4 B is an odd number of
                                         // 16-B slots -> conflict-free ds_read_b128 per lane group
    static constexpr size_t LDS_BYTES = 2ull * (BM + BN) * LDSK * sizeof(float);
    static constexpr size_t LDS_BYTES_DMA = 2ull * (BM + BN) * BK * sizeof(float);  // PIPE = 2
    static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "wave tile must be 32-aligned");
    static_assert((BM * BK) % (4 * NT) == 0 && (BN * BK) % (4 * NT) == 0, "loader divisibility");
};

template <int MODE, int ROWS, int BK, int NT>
struct TileLoader {
    static constexpr int NV = ROWS * BK / NT;  // floats per thread
    float v[NV];

    template <bool CLAMP = false>
    __device__ __forceinline__ void load(const float* __restrict__ ptr, int nrows, int ld, int row0,
                                         int k0, int tid, int T, int S, int Cf) {
        if constexpr (MODE == ROWS_MOTION) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int idx = tid + i * NT;
                const int r = idx % ROWS, k = idx / ROWS;
                const int row = row0 + r, c = k0 + k;
                float val = 0.f;
                if (row < nrows && c < Cf) {
                    const int b = row / T, t = row - b * T;
                    val = ptr[((size_t)b * Cf + c) * T + t];
                }
                v[i] = val;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NV / 4; ++i) {
                const int idx = tid + i * NT;
                const int r = idx / (BK / 4), c4 = idx % (BK / 4);
                int row = row0 + r;
                float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
                // CLAMP: rows past the end re-read the last valid row (branch-free, so all loads
                // of a tile issue back to back); their products land in rows the epilogue drops.
                if constexpr (CLAMP) row = row < nrows ? row : nrows - 1;
                if (CLAMP || row < nrows) {
                    size_t phys = row;
                    if constexpr (MODE == ROWS_TOK) {
                        const int b = row / T;
                        phys = (size_t)b * S + 1 + (row - b * T);
                    }
                    val = *reinterpret_cast<const float4*>(ptr + phys * ld + k0 + c4 * 4);
                }
                v[4 * i + 0] = val.x; v[4 * i + 1] = val.y; v[4 * i + 2] = val.z; v[4 * i + 3] = val.w;
            }
        }
    }

    __device__ __forceinline__ void store(float* lds, int tid) const {
        constexpr int LDSK = BK + 4;
        if constexpr (MODE == ROWS_MOTION) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int idx = tid + i * NT;
                lds[(idx % ROWS) * LDSK + idx / ROWS] = v[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < NV / 4; ++i) {
                const int idx = tid + i * NT;
                const int r = idx / (BK / 4), c4 = idx % (BK / 4);
                *reinterpret_cast<float4*>(&lds[r * LDSK + c4 * 4]) =
                    make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
            }
        }
    }
};

template <class TC, int AMODE, int BMODE, int EPI, int PIPE = 0>
__global__ __launch_bounds__(TC::NT) void gemm_nt_kernel(const GemmParams p) {
    constexpr int BM = TC::BM, BN = TC::BN, BK = TC::BK, LDSK = TC::LDSK;
    constexpr int TM = TC::TM, TN = TC::TN, NT = TC::NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                     // [2][BM][LDSK]
    float* Ws = smem + 2 * BM * LDSK;     // [2][BN][LDSK]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave / TC::WN, wn = wave % TC::WN;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    TileLoader<AMODE, BM, BK, NT> la;
    TileLoader<BMODE, BN, BK, NT> lw;
    const int nk = p.K / BK;

    // Residual / aux operand of the epilogue, prefetched under the last K tile (PIPE = 1).
    constexpr bool PRE = PIPE >= 1 && (EPI == EPI_RESID || EPI == EPI_ACCUM || EPI == EPI_GELUGRAD);
    float pre[PRE ? TM : 1][PRE ? TN : 1][16];

    if constexpr (PIPE == 0) {
        la.load(p.A, p.M, p.lda, m0, 0, tid, p.T, p.S, p.Cf);
        lw.load(p.W, p.N, p.ldw, n0, 0, tid, p.T, p.S, p.Cf);
        la.store(As, tid);
        lw.store(Ws, tid);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) {
                la.load(p.A, p.M, p.lda, m0, (kt + 1) * BK, tid, p.T, p.S, p.Cf);
                lw.load(p.W, p.N, p.ldw, n0, (kt + 1) * BK, tid, p.T, p.S, p.Cf);
            }
        const float* a_base = As + cur * BM * LDSK + (wm * TM * 32 + l31) * LDSK + hi * 4;
            const float* w_base = Ws + cur * BN * LDSK + (wn * TN * 32 + l31) * LDSK + hi * 4;
#pragma unroll
            for (int kc = 0; kc < BK / 8; ++kc) {
                // Lane (row, hi) takes k = kc*8 + 4*hi + j for j = 0..3 from BOTH operands, so MFMA
                // step j multiplies matching k's: the K order inside a chunk of 8 is permuted, the sum
                // is not.
                float4 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[i] = *reinterpret_cast<const float4*>(a_base + i * 32 * LDSK + kc * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[j] = *reinterpret_cast<const float4*>(w_base + j * 32 * LDSK + kc * 8);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(a[i].x, b[j].x, acc[i][j]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(a[i].y, b[j].y, acc[i][j]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(a[i].z, b[j].z, acc[i][j]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(a[i].w, b[j].w, acc[i][j]);
            }
            if (kt + 1 < nk) {
                la.store(As + (cur ^ 1) * BM * LDSK, tid);
                lw.store(Ws + (cur ^ 1) * BN * LDSK, tid);
            }
            __syncthreads();
        }
    } else if constexpr (PIPE == 1) {
        // Software pipeline, one register set: while tile kt is multiplied out of LDS[kt&1], tile
        // kt+1 (fetched during the previous iteration) is written to LDS[(kt+1)&1] at the TOP of the
        // iteration and the global loads of tile kt+2 are issued right behind it, so every load has
        // a whole K tile of MFMAs to land under and the barrier never waits on memory.
        la.template load<true>(p.A, p.M, p.lda, m0, 0, tid, p.T, p.S, p.Cf);
        lw.template load<true>(p.W, p.N, p.ldw, n0, 0, tid, p.T, p.S, p.Cf);
        la.store(As, tid);
        lw.store(Ws, tid);
        if (nk > 1) {
            la.template load<true>(p.A, p.M, p.lda, m0, BK, tid, p.T, p.S, p.Cf);
            lw.template load<true>(p.W, p.N, p.ldw, n0, BK, tid, p.T, p.S, p.Cf);
        }
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) {
                la.store(As + (cur ^ 1) * BM * LDSK, tid);
                lw.store(Ws + (cur ^ 1) * BN * LDSK, tid);
            }
            if (kt + 2 < nk) {
                la.template load<true>(p.A, p.M, p.lda, m0, (kt + 2) * BK, tid, p.T, p.S, p.Cf);
                lw.template load<true>(p.W, p.N, p.ldw, n0, (kt + 2) * BK, tid, p.T, p.S, p.Cf);
            }
            if constexpr (PRE) {
                if (kt == nk - 1) {
                    const float* src = EPI == EPI_GELUGRAD ? p.aux : p.R;
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const int n = n0 + (wn * TN + j) * 32 + l31;
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                int m = m0 + (wm * TM + i) * 32 + mfma32_row(r, lane);
                                m = m < p.M ? m : p.M - 1;
                                pre[i][j][r] = src[(size_t)m * p.ldc + (n < p.N ? n : p.N - 1)];
                            }
                        }
                }
            }
        const float* a_base = As + cur * BM * LDSK + (wm * TM * 32 + l31) * LDSK + hi * 4;
            const float* w_base = Ws + cur * BN * LDSK + (wn * TN * 32 + l31) * LDSK + hi * 4;
#pragma unroll
            for (int kc = 0; kc < BK / 8; ++kc) {
                // Lane (row, hi) takes k = kc*8 + 4*hi + j for j = 0..3 from BOTH operands, so MFMA
                // step j multiplies matching k's: the K order inside a chunk of 8 is permuted, the sum
                // is not.
                float4 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[i] = *reinterpret_cast<const float4*>(a_base + i * 32 * LDSK + kc * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[j] = *reinterpret_cast<const float4*>(w_base + j * 32 * LDSK + kc * 8);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(a[i].x, b[j].x, acc[i][j]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(a[i].y, b[j].y, acc[i][j]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(a[i].z, b[j].z, acc[i][j]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(a[i].w, b[j].w, acc[i][j]);
            }
            __syncthreads();
        }
    }

    if constexpr (PIPE == 2) {
        // LDS-DMA pipeline: tiles go HBM/L2 -> LDS with global_load_lds_dwordx4 (no staging VGPRs,
        // no ds_write pass).  The LDS image is lane-linear per wave-instruction (8 rows x 128 B),
        // so the bank-conflict swizzle sits on the SOURCE address: the 16-B chunk c of row r is
        // stored at chunk position c ^ ((r >> 1) & 7); a ds_read_b128 lane group (16 rows, one
        // logical chunk) then covers all 16 slots of the 256-B bank row.
        static_assert(AMODE != ROWS_MOTION && BMODE != ROWS_MOTION && BK == 32, "glds path");
        constexpr int NW = NT / 64;
        char* lds_bytes = reinterpret_cast<char*>(smem);
        const int lrow = lane >> 3, lp = lane & 7;
        auto issue = [&](int kt, int buf) {
#pragma unroll
            for (int g0 = 0; g0 < BM / 8; g0 += NW) {
                const int g = g0 + wave;
                const int row = g * 8 + lrow;
                const int c = lp ^ ((row >> 1) & 7);
                int grow = m0 + row;
                grow = grow < p.M ? grow : p.M - 1;
                size_t phys = grow;
                if constexpr (AMODE == ROWS_TOK) {
                    const int b = grow / p.T;
                    phys = (size_t)b * p.S + 1 + (grow - b * p.T);
                }
                const float* src = p.A + phys * p.lda + kt * BK + c * 4;
                char* dst = lds_bytes + (size_t)buf * BM * 128 + g * 1024;
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)src,
                    (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
#pragma unroll
            for (int g0 = 0; g0 < BN / 8; g0 += NW) {
                const int g = g0 + wave;
                const int row = g * 8 + lrow;
                const int c = lp ^ ((row >> 1) & 7);
                int grow = n0 + row;
                grow = grow < p.N ? grow : p.N - 1;
                size_t phys = grow;
                if constexpr (BMODE == ROWS_TOK) {
                    const int b = grow / p.T;
                    phys = (size_t)b * p.S + 1 + (grow - b * p.T);
                }
                const float* src = p.W + phys * p.ldw + kt * BK + c * 4;
                char* dst = lds_bytes + (size_t)(2 * BM + buf * BN) * 128 + g * 1024;
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)src,
                    (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
        };
        const int swz = (l31 >> 1) & 7;
        int poff[BK / 8];  // byte offset of this lane's logical chunk (2*kc + hi) inside a row
#pragma unroll
        for (int kc = 0; kc < BK / 8; ++kc) poff[kc] = ((2 * kc + hi) ^ swz) * 16;

        issue(0, 0);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) (expcnt/lgkmcnt untouched)
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) issue(kt + 1, cur ^ 1);
            if constexpr (PRE) {
                if (kt == nk - 1) {
                    const float* src = EPI == EPI_GELUGRAD ? p.aux : p.R;
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const int n = n0 + (wn * TN + j) * 32 + l31;
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                int m = m0 + (wm * TM + i) * 32 + mfma32_row(r, lane);
                                m = m < p.M ? m : p.M - 1;
                                pre[i][j][r] = src[(size_t)m * p.ldc + (n < p.N ? n : p.N - 1)];
                            }
                        }
                }
            }
            const char* a_rows = lds_bytes + (size_t)cur * BM * 128 + (wm * TM * 32 + l31) * 128;
            const char* w_rows = lds_bytes + (size_t)(2 * BM + cur * BN) * 128 + (wn * TN * 32 + l31) * 128;
#pragma unroll
            for (int kc = 0; kc < BK / 8; ++kc) {
                float4 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[i] = *reinterpret_cast<const float4*>(a_rows + i * 32 * 128 + poff[kc]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[j] = *reinterpret_cast<const float4*>(w_rows + j * 32 * 128 + poff[kc]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(a[i].x, b[j].x, acc[i][j]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(a[i].y, b[j].y, acc[i][j]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(a[i].z, b[j].z, acc[i][j]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(a[i].w, b[j].w, acc[i][j]);
            }
            __syncthreads();  // with an LDS-DMA in flight hipcc puts s_waitcnt vmcnt(0) in front
        }
    }

    // ---- epilogue: lane holds column n (l31) of 16 rows per 32x32 fragment -------------------
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + l31;
            const int mbase = m0 + (wm * TM + i) * 32;
            if (n >= p.N) continue;
            float bn = 0.f;
            if constexpr (EPI != EPI_MOTION && EPI != EPI_TOKOUT && EPI != EPI_GELUGRAD &&
                          EPI != EPI_ACCUM) {
                if (p.bias) bn = p.bias[n];
            }
            int nb = 0, ntt = 0;
            float gs = 1.0f;
            if constexpr (EPI == EPI_MOTION) {
                nb = n / p.T;
                ntt = n - nb * p.T;
                if (p.gs_bits) gs = 1.0f / grad_scale_from_bits(*p.gs_bits);  // exact: power of two
            }
            if constexpr (EPI == EPI_TOKOUT) {
                if (p.gs_bits) gs = grad_scale_from_bits(*p.gs_bits);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mbase + mfma32_row(r, lane);
                if (m >= p.M) continue;
                const float v = acc[i][j][r];
                if constexpr (EPI == EPI_PLAIN) {
                    p.C[(size_t)m * p.ldc + n] = v + bn;
                } else if constexpr (EPI == EPI_GELU) {
                    const float pre = v + bn;
                    if (p.aux) p.aux[(size_t)m * p.ldc + n] = pre;
                    p.C[(size_t)m * p.ldc + n] = gelu_erf(pre);
                } else if constexpr (EPI == EPI_SILU) {
                    p.C[(size_t)m * p.ldc + n] = silu(v + bn);
                } else if constexpr (EPI == EPI_RESID) {
                    const float rr = PRE ? pre[PRE ? i : 0][PRE ? j : 0][r] : p.R[(size_t)m * p.ldc + n];
                    p.C[(size_t)m * p.ldc + n] = (v + bn) + rr;
                } else if constexpr (EPI == EPI_ACCUM) {
                    const float rr = PRE ? pre[PRE ? i : 0][PRE ? j : 0][r] : p.R[(size_t)m * p.ldc + n];
                    p.C[(size_t)m * p.ldc + n] = v + rr;
                } else if constexpr (EPI == EPI_GELUGRAD) {
                    const float ax = PRE ? pre[PRE ? i : 0][PRE ? j : 0][r] : p.aux[(size_t)m * p.ldc + n];
                    p.C[(size_t)m * p.ldc + n] = v * gelu_erf_grad(ax);
                } else if constexpr (EPI == EPI_INPROJ) {
                    const int b = m / p.T, t = m - b * p.T;
                    const float o = (v + bn) + p.pe[(size_t)(1 + t) * p.ldc + n];
                    p.C[((size_t)b * p.S + 1 + t) * p.ldc + n] = o;
                    if (p.Bdup) p.C[((size_t)(b + p.Bdup) * p.S + 1 + t) * p.ldc + n] = o;
                } else if constexpr (EPI == EPI_TOKOUT) {
                    const int b = m / p.T, t = m - b * p.T;
                    p.C[((size_t)b * p.S + 1 + t) * p.ldc + n] = v * gs;
                } else if constexpr (EPI == EPI_MOTION) {
                    // m = feature c, n = (b, t): lanes run along t -> 128-B contiguous stores
                    const float bm = p.bias ? p.bias[m] : 0.f;
                    p.C[((size_t)nb * p.Cf + m) * p.T + ntt] = (v * gs) * p.out_scale + bm;
                }
            }
        }
    }
}

}  // namespace cmdi
