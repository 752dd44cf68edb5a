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
    static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "wave tile must be 32-aligned");
    static_assert((BM * BK) % (4 * NT) == 0 && (BN * BK) % (4 * NT) == 0, "loader divisibility");
};

template <int MODE, int ROWS, int BK, int NT>
struct TileLoader {
    static constexpr int NV = ROWS * BK / NT;  // floats per thread
    float v[NV];

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
                const int row = row0 + r;
                float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < nrows) {
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

template <class TC, int AMODE, int BMODE, int EPI>
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
            if constexpr (EPI == EPI_MOTION) {
                nb = n / p.T;
                ntt = n - nb * p.T;
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
                    p.C[(size_t)m * p.ldc + n] = (v + bn) + p.R[(size_t)m * p.ldc + n];
                } else if constexpr (EPI == EPI_ACCUM) {
                    p.C[(size_t)m * p.ldc + n] = v + p.R[(size_t)m * p.ldc + n];
                } else if constexpr (EPI == EPI_GELUGRAD) {
                    p.C[(size_t)m * p.ldc + n] = v * gelu_erf_grad(p.aux[(size_t)m * p.ldc + n]);
                } else if constexpr (EPI == EPI_INPROJ) {
                    const int b = m / p.T, t = m - b * p.T;
                    const float o = (v + bn) + p.pe[(size_t)(1 + t) * p.ldc + n];
                    p.C[((size_t)b * p.S + 1 + t) * p.ldc + n] = o;
                    if (p.Bdup) p.C[((size_t)(b + p.Bdup) * p.S + 1 + t) * p.ldc + n] = o;
                } else if constexpr (EPI == EPI_TOKOUT) {
                    const int b = m / p.T, t = m - b * p.T;
                    p.C[((size_t)b * p.S + 1 + t) * p.ldc + n] = v;
                } else if constexpr (EPI == EPI_MOTION) {
                    // m = feature c, n = (b, t): lanes run along t -> 128-B contiguous stores
                    const float bm = p.bias ? p.bias[m] : 0.f;
                    p.C[((size_t)nb * p.Cf + m) * p.T + ntt] = v * p.out_scale + bm;
                }
            }
        }
    }
}

}  // namespace cmdi
