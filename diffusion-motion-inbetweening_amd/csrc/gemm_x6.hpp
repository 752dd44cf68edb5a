// fp32 "NT" GEMM with EXACT operands on the bf16 matrix pipe:  C = epi(A[M,K] · W[N,K]^T), fp32 accumulate.
//
// Why: gfx950's fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 rate and — measured — pulls the chip down
// to 1.76 GHz, so the exact-fp32 engine sits at its power-limited roofline (110 of 157 TFLOP/s).  A binary32 value has
// 24 significant bits = three bf16 mantissas:
//     x = b0 + b1 + b2,   b0 = bf16(x),  b1 = bf16(x - b0),  b2 = bf16(x - b0 - b1)        (EXACT for normal x; bf16 has
//                                                                                           fp32's exponent range, so
//                                                                                           there is no range limit)
// and a product of two such numbers is the sum of nine bf16 x bf16 partial products, each EXACT in the fp32 accumulator
// (16-bit product mantissa).  Six of them are kept,
//     acc0 += a0·w0          acc1 += a0·w1 + a1·w0 + a1·w1 + a0·w2 + a2·w0          C = acc0 + acc1
// the three dropped ones (a1·w2, a2·w1, a2·w2) are below 2^-26 of the product — a quarter of fp32's own rounding
// unit — so results are fp32-class WITHOUT operand truncation (unlike the 22-bit f16x3 mode, gemm_h3.hpp).  The small
// terms accumulate apart from the leading one, so their roundings stay 2^-8 below it.
// Six v_mfma_f32_32x32x16_bf16 per fp32 product: 2500 / 6 = 417 TFLOP/s fp32-equivalent peak, 2.65x the fp32 MFMA.
//
// Operands: A is plain fp32 [M][lda] — split into planes on the fly while a K step is staged through registers into LDS
// (one pass per block, VALU under the other wave's MFMAs); W is pre-split once at cmdi_finalize_weights into
// [N][K/32][3 planes][32] bf16 (192 contiguous bytes per row and K step).  LDS rows are 208 B (13 slots of 16 B: an odd
// slot count makes every ds_read_b128 lane group conflict-free).  Persistent grid: a block walks tiles
// blockIdx.x, blockIdx.x + gridDim.x, ... and its register-staged loads run one K step ahead ACROSS tile boundaries, so
// the next tile's first two K steps are in flight / in LDS while the epilogue stores.
#pragma once
#include "common.hpp"
#include "gemm_params.hpp"

namespace cmdi {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// x -> (b0, b1, b2), exact for every finite binary32 whose pieces do not underflow (|x| >= 2^-109)
__device__ __forceinline__ void split_bf16x3(float x, __bf16& b0, __bf16& b1, __bf16& b2) {
    b0 = (__bf16)x;                       // v_cvt_pk_bf16_f32: round to nearest even
    const float r1 = x - (float)b0;       // exact (Sterbenz-like: b0 carries the leading 8 bits of x)
    b1 = (__bf16)r1;
    const float r2 = r1 - (float)b1;      // exact
    b2 = (__bf16)r2;                      // at most 8 significant bits remain: exact
}

// the six partial products of one k-substep (16 columns) for one A fragment and TN W fragments; consecutive MFMAs
// alternate between the column fragments
template <int TN>
__device__ __forceinline__ void x6_products(const bf16x8 (&a)[3], const bf16x8 (&w)[TN][3], f32x16 (&acc0)[TN],
                                            f32x16 (&acc1)[TN]) {
#pragma unroll
    for (int j = 0; j < TN; ++j) acc0[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], w[j][0], acc0[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < TN; ++j) acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], w[j][1], acc1[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < TN; ++j) acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], w[j][0], acc1[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < TN; ++j) acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], w[j][1], acc1[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < TN; ++j) acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], w[j][2], acc1[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < TN; ++j) acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], w[j][0], acc1[j], 0, 0, 0);
}

struct X6Tile {
    static constexpr int BM = 128, BN = 128, BK = 32, WM = 4, WN = 2;
    static constexpr int NW = WM * WN, NT = 64 * NW;          // 8 waves, 32 x 64 outputs each
    static constexpr int TN = BN / WN / 32;                    // 2 column fragments per wave (TM = 1)
    static constexpr int ROWB = 3 * BK * 2 + 16;               // 208 bytes per LDS row: 3 planes x 32 bf16 + one pad slot
    static constexpr int STAGE = (BM + BN) * ROWB;             // 53,248 B
    static constexpr size_t LDS_BYTES = 2ull * STAGE;          // 104 KiB: one block per CU
};

// VAR 0: the compiler's own schedule, loads / stores of the staging pipeline behind their `if`s.
// VAR 1: branch-free K step (the tail re-stages a valid step nobody reads) in ONE scheduling region, pinned with
//        sched_group_barrier: the 9 fragment reads of k-substep 0 first, then one MFMA per remaining read, then the
//        split arithmetic of the next K step (VALU), its LDS writes and the global loads of the step after it woven
//        between the remaining MFMAs — the matrix pipe covers the staging work of the same wave.
// VAR 2: the K step is rotated so that its ONE barrier sits in the middle of the MFMA stream: first half = the products of
//        k-substep 0 (fragments read during the previous half) + this step's k-substep-1 reads + staging of step it+1;
//        barrier; second half = the products of k-substep 1 + the k-substep-0 reads of step it+1 (from the stage the
//        barrier just released) + the global loads of step it+2.  The matrix pipe has 12 MFMAs on either side of the
//        barrier, so neither the LDS latency after it nor the arrival skew before it is exposed.
// CONV (round 5, EPI_PLAIN / EPI_RESID): the A operand is a 1-D convolution's tap-shifted fp32 row matrix and the outputs
// follow the convolution's row rule (GemmParams taps / a_row_mul / c_row_mul / c_row_add / tp / t_lo / t_hi / C2) — the U-Net's
// convolutions with EXACT operands (no f16 range limit): only the A address of a K step and the epilogue's row map change.
template <int EPI, int VAR, bool CONV = false>
__global__ __launch_bounds__(X6Tile::NT, 2) void gemm_x6_kernel(const GemmParams p) {
    using TC = X6Tile;
    constexpr int BM = TC::BM, BN = TC::BN, BK = TC::BK, TN = TC::TN, ROWB = TC::ROWB, STAGE = TC::STAGE;
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave / TC::WN, wn = wave % TC::WN;

    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int n_tiles = tiles_m * tiles_n;
    const int nk = p.K / BK;
    const __bf16* Wx = reinterpret_cast<const __bf16*>(p.Wx);

    // staging roles: A — thread owns 8 consecutive k of one row (two float4 loads, three 16-B plane writes);
    //                W — thread owns three of the 1536 16-B slots of the [128 rows][12 slots] pre-split K step
    const int a_row = tid >> 2, a_c8 = tid & 3;
    float4 ra0, ra1;
    uint4 rw0, rw1, rw2;      // (named, not an array: an array captured by the lambdas below ends up in scratch)
    const int w_row0 = tid / 12, w_slot0 = tid - w_row0 * 12;                                   // items tid, tid + 512,
    const int w_row1 = (tid + TC::NT) / 12, w_slot1 = (tid + TC::NT) - w_row1 * 12;             // tid + 1024 of the
    const int w_row2 = (tid + 2 * TC::NT) / 12, w_slot2 = (tid + 2 * TC::NT) - w_row2 * 12;     // 128 x 12 slot grid
    auto tile_origin = [&](int t, int& m0, int& n0) {
        const int bid = xcd_remap(t, n_tiles);
        m0 = (bid / tiles_n) * BM;
        n0 = (bid % tiles_n) * BN;
    };
    auto load_step = [&](int m0, int n0, int kt) {
        int grow = m0 + a_row;
        grow = grow < p.M ? grow : p.M - 1;                    // clamp: rows past the end land in outputs nobody stores
        const float* src;
        if constexpr (CONV) {     // chunk-major K: the taps of a 32-channel chunk are consecutive K steps (as gemm_h3.hpp)
            const int taps = p.taps > 0 ? p.taps : 1;
            const int chunk = kt / taps, tap = kt - chunk * taps;
            src = p.A + ((size_t)grow * (p.a_row_mul ? p.a_row_mul : 1) + tap) * p.lda + chunk * BK + a_c8 * 8;
        } else {
            src = p.A + (size_t)grow * p.lda + kt * BK + a_c8 * 8;
        }
        ra0 = *reinterpret_cast<const float4*>(src);
        ra1 = *reinterpret_cast<const float4*>(src + 4);
        auto wload = [&](int row, int slot) {
            int wrow = n0 + row;
            wrow = wrow < p.N ? wrow : p.N - 1;
            return *reinterpret_cast<const uint4*>(Wx + ((size_t)wrow * nk + kt) * 96 + slot * 8);
        };
        rw0 = wload(w_row0, w_slot0);
        rw1 = wload(w_row1, w_slot1);
        rw2 = wload(w_row2, w_slot2);
    };
    auto store_step = [&](int buf) {
        char* st = lds + buf * STAGE;
        const float v[8] = {ra0.x, ra0.y, ra0.z, ra0.w, ra1.x, ra1.y, ra1.z, ra1.w};
        bf16x8 p0, p1, p2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            __bf16 b0, b1, b2;
            split_bf16x3(v[e], b0, b1, b2);
            p0[e] = b0; p1[e] = b1; p2[e] = b2;
        }
        char* arow = st + a_row * ROWB + a_c8 * 16;
        *reinterpret_cast<bf16x8*>(arow) = p0;
        *reinterpret_cast<bf16x8*>(arow + 64) = p1;
        *reinterpret_cast<bf16x8*>(arow + 128) = p2;
        *reinterpret_cast<uint4*>(st + (BM + w_row0) * ROWB + w_slot0 * 16) = rw0;
        *reinterpret_cast<uint4*>(st + (BM + w_row1) * ROWB + w_slot1 * 16) = rw1;
        *reinterpret_cast<uint4*>(st + (BM + w_row2) * ROWB + w_slot2 * 16) = rw2;
    };

    // this block's flattened (tile, K step) sequence
    const int my_tiles = (int)blockIdx.x < n_tiles ? (n_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    if (my_tiles == 0) return;
    const int total = my_tiles * nk;
    int ld_tile = blockIdx.x, ld_kt = 0, ld_m0, ld_n0;       // position of the NEXT step to load
    tile_origin(ld_tile, ld_m0, ld_n0);
    auto advance_load = [&]() {
        if (++ld_kt == nk) {
            ld_kt = 0;
            if (ld_tile + (int)gridDim.x < n_tiles) {      // past the last tile: stay (VAR 1 re-stages valid memory)
                ld_tile += gridDim.x;
                tile_origin(ld_tile, ld_m0, ld_n0);
            }
        }
    };

    load_step(ld_m0, ld_n0, 0);
    advance_load();
    store_step(0);
    if (total > 1) { load_step(ld_m0, ld_n0, ld_kt); advance_load(); }
    __syncthreads();

    const int frag_a = (wm * 32 + l31) * ROWB + hi * 16;                 // + plane * 64 + ks * 32
    const int frag_w = (BM + wn * (TN * 32) + l31) * ROWB + hi * 16;     // + j * 32 * ROWB + plane * 64 + ks * 32

    bf16x8 f0a[3], f0w[TN][3];          // VAR 2: k-substep-0 fragments of the step about to be multiplied
    if constexpr (VAR == 2) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            f0a[pl] = *reinterpret_cast<const bf16x8*>(lds + frag_a + pl * 64);
#pragma unroll
            for (int j = 0; j < TN; ++j) f0w[j][pl] = *reinterpret_cast<const bf16x8*>(lds + frag_w + j * 32 * ROWB + pl * 64);
        }
    }

    int it = 0;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        int m0, n0;
        tile_origin(tile, m0, n0);
        f32x16 acc0[TN], acc1[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[j][r] = 0.f; acc1[j][r] = 0.f; }

        if constexpr (VAR == 2) {
            for (int kt = 0; kt < nk; ++kt, ++it) {
                const char* st = lds + (it & 1) * STAGE;
                const char* sn = lds + ((it + 1) & 1) * STAGE;
                // ---- first half: products of k-substep 0 | reads of k-substep 1 | stage step it+1 -------------------
                bf16x8 a1[3], w1[TN][3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    a1[pl] = *reinterpret_cast<const bf16x8*>(st + frag_a + pl * 64 + 32);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        w1[j][pl] = *reinterpret_cast<const bf16x8*>(st + frag_w + j * 32 * ROWB + pl * 64 + 32);
                }
                x6_products<TN>(f0a, f0w, acc0, acc1);
                store_step((it + 1) & 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 3 + 3 * TN, 0);     // the 9 reads up front
#pragma unroll
                for (int q = 0; q < 6 * TN - 2; ++q) {                           // split arithmetic under the MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x200, 6, 0);               // stage it+1 -> LDS
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __syncthreads();
                // ---- second half: products of k-substep 1 | k-substep-0 reads of step it+1 | request step it+2 -----
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    f0a[pl] = *reinterpret_cast<const bf16x8*>(sn + frag_a + pl * 64);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        f0w[j][pl] = *reinterpret_cast<const bf16x8*>(sn + frag_w + j * 32 * ROWB + pl * 64);
                }
                load_step(ld_m0, ld_n0, ld_kt);
                advance_load();
                x6_products<TN>(a1, w1, acc0, acc1);
                __builtin_amdgcn_sched_group_barrier(0x100, 3 + 3 * TN, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 5, 0);               // global loads of step it+2
                __builtin_amdgcn_sched_group_barrier(0x008, 6 * TN - 2, 0);
            }
        } else
        for (int kt = 0; kt < nk; ++kt, ++it) {
            // step it+1 (in registers since the previous iteration) goes to the other stage, step it+2 is requested
            if constexpr (VAR == 0) {
                if (it + 1 < total) store_step((it + 1) & 1);
                if (it + 2 < total) { load_step(ld_m0, ld_n0, ld_kt); advance_load(); }
            }
            const char* st = lds + (it & 1) * STAGE;
            bf16x8 a[2][3], w[2][TN][3];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    a[ks][pl] = *reinterpret_cast<const bf16x8*>(st + frag_a + pl * 64 + ks * 32);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        w[ks][j][pl] = *reinterpret_cast<const bf16x8*>(st + frag_w + j * 32 * ROWB + pl * 64 + ks * 32);
                }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) x6_products<TN>(a[ks], w[ks], acc0, acc1);
            if constexpr (VAR == 1) {
                store_step((it + 1) & 1);
                load_step(ld_m0, ld_n0, ld_kt);
                advance_load();
                __builtin_amdgcn_sched_group_barrier(0x100, 3 + 3 * TN, 0);        // fragment reads of k-substep 0
#pragma unroll
                for (int q = 0; q < 3 + 3 * TN; ++q) {                              // k-substep 1's reads under MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#pragma unroll
                for (int q = 0; q < 9; ++q) {                                       // split arithmetic under MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x200, 6, 0);                  // stage it+1 -> LDS
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 5, 0);                  // request stage it+2
                __builtin_amdgcn_sched_group_barrier(0x008, 12 * TN - (3 + 3 * TN) - 9 - 2, 0);
            }
            __syncthreads();
        }

        // ---- epilogue.  The stage the last K step was multiplied from is idle (the next tile's first K step sits in the
        // other one, its second is in flight): each wave transposes its 32 x 64 accumulators through a private 4.5-KiB
        // slice of it, 32 columns at a time, and leaves as row-major float4 — every global instruction covers eight whole
        // 128-byte row segments.  Interior tiles take the predicate-free path (per-row branches would make hipcc drain
        // vmcnt(0) in front of every store).
        {
            constexpr int LD = 36;                                  // floats per transposed row: 9 slots of 16 B (odd)
            float* wl = reinterpret_cast<float*>(lds + ((it - 1) & 1) * STAGE) + wave * (32 * LD);
            const int rl = lane >> 3, c4 = (lane & 7) * 4;          // this lane's row (of 8 per instruction) and 4 columns
            const bool interior = !CONV && m0 + BM <= p.M && n0 + BN <= p.N;
            if constexpr (CONV) {
                // convolution rows: output row mo = m * c_row_mul + c_row_add, written only where its position inside the
                // tp-row frame is a valid frame (halo rows stay zero); C and / or C2, the residual R[mo][r_ld] added first
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        wl[mfma32_row(r, lane) * LD + l31] = acc0[j][r] + acc1[j][r];
                    const int n = n0 + (wn * TN + j) * 32 + c4;
                    const bool nok = n < p.N;
                    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (p.bias && nok) bias4 = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = q * 8 + rl;
                        const float4 t = *reinterpret_cast<const float4*>(wl + row * LD + c4);
                        const int m = m0 + wm * 32 + row;
                        if (m >= p.M || !nok) continue;
                        const int mo = p.c_row_mul ? m * p.c_row_mul + p.c_row_add : m;
                        if (p.tp) {
                            const int pos = mo % p.tp;
                            if (pos < p.t_lo || pos >= p.t_hi) continue;
                        }
                        float v[4] = {t.x + bias4.x, t.y + bias4.y, t.z + bias4.z, t.w + bias4.w};
                        if constexpr (EPI == EPI_RESID) {
                            const float4 x = *reinterpret_cast<const float4*>(p.R + (size_t)mo * (p.r_ld ? p.r_ld : p.ldc) + n);
                            v[0] += x.x; v[1] += x.y; v[2] += x.z; v[3] += x.w;
                        }
                        if (p.C) *reinterpret_cast<float4*>(p.C + (size_t)mo * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
                        if (p.C2) *reinterpret_cast<float4*>(p.C2 + (size_t)mo * p.ldc2 + n) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
                __syncthreads();
                continue;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    wl[mfma32_row(r, lane) * LD + l31] = acc0[j][r] + acc1[j][r];
                const int n = n0 + (wn * TN + j) * 32 + c4;
                const bool nok = interior || n < p.N;              // N % 4 == 0: a float4 is all in or all out
                float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (EPI != EPI_GELUGRAD && EPI != EPI_ACCUM) {
                    if (p.bias && nok) bias4 = *reinterpret_cast<const float4*>(p.bias + n);
                }
                float4 t[4], x[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = q * 8 + rl;
                    t[q] = *reinterpret_cast<const float4*>(wl + row * LD + c4);
                    int m = m0 + wm * 32 + row;
                    m = (interior || m < p.M) ? m : p.M - 1;
                    const size_t off = (size_t)m * p.ldc + (nok ? n : 0);
                    if constexpr (EPI == EPI_RESID || EPI == EPI_ACCUM) x[q] = *reinterpret_cast<const float4*>(p.R + off);
                    if constexpr (EPI == EPI_GELUGRAD) x[q] = *reinterpret_cast<const float4*>(p.aux + off);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = m0 + wm * 32 + q * 8 + rl;
                    if (!interior && (m >= p.M || !nok)) continue;
                    const size_t off = (size_t)m * p.ldc + n;
                    float v[4] = {t[q].x + bias4.x, t[q].y + bias4.y, t[q].z + bias4.z, t[q].w + bias4.w};
                    if constexpr (EPI == EPI_GELU) {
                        if (p.aux) *reinterpret_cast<float4*>(p.aux + off) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                    } else if constexpr (EPI == EPI_RESID || EPI == EPI_ACCUM) {
                        v[0] += x[q].x; v[1] += x[q].y; v[2] += x[q].z; v[3] += x[q].w;
                    } else if constexpr (EPI == EPI_GELUGRAD) {
                        v[0] *= gelu_erf_grad(x[q].x); v[1] *= gelu_erf_grad(x[q].y);
                        v[2] *= gelu_erf_grad(x[q].z); v[3] *= gelu_erf_grad(x[q].w);
                    }
                    *reinterpret_cast<float4*>(p.C + off) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
            __syncthreads();   // the next K step's staging writes this stage again
        }
    }
}

}  // namespace cmdi
