// Persistent, phase-alternating ("ping-pong") member of the split-f16 GEMM family:  C = epi(A[M,K] · W[N,K]^T), same operand
// format, same products, same per-output accumulation order — hence the same bits — as gemm_h3.hpp.
//
// Why a second kernel: gemm_h3's one-tile-per-block grid (128 x 128 tile, 8 waves of 32 x 64, two blocks per CU) delivers
// 730-850 TFLOP/s of executed f16 matrix work on the denoiser's shapes, while the matrix pipe alone sustains 1740 TFLOP/s on
// random f16 operands under the socket's power cap (tools/probes/mfma_power.hip) — the tile structure, not the pipe and not
// the cap, was the limit: every K step ends in "wait for my DMA, barrier", all eight waves read fragments at the same time and
// multiply at the same time, and prologue + epilogue are 20 % of a block's life.
//
// Structure (MI355X guide, "256² 8-phase template", adapted to split rows and to K = 512):
// * ONE block of 8 waves per CU (2 waves per SIMD, <= 256 registers each), tile 128 x 256, wave (wm, wn) of the 2 x 4 grid owns
//   64 rows x two 32-column fragments {wn 32, 128 + wn 32}: 64 x 64 outputs = 2 x 2 accumulator fragment pairs, 16 fragment
//   reads for 24 MFMAs per K step (gemm_h3: 12 for 12).
// * The block's waves form two groups (wm = 0 / 1: one wave of each group per SIMD) that run the SAME instruction stream one
//   barrier apart.  A K step of 32 columns is four barrier intervals per wave — [read the k-substep-0 fragments] | [12 MFMAs
//   with 3 DMA requests between them] | [read k-substep 1 + counted wait] | [12 MFMAs + 3 requests] — so in every interval one
//   wave of a SIMD multiplies while its partner reads LDS: the matrix pipe always has a wave in a product interval.  (First
//   version: requests in the reading intervals — there an LDS-DMA piece costs 100-185 cycles of issue, a reading interval
//   took twice a product interval, and the kernel ran at the speed of gemm_h3.)
// * LDS-DMA ring of THREE 48-KiB stages, requests two K steps ahead, one counted `s_waitcnt vmcnt(N)` per K step (never 0 in
//   the loop): a request has 4-8 intervals (1.5-3 k cycles) to land.
// * PERSISTENT: 256 blocks walk a work list; the request stream runs two K steps ahead ACROSS tile boundaries, so a tile's
//   first two K steps are in LDS before its first product, and the epilogue's stores drain under the next tile's products.
//   Work list per XCD (blocks id % 8 share an L2): the XCD's contiguous chunk of the N-fast tile order — so the few A row
//   panels being read at one time are shared by the CUs of one L2 and the whole W matrix (<= 3 MiB) stays resident — in rounds
//   of (blocks per XCD) tiles; a last partial round of at most half the blocks is cut into 128 x 128 HALF tiles (the same
//   wave grid with one column fragment per wave), which halves the tail.  No K splitting anywhere: an output's sum runs over
//   k in the same order whatever the tile, so results are bitwise those of gemm_h3 (and independent of M, i.e. of the batch).
// * Epilogue per wave through a private 4-KiB slice of the stage that was multiplied last, one 32 x 32 block at a time;
//   arithmetic copied from gemm_h3's interior path line by line.
//
// Supported: plain GEMM addressing (no convolution taps, no split-K), N % 256 == 0, K % 32 == 0, K >= 64, epilogues
// H3_PLAIN_SPLIT / H3_GELU_SPLIT / H3_GELUGRAD_SPLIT / H3_RESID / H3_PLAIN with every folded-LayerNorm option of H3Params.
#pragma once
#include <type_traits>

#include "gemm_h3.hpp"

namespace cmdi {

struct H3PTile {
    static constexpr int BM = 128, BN = 256, NW = 8, NT = 512;
    static constexpr int STAGE_ROWS = BM + BN;               // A rows, then W rows (row = one 128-B K-step line)
    static constexpr int STAGE = STAGE_ROWS * 128;            // 48 KiB
    static constexpr int NSTAGE = 3;
    static constexpr int STATS = BM * 8;                      // (mean, rstd) of the tile's rows; two buffers (tile parity)
    static constexpr size_t LDS_BYTES = (size_t)NSTAGE * STAGE + 2 * STATS;   // 149,504 B: one block per CU
};

template <int A, int B, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (A < B) {
        f(std::integral_constant<int, A>{});
        static_for<A + 1, B>(f);
    }
}

// ABL (probes library only, tools/h3p_ablate.py): 1 = no requests inside the loop, 2 = fragments from registers instead of LDS,
// 4 = no epilogue, 8 = cycle stamps (tools/h3p_timeline.py).  Compile-time: a runtime switch costs the loop 15-20 %.
// CONV (H3_PLAIN only, round 4): the A operand is a 1-D convolution's tap-shifted row matrix (H3Params a_ld / a_row_mul / cpt:
// K step kt reads chunk kt % cpt of the row `kt / cpt` frames further on — only the SCALAR offset of the A requests changes)
// and the output rows follow c_row_mul / c_row_add / tp / t_lo / t_hi (halo rows of a framed sequence are not written).
template <int EPI, int ABL = 0, bool CONV = false>
__global__ __launch_bounds__(H3PTile::NT, 2) void gemm_h3p_kernel(const H3Params p) {
    using TC = H3PTile;
    constexpr bool NO_DMA = (ABL & 1) != 0, NO_READ = (ABL & 2) != 0, NO_EPI = (ABL & 4) != 0, STAMPS = (ABL & 8) != 0;
    constexpr bool NO_WAIT = (ABL & 128) != 0;       // ablation: requests are never waited for (wrong results, timing only)
    constexpr bool HALF_DMA = (ABL & 256) != 0;      // ablation: only half of the pieces are requested
    constexpr bool NO_STORE = (ABL & 512) != 0;      // ablation: the epilogue computes but does not store
    constexpr bool TWO_INT = (ABL & 32) == 0;        // K step = TWO barrier intervals (ABL & 32: the first version's four, for A/B)        // structure variants under test: K step = two barrier intervals instead of four
    constexpr int TAILN = (ABL & 64) ? 4 : 2;        // MFMAs behind an interval's closing barrier
    constexpr int STAGE = TC::STAGE;
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;
    const int M = p.M;
    const int tiles_n = p.N / TC::BN, tiles_m = (M + TC::BM - 1) / TC::BM, n_tiles = tiles_m * tiles_n;
    const int nk = p.K / 32;

    // ---- this block's work list (see the header): items lb, lb + P, ... of its XCD's list ------------------------------
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, P = gridDim.x >> 3;
    const int t_begin = (int)((long)n_tiles * xcd / 8), n_x = (int)((long)n_tiles * (xcd + 1) / 8) - t_begin;
    const int n_round = (n_x / P) * P, rem = n_x - n_round;
    const bool halves = rem > 0 && 2 * rem <= P;
    const int n_items = n_round + (halves ? 2 * rem : rem);
    if (lb >= n_items) return;
    auto decode = [&](int item, int& m0, int& n0, int& tn) {
        int tile = t_begin + item, half = 0;
        tn = 2;
        if (halves && item >= n_round) {
            const int h = item - n_round;
            tile = t_begin + n_round + (h >> 1);
            half = h & 1;
            tn = 1;
        }
        const int mt = tile / tiles_n;
        // (block-uniform by construction; the integer division runs on the vector unit — say so, or everything derived from
        // it is kept in vector registers and branched on per lane)
        m0 = __builtin_amdgcn_readfirstlane(mt * TC::BM);
        n0 = __builtin_amdgcn_readfirstlane((tile - mt * tiles_n) * TC::BN + half * 128);
        tn = __builtin_amdgcn_readfirstlane(tn);
    };

    // ---- LDS-DMA requests.  A wave owns a CONTIGUOUS run of 48 of the stage's 384 rows: 6 lane-linear 1-KiB pieces of 8 rows,
    // bank swizzle on the SOURCE address (slot c of row r is stored at c ^ ((r >> 1) & 7), as gemm_h3).  Stage row r < 128 is
    // A row m0 + r, row 128 + c is W row n0 + c.  (A half tile reads W rows n0 .. n0 + 127 only; it still requests all 256 —
    // one request count for every K step — clamped to the matrix.)  Sources are 32-bit byte offsets from p.A / p.W.
    const int prow = lane >> 3, pslot = lane & 7;
    const unsigned ldk_b = 4u * (unsigned)p.K;                 // bytes per split row
    const unsigned lda_b = CONV && p.a_ld ? 2u * (unsigned)p.a_ld : ldk_b;                       // bytes per A row
    const unsigned a_rstep = CONV && p.a_row_mul ? lda_b * (unsigned)p.a_row_mul : lda_b;        // ... per output row
    [[maybe_unused]] const int cpt = CONV && p.cpt ? p.cpt : 0x40000000;
    [[maybe_unused]] const int ktaps = CONV && p.cpt && p.taps > 0 ? p.taps : 1;
    unsigned src_off[6];
    auto set_src = [&](int m0, int n0) {
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int r = wave * 48 + q * 8 + prow;
            const unsigned sw = (unsigned)((pslot ^ ((r >> 1) & 7)) << 4);
            if (r < TC::BM) {
                int grow = m0 + r;
                grow = grow < M ? grow : M - 1;
                if constexpr (CONV) {
                    if (p.rc_tv) { const int sq = grow / p.rc_tv; grow = sq * p.tp + p.t_lo + (grow - sq * p.rc_tv); }   // logical -> physical row
                }
                src_off[q] = (unsigned)grow * a_rstep + sw;
            } else {
                int c = n0 + r - TC::BM;
                c = c < p.N ? c : p.N - 1;
                src_off[q] = (unsigned)c * ldk_b + sw;
            }
        }
    };
    // pieces [Q0, Q1) of K step kt into stage s: `buffer_load_dwordx4 ... offen lds` — base in a buffer descriptor, the lane's
    // 32-bit offset in a register that only changes with the tile, the K step in the scalar offset: one scalar M0 write and
    // one request per piece, no vector arithmetic (a flat global_load_lds needs a 64-bit address per lane and piece)
    const auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.A), 0, 0x7fffffff, 0x00020000);
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.W), 0, 0x7fffffff, 0x00020000);
    int iss_aoff = 0, iss_tap = 0, iss_chunk = 0;           // CONV: scalar offset of the A requests of the next K step to request
    auto issue_range = [&](auto q0_c, auto q1_c, int kt, int s) {
        constexpr int Q0 = decltype(q0_c)::value, Q1 = decltype(q1_c)::value;
        const int a_soff = CONV ? __builtin_amdgcn_readfirstlane(iss_aoff) : kt * 128;
#pragma unroll
        for (int q = Q0; q < Q1; ++q) {
            const int r0 = wave * 48 + q * 8;                 // wave-uniform: the piece lies entirely in A or entirely in W
            auto dst = (__attribute__((address_space(3))) void*)(lds + s * STAGE + r0 * 128);
            if (r0 < TC::BM) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, dst, 16, (int)src_off[q], a_soff, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, dst, 16, (int)src_off[q], kt * 128, 0, 0);
        }
    };
    using Q0 = std::integral_constant<int, 0>;
    using Q3 = std::integral_constant<int, 3>;
    using Q6 = std::integral_constant<int, 6>;

    // ---- fragment addresses ------------------------------------------------------------------------------------------
    const int swz = (l31 >> 1) & 7;
    int off_hi[2], off_lo[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        off_hi[ks] = ((2 * ks + hi) ^ swz) * 16;
        off_lo[ks] = ((4 + 2 * ks + hi) ^ swz) * 16;
    }
    const int a_row = (wm * 64 + l31) * 128;                    // + i * 32 * 128
    const int w_row = (TC::BM + wn * 32 + l31) * 128;           // + j * 128 * 128

    // ---- folded LayerNorm: (mean, rstd) of a tile's rows, 16 rows per wave (same arithmetic, same bits as gemm_h3_body) --
    float2* stats0 = reinterpret_cast<float2*>(lds + TC::NSTAGE * STAGE);
    auto row_stats = [&](int m0, int n0, int parity) {
        if (p.ln_part && lane < 16) {
            const int row = wave * 16 + lane;
            int grow = m0 + row;
            grow = grow < M ? grow : M - 1;
            const float4* pp = reinterpret_cast<const float4*>(p.ln_part + (size_t)grow * 32);
            float mean_b[16], m2 = 0.f, mean = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 v = pp[q];
                mean_b[2 * q] = v.x * (1.0f / 32.0f); mean_b[2 * q + 1] = v.z * (1.0f / 32.0f);
                m2 += v.y + v.w;
                mean += v.x + v.z;
            }
            mean *= (1.0f / 512.0f);
#pragma unroll
            for (int q = 0; q < 16; ++q) { const float dq = mean_b[q] - mean; m2 = __builtin_fmaf(32.0f * dq, dq, m2); }
            const float2 ms = make_float2(mean, 1.0f / sqrtf(__builtin_fmaf(m2, 1.0f / 512.0f, 1e-5f)));
            stats0[parity * TC::BM + row] = ms;
            if (p.ln_stats && n0 == 0 && m0 + row < M) *reinterpret_cast<float2*>(p.ln_stats + 2 * (size_t)(m0 + row)) = ms;   // (as gemm_h3)
        }
    };

    // ---- request stream: position of the NEXT K step to request ------------------------------------------------------------
    int item = lb, m0, n0, tn;
    decode(item, m0, n0, tn);
    int iss_item = item, iss_kt = 0;
    int more = 1;                          // 0 once every K step of the block's last item has been requested
    set_src(m0, n0);
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };   // block-uniform bookkeeping stays on the scalar unit
    auto advance_issue = [&]() {
        iss_kt = uni(iss_kt + 1);
        if constexpr (CONV) {     // (tap, chunk) of K step kt, kept incrementally on the scalar unit
#if CMDI_CONV_KORDER
            // chunk-major (round 5, as gemm_h3.hpp): kt = chunk * taps + tap — the taps of a chunk are consecutive K steps
            iss_tap = uni(iss_tap + 1);
            if (iss_tap == ktaps) { iss_tap = 0; iss_chunk = uni(iss_chunk + 1); }
#else
            iss_chunk = uni(iss_chunk + 1);
            if (iss_chunk == cpt) { iss_chunk = 0; iss_tap = uni(iss_tap + 1); }
#endif
            if (iss_kt == nk) { iss_chunk = 0; iss_tap = 0; }
            iss_aoff = uni((int)((unsigned)iss_tap * lda_b) + iss_chunk * 128);
        }
        if (iss_kt == nk) {
            iss_kt = 0;
            iss_item = uni(iss_item + P);
            if (iss_item < n_items) {
                int am, an, at;
                decode(iss_item, am, an, at);
                set_src(am, an);
            } else {
                more = 0;
            }
        }
    };

    // ---- prologue: statistics of the first tile, K steps 0 and 1 requested, K step 0 landed --------------------------------
    row_stats(m0, n0, 0);
    issue_range(Q0{}, Q6{}, iss_kt, 0);
    advance_issue();
    issue_range(Q0{}, Q6{}, iss_kt, 1);       // (nk >= 2: the same item)
    advance_issue();
    wait_vmcnt<6>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (wm == 1) __builtin_amdgcn_s_barrier();       // the second group runs one barrier behind the first
    __builtin_amdgcn_sched_barrier(0);

    int s_cur = 0, s_iss = 2;                          // stage being multiplied / stage being requested into
    int parity = 0;
    bool overflow = false;

    f32x16 acc0[2][2], acc1[2][2];
    [[maybe_unused]] long long ts[15] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};     // cycle stamps of K step 8 of the block's first tile
    [[maybe_unused]] int probe_kt = -1;
#define H3P_STAMP(i) do { if constexpr (STAMPS) { if (probe_kt == 8) ts[i] = __builtin_readcyclecounter(); } } while (0)
    [[maybe_unused]] h8 keep_h[8];          // (ABL & 2 only) register-resident stand-ins for the fragments
    if constexpr (NO_READ) {
#pragma unroll
        for (int q = 0; q < 8; ++q) keep_h[q] = *reinterpret_cast<const h8*>(lds + (lane * 8 + q) * 16);
    }
    // One K step of the tile kind TN (2 = full, 1 = half): four barrier intervals, see the header.
    auto kstep = [&](auto tn_c, bool first) {
        constexpr int TN = decltype(tn_c)::value;
        constexpr int NMM = 6 * TN;                     // MFMAs per product interval
        const char* st = lds + s_cur * STAGE;
        const int kt_req = uni(iss_kt);
        const bool req = uni(more) != 0 && !NO_DMA;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // ---- reading interval: this k-substep's fragments (the partner wave of this SIMD is multiplying) ----------------
            H3P_STAMP(ks * 5 + 0);
            h8 ah[2], al[2], wh[TN], wl[TN];
            if constexpr (!NO_READ) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ah[i] = *reinterpret_cast<const h8*>(st + a_row + i * 4096 + off_hi[ks]);
                    al[i] = *reinterpret_cast<const h8*>(st + a_row + i * 4096 + off_lo[ks]);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    wh[j] = *reinterpret_cast<const h8*>(st + w_row + j * 16384 + off_hi[ks]);
                    wl[j] = *reinterpret_cast<const h8*>(st + w_row + j * 16384 + off_lo[ks]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) { ah[i] = keep_h[i]; al[i] = keep_h[2 + i]; }
#pragma unroll
                for (int j = 0; j < TN; ++j) { wh[j] = keep_h[4 + j]; wl[j] = keep_h[6 + j]; }
            }
            if (ks == 1) {
                // this wave's pieces of K step (current + 1) have landed; the three requested in this K step's first reading
                // interval may stay in flight (a tile's first K step requests nothing there, see below)
                // (a later tile's first K step waits for nothing: the pieces it is about to read landed before the previous
                // epilogue's first store — that epilogue opens with vmcnt(0) — and a vmcnt(0) here would wait for the write
                // latency of the whole epilogue: ~9 k cycles per tile, measured)
                if constexpr (!NO_WAIT) {
                    if (first) { if (item == lb) wait_vmcnt<0>(); }
                    else if (req) wait_vmcnt<3>();
                    else wait_vmcnt<0>();
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // requests of K step (current + 2), half per reading interval, AFTER the fragments have arrived: an LDS-DMA piece
            // costs 25-60 cycles of issue here, 100-185 with fragment reads in flight (first version of this kernel) and ~50
            // between MFMAs (second version: the product interval grew from 384 to 470-520 cycles) — and here the wave would
            // otherwise sit at the barrier until its partner's products are through.  A tile's first K step requests nothing
            // in its first interval: the stage it requests into is the other group's epilogue scratch until then.
            if (ks == 0) {
                if (req && !first) issue_range(Q0{}, Q3{}, kt_req, s_iss);
            } else if (req) {
                if (first) issue_range(Q0{}, Q3{}, kt_req, s_iss);
                if constexpr (!HALF_DMA) issue_range(Q3{}, Q6{}, kt_req, s_iss);
            }
            H3P_STAMP(ks * 5 + 1);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            H3P_STAMP(ks * 5 + 2);
            // ---- product interval: 6 TN MFMAs.  Its closing barrier sits in front of the last two: the partner, released there,
            // starts its products while this wave's tail still occupies the pipe — a barrier's release latency (~80 cycles
            // measured) is not exposed.  (Barriers order LDS traffic only; MFMAs work on registers and may sit on either side.)
            __builtin_amdgcn_s_setprio(1);
            auto mm = [&](auto idx_c) {
                constexpr int idx = decltype(idx_c)::value;
                constexpr int kind = idx / (2 * TN), i = (idx % (2 * TN)) / TN, j = idx % TN;
                if constexpr (kind == 0) acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wh[j], acc0[i][j], 0, 0, 0);
                if constexpr (kind == 1) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wl[j], acc1[i][j], 0, 0, 0);
                if constexpr (kind == 2) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], wh[j], acc1[i][j], 0, 0, 0);
            };
            constexpr int TAIL = NMM - (TN == 2 ? TAILN : 2);
            static_for<0, TAIL>([&](auto ic) { mm(ic); });
            H3P_STAMP(ks * 5 + 3);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            H3P_STAMP(ks * 5 + 4);
            static_for<TAIL, NMM>([&](auto ic) { mm(ic); });
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
        advance_issue();
        s_cur = uni(s_cur == 2 ? 0 : s_cur + 1);
        s_iss = uni(s_iss == 2 ? 0 : s_iss + 1);
    };

    // Variant under test (ABL & 32): a K step as TWO barrier intervals — [all 16 fragment reads + counted wait + the 6 requests]
    // | [24 MFMAs] — half the barriers per product, 64 fragment registers instead of 32.  A tile's first K step (whose request
    // target is the other group's epilogue scratch during its reading interval) issues its requests between its MFMAs instead.
    auto kstep2 = [&](auto tn_c, bool first) {
        constexpr int TN = decltype(tn_c)::value;
        constexpr int NMM = 12 * TN;
        const char* st = lds + s_cur * STAGE;
        const int kt_req = uni(iss_kt);
        const bool req = uni(more) != 0 && !NO_DMA;
        h8 ah[2][2], al[2][2], wh[2][TN], wl[2][TN];
        if constexpr (!NO_READ) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ah[ks][i] = *reinterpret_cast<const h8*>(st + a_row + i * 4096 + off_hi[ks]);
                    al[ks][i] = *reinterpret_cast<const h8*>(st + a_row + i * 4096 + off_lo[ks]);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    wh[ks][j] = *reinterpret_cast<const h8*>(st + w_row + j * 16384 + off_hi[ks]);
                    wl[ks][j] = *reinterpret_cast<const h8*>(st + w_row + j * 16384 + off_lo[ks]);
                }
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < 2; ++i) { ah[ks][i] = keep_h[i]; al[ks][i] = keep_h[2 + i]; }
#pragma unroll
                for (int j = 0; j < TN; ++j) { wh[ks][j] = keep_h[4 + j]; wl[ks][j] = keep_h[6 + j]; }
            }
        }
        // this wave's pieces of K step (current + 1) have landed: nothing younger is outstanding yet (a tile's first K step: they
        // landed before the epilogue, which drains the queue before its first store; only its stores may be in flight)
        if (!first || item == lb) wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (req && !first) issue_range(Q0{}, Q6{}, kt_req, s_iss);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        auto mm = [&](auto idx_c) {
            constexpr int idx = decltype(idx_c)::value;
            constexpr int ks = idx / (6 * TN), r = idx % (6 * TN);
            constexpr int kind = r / (2 * TN), i = (r % (2 * TN)) / TN, j = r % TN;
            if constexpr (kind == 0) acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks][i], wh[ks][j], acc0[i][j], 0, 0, 0);
            if constexpr (kind == 1) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks][i], wl[ks][j], acc1[i][j], 0, 0, 0);
            if constexpr (kind == 2) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks][i], wh[ks][j], acc1[i][j], 0, 0, 0);
        };
        constexpr int TAIL = NMM - (TN == 2 ? TAILN : 2);
        if (req && first) {
            // (once per tile) the six requests between the first MFMAs
            static_for<0, 6>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                mm(std::integral_constant<int, q>{});
                __builtin_amdgcn_sched_barrier(0);
                issue_range(std::integral_constant<int, q>{}, std::integral_constant<int, q + 1>{}, kt_req, s_iss);
                __builtin_amdgcn_sched_barrier(0);
            });
            static_for<6, TAIL>([&](auto ic) { mm(ic); });
        } else {
            static_for<0, TAIL>([&](auto ic) { mm(ic); });
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        static_for<TAIL, NMM>([&](auto ic) { mm(ic); });
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        advance_issue();
        s_cur = uni(s_cur == 2 ? 0 : s_cur + 1);
        s_iss = uni(s_iss == 2 ? 0 : s_iss + 1);
    };
    auto kstep_any = [&](auto tn_c, bool first) {
        if constexpr (TWO_INT) kstep2(tn_c, first); else kstep(tn_c, first);
    };

    auto epilogue = [&](auto tn_c, auto edge_c) {
        constexpr int TN = decltype(tn_c)::value;
        constexpr bool EDGE = decltype(edge_c)::value;
        // scratch: the stage multiplied last (s_cur has already moved on: it is the one before, = the next request target)
        float* wl = reinterpret_cast<float*>(lds + s_iss * STAGE) + wave * 1024;
        const float2* rs = p.ln_part ? stats0 + parity * TC::BM : nullptr;
        const int cl = (lane & 3) * 8;
        // acc0 + acc1 * 2^-11 first, in place (an exact scaling and one rounding, as in gemm_h3): 64 registers come free for the
        // operand rows below
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc0[i][j][r] = acc0[i][j][r] + acc1[i][j][r] * kLoInv;
        wait_vmcnt<0>();      // every request issued so far has landed (the next tile's first K steps rely on it, see kstep2)
        H3PCols cols[TN];
        H3PRows rows[2][TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) cols[j] = h3p_load_cols<EPI>(p, n0 + j * 128 + wn * 32 + cl);
        if constexpr (EPI == H3_RESID || EPI == H3_GELUGRAD_SPLIT) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    h3p_load_rows<EPI, EDGE>(p, m0 + wm * 64 + i * 32, n0 + j * 128 + wn * 32 + cl, lane, rows[i][j]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                h3p_epi_block<EPI, EDGE, NO_STORE, false, CONV>(p, acc0[i][j], acc0[i][j], m0 + wm * 64 + i * 32, n0 + j * 128 + wn * 32,
                                         rs ? rs + wm * 64 + i * 32 : nullptr, wl, lane, cols[j], rows[i][j], overflow);
    };

    // Around a tile's epilogue the two groups fall into step: the first group waits one barrier for the second group's last
    // products (the pipe is busy with them), then BOTH run their epilogues in the same interval, and the second group takes one
    // extra barrier afterwards, which puts it one interval behind again.  (Without this the epilogues ran one after the other,
    // each under a partner that had nothing left to multiply: 16 k idle cycles per tile, measured.)
    auto epi_enter = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        if (wm == 0) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto epi_leave = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        if (wm == 1) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    for (;;) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }
        const bool edge = m0 + TC::BM > M;
        const int next = item + P;
        int nm0 = 0, nn0 = 0, ntn = 2;
        if (next < n_items) decode(next, nm0, nn0, ntn);
        if (tn == 2) {
            const bool probing = STAMPS && item == lb;
            if constexpr (STAMPS) { if (probing) ts[10] = __builtin_readcyclecounter(); }
            for (int kt = 0; kt < nk; ++kt) {
                if constexpr (STAMPS) probe_kt = probing ? kt : -1;
                kstep_any(std::integral_constant<int, 2>{}, kt == 0);
            }
            if constexpr (STAMPS) { probe_kt = -1; if (probing) ts[11] = __builtin_readcyclecounter(); }
            if (next < n_items) row_stats(nm0, nn0, parity ^ 1);
            if constexpr (STAMPS) { if (probing) ts[12] = __builtin_readcyclecounter(); }
            epi_enter();
            if (NO_EPI && acc0[0][0][0] != 12345.678f) { /* ablation: no epilogue */ }
            else if (edge) epilogue(std::integral_constant<int, 2>{}, std::true_type{});
            else epilogue(std::integral_constant<int, 2>{}, std::false_type{});
            if constexpr (STAMPS) { if (probing) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ts[13] = __builtin_readcyclecounter(); } }
            epi_leave();
        } else {
            for (int kt = 0; kt < nk; ++kt) kstep_any(std::integral_constant<int, 1>{}, kt == 0);
            if (next < n_items) row_stats(nm0, nn0, parity ^ 1);
            epi_enter();
            if (edge) epilogue(std::integral_constant<int, 1>{}, std::true_type{});
            else epilogue(std::integral_constant<int, 1>{}, std::false_type{});
            epi_leave();
        }
        if (next >= n_items) break;
        item = uni(next); m0 = nm0; n0 = nn0; tn = ntn;
        parity = uni(parity ^ 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (wm == 0) __builtin_amdgcn_s_barrier();       // pairs with the second group's last barrier
    if constexpr (EPI != H3_PLAIN) {
        if (overflow && p.range_flag) atomicOr(p.range_flag, 1);
    }
    if constexpr (STAMPS) {
      if (p.dbg_buf && lane == 0) {
        long long* o = p.dbg_buf + ((size_t)blockIdx.x * 8 + wave) * 16;
        const long long end = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < 14; ++i) o[i] = ts[i];
        o[14] = end;
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        o[15] = hwid;
      }
    }
#undef H3P_STAMP
}

}  // namespace cmdi
