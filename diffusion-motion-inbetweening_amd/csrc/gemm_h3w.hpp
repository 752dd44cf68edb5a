// Weight-stationary form of the split-f16 NT GEMM for K = 512 (round 6; VERDICT r5 task 2):  C = epi(A[M,512] · W[N,512]^T).
//
// gemm_h3.hpp stages BOTH operands of every K step through LDS and reads one fragment per MFMA; its K step is bound by the CU's
// one LDS pipe (64 KiB of LDS-DMA writes + 192 KiB of fragment reads against 1,536 cycles of MFMA issue).  Here the dataflow
// is turned round: a wave OWNS 32 output columns for the whole K — its W fragments (hi and lo plane, 32 k16-steps each) are
// 64 x 128 bit = 256 registers per lane, the entire ACCUMULATION file, loaded once per pass — and only A streams through LDS:
//     block  = 4 waves, ONE per SIMD (512 registers each), a 128-column strip of W resident in the CU's register files
//     A      = 32-row tiles (32 x 2 KiB = 64 KiB: both planes, all 16 chunks) in a ring of two LDS slots, filled by LDS-DMA a
//              tile ahead; per tile a wave reads 64 fragments for 96 MFMAs (2 per 3 instead of 1 per 1), the CU moves
//              64 KiB in + 256 KiB out of LDS per 3,072 cycles of MFMA issue (the tiled kernel: 256 KiB per 1,536)
//     work   = XCD x owns the row tiles [T x / 8, T (x + 1) / 8); its blocks split them into R sub-ranges x the N / 128 strips,
//              in P passes when there are more (strip, sub-range) units than blocks (in_proj: 12 strips x 8 sub-ranges on 32
//              blocks = 3 passes, W reloaded per pass) — rows per block, not tiles per slot: no launch quantisation beyond
//              one 32-row tile, and the blocks that share A rows sit on ONE XCD (its L2 serves the re-reads)
//     W      = read from a fragment-ordered copy (pack_w_h3w_kernel, made once at cmdi_finalize_weights): every load instruction
//              of a wave is one contiguous KiB
// Products and their order per output are those of every gemm_h3 tile (k ascending in steps of 16; acc0 += a_hi w_hi;
// acc1 += a_hi w_lo, then a_lo w_hi) and the epilogue arithmetic is h3_epilogue's: the same bits
// (test_gemm_h3w_is_bitwise_the_tiled_kernel).
//
// The instruction stream of a tile is hand-issued (inline asm).  hipcc left alone reads every A fragment just in time behind
// lgkmcnt(0), copies the AGPR-resident W fragments through a scratch quadruple (4 v_accvgpr_mov per MFMA) and breaks the stream
// into basic blocks around the requests.  Here: W fragments are "a"-constrained operands (MFMA B operands straight from the
// accumulation registers), A fragments are read two k16-steps ahead into a ring of three register sets under counted
// lgkmcnt, the next tile's 16 LDS-DMA pieces ride behind the first 16 steps, and the first tile of a pass starts under the W
// loads still in flight (counted vmcnt per step).  Measured (profiles/r06_h3w_*): the stream runs at 3,190 cycles per tile
// (floor 3,072), an epilogue run BEHIND it costs another 2,200-3,200 with the matrix pipe idle — one wave per SIMD has nobody
// to cover it — so the epilogue of an interior tile is DEFERRED and cut into slices that ride in the MFMA gaps of the next
// tile's stream (H3WDefer).
#pragma once
#include <type_traits>

#include "gemm_h3.hpp"

namespace cmdi {

typedef int i32x4 __attribute__((ext_vector_type(4)));

// Compile-time switches of experiment builds (tools/variant_build.sh; 0 in every shipped library): 1 = no requests inside the
// stream (results wrong), 16 = cycle stamps inside the stream of a block's third tile, 32 = no deferred epilogue
#ifndef H3W_ABL
#define H3W_ABL 0
#endif

// tile descriptor in the vocabulary of h3_epilogue (gemm_h3.hpp): 32 x 128 block tile, wave w = columns [32 w, 32 w + 32)
struct H3WTile {
    static constexpr int BM = 32, BN = 128, WM = 1, WN = 4, TM = 1, TN = 1, NW = 4, NT = 256, EPI8 = 0;
    static constexpr int SLOT = 32 * 2048;                       // one A tile: 32 rows x (512 columns x 2 planes x 2 B)
    static constexpr int EPI_SCRATCH = 4 * 32 * 32 * 4;          // per-wave transpose slices of the epilogue
    // h3_epilogue reads (mean, rstd) of the tile's rows at lds + MAIN_BYTES; the kernel passes lds = scratch + par3 * 256, so the
    // statistics live in THREE buffers: tile t's are read by its deferred epilogue during the stream of tile t + 1, while wave 0
    // already writes tile t + 2's
    static constexpr size_t MAIN_BYTES = EPI_SCRATCH + 2 * 256;
    static constexpr size_t SCRATCH_BYTES = MAIN_BYTES + 3 * 256;
    // folded LayerNorm: the 16 partial statistics of a tile's 32 rows (128 B per row) arrive by LDS-DMA with the tile's A rows —
    // two buffers — and wave 0 turns them into (mean, rstd) behind the barrier that makes them visible (no registers held across
    // a stream, no global-load latency on the barrier path)
    static constexpr int PART = 32 * 128;
    static constexpr size_t LDS_BYTES = 2 * (size_t)SLOT + SCRATCH_BYTES + 2 * (size_t)PART;
};

// (mean, rstd) of one row of the folded LayerNorm from its 16 partial statistics — the arithmetic of gemm_h3_body, explicit fmaf
__device__ __forceinline__ float2 h3_row_stats_from_partials(const float4* pp) {
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
    return make_float2(mean, 1.0f / sqrtf(__builtin_fmaf(m2, 1.0f / 512.0f, 1e-5f)));
}

// one LDS-DMA piece: 8 rows x 128 B (one chunk) -> LDS at M0 + lane * 16.  M0 is written here (s_nop: one wait state between a
// scalar write of M0 and the LDS-DMA that reads it — the hazard pass does not look into asm).  hipcc keeps no value in M0 across
// statements (a reserved register, refused in clobber lists; gfx9 LDS instructions do not read it), and nothing it generates
// for this kernel writes it (tools/audit_asm_loads.py --m0 checks the compiled ISA): a stream sets M0 for its first piece and
// ADVANCES it for the following ones (h3w_dma_next) — sixteen destinations per tile cost one scalar add each instead of a
// register each (under this kernel's scalar-register pressure hipcc parked them in vector registers, one v_readfirstlane per piece).
__device__ __forceinline__ void h3w_dma(i32x4 rsrc, unsigned voff, int soff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(__builtin_amdgcn_readfirstlane((int)lds_dst)), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}
// the next piece of a sequence: LDS destination + ADVANCE, source + SRC bytes; `soff` is the running scalar offset (one register
// for the whole sequence: sixteen "s" constants would be hoisted out of the tile loop and kept)
template <int ADVANCE, int SRC = 128>
__device__ __forceinline__ void h3w_dma_next(i32x4 rsrc, unsigned voff, int& soff) {
    asm volatile("s_add_u32 m0, m0, %3\n\ts_add_u32 %0, %0, %4\n\tbuffer_load_dwordx4 %1, %2, %0 offen lds"
                 : "+s"(soff) : "v"(voff), "s"(rsrc), "n"(ADVANCE), "n"(SRC) : "memory", "scc");
}

// ---- the deferred epilogue ------------------------------------------------------------------------------------------------
// h3_epilogue's interior ("fast") path for the epilogues without operand loads — bias / folded-LayerNorm correction, optional
// pre-activation stash, GELU, hi / lo split, range check, whole-line split stores — as 40 slices: 16 x one accumulator pair into
// the wave's transpose scratch, 4 x one row batch read back, 4 x (4 element slices + 1 store slice).  slice<G>() is called
// behind MFMA number G (0..95) of the NEXT tile's stream; the slice table below says which gaps carry work.  Same expressions
// as h3_epilogue in the same order per value: the same bits.
struct H3WNoFill {
    static constexpr int kStoresBehindRequests = 0;
    static constexpr int lds_ops(int) { return 0; }
    template <int G> __device__ __forceinline__ void slice() {}
};
// a deferred epilogue bound to the accumulator set of the tile it finishes
template <class D>
struct H3WFill {
    static constexpr int kStoresBehindRequests = D::kStoresBehindRequests;
    static constexpr int lds_ops(int G) { return D::lds_ops(G); }
    D& d;
    const f32x16& a0;
    const f32x16& a1;
    template <int G> __device__ __forceinline__ void slice() { d.template slice<G>(a0, a1); }
};

template <int EPI>
struct H3WDefer {
    static constexpr bool kSupported = (EPI == H3_PLAIN_SPLIT || EPI == H3_GELU_SPLIT) && !(H3W_ABL & 32);
    // vector-memory operations this epilogue is guaranteed to issue behind the stream's last request (step 16 = gap 50): the two
    // split stores of iterations 1, 2, 3 (gaps >= 54); stash stores, if any, only add to them
    static constexpr int kStoresBehindRequests = 6;
    // LDS operations of the slice behind MFMA G (one transpose store; two read-backs): the stream's counted fragment waits add
    // them — without that a wait "all but the newest two" lands on this step's OWN fragment reads as soon as a slice has put
    // an LDS operation behind them, and every step of the transpose phase paid an LDS round trip (+ 530 cycles per tile)
    static constexpr int lds_ops(int G) { return G < 16 ? 1 : (G >= 18 && G < 22 ? 2 : 0); }
    const H3Params& p;
    const int lane, wave;
    // per pass (lane constants of the strip)
    float4 bias4, c14;
    int n;
    bool fold, has_aux;        // folded LayerNorm on the A operand (ln_part + ln_c1); pre-activation stash wanted
    // per tile (the accumulators stay where the stream left them: the kernel alternates between two sets, H3WFill binds one)
    float* wl;
    const float2* row_stats;
    char* dst0;                // split output of this lane's first row (row m0 + lane / 8) at its four columns
    float* aux0;
    unsigned dst_step, aux_step;    // bytes per 8 rows
    float4 tt[4];
    float2 rst[4];
    float v[4];
    h4 oh, ol;
    unsigned ovf;              // max over every value split so far of (bits << 1): >= (bits of 65504.f) << 1 <=> !(|x| < 65504), NaN included
    unsigned lane_dst;         // per pass: byte offset of this lane's (row lane / 8, its four columns) inside a 32-row output block
    unsigned ld_bytes;         // row pitch of the split output (launch_gemm_h3w refuses pitches beyond 32 bits)
    long long head_base;       // cs_head_major: byte offset of this lane's (q|k|v, head) block

    __device__ __forceinline__ H3WDefer(const H3Params& p_, int lane_, int wave_) : p(p_), lane(lane_), wave(wave_), ovf(0) {}

    // Lane constants of the pass at strip n0.  prefetch_pass() runs IN FRONT of the pass's first stream — in which the epilogue of
    // the previous pass's last tile may still ride on the old constants — and only issues the loads; commit_pass() takes them over
    // BEHIND that stream (whose closing wait has covered the loads: no compiler wait for them lands inside a stream, where it
    // would also wait for every hand-issued request in flight).  Without folded LayerNorm the slices run the SAME formula on
    // c1 = 0, (mean, rstd) = (0, 1): fma(1, fma(-0, 0, t), b) = t + b exactly, no select per value.
    float4 nbias4, nc14;
    int nn;
    __device__ __forceinline__ void prefetch_pass(int n0) {
        nn = n0 + wave * 32 + (lane & 7) * 4;
        nbias4 = p.bias ? *reinterpret_cast<const float4*>(p.bias + nn) : make_float4(0.f, 0.f, 0.f, 0.f);
        nc14 = p.ln_c1 ? *reinterpret_cast<const float4*>(p.ln_c1 + nn) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __device__ __forceinline__ void commit_pass() {
        n = nn; bias4 = nbias4; c14 = nc14;
        fold = p.ln_c1 != nullptr && p.ln_part != nullptr;
        has_aux = p.aux != nullptr;
        // output addressing of the pass: row pitch and this lane's offset inside a 32-row block (a tile then adds ONE wave-uniform
        // product m0 * pitch — scalar unit — and a store slice adds it * 8 rows)
        const int rl = lane >> 3;
        ld_bytes = 2u * (unsigned)(p.cs_ld ? p.cs_ld : 2 * p.N);
        lane_dst = (unsigned)rl * ld_bytes + 2u * (unsigned)split_pos(n);
        head_base = 0;
        if constexpr (EPI == H3_PLAIN_SPLIT) {
            if (p.cs_head_major) { ld_bytes = 512; lane_dst = (unsigned)rl * 512u + 2u * (unsigned)split_pos(n & 127); head_base = (long long)(n >> 7) * p.M * 512; }
        }
        dst_step = 8 * ld_bytes;
        aux_step = 8u * (unsigned)p.ldc * 4u;
    }
    __device__ __forceinline__ void begin_tile(int m0, char* lds_epi) {
        wl = reinterpret_cast<float*>(lds_epi) + wave * (32 * 32);
        row_stats = reinterpret_cast<const float2*>(lds_epi + H3WTile::MAIN_BYTES);
        const long long row0 = (long long)__builtin_amdgcn_readfirstlane(m0);
        dst0 = reinterpret_cast<char*>(p.Cs) + (row0 * (long long)ld_bytes + head_base + lane_dst);
        aux0 = has_aux ? p.aux + ((row0 + (lane >> 3)) * p.ldc + n) : nullptr;
    }
    __device__ __forceinline__ void finish() {
        if (ovf >= (0x477fe000u << 1) && p.range_flag) atomicOr(p.range_flag, 1);
        ovf = 0;
    }

    // gap -> slice.  Transpose r behind MFMA r (0..15); row batch it read back behind MFMA 18 + it; iteration it owns the gaps
    // B = 24 + 18 it .. B + 17: the row's four values at B, the stash store at B + 1, GELU in six stages on the four values at
    // once (four independent chains per gap: a lone wave issues a dependent VALU chain at about 8 cycles per instruction, so one
    // element's chain alone in a gap ran past the MFMA it was meant to hide under) at B + 2 .. B + 7, the hi / lo split as two
    // element pairs x two halves at B + 8 .. B + 11, the stores at B + 12.  Every slice ends with its results pinned (empty asm):
    // the compiler may not slide work past the next MFMA statement.
    float gu[4], gt[4], gp[4];
    // gaps of an iteration (relative to its first): the split's four half-slices and the stores — packed behind GELU's six stages,
    // spread evenly without them
    static constexpr int kSplit0 = EPI == H3_GELU_SPLIT ? 8 : 3, kSplitStep = EPI == H3_GELU_SPLIT ? 1 : 3;
    static constexpr int kStore = kSplit0 + 4 * kSplitStep;
    template <int G> __device__ __forceinline__ void slice(const f32x16& a0, const f32x16& a1) {
        [[maybe_unused]] const int l31 = lane & 31, rl = lane >> 3, cl = (lane & 7) * 4;
        if constexpr (G < 16) {
            constexpr int r = G;
            wl[mfma32_row(r, lane) * 32 + l31] = a0[r] + a1[r] * kLoInv;
        } else if constexpr (G >= 18 && G < 22) {
            constexpr int it = G - 18;
            tt[it] = *reinterpret_cast<const float4*>(wl + (it * 8 + rl) * 32 + cl);
            const float2 rs = row_stats[rl + it * 8];
            rst[it] = make_float2(fold ? rs.x : 0.f, fold ? rs.y : 1.f);
        } else if constexpr (G >= 24) {
            constexpr int it = (G - 24) / 18, m = (G - 24) % 18;
            const float4 t = tt[it];
            const float2 rs2 = rst[it];
            if constexpr (m == 0) {
                // the four values of the row: the folded LayerNorm's correction (gemm_params.hpp), = bias add without fold
                v[0] = __builtin_fmaf(rs2.y, __builtin_fmaf(-rs2.x, c14.x, t.x), bias4.x);
                v[1] = __builtin_fmaf(rs2.y, __builtin_fmaf(-rs2.x, c14.y, t.y), bias4.y);
                v[2] = __builtin_fmaf(rs2.y, __builtin_fmaf(-rs2.x, c14.z, t.z), bias4.z);
                v[3] = __builtin_fmaf(rs2.y, __builtin_fmaf(-rs2.x, c14.w, t.w), bias4.w);
                asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
            } else if constexpr (m == 1) {
                if (__builtin_expect(has_aux, 0)) {   // pre-activation stash of a forward pass that keeps activations
                    float* dst = reinterpret_cast<float*>(reinterpret_cast<char*>(aux0) + (size_t)(it * aux_step));
#if CMDI_AUX_SC1
                    h3_store_f4(dst, make_float4(v[0], v[1], v[2], v[3]));
#else
                    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
#endif
                }
            } else if constexpr (EPI == H3_GELU_SPLIT && m >= 2 && m <= 7) {
                // gelu_erf(x) = 0.5 x (1 + erf_poly(x / sqrt 2)) (common.hpp), the same operations in the same order
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (m == 2) {
                        gu[e] = v[e] * 0.70710678118654752440f;
                        gt[e] = fminf(fabsf(gu[e]), 3.95f);
                        gp[e] = __builtin_fmaf(3.1441086321137846e-05f, gt[e], -0.0003088079974986613f);
                    } else if constexpr (m == 3) {
                        gp[e] = __builtin_fmaf(gp[e], gt[e], 0.0010324155446141958f);
                        gp[e] = __builtin_fmaf(gp[e], gt[e], 0.0005369179998524487f);
                    } else if constexpr (m == 4) {
                        gp[e] = __builtin_fmaf(gp[e], gt[e], -0.01958395168185234f);
                        gp[e] = __builtin_fmaf(gp[e], gt[e], 0.10291960835456848f);
                    } else if constexpr (m == 5) {
                        gp[e] = __builtin_fmaf(gp[e], gt[e], 0.636597752571106f);
                        gp[e] = __builtin_fmaf(gp[e], gt[e], 1.128380298614502f);
                    } else if constexpr (m == 6) {
                        gp[e] = __expf(-gt[e] * gp[e]);
                    } else {
                        v[e] = 0.5f * v[e] * (1.0f + copysignf(1.0f - gp[e], gu[e]));
                    }
                }
                if constexpr (m == 7) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
                else asm volatile("" : "+v"(gp[0]), "+v"(gp[1]), "+v"(gp[2]), "+v"(gp[3]));
            } else if constexpr (m >= kSplit0 && m < kStore && (m - kSplit0) % kSplitStep == 0) {
                constexpr int h = (m - kSplit0) / kSplitStep, e0 = h / 2 * 2;
                if constexpr (h % 2 == 0) {            // hi = f16(x) of one rounded fp32 value (split_f16's barrier), range check
#pragma unroll
                    for (int e = e0; e < e0 + 2; ++e) {
                        asm volatile("" : "+v"(v[e]));
                        oh[e] = (_Float16)v[e];
                        const unsigned b2 = __float_as_uint(v[e]) << 1;
                        ovf = ovf > b2 ? ovf : b2;
                    }
                    asm volatile("" : "+v"(oh), "+v"(ovf));
                } else {                               // lo = f16((x - hi) 2^11)
#pragma unroll
                    for (int e = e0; e < e0 + 2; ++e) ol[e] = (_Float16)((v[e] - (float)oh[e]) * kLoScale);
                    asm volatile("" : "+v"(ol));
                }
            } else if constexpr (m == kStore) {
                _Float16* dst = reinterpret_cast<_Float16*>(dst0 + (size_t)(it * dst_step));
                h3_store_h4(dst, oh);
                h3_store_h4(dst + 32, ol);
            }
        }
    }
};

// ---- the stream of one tile ------------------------------------------------------------------------------------------------
// Issue order behind the MFMAs of step s: [request of chunk s, s < 16] [partial-statistics request, s = 16].
struct H3WStreamArgs {
    unsigned slot_base;          // LDS byte address of this tile's A slot
    i32x4 rsrc;                  // A rows
    unsigned voff;               // this lane's byte offset into them for the NEXT tile's requests
    unsigned dma_dst;            // LDS byte address of this wave's first piece of the next tile (other slot + wave * 1024)
    i32x4 rsrc_part;             // partial LayerNorm statistics (or a stand-in, see the kernel)
    unsigned part_voff, part_dst;
};

// ---- a pass's W fragments: global -> LDS by LDS-DMA -> accumulation registers by ds_read --------------------------------------
// NOT by loads into registers: the register-return path of a CU sustains 20-24 B/clk (profiles/r04_ingest_rate.txt: 64 lanes x 16
// B come back through the vector-register write port), LDS-DMA 60-110 — the 64 W loads of a wave measured 11-13k cycles per pass
// (profiles/r06_h3w_timeline_*), four times the pass's first tile.  The wave stages its own 64 KiB (fragment-ordered copy: piece i
// = instruction i = one contiguous KiB, lane-linear — conflict-free ds_read_b128) through a PRIVATE 16-KiB quarter of the A slot
// that is idle at a pass boundary, as two 8-KiB halves in flight: wait for a half, read its 8 fragments, request the half after
// next.  No block barrier inside (the caller puts one behind it: the slot then goes back to the A requests of all waves).
template <int R>
__device__ __forceinline__ void h3w_stage_read(h8 (&wh)[32], h8 (&wl)[32], unsigned addr) {
    // half-round R: pieces 8R .. 8R + 7 = k16-steps 4R .. 4R + 3 (lo plane, hi plane each), at `addr` + (R & 1) * 8 KiB
    constexpr int o = (R & 1) * 8192, s = 4 * R;
    asm volatile("ds_read_b128 %0, %8 offset:%9\n\tds_read_b128 %1, %8 offset:%10\n\tds_read_b128 %2, %8 offset:%11\n\t"
                 "ds_read_b128 %3, %8 offset:%12\n\tds_read_b128 %4, %8 offset:%13\n\tds_read_b128 %5, %8 offset:%14\n\t"
                 "ds_read_b128 %6, %8 offset:%15\n\tds_read_b128 %7, %8 offset:%16\n\ts_waitcnt lgkmcnt(0)"
                 : "=&a"(wl[s]), "=&a"(wh[s]), "=&a"(wl[s + 1]), "=&a"(wh[s + 1]), "=&a"(wl[s + 2]), "=&a"(wh[s + 2]), "=&a"(wl[s + 3]),
                   "=&a"(wh[s + 3])
                 : "v"(addr), "n"(o), "n"(o + 1024), "n"(o + 2048), "n"(o + 3072), "n"(o + 4096), "n"(o + 5120), "n"(o + 6144),
                   "n"(o + 7168)
                 : "memory");
}
template <int R>
__device__ __forceinline__ void h3w_stage_round(h8 (&wh)[32], h8 (&wl)[32], i32x4 rsrc_w, unsigned wv, unsigned region, int& soff) {
    // half R has landed when at most the 8 pieces of half R + 1 are still in flight
    if constexpr (R < 7) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    h3w_stage_read<R>(wh, wl, region + wv);
    if constexpr (R + 2 < 8) {      // the half after next, into the 8 KiB just read (M0: back to the start of that half)
        soff += 1024;
        h3w_dma(rsrc_w, wv, soff, region + (R & 1) * 8192);
#pragma unroll
        for (int q = 1; q < 8; ++q) h3w_dma_next<1024, 1024>(rsrc_w, wv, soff);
    }
    if constexpr (R + 1 < 8) h3w_stage_round<R + 1>(wh, wl, rsrc_w, wv, region, soff);
}
__device__ __forceinline__ void h3w_stage_w(h8 (&wh)[32], h8 (&wl)[32], i32x4 rsrc_w, unsigned wv, unsigned region) {
    int soff = 0;
    h3w_dma(rsrc_w, wv, 0, region);
#pragma unroll
    for (int q = 1; q < 16; ++q) h3w_dma_next<1024, 1024>(rsrc_w, wv, soff);
    h3w_stage_round<0>(wh, wl, rsrc_w, wv, region, soff);
}

// k16-step S: [wait] MFMA | read | MFMA | read | MFMA — one companion instruction per MFMA, so that each issues under a running
// MFMA (a lone wave per SIMD hides about five issue slots per MFMA and nothing else), and behind each MFMA one slice of the
// deferred epilogue.  The reads fetch the fragments of step S + 2 into the register set step S - 1 has just released; at the
// wait of step S the two reads of step S + 1 may still be in flight.  Step 0 starts the sums (srcC = 0: no zeroing pass, and no
// VALU write in front of an MFMA the compiler cannot see).
template <int S, class Fill>
__device__ __forceinline__ void h3w_step(f32x16& c0, f32x16& c1, h8 (&fh)[3], h8 (&fl)[3], const h8& wh, const h8& wl,
                                         const unsigned (&ad)[4], Fill& fill) {
    constexpr int cur = S % 3, nxt = (S + 2) % 3;
    // LDS operations that may stay in flight at the wait of step S: the two fragment reads of step S + 1 and whatever the
    // epilogue slices of step S - 1 issued (all of it younger than the fragments of step S: the waits carry memory clobbers, the
    // compiler cannot move its LDS operations out of the step it wrote them in)
    constexpr int LG = (S + 1 < 32 ? 2 : 0) + (S >= 1 ? Fill::lds_ops(3 * S - 3) + Fill::lds_ops(3 * S - 2) + Fill::lds_ops(3 * S - 1) : 0);
    static_assert(LG <= 15, "lgkmcnt is a 4-bit counter");
    constexpr int off = ((S + 2) >> 1) * 4096;
    constexpr bool RD = S + 2 < 32;
    // MFMA 1: acc1 (+)= a_hi w_lo, and the read of the hi-plane fragment of step S + 2
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(LG) : "memory");
    if constexpr (S == 0) {
        asm volatile("v_mfma_f32_32x32x16_f16 %[c1], %[fh], %[wl], 0\n\tds_read_b128 %[nh], %[ah] offset:%[off]"
                     : [c1] "=&v"(c1), [nh] "=&v"(fh[nxt])
                     : [fh] "v"(fh[cur]), [wl] "a"(wl), [ah] "v"(ad[0]), [off] "n"(off));
    } else if constexpr (RD) {
        asm volatile("v_mfma_f32_32x32x16_f16 %[c1], %[fh], %[wl], %[c1]\n\tds_read_b128 %[nh], %[ah] offset:%[off]"
                     : [c1] "+v"(c1), [nh] "=&v"(fh[nxt])
                     : [fh] "v"(fh[cur]), [wl] "a"(wl), [ah] "v"(ad[(S & 1) * 2]), [off] "n"(off));
    } else {
        asm volatile("v_mfma_f32_32x32x16_f16 %[c1], %[fh], %[wl], %[c1]" : [c1] "+v"(c1) : [fh] "v"(fh[cur]), [wl] "a"(wl));
    }
    fill.template slice<3 * S>();
    // MFMA 2: acc0 (+)= a_hi w_hi, and the read of the lo-plane fragment
    if constexpr (S == 0) {
        asm volatile("v_mfma_f32_32x32x16_f16 %[c0], %[fh], %[wh], 0\n\tds_read_b128 %[nl], %[al] offset:%[off]"
                     : [c0] "=&v"(c0), [nl] "=&v"(fl[nxt])
                     : [fh] "v"(fh[cur]), [wh] "a"(wh), [al] "v"(ad[1]), [off] "n"(off));
    } else if constexpr (RD) {
        asm volatile("v_mfma_f32_32x32x16_f16 %[c0], %[fh], %[wh], %[c0]\n\tds_read_b128 %[nl], %[al] offset:%[off]"
                     : [c0] "+v"(c0), [nl] "=&v"(fl[nxt])
                     : [fh] "v"(fh[cur]), [wh] "a"(wh), [al] "v"(ad[(S & 1) * 2 + 1]), [off] "n"(off));
    } else {
        asm volatile("v_mfma_f32_32x32x16_f16 %[c0], %[fh], %[wh], %[c0]" : [c0] "+v"(c0) : [fh] "v"(fh[cur]), [wh] "a"(wh));
    }
    fill.template slice<3 * S + 1>();
    // MFMA 3: acc1 += a_lo w_hi
    asm volatile("v_mfma_f32_32x32x16_f16 %[c1], %[fl], %[wh], %[c1]" : [c1] "+v"(c1) : [fl] "v"(fl[cur]), [wh] "a"(wh));
}

template <int S, class Fill>
struct H3WSteps {
    static __device__ __forceinline__ void run(f32x16& c0, f32x16& c1, h8 (&fh)[3], h8 (&fl)[3], const h8 (&wh)[32],
                                               const h8 (&wl)[32], const unsigned (&ad)[4], const H3WStreamArgs& a, int& soff,
                                               Fill& fill, unsigned long long (&tk)[8]) {
        h3w_step<S>(c0, c1, fh, fl, wh[S], wl[S], ad, fill);
        if constexpr ((H3W_ABL & 16) && (S % 8 == 7 || S == 0)) asm volatile("s_memtime %0" : "=s"(tk[S == 0 ? 1 : 2 + S / 8]));
        // the next tile's piece of chunk S (this wave's 8 rows of it), behind the step's last MFMA: 16 pieces in the first 16
        // steps, the last one >= 1,500 cycles ahead of the wait at the end of the tile
        if constexpr (S == 0 && !(H3W_ABL & 1)) { soff = 0; h3w_dma(a.rsrc, a.voff, 0, a.dma_dst); }
        else if constexpr (S < 16 && !(H3W_ABL & 1)) h3w_dma_next<4096>(a.rsrc, a.voff, soff);
        // ... and its rows' partial LayerNorm statistics (8 rows x 128 B: one piece per wave; without folded LayerNorm the
        // request fetches a KiB of A rows instead and nobody reads it)
        if constexpr (S == 16 && !(H3W_ABL & 1)) h3w_dma(a.rsrc_part, a.part_voff, 0, a.part_dst);
        fill.template slice<3 * S + 2>();
        if constexpr (S + 1 < 32) H3WSteps<S + 1, Fill>::run(c0, c1, fh, fl, wh, wl, ad, a, soff, fill, tk);
    }
};

// The stream of one tile.  `ad_lane`: this lane's four fragment offsets inside a slot (hi / lo plane of k-substep 0, then of
// k-substep 1); `fill`: the deferred epilogue of the previous tile (or H3WNoFill).
template <class Fill>
__device__ __forceinline__ void h3w_tile_stream(f32x16& c0, f32x16& c1, const h8 (&wh)[32], const h8 (&wl)[32],
                                                const unsigned (&ad_lane)[4], const H3WStreamArgs& a, Fill& fill,
                                                unsigned long long (&tk)[8]) {
    if constexpr (H3W_ABL & 16) asm volatile("s_memtime %0" : "=s"(tk[0]));
    unsigned ad[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) ad[q] = a.slot_base + ad_lane[q];
    h8 fh[3], fl[3];
    // fragments of steps 0 and 1 (chunk 0, both k-substeps)
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7"
                 : "=&v"(fh[0]), "=&v"(fl[0]), "=&v"(fh[1]), "=&v"(fl[1])
                 : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]));
    int soff = 0;
    H3WSteps<0, Fill>::run(c0, c1, fh, fl, wh, wl, ad, a, soff, fill, tk);
    // End of the stream: (1) this wave's pieces of the next tile have landed — operations issued BEHIND the last request (step
    // 16) may stay in flight: a riding epilogue's six split stores of its iterations 1-3 (waiting for their write-through cost
    // 520 cycles per tile); (2) the accumulators may be read by ordinary instructions: an MFMA's result is not interlocked
    // against a VALU / LDS / VMEM read that follows within passes + 3 wait states — the compiler inserts those for its own
    // MFMAs and cannot see these
    asm volatile("s_waitcnt vmcnt(%2)\n\ts_nop 7\n\ts_nop 7" : "+v"(c0), "+v"(c1) : "n"(Fill::kStoresBehindRequests) : "memory");
    if constexpr (H3W_ABL & 16) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tk[6]));
}

// split rows W[N][1024 halves] -> fragment order: [strip N / 128][wave 4][instruction i = 2 s + plane (0 lo, 1 hi)][lane 64][8
// halves]: lane (l31, hi) of wave w holds, for k16-step s, columns 16 s + 8 hi .. + 8 of row strip * 128 + 32 w + l31 — the B
// operand of v_mfma_f32_32x32x16_f16.  One thread per 16-byte fragment.
__global__ __launch_bounds__(256) void pack_w_h3w_kernel(const _Float16* __restrict__ W, _Float16* __restrict__ Wp, int N) {
    const int idx = blockIdx.x * 256 + threadIdx.x;        // fragment index = ((strip * 4 + wave) * 64 + i) * 64 + lane
    if (idx >= N * 128) return;
    const int lane = idx & 63, i = (idx >> 6) & 63, sw = idx >> 12;
    const int s = i >> 1, hi_plane = i & 1;
    const int n = sw * 32 + (lane & 31);
    const int col = (s >> 1) * 64 + (hi_plane ? 0 : 32) + (s & 1) * 16 + (lane >> 5) * 8;
    *reinterpret_cast<h8*>(Wp + (size_t)idx * 8) = *reinterpret_cast<const h8*>(W + (size_t)n * 1024 + col);
}

template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_h3w_kernel(const H3Params p, int passes, int subs, const float* part_src) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using TC = H3WTile;
    using Defer = H3WDefer<EPI>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int M = p.M;

    // ---- work of this block -------------------------------------------------------------------------------------------
    const int nb = gridDim.x >> 3, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;     // block id % 8 = XCD (observed; speed only)
    const int tiles_all = (M + 31) >> 5;
    const int tx0 = (int)((long)tiles_all * xcd / 8), tx1 = (int)((long)tiles_all * (xcd + 1) / 8);
    const int sub = j % subs;
    const int t_begin = tx0 + (int)((long)(tx1 - tx0) * sub / subs), t_end = tx0 + (int)((long)(tx1 - tx0) * (sub + 1) / subs);
    const int n_tiles = t_end - t_begin;
    if (n_tiles <= 0) return;

    char* scratch = lds + 2 * TC::SLOT;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

    // fragment offsets of this lane inside a slot: row l31, slot (2 ks + hi) of the hi plane / (4 + 2 ks + hi) of the lo plane,
    // swizzled as the requests store them (gemm_h3.hpp)
    const int swz = (l31 >> 1) & 7;
    unsigned ad_lane[4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        ad_lane[2 * ks] = (unsigned)(l31 * 128 + ((2 * ks + hi) ^ swz) * 16);
        ad_lane[2 * ks + 1] = (unsigned)(l31 * 128 + ((4 + 2 * ks + hi) ^ swz) * 16);
    }
    // requests: this wave's 8 rows of a chunk (row = lane / 8, 16-B slot = lane % 8, source slot swizzled)
    const int prow = wave * 8 + (lane >> 3), pslot = lane & 7;
    const unsigned long long a_addr = (unsigned long long)(size_t)p.A;
    H3WStreamArgs sa;
    sa.rsrc = i32x4{(int)(unsigned)a_addr, (int)(unsigned)((a_addr >> 32) & 0xffff), -1, 0x00020000};
    auto row_voff = [&](int t) __attribute__((always_inline)) {
        int grow = t * 32 + prow;
        grow = grow < M ? grow : M - 1;
        return (unsigned)grow * 2048u + (unsigned)((pslot ^ ((prow >> 1) & 7)) << 4);
    };
    // folded LayerNorm: the raw partial statistics of a tile's rows are requested with its A rows (one 8-row piece per wave) into
    // part buffer `par`; behind the barrier that publishes them wave 0 — one row per lane of its lower half — turns them into
    // (mean, rstd) in statistics buffer par3 (read by that tile's epilogue, deferred or not, behind the NEXT barrier).
    // (`part_src` = p.ln_part, or — without folded LayerNorm — the A rows themselves, chosen on the host: the request is issued
    // either way, the counted waits of a cold start depend on it, and a device-side select of the two pointers ends up in
    // vector registers, which an "s" operand then gets unchanged)
    char* part_lds = scratch + TC::SCRATCH_BYTES;
    const unsigned long long part_addr = (unsigned long long)(size_t)part_src;
    sa.rsrc_part = i32x4{(int)(unsigned)part_addr, (int)(unsigned)((part_addr >> 32) & 0xffff), -1, 0x00020000};
    auto part_voff_of = [&](int t) __attribute__((always_inline)) {
        int grow = t * 32 + prow;
        grow = grow < M ? grow : M - 1;
        return (unsigned)grow * 128u + (unsigned)(pslot << 4);
    };
    auto stats_from_lds = [&](int t, int par_, int par3_, bool write_out) __attribute__((always_inline)) {
        if (p.ln_part && tid < 32) {
            const float2 ms = h3_row_stats_from_partials(reinterpret_cast<const float4*>(part_lds + par_ * TC::PART + tid * 128));
            reinterpret_cast<float2*>(scratch + par3_ * 256 + TC::MAIN_BYTES)[tid] = ms;
            // a stashing forward pass keeps (mean, rstd) of every LayerNorm for the backward: the first column strip writes them
            if (p.ln_stats && write_out && t * 32 + tid < M) *reinterpret_cast<float2*>(p.ln_stats + 2 * (size_t)(t * 32 + tid)) = ms;
        }
    };
    // this wave's 64 KiB of the fragment-ordered W of the strip at column n0 (pack_w_h3w_kernel)
    auto w_rsrc = [&](int n0) __attribute__((always_inline)) {
        const unsigned long long w_addr = (unsigned long long)(size_t)p.Wp + ((size_t)(n0 >> 7) * 4 + wave) * 65536ull;
        return i32x4{(int)(unsigned)w_addr, (int)(unsigned)((w_addr >> 32) & 0xffff), -1, 0x00020000};
    };
    auto strip_of = [&](int pass) __attribute__((always_inline)) { return ((pass * nb + j) / subs) * 128; };

    int par = 0, par3 = 0;                          // slot of the current tile; its statistics buffer (of three)
    // probes build, dbg & 16: cycle stamps of wave 0 — [0] start, [1] first barrier passed, then per tile (stream end, barrier
    // passed, epilogue end), 64 slots per block; [63] = the 100 MHz clock at the start
    [[maybe_unused]] int n_stamp = 0;
    [[maybe_unused]] long long* stamps = nullptr;
    auto stamp = [&]() __attribute__((always_inline)) {
        if ((CMDI_DBG(p) & 16) && stamps && tid == 0 && n_stamp < 62) stamps[n_stamp++] = (long long)__builtin_readcyclecounter();
    };
    if ((CMDI_DBG(p) & 16) && p.dbg_buf) {
        stamps = p.dbg_buf + (size_t)blockIdx.x * 64;
        if (tid == 0) { stamps[63] = (long long)__builtin_amdgcn_s_memrealtime(); stamps[62] = 0; }
        stamp();
    }
    const unsigned part_base = lds_base + 2 * TC::SLOT + (unsigned)TC::SCRATCH_BYTES;

    Defer defer(p, lane, wave);
    H3WNoFill nofill;
    f32x16 acc0[2][1][1], acc1[2][1][1];            // two accumulator sets: tiles alternate
    h8 wh[32], wl[32];                              // the pass's W fragments: the 256 accumulation registers

    // the first tile of the first pass: its requests (and its rows' partial statistics) into slot 0
#pragma unroll
    for (int c = 0; c < 16; ++c) h3w_dma(sa.rsrc, row_voff(t_begin), c * 128, lds_base + c * 4096 + wave * 1024);
    h3w_dma(sa.rsrc_part, part_voff_of(t_begin), 0, part_base + wave * 1024);

    bool pending = false;                           // a deferred epilogue waits for the next stream to ride in
    int kset = 0;                                   // accumulator set of the next tile
    for (int pass = 0; pass < passes; ++pass) {
        const int n0 = strip_of(pass);
        if (n0 >= p.N) break;
        const bool last_pass = pass + 1 >= passes || strip_of(pass + 1) >= p.N;
        // ---- the pass's W fragments, staged through this wave's quarter of the idle A slot (the one the previous tile has just
        // left); the barrier behind it hands the slot back to the A requests of all waves — and, in the first pass, publishes the
        // first tile's pieces (older than the staging requests, whose last wait is vmcnt(0))
        h3w_stage_w(wh, wl, w_rsrc(n0), (unsigned)lane * 16u, lds_base + (par ^ 1) * TC::SLOT + wave * 16384);
        __builtin_amdgcn_s_barrier();
        if (pass == 0) stamp();
        // (mean, rstd) of the pass's first tile (every pass: the statistics buffers rotate)
        stats_from_lds(t_begin, par, par3, n0 == 0);

        // one tile.  K = the accumulator set it sums into (tiles alternate, so the deferred epilogue of the tile before reads the
        // OTHER set in place); FILL: that epilogue rides in this stream
        auto tile = [&](auto K_, auto FILL_, int it) __attribute__((always_inline)) {
            constexpr int K = decltype(K_)::value;
            constexpr bool FILL = decltype(FILL_)::value;
            const int t = t_begin + it;
            // the tile after this one: the next of the sub-range, or the first of the next pass, or (nothing left) this one again
            const int t_next = it + 1 < n_tiles ? t + 1 : (last_pass ? t : t_begin);
            sa.slot_base = lds_base + par * TC::SLOT;
            sa.dma_dst = lds_base + (par ^ 1) * TC::SLOT + wave * 1024;
            sa.voff = row_voff(t_next);
            sa.part_voff = part_voff_of(t_next);
            sa.part_dst = part_base + (par ^ 1) * TC::PART + wave * 1024;
            const int par3_next = par3 == 2 ? 0 : par3 + 1;
            // first tile of a pass: the epilogue riding in its stream (if any) still belongs to the pass before; this pass's lane
            // constants are requested in front of the stream and taken over behind it
            if constexpr (Defer::kSupported) { if (it == 0) defer.prefetch_pass(n0); }
            [[maybe_unused]] unsigned long long tk[8] = {};   // H3W_ABL & 16: stream begin, steps 0 / 7 / 15 / 23 / 31 issued, end
            if constexpr (FILL) {
                H3WFill<Defer> fill{defer, acc0[K ^ 1][0][0], acc1[K ^ 1][0][0]};
                h3w_tile_stream(acc0[K][0][0], acc1[K][0][0], wh, wl, ad_lane, sa, fill, tk);
            } else {
                h3w_tile_stream(acc0[K][0][0], acc1[K][0][0], wh, wl, ad_lane, sa, nofill, tk);
            }
            if constexpr ((H3W_ABL & 16) != 0) {
                if ((CMDI_DBG(p) & 16) && stamps && tid == 0 && it == 2)
                    for (int q = 0; q < 7; ++q) stamps[48 + q] = (long long)tk[q];
            }
            stamp();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();      // every wave has left this slot; every wave's pieces of the next tile are in LDS
            stamp();
            // the next tile's (mean, rstd) — unless it is the first tile of the next pass, which computes its own above
            if (it + 1 < n_tiles) stats_from_lds(t_next, par ^ 1, par3_next, n0 == 0);
            if constexpr (Defer::kSupported) { if (it == 0) defer.commit_pass(); }
            // interior tile with another stream behind it (the pass's next tile or the next pass's first): its epilogue rides
            // there; otherwise here
            pending = false;
            if constexpr (Defer::kSupported) {
                if ((it + 1 < n_tiles || !last_pass) && t * 32 + 32 <= M) {
                    defer.begin_tile(t * 32, scratch + par3 * 256);
                    pending = true;
                }
            }
            if (!pending) {
                if constexpr (Defer::kSupported) defer.finish();
                int n0_here = n0;      // (opaque: the lane constants of this epilogue are loaded where it runs — once per block —
                asm volatile("" : "+s"(n0_here));   //  instead of being hoisted out of the tile loop into registers every stream then carries)
                h3_epilogue<TC, EPI>(p, acc0[K], acc1[K], t * 32, n0_here, M, 0, scratch + par3 * 256);
            }
            stamp();
            par ^= 1;
            par3 = par3_next;
            kset = K ^ 1;
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        for (int it = 0; it < n_tiles; ++it) {
            if (kset) { if (pending) tile(I1{}, std::true_type{}, it); else tile(I1{}, std::false_type{}, it); }
            else { if (pending) tile(I0{}, std::true_type{}, it); else tile(I0{}, std::false_type{}, it); }
        }
    }
    if ((CMDI_DBG(p) & 16) && stamps && tid == 0) stamps[62] = n_stamp;
}

}  // namespace cmdi
