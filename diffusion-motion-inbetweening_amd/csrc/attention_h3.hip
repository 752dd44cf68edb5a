// Self-attention core on the f16 matrix pipe with fp32-equivalent split-f16 products (same math as
// attention_f32.hip: torch MultiheadAttention called from model/mdm.py:284, SURVEY.md Appendix A.2;
// operand format and error analysis in gemm_h3.hpp).
//
// Input: qkv as SPLIT ROWS [B'·S][2·3d] (written by the in_proj GEMM's epilogue): per 32-column chunk
// 32 hi halves then 32 lo halves (lo = (x - hi)·2^11), so one head of Q, K or V of one token is 512
// contiguous bytes.  d_head = 128, S <= 224.
//
// Work split: grid = (B'·H, ceil(S/256)); 8 waves per block (one block per (sequence, head) and CU:
// K / V are staged once, not once per query half); a wave owns 32 queries.
// K / V stream through LDS in 32-key stages (LDS-DMA, ring of 4 stages x 32 KiB, 3 in flight).
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
#include <cstdlib>

#include "common.hpp"
#include "gemm_h3.hpp"
#include "kernels.hpp"

namespace cmdi {

namespace {
constexpr int DH = 128;
constexpr int KBLK = 32;                 // keys per stage
constexpr int NWAVE = 8;                 // 256 queries per block: one block per (sequence, head) at S <= 224
constexpr int ROWB = 512;                // bytes of one head of one token: 4 chunks x (64 B hi + 64 B lo)
constexpr int TILE = KBLK * ROWB;        // 16 KiB per K or V tile
constexpr int STAGE = 2 * TILE;
constexpr int NSTG = 4;                  // LDS ring: 3 stages (96 KiB) in flight ahead of the one being used
constexpr int kAttnStagDefault = 1;      // two wave groups half a stage apart (attention_h3_kernel); probes build: CMDI_ATTN_STAG
constexpr int kAttnSplitDefault = -1;    // -1 = by grid size, see launch_attention_h3 (at one block per CU, B = 32: 36.7 vs 35.1 us
                                         // per layer — no gain; below half a chip of (sequence, head) pairs the split wins)

typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));

// 16-B slot swizzle of a tile row (32 slots per key): slot t of key k is stored at t ^ kswz(k).
// kswz is a bijection of k & 15 (K: the 16 keys of a ds_read_b128 lane group hit 16 different slots
// of the 256-B bank row) whose bits 2-3 are k & 3 (V: the 4 keys of a transpose-read group land in
// 4 different 64-B quarters).
__device__ __forceinline__ int kswz(int k) { return ((k & 3) << 2) | ((k >> 2) & 3); }

// One 32-key stage of K and V by LDS-DMA: 32 pieces of 1 KiB (2 keys each), 32 / NW per wave: `buffer_load ... offen lds`
// through a descriptor that starts at the sequence's first row and ENDS WITH THE SEQUENCE'S LAST ROW (token-major layout).
// The lane's byte offset inside a stage is a constant of the kernel (kv_lane_offset); the stage position is ADDED TO THAT
// VECTOR OFFSET (round 4, ADVICE r3: the scalar offset of a raw buffer access is excluded from the hardware's range check,
// the vector offset is not), so rows past the sequence read as zeros whatever lies behind them in memory — a neighbouring
// sequence's overflowed rows or a NaN pattern behind the tensor cannot reach the P·V product (0·NaN).  Their scores are
// masked by a select and their P is exactly 0.  (Rounds 1-2: flat global_load_lds with per-piece clamping, ~15 vector
// instructions per piece inside the key loop; one v_add per piece now.)  Head-major layout: the descriptor ends with the
// last head's V matrix (rows of the next sequence are finite split rows there; past the tensor the hardware returns zeros).
template <int NW = NWAVE>
__device__ __forceinline__ unsigned kv_lane_offset(int it, size_t ld, long koff, long voff, int wave, int lane) {
    const int pc = it * NW + wave;               // 0..15 K pieces, 16..31 V pieces (2 keys each)
    const int mat = pc >> 4, g = pc & 15;
    const int kl = 2 * g + (lane >> 5);
    const int t = (lane & 31) ^ kswz(kl);
    return (unsigned)(((size_t)kl * ld + (size_t)(mat ? voff : koff) + t * 8) * 2);
}
__device__ __forceinline__ h8 tr_pair(const char* p0, const char* p1) {
    const s4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)p0);
    const s4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)p1);
    return __builtin_bit_cast(h8, (s8v)__builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}
}  // namespace

// NW waves per block (32 queries each) and an LDS ring of NS K/V stages.  (8, 4): one block per (sequence, head) and per
// CU, K / V staged once, three stages in flight.  (4, 2): the queries of a (sequence, head) are split over two blocks of
// 4 waves with a 64-KiB ring each, so TWO blocks share a CU and one's load bursts / epilogue sit under the other's key
// loop (the pair lands on one XCD: block ids differ by gridDim.x, a multiple of 8, so the second K / V read hits its L2).
// STAG != 0: the waves of a block run in two groups half a stage apart.  All waves of a block otherwise move in lockstep
// (one barrier per 32-key stage), so the two waves that share a SIMD both issue MFMAs (Sᵀ = K·Qᵀ), then both do the
// softmax on the VALU with the matrix pipe idle, then both issue MFMAs again (Oᵀ += Vᵀ·Pᵀ).  The lagging group runs
// [Vᵀ·Pᵀ of the previous stage, K·Qᵀ, softmax] inside a barrier interval where the leading group runs [K·Qᵀ, softmax,
// Vᵀ·Pᵀ]: one group's softmax falls under the other's products.  The ring then keeps the previous stage alive, so the
// look-ahead is NS - 2 stages instead of NS - 1.  STAG = 1: waves NW/2.. lag; STAG = 2: odd waves lag.
// PERSIST (round 6): gridDim.x blocks walk the n_items (sequence, head) pairs — block x takes the contiguous range
// [n_items x / gridDim.x, n_items (x + 1) / gridDim.x) — instead of one block per pair.  At B >= 128 a CU is handed 8-32 pairs
// one after the other; as separate blocks each pays its dispatch and runs its load burst (Q + the first K/V stages) with nothing
// else on the CU, and its store tail likewise.  Inside one block the next pair's requests are issued right behind the
// epilogue's LDS read-back (one barrier: the ring is every wave's staging area), so they travel under the previous pair's
// output stores; a wave runs the same instruction sequence per pair: the same bits.
template <bool STASH, int NW = NWAVE, int NS = NSTG, int STAG = 0, bool PERSIST = false>
__global__ __launch_bounds__(64 * NW, 2) void attention_h3_kernel(const _Float16* __restrict__ qkv,
                                                              float* __restrict__ out,
                                                              _Float16* __restrict__ out_s,
                                                              int* __restrict__ range_flag,
                                                              float* __restrict__ row_stats, int S,
                                                              int H, float scale, int dbg_arg, long head_rows, int n_items) {
#ifdef CMDI_PROBES
    const int dbg = dbg_arg;   // bench-only ablations / cycle stamps (probes build only, see gemm_h3.hpp)
#else
    constexpr int dbg = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) char lds[];  // [2][K tile | V tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    long long t0 = 0, t1 = 0, t2 = 0;
    if (dbg & 16) t0 = __builtin_readcyclecounter();
    const int item0 = PERSIST ? (int)((long)n_items * blockIdx.x / gridDim.x) : (int)blockIdx.x;
    const int item1 = PERSIST ? (int)((long)n_items * (blockIdx.x + 1) / gridDim.x) : (int)blockIdx.x + 1;
    for (int bh = item0; bh < item1; ++bh) {
    const int b = bh / H, h = bh % H;
    const int d_model = H * DH;
    // token-major qkv: row = token, 6 d halves, q | k | v column blocks.  head_rows > 0: HEAD-major (written so by the
    // in_proj epilogue, gemm_params.hpp cs_head_major): operand w of head h is its own [head_rows][256-half] matrix, so
    // the K / V of a (sequence, head) are one contiguous 100-KB stream each instead of 512-B pieces 6 KiB apart
    const size_t ld = head_rows ? 256 : 6 * (size_t)d_model;     // halves per token row
    const long hm = head_rows * 256;                              // halves per head matrix
    const long qoff = head_rows ? (long)(0 * H + h) * hm : 2 * h * DH;
    const long koff = head_rows ? (long)(1 * H + h) * hm : 2 * (d_model + h * DH);
    const long voff = head_rows ? (long)(2 * H + h) * hm : 2 * (2 * d_model + h * DH);
    const _Float16* base = qkv + (size_t)b * S * ld;
    const int q0 = (blockIdx.y * NW + wave) * 32;
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
    float m_run = -INFINITY, l_run = 0.f;   // m_run in log2 units
    const float scale2 = scale * 1.4426950408889634f;

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

    // K/V of a whole (sequence, head) is only 7 stages, each a fabric round trip: keep NSTG - 1 stages
    // in flight (counted vmcnt, raw barrier) so the loop is not one memory latency per 32 keys.
    constexpr int PCS = 32 / NW;   // LDS-DMA pieces per wave per stage
    constexpr int AHEAD = STAG ? NS - 2 : NS - 1;   // stages requested ahead of the one being multiplied
    const bool lag = STAG == 1 ? wave >= NW / 2 : (STAG == 2 ? (wave & 1) != 0 : false);
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    unsigned kv_off[PCS];
#pragma unroll
    for (int it = 0; it < PCS; ++it) kv_off[it] = kv_lane_offset<NW>(it, ld, koff, voff, uwave, lane);
    const unsigned row_bytes = (unsigned)(ld * 2);
    // token-major: the descriptor ends with the last row of the launch's last sequence; head-major: with the V matrix of the
    // last head (operand matrices of head_rows rows each)
    const size_t span = head_rows ? (size_t)3 * H * hm * 2 - (size_t)b * S * row_bytes
                                  : (size_t)S * row_bytes;
    const auto kv_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(base), 0,
                                                           (int)(span < 0xffffffffull ? span : 0xffffffffull), 0x00020000);
    auto stage_kv = [&](char* stage, int key0) {
#pragma unroll
        for (int it = 0; it < PCS; ++it) {
            const int pc = it * NW + uwave;
            char* dst = stage + (pc >> 4) * TILE + (pc & 15) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(kv_rsrc, (__attribute__((address_space(3))) void*)dst, 16,
                                                     (int)(kv_off[it] + key0 * row_bytes), 0, 0, 0);
        }
    };
#pragma unroll
    for (int st = 0; st < AHEAD; ++st)
        if (st < nkb) stage_kv(lds + st * STAGE, st * KBLK);
    {
        const int ahead = (nkb - 1 < AHEAD - 1 ? nkb - 1 : AHEAD - 1);   // stages allowed to stay in flight
        if (ahead >= 2) wait_vmcnt<2 * PCS>(); else if (ahead == 1) wait_vmcnt<PCS>(); else wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (dbg & 16) t1 = __builtin_readcyclecounter();

    f32x16 a0, a1, a2;
    float s[16];
    // ---- scores of stage kb: Sᵀ = K·Qᵀ ------------------------------------------------------------------
    auto scores = [&](int kb) {
        const char* kt = lds + (kb % NS) * STAGE;
#pragma unroll
        for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; a2[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if ((dbg & 1) && ks > 0) break;   // bench-only ablation: 1/8 of the QK^T MFMAs
            const int t = (ks >> 1) * 8 + (ks & 1) * 2 + hi;
            const h8 kh = *reinterpret_cast<const h8*>(kt + k_row + ((t ^ fk) << 4));
            const h8 kl = *reinterpret_cast<const h8*>(kt + k_row + (((t + 4) ^ fk) << 4));
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], a2, 0, 0, 0);
        }
    };
    // ---- online softmax of stage kb (per query = per lane column): a0..a2 -> p in s[], m_run / l_run / o rescaled ----
    auto softmax = [&](int kb) {
        // scores in log2 units: p = 2^(s2 - m2) with s2 = s * log2(e) (v_exp_f32 is a base-2
        // exponential; the absolute error of p stays below 2.2e-8 because |x| 2^-24 e^-|x| <= 2^-24 / e)
        float mloc = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = (a0[r] + (a1[r] + a2[r]) * kLoInv) * scale2;
            mloc = fmaxf(mloc, s[r]);
        }
        if (kb == nkb - 1) {   // only the last block can hold keys past S (wave-uniform branch)
            mloc = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * KBLK + mfma32_row(r, lane);
                s[r] = key < S ? s[r] : -INFINITY;
                mloc = fmaxf(mloc, s[r]);
            }
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        // Lazy reference maximum: m_run moves only when a block's maximum exceeds it by more than 2^LAZY (log2 units), so
        // p = 2^(s - m_run) stays below 2^LAZY — the hi / lo split of p keeps its 22 bits relative to that — and the 64
        // accumulator values are rescaled only then (wave-uniform branch; after the first block it is rarely taken).
        constexpr float LAZY = 8.0f;
        const bool moved = mloc > m_run + LAZY;          // first block: m_run = -inf
        if (__builtin_amdgcn_ballot_w64(moved) != 0) {
            const float m_new = moved ? mloc : m_run;
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // 1 where nothing moved, 0 on the first block
            l_run *= alpha;
            m_run = m_new;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        }
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = (dbg & 2) ? (s[r] - m_run) * 1e-3f : __builtin_amdgcn_exp2f(s[r] - m_run);   // masked keys: 2^-inf = 0
            psum += s[r];
        }
        l_run += psum;                                   // partial over this lane's keys
    };
    // ---- Oᵀ += Vᵀ · Pᵀ of stage kb (p in s[]) -------------------------------------------------------------
    auto pv = [&](int kb) {
        const char* vt = lds + (kb % NS) * STAGE + TILE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if ((dbg & 4) && kk > 0) break;   // bench-only ablation: half of the PV work
            h8 ph, pl, ps;   // p_hi (round to nearest), p - p_hi (unscaled), p_hi * 2^-11
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                _Float16 a, b;
                split_f16_unscaled(s[8 * kk + e], a, b);      // (gemm_h3.hpp: the one pinned split of an unscaled operand)
                ph[e] = a;
                pl[e] = b;
            }
            ps = ph * (_Float16)kLoInv;   // packed f16 multiply: exact (power of two) above 2^-14
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
    };

    for (int kb = 0; kb < nkb; ++kb) {
        if (kb + AHEAD < nkb)
            stage_kv(lds + ((kb + AHEAD) % NS) * STAGE, (kb + AHEAD) * KBLK);
        if (active) {
            if (!lag) {
                scores(kb);
                softmax(kb);
                pv(kb);
            } else {
                if (kb > 0) pv(kb - 1);
                scores(kb);
                softmax(kb);
            }
        }
        {   // stage kb+1 must have landed; later stages may stay in flight across the barrier
            const int last = nkb - 1 < kb + AHEAD ? nkb - 1 : kb + AHEAD;   // newest stage issued
            const int ahead = last - (kb + 1);
            if (ahead >= 2) wait_vmcnt<2 * PCS>(); else if (ahead == 1) wait_vmcnt<PCS>(); else wait_vmcnt<0>();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (active && lag) pv(nkb - 1);

    if (dbg & 16) t2 = __builtin_readcyclecounter();
    // the ring is free from here on (round 3: the split output leaves through it as whole rows): every wave — the lagging
    // group reads its last V tile after the loop's last barrier — must be past its last LDS read first
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (active) {
        const float lsum = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / lsum;
        if (out_s) {
            // Split rows for the out_proj GEMM.  A lane holds ONE query's values at dims 32 db + 8 g4 + 4 hi + e: stored from
            // registers that is 32 dwordx2 stores per lane, each instruction touching 32 rows with 16 bytes (rounds 1-2; the
            // store tail was ~9 % of a block's life).  Now: the wave's 32 queries x 512 B (the head's 4 chunks of 64 B hi |
            // 64 B lo) are assembled in a private LDS slice (rows padded to 528 B: conflict-free ds_write_b64 per 16-lane
            // group) and leave as 16 dwordx4 stores per lane, every instruction two whole 512-byte rows.
            constexpr int RSTR = 528;
            char* ws = lds + wave * (32 * RSTR);
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
                        overflow |= qok && !(fabsf(v) < 65504.0f);
                    }
                    char* wp = ws + l31 * RSTR + db * 128 + (g4 * 8 + 4 * hi) * 2;
                    *reinterpret_cast<h4*>(wp) = oh;
                    *reinterpret_cast<h4*>(wp + 64) = ol;
                }
            if (overflow && range_flag) atomicOr(range_flag, 1);
            // (wave-private slice: LDS operations of one wave execute in order, no barrier needed)
            _Float16* ob = out_s + ((size_t)b * S + q0) * (2 * d_model) + split_pos(h * DH);
#pragma unroll
            for (int pc = 0; pc < 16; ++pc) {
                const int row = 2 * pc + hi;
                const uint4 v = *reinterpret_cast<const uint4*>(ws + row * RSTR + l31 * 16);
                if (q0 + row < S)
                    *reinterpret_cast<uint4*>(reinterpret_cast<char*>(ob + (size_t)row * (2 * d_model)) + l31 * 16) = v;
            }
        }
        if (qok) {
            if constexpr (STASH) {
                if (hi == 0) {
                    row_stats[((size_t)bh * S + q) * 2] = m_run * 0.6931471805599453f;   // back to ln units
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
        }
    }
    if constexpr (PERSIST) {
        // the next pair's first stages go into the ring every wave has just used as its output staging area
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    }   // (sequence, head) pairs of this block
    if constexpr (!STASH) {
        if ((dbg & 16) && row_stats && tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            long long* o = reinterpret_cast<long long*>(row_stats) + (size_t)blockIdx.x * 4;
            o[0] = t1 - t0; o[1] = t2 - t1; o[2] = __builtin_readcyclecounter() - t2; o[3] = 0;
        }
    }
}

template <int NW, int NS, int STAG = 0, bool PERSIST = false>
static hipError_t launch_attention_h3_cfg(const _Float16* qkv_split, float* out, _Float16* out_split, int* range_flag,
                                          float* row_stats, int n_seq, int S, int H, int dbg, long head_rows,
                                          hipStream_t stream, int blocks = 0) {
    dim3 grid(PERSIST ? blocks : n_seq * H, (S + 32 * NW - 1) / (32 * NW));
    const float scale = 1.0f / sqrtf((float)DH);
    constexpr size_t lds_ring = (size_t)NS * STAGE, lds_epi = (size_t)NW * 32 * 528;   // K/V ring | output rows (epilogue)
    constexpr size_t lds = lds_ring > lds_epi ? lds_ring : lds_epi;
    static PerDevice<bool> attr_done_dev;
    bool& attr_done = attr_done_dev.get();
    if (!attr_done) {
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_h3_kernel<true, NW, NS, STAG, PERSIST>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_h3_kernel<false, NW, NS, STAG, PERSIST>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e1 != hipSuccess) return e1;
        if (e2 != hipSuccess) return e2;
        attr_done = true;
    }
    if (row_stats && !(dbg & 16))
        hipLaunchKernelGGL((attention_h3_kernel<true, NW, NS, STAG, PERSIST>), grid, dim3(64 * NW), lds, stream, qkv_split,
                           out, out_split, range_flag, row_stats, S, H, scale, dbg, head_rows, n_seq * H);
    else
        hipLaunchKernelGGL((attention_h3_kernel<false, NW, NS, STAG, PERSIST>), grid, dim3(64 * NW), lds, stream, qkv_split,
                           out, out_split, range_flag, row_stats, S, H, scale, dbg, head_rows, n_seq * H);
    return hipGetLastError();
}

// (bench-only: with dbg & 16 and STASH == false, row_stats receives 4 cycle stamps per block)
hipError_t launch_attention_h3(const _Float16* qkv_split, float* out, _Float16* out_split,
                               int* range_flag, float* row_stats, int n_seq, int S, int H,
                               hipStream_t stream, bool head_major) {
    if (S < 1 || S > 224) return hipErrorInvalidValue;
    const long head_rows = head_major ? (long)n_seq * S : 0;
#ifdef CMDI_PROBES
    static const int dbg = std::getenv("CMDI_ATTN_DBG") ? std::atoi(std::getenv("CMDI_ATTN_DBG")) : 0;
#else
    constexpr int dbg = 0;
#endif
    // schedule: CMDI_ATTN_SPLIT = 1 -> two 4-wave blocks per (sequence, head), two blocks per CU; 0 -> one 8-wave block;
    // unset -> by grid size (round 4): while the (sequence, head) pairs fill at most half the CUs (B <= 16 with CFG) the two
    // halves of a pair's queries run on two CUs, one wave per SIMD instead of two: 23.6 -> 19 us per layer at B = 2,
    // 25 -> 17 us at B = 10 (0.80 -> 0.76 and 0.99 -> 0.92 ms per step).  A wave computes its 32 queries with the same
    // instruction sequence in both schedules: same bits (test_attention_split_schedule_is_bitwise_identical).
    static const int split_env = std::getenv("CMDI_ATTN_SPLIT") ? std::atoi(std::getenv("CMDI_ATTN_SPLIT")) : kAttnSplitDefault;
    bool split = split_env > 0, auto_split = false;
    if (split_env < 0) {
        static PerDevice<int> cus_dev;
        int& cus = cus_dev.get();
        if (!cus) {
            int dev = 0, n = 256;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
            cus = n > 0 ? n : 256;
        }
        split = auto_split = 2L * n_seq * H <= cus;
    }
    // (by grid size: every half block has a CU — and its 160 KiB of LDS — to itself, so it keeps the 4-stage K/V ring of the
    // 8-wave form, three stages in flight instead of one; CMDI_ATTN_SPLIT=1 is the 2-stage form that fits twice per CU)
    if (auto_split && S > 128)
        return launch_attention_h3_cfg<4, 4>(qkv_split, out, out_split, range_flag, row_stats, n_seq, S, H, dbg, head_rows, stream);
    if (split && S > 128)
        return launch_attention_h3_cfg<4, 2>(qkv_split, out, out_split, range_flag, row_stats, n_seq, S, H, dbg, head_rows, stream);
#ifdef CMDI_PROBES
    static const int stag = std::getenv("CMDI_ATTN_STAG") ? std::atoi(std::getenv("CMDI_ATTN_STAG")) : kAttnStagDefault;
    if (stag == 0)
        return launch_attention_h3_cfg<NWAVE, NSTG, 0>(qkv_split, out, out_split, range_flag, row_stats, n_seq, S, H, dbg, head_rows, stream);
    if (stag == 2)
        return launch_attention_h3_cfg<NWAVE, NSTG, 2>(qkv_split, out, out_split, range_flag, row_stats, n_seq, S, H, dbg, head_rows, stream);
#endif
    // many pairs per CU (B >= 64 with CFG): one persistent block per CU walks its share — built and measured in round 6
    // (profiles/r06_attention_persistent.txt): at C4 (2,048 pairs) the kernel takes 296 us against 280 us for one block per pair
    // (the hardware hands a CU its next block as fast as the loop does, and the loop costs a barrier, 16 spilled row pointers per
    // pair and the serialisation of one pair's store tail with the next pair's Q loads), the step 14.38 vs 14.36 ms.  Off by
    // default; CMDI_ATTN_PERSIST=1 selects it (bitwise the default: test_attention_persistent_schedule_is_bitwise_identical).
    static const int persist_env = std::getenv("CMDI_ATTN_PERSIST") ? std::atoi(std::getenv("CMDI_ATTN_PERSIST")) : 0;
    const int cus_p = device_cu_count();
    if (persist_env && S > 128 && S <= 256 && (long)n_seq * H >= 2L * cus_p && !(dbg & 16))
        return launch_attention_h3_cfg<NWAVE, NSTG, kAttnStagDefault, true>(qkv_split, out, out_split, range_flag, row_stats, n_seq, S, H, dbg,
                                                                        head_rows, stream, cus_p);
    return launch_attention_h3_cfg<NWAVE, NSTG, kAttnStagDefault>(qkv_split, out, out_split, range_flag, row_stats, n_seq, S, H, dbg, head_rows, stream);
}

}  // namespace cmdi
