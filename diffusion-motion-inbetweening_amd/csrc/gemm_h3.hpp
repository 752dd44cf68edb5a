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
// Operand format ("split rows"): row r of a [rows, K] matrix is 2K halves, interleaved in chunks of
// 32 columns — [hi k0..31 | lo k0..31 | hi k32..63 | lo k32..63 | ...] — so a split matrix occupies
// exactly the bytes of its fp32 original and one K step of 32 columns is ONE contiguous 128-byte
// line per row holding both planes (full-line LDS-DMA fetches, 8 rows per 1-KiB wave-instruction).
//
// Epilogue: the accumulators (lane = one column of 16 rows per fragment) are transposed through the
// wave's private slice of the now idle LDS stages, 32 rows at a time, and leave as row-major float4:
// every global instruction (bias / residual loads, C stores) covers whole 128- or 256-byte row
// segments (split outputs: the hi and the lo half of whole 128-byte chunks).
#pragma once
#include "common.hpp"
#include "gemm_params.hpp"

namespace cmdi {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

// Bench-only instrumentation (ablations, cycle stamps) exists only in the probes build (build.py --probes,
// -DCMDI_PROBES -> libcondmdi_hip_probes.so, used by tools/); in the product library the expression is the constant 0
// and every branch on it is compiled out — no environment variable can make a shipped kernel skip work.
#ifdef CMDI_PROBES
#define CMDI_DBG(p) ((p).dbg)
#else
#define CMDI_DBG(p) 0
#endif
// Compile-time ablations of the K step (tools/probes/kstep.hip only; 0 everywhere else): 1 = no operand requests inside the
// loop, 2 = no MFMAs (the fragments are xor-ed into a sink), 4 = no fragment reads (loop-invariant registers), 8 = no
// end-of-step waits / barrier, 32 = requests behind the products instead of in front of the fragment reads
#ifndef CMDI_KABL
#define CMDI_KABL 0
#endif

constexpr float kLoScale = 2048.0f;          // 2^11
constexpr float kLoInv = 1.0f / 2048.0f;

// position (in halves) of column k's hi value inside a split row; its lo value is 32 further on
__host__ __device__ __forceinline__ int split_pos(int k) { return ((k >> 5) << 6) + (k & 31); }

__device__ __forceinline__ void split_f16(float x, _Float16& hi, _Float16& lo) {
    // x must be ONE rounded fp32 value for both lines below.  Without this barrier hipcc (-ffp-contract=fast) fuses a
    // product that feeds x into the conversions — the residual was taken against RN16(exact a*b) (v_fma_mixlo_f16 +
    // v_fma_mix_f32) while the stored hi was RN16(RN32(a*b)) (v_cvt_pk_f16_f32): whenever the two roundings part, hi and lo
    // belong to different splits and the value is off by a whole f16 ulp.  Found in round 2 behind the GELU epilogue
    // (rel-L2 1.65e-6 instead of 1.7e-7 on linear1, tools/x6_bench.py); every caller whose x is a product was exposed.
    asm volatile("" : "+v"(x));
    hi = (_Float16)x;                         // round to nearest even
    lo = (_Float16)((x - (float)hi) * kLoScale);
}

// The same for an operand that needs no lo-scaling (attention probabilities, 0 <= x <= 2^8): hi = f16(x), rest = f16(x - hi).
// EVERY split of a computed value goes through one of these two helpers — the pin is what keeps hi and its remainder on the
// same rounding of x (rounds 2 and 5 each found a hand-rolled split whose product hipcc had contracted into the conversions);
// tests/test_host_logic.py::test_split_conversions_are_pinned scans the compiled ISA for v_fma_mix*_f16 outside a whitelist.
__device__ __forceinline__ void split_f16_unscaled(float x, _Float16& hi, _Float16& rest) {
    asm volatile("" : "+v"(x));
    hi = (_Float16)x;
    rest = (_Float16)(x - (float)hi);
}

// Output stores carry `sc1` (device scope: written through the XCD's L2 instead of parked there as dirty lines).  The
// outputs of these GEMMs (26-78 MB) are never re-read by the kernel that writes them, and as dirty L2 lines they evict the
// weight panel every block of that XCD re-reads: measured at C2 (in-run counters of the in_proj GEMM) 187 -> 162 MB of L2
// misses + write-backs per launch (1.76 -> 1.52 x the algorithmic bytes), kernel -1 %, step -0.6 ... -2.8 % by box.
// `nt` stores are 16 % slower, `sc0 sc1` and `nt` operand requests slower still (profiles/r03_store_policy.txt).
// The persistent kernel (gemm_h3_epi.hpp) keeps plain stores: there sc1 costs 2 % (in_proj at B=256: 530 vs 520 us), and on
// the attention kernel's output rows it is neutral.  CMDI_OUT_SC1=0 (experiment builds, tools/st_policy_build.sh) restores
// plain stores here too.
// (Inline asm, because no builtin carries the scope bit on a 128-bit store.  The compiler's hazard recognizer does not look
// inside: a store of more than 8 bytes must not be followed at once by a VALU write of its data registers — it reads them
// a cycle late — hence the s_nop behind the dwordx4 form.  Found the hard way: with sc1 stores in the
// persistent kernel's epilogue its split rows came out wrong until the s_nop was there.)
#ifndef CMDI_OUT_SC1
#define CMDI_OUT_SC1 1
#endif
#ifndef CMDI_H3_SETPRIO
#define CMDI_H3_SETPRIO 1   // wave priority during the fragment reads + products of a K step (in_proj 75.7 -> 74.5 us: the waves
                            // in their product phase issue ahead of the co-resident block's epilogue / request / wait phases)
#endif
#ifndef CMDI_AUX_SC1
#define CMDI_AUX_SC1 1    // the stashing forward's pre-activation rows (reconstruction guidance) as well
#endif
#ifndef CMDI_A_AUX
#define CMDI_A_AUX 0     // cache policy bits of the A / W requests (gfx940+: 1 = sc0, 2 = nt, 16 = sc1)
#endif
#ifndef CMDI_W_AUX
#define CMDI_W_AUX 0
#endif
#ifndef CMDI_CONV_KORDER
#define CMDI_CONV_KORDER 1   // K order of a convolution's tap-shifted GEMM: 1 = chunk-major (taps of a chunk consecutive), 0 = tap-major
#endif
typedef float f4v_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void h3_store_f4(float* dst, float4 v) {
#if CMDI_OUT_SC1
    const f4v_ t = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(t) : "memory");
#else
    *reinterpret_cast<float4*>(dst) = v;
#endif
}
__device__ __forceinline__ void h3_store_h4(_Float16* dst, h4 v) {
#if CMDI_OUT_SC1
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
#else
    *reinterpret_cast<h4*>(dst) = v;
#endif
}

}  // namespace cmdi
#include "gemm_h3_epi.hpp"
namespace cmdi {

// (Round 2 measured four more K-loop schedules on this tile — requests between the trailing MFMAs, wait + barrier pinned behind
// the last MFMA, the K step rotated around its barrier, the A operand from registers; all lost to the one below and were
// removed in round 3: DESIGN.md "GEMM design".)
template <int BM_, int BN_, int WM_, int WN_, int NSTAGE_, int MINW_, int EPI8_ = 0, int LPS_ = 1>
struct H3Tile {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, NSTAGE = NSTAGE_, MINW = MINW_;
    static constexpr int EPI8 = EPI8_;             // interior tiles of the split epilogues without residual: 8 columns per lane
    // LPS (round 5, small M): 128-byte lines per row and ring stage.  A block that is alone on its CU pays ~1,000-1,400 cycles
    // per K step whatever the tile (requests, fragment reads and MFMAs of its lock-stepped waves add up: r04_kstep_ablation.md);
    // with LPS lines per stage it requests, waits and meets at the barrier once per LPS * 32 columns — a "fat" K step of LPS
    // ordinary ones back to back, each on its own 128-byte line of the row.  The products and their order per output are
    // untouched (k ascending in steps of 16): the same bits as every other tile.
    static constexpr int LPS = LPS_;
    static constexpr int BK = 32;                  // columns per K step = one 128-B line per row
    static constexpr int NW = WM * WN, NT = 64 * NW;
    static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static constexpr int STAGE = (BM + BN) * 128;  // bytes: A rows then W rows, 128 B each
    static constexpr int PW = (BM + BN) / 8 / NW;  // LDS-DMA pieces (1 KiB = 8 rows) per wave per stage
    static constexpr size_t EPI_BYTES = (size_t)NW * 32 * 32 * TN * 4 + 2 * WN * BM * 4;  // transpose + LN sums
    static constexpr size_t MAIN_BYTES = (size_t)NSTAGE * LPS * STAGE > EPI_BYTES ? (size_t)NSTAGE * LPS * STAGE : EPI_BYTES;
    static_assert(LPS == 1 || NSTAGE == 2, "fat K steps run on a ring of two stage groups");
    static constexpr size_t LDS_BYTES = MAIN_BYTES + (size_t)BM * 8;   // + (mean, rstd) of the tile's rows (folded LayerNorm)
    static_assert(NSTAGE == 2 || NSTAGE == 3, "NSTAGE");
    static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "wave tile must be 32-aligned");
    static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "DMA pieces per wave");
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Epilogue of one output tile whose fp32 accumulators sit in acc0 (hi·hi) / acc1 (cross terms): transposed through the
// wave's private slice of the idle LDS stages and written as row-major float4 / half4 (see the header).  Wave (wm, wn) of
// the WM x WN wave grid owns rows [wm TM 32, +TM 32) x columns [wn TN 32, +TN 32) of the tile at (m0, n0).
template <class TC, int EPI>
__device__ __forceinline__ void h3_epilogue(const H3Params& p, f32x16 (&acc0)[TC::TM][TC::TN], f32x16 (&acc1)[TC::TM][TC::TN],
                                            int m0, int n0, int M, int kslice, char* lds) {
    constexpr int BM = TC::BM, BN = TC::BN, TM = TC::TM, TN = TC::TN, NW = TC::NW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31;
    const int wm = wave / TC::WN, wn = wave % TC::WN;
    const float2* row_stats = reinterpret_cast<const float2*>(lds + TC::MAIN_BYTES);
    (void)BM; (void)BN; (void)NW; (void)l31; (void)wm; (void)tid;
    // ---- epilogue ---------------------------------------------------------------------------------
    bool overflow = false;
    if ((CMDI_DBG(p) & 2) && acc0[0][0][0] != 12345.678f) return;
    if constexpr (EPI == H3_RESID_LN) {
        // y = LayerNorm((v + bias) + R) over the full row (the block tile spans all N = BN columns):
        // nn.TransformerEncoderLayer.norm1/norm2 (post-norm, eps 1e-5, biased variance, two-pass as
        // in layernorm_kernel).  Row sums: 32 lanes of a row by shuffles, the WN waves through LDS.
        static_assert(TM == 1, "LN epilogue: one 32-row fragment per wave");
        constexpr int ROWLEN = 32 * TN, LPR = ROWLEN / 4, RPI = 64 / LPR, NIT = 32 / RPI;
        static_assert(LPR == 32, "LN epilogue written for 128-column wave tiles");
        float* wl = reinterpret_cast<float*>(lds) + wave * (32 * ROWLEN);
        float* red = reinterpret_cast<float*>(lds) + NW * (32 * ROWLEN);   // [2][WN][BM]
        const int rl = lane / LPR, cl = (lane % LPR) * 4;
        const int n = wn * ROWLEN + cl;
        const float4 bias4 = p.bias ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 g4 = *reinterpret_cast<const float4*>(p.ln_g + n);
        const float4 b4 = *reinterpret_cast<const float4*>(p.ln_b + n);
        const float inv_n = 1.0f / (float)BN;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                wl[mfma32_row(r, lane) * ROWLEN + j * 32 + l31] = acc0[0][j][r] + acc1[0][j][r] * kLoInv;
        float x[NIT][4];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = it * RPI + rl;
            int m = m0 + wm * 32 + row;
            const bool mok = m < M;
            m = mok ? m : M - 1;
            const float4 t = *reinterpret_cast<const float4*>(wl + row * ROWLEN + cl);
            const float4 rr = *reinterpret_cast<const float4*>(p.R + (size_t)m * BN + n);
            x[it][0] = (t.x + bias4.x) + rr.x; x[it][1] = (t.y + bias4.y) + rr.y;
            x[it][2] = (t.z + bias4.z) + rr.z; x[it][3] = (t.w + bias4.w) + rr.w;
            if (p.aux && mok)
                *reinterpret_cast<float4*>(p.aux + (size_t)m * BN + n) = make_float4(x[it][0], x[it][1], x[it][2], x[it][3]);
            float sm = (x[it][0] + x[it][1]) + (x[it][2] + x[it][3]);
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
            if ((lane & 31) == 0) red[wn * BM + wm * 32 + row] = sm;
        }
        __syncthreads();
        float mean[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = wm * 32 + it * RPI + rl;
            float sm = 0.f;
#pragma unroll
            for (int w = 0; w < TC::WN; ++w) sm += red[w * BM + row];
            mean[it] = sm * inv_n;
            const float a = x[it][0] - mean[it], b = x[it][1] - mean[it], c = x[it][2] - mean[it], d = x[it][3] - mean[it];
            float q = (a * a + b * b) + (c * c + d * d);
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
            if ((lane & 31) == 0) red[(TC::WN + wn) * BM + row] = q;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = wm * 32 + it * RPI + rl;
            const int m = m0 + row;
            float q = 0.f;
#pragma unroll
            for (int w = 0; w < TC::WN; ++w) q += red[(TC::WN + w) * BM + row];
            const float rstd = 1.0f / sqrtf(q * inv_n + 1e-5f);
            if (m >= M) continue;
            if (p.ln_stats && wn == 0 && (lane & 31) == 0) {
                p.ln_stats[2 * (size_t)m] = mean[it];
                p.ln_stats[2 * (size_t)m + 1] = rstd;
            }
            float y[4];
            y[0] = (x[it][0] - mean[it]) * rstd * g4.x + b4.x;
            y[1] = (x[it][1] - mean[it]) * rstd * g4.y + b4.y;
            y[2] = (x[it][2] - mean[it]) * rstd * g4.z + b4.z;
            y[3] = (x[it][3] - mean[it]) * rstd * g4.w + b4.w;
            *reinterpret_cast<float4*>(p.C + (size_t)m * BN + n) = make_float4(y[0], y[1], y[2], y[3]);
            if (p.Cs) {
                h4 oh, ol;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    _Float16 a, b;
                    split_f16(y[e], a, b);
                    oh[e] = a; ol[e] = b;
                    overflow |= !(fabsf(y[e]) < 65504.0f);
                }
                _Float16* dst = p.Cs + (size_t)m * (2 * BN) + split_pos(n);
                h3_store_h4(dst, oh);
                h3_store_h4(dst + 32, ol);
            }
        }
        if (overflow && p.range_flag) atomicOr(p.range_flag, 1);
    } else {
        constexpr int ROWLEN = 32 * TN;          // floats per row of the wave's tile
        constexpr int LPR = ROWLEN / 4;          // lanes per row (float4 each)
        constexpr int RPI = 64 / LPR;            // rows per wave-instruction
        float* wl = reinterpret_cast<float*>(lds) + wave * (32 * ROWLEN);
        const int rl = lane / LPR, cl = (lane % LPR) * 4;
        const int n = n0 + wn * ROWLEN + cl;     // first of this lane's 4 consecutive columns
        const bool nok = n < p.N;
        float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (EPI != H3_MOTION && EPI != H3_CONV_GN && p.bias && nok && kslice == 0) bias4 = *reinterpret_cast<const float4*>(p.bias + n);
        const int npos = split_pos(n);
        // folded LayerNorm (gemm_params.hpp): column sums of the gamma-folded weights, gamma / beta of a normalised residual
        float4 c14 = make_float4(0.f, 0.f, 0.f, 0.f), rg4 = c14, rb4 = c14;
        if (p.ln_c1 && nok) c14 = *reinterpret_cast<const float4*>(p.ln_c1 + n);
        if constexpr (EPI == H3_RESID) {
            if (p.ln_rg && nok) {
                rg4 = *reinterpret_cast<const float4*>(p.ln_rg + n);
                rb4 = *reinterpret_cast<const float4*>(p.ln_rb + n);
            }
        }
        int mo_b = 0, mo_s = 0;                  // H3_MOTION: (sequence, token) of this lane's first column
        if constexpr (EPI == H3_MOTION) { mo_b = n / p.tok_S; mo_s = n - mo_b * p.tok_S; }
        if constexpr (EPI == H3_CONV_GN) {
            // GroupNorm over (gn_cg channels) x (the sequence's valid frames), statistics straight from the fp32
            // accumulators: the tile is ONE framed sequence (BM == tp) and a wave's 32*TN columns lie in one group
            __shared__ float red[2][16];
            const int grp = (wn * ROWLEN) / p.gn_cg;               // this wave's group inside the tile
            const float inv_n = 1.0f / (float)((p.t_hi - p.t_lo) * p.gn_cg);
            float cb[TN], cg_[TN], cbeta[TN], csc[TN], csh[TN];
            const int seq = m0 / p.tp;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * ROWLEN + j * 32 + l31;
                const bool ok = col < p.N;
                cb[j] = ok && p.bias ? p.bias[col] : 0.f;
                cg_[j] = ok ? p.ln_g[col] : 0.f;
                cbeta[j] = ok ? p.ln_b[col] : 0.f;
                csc[j] = ok && p.gn_ss ? p.gn_ss[(size_t)seq * p.gn_ss_ld + col] : 0.f;
                csh[j] = ok && p.gn_ss ? p.gn_ss[(size_t)seq * p.gn_ss_ld + p.N + col] : 0.f;
            }
            float s1 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (wm * TM + i) * 32 + mfma32_row(r, lane);
                    const bool valid = row >= p.t_lo && row < p.t_hi;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float v = acc0[i][j][r] + acc1[i][j][r] * kLoInv + cb[j];
                        acc0[i][j][r] = v;
                        if (valid) s1 += v;
                    }
                }
            s1 = wave_sum(s1);
            if (lane == 0) red[0][wave] = s1;
            __syncthreads();
            float mean = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w)
                if (((w % TC::WN) * ROWLEN) / p.gn_cg == grp) mean += red[0][w];
            mean *= inv_n;
            float s2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (wm * TM + i) * 32 + mfma32_row(r, lane);
                    if (row >= p.t_lo && row < p.t_hi) {
#pragma unroll
                        for (int j = 0; j < TN; ++j) { const float d = acc0[i][j][r] - mean; s2 += d * d; }
                    }
                }
            s2 = wave_sum(s2);
            if (lane == 0) red[1][wave] = s2;
            __syncthreads();
            float var = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w)
                if (((w % TC::WN) * ROWLEN) / p.gn_cg == grp) var += red[1][w];
            const float rstd = 1.0f / sqrtf(var * inv_n + 1e-5f);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float y = (acc0[i][j][r] - mean) * rstd * cg_[j] + cbeta[j];
                        if (p.gn_ss) y = y * (1.f + csc[j]) + csh[j];
                        acc0[i][j][r] = mish_f(y);
                        acc1[i][j][r] = 0.f;
                    }
            bias4 = make_float4(0.f, 0.f, 0.f, 0.f);   // already inside the normalised values
        }
        const bool fast = !p.c_row_mul && !p.tp && m0 + BM <= M && n0 + BN <= p.N;
        (void)fast;
        if constexpr (EPI == H3_PLAIN_SPLIT || EPI == H3_GELU_SPLIT) {
            // interior tiles, round 3: 32 x 32 blocks with EIGHT columns per lane (gemm_h3_epi.hpp) — the hi and the lo half of a
            // split row leave as one dwordx4 store each instead of two dwordx2 (half the store instructions through the
            // address path this block's neighbour on the CU is feeding its LDS-DMA requests through).  Same arithmetic per
            // element, same bits.
            if (TC::EPI8 && fast && !p.cs_head_major) {
                float* wb = reinterpret_cast<float*>(lds) + wave * (32 * ROWLEN);
                const int cl8 = (lane & 3) * 8;
                bool ovf = false;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int nb = n0 + wn * ROWLEN + j * 32;
                        const H3PCols cols = h3p_load_cols<EPI>(p, nb + cl8);
                        H3PRows rows;
                        h3p_epi_block<EPI, false, false, true>(p, acc0[i][j], acc1[i][j], m0 + (wm * TM + i) * 32, nb,
                                                               p.ln_part ? row_stats + (wm * TM + i) * 32 : nullptr, wb, lane, cols,
                                                               rows, ovf);
                    }
                if (ovf && p.range_flag) atomicOr(p.range_flag, 1);
                return;
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    wl[mfma32_row(r, lane) * ROWLEN + j * 32 + l31] = acc0[i][j][r] + acc1[i][j][r] * kLoInv;
            if constexpr (EPI == H3_PLAIN || EPI == H3_PLAIN_SPLIT || EPI == H3_GELU_SPLIT || EPI == H3_GELUGRAD_SPLIT ||
                          EPI == H3_RESID) {
                // Interior tile of a plain GEMM (every transformer launch but its last row panel): no per-row predicates, and
                // every read — the transposed rows, their statistics, the residual rows — is issued before the first store.
                // (With stores in flight the compiler waits vmcnt(0) in front of each LDS read, because the K loop's LDS-DMA
                // shares the counter: row-by-row that serialised the epilogue on the store latency.)  Same arithmetic, same
                // bits as the general loop below.
                if (fast) {
                    // (the residual epilogue goes in two batches of rows: its prefetched operands would not fit the 128 VGPRs
                    // that 4 waves per SIMD leave)
                    constexpr int NBATCH = EPI == H3_RESID ? 2 : 1, NIT = 32 / RPI / NBATCH;
#pragma unroll
                    for (int bt = 0; bt < NBATCH; ++bt) {
                    const int rbase = (wm * TM + i) * 32 + bt * NIT * RPI + rl;
                    float4 tt[NIT];
                    float2 rst[NIT];
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        tt[it] = *reinterpret_cast<const float4*>(wl + ((bt * NIT + it) * RPI + rl) * ROWLEN + cl);
                        rst[it] = p.ln_part ? row_stats[rbase + it * RPI] : make_float2(0.f, 1.f);
                    }
                    [[maybe_unused]] h4 rh[NIT], rlo[NIT];
                    [[maybe_unused]] float4 rr[NIT];
                    if constexpr (EPI == H3_RESID) {
                        if (p.Rs) {
#pragma unroll
                            for (int it = 0; it < NIT; ++it) {
                                const _Float16* rs = p.Rs + (size_t)(m0 + rbase + it * RPI) * (2 * p.N) + npos;
                                rh[it] = *reinterpret_cast<const h4*>(rs);
                                rlo[it] = *reinterpret_cast<const h4*>(rs + 32);
                            }
                        } else {
#pragma unroll
                            for (int it = 0; it < NIT; ++it) {
                                const size_t m = (size_t)(m0 + rbase + it * RPI);
                                rr[it] = *reinterpret_cast<const float4*>(p.R + (p.r_ld ? m * p.r_ld + n : m * p.ldc + n));
                            }
                        }
                    }
                    if constexpr (EPI == H3_GELUGRAD_SPLIT) {
#pragma unroll
                        for (int it = 0; it < NIT; ++it)
                            rr[it] = *reinterpret_cast<const float4*>(p.aux + (size_t)(m0 + rbase + it * RPI) * p.ldc + n);
                    }
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const int m = m0 + rbase + it * RPI;
                        const float4 t = tt[it];
                        const float2 rs2 = rst[it];
                        float v[4] = {t.x + bias4.x, t.y + bias4.y, t.z + bias4.z, t.w + bias4.w};
                        if (p.ln_c1) {
                            v[0] = __builtin_fmaf(rs2.y, __builtin_fmaf(-rs2.x, c14.x, t.x), bias4.x);
                            v[1] = __builtin_fmaf(rs2.y, __builtin_fmaf(-rs2.x, c14.y, t.y), bias4.y);
                            v[2] = __builtin_fmaf(rs2.y, __builtin_fmaf(-rs2.x, c14.z, t.z), bias4.z);
                            v[3] = __builtin_fmaf(rs2.y, __builtin_fmaf(-rs2.x, c14.w, t.w), bias4.w);
                        }
                        const size_t off = (size_t)m * p.ldc + n;
                        if constexpr (EPI == H3_PLAIN) {
                            *reinterpret_cast<float4*>(p.C + (size_t)kslice * p.slice_stride + off) = make_float4(v[0], v[1], v[2], v[3]);
                            continue;
                        }
                        if constexpr (EPI == H3_RESID) {
                            if (p.Rs) {
                                float r4[4];
#pragma unroll
                                for (int e = 0; e < 4; ++e) r4[e] = (float)rh[it][e] + (float)rlo[it][e] * kLoInv;
                                if (p.ln_rg) {
                                    r4[0] = __builtin_fmaf((r4[0] - rs2.x) * rs2.y, rg4.x, rb4.x);
                                    r4[1] = __builtin_fmaf((r4[1] - rs2.x) * rs2.y, rg4.y, rb4.y);
                                    r4[2] = __builtin_fmaf((r4[2] - rs2.x) * rs2.y, rg4.z, rb4.z);
                                    r4[3] = __builtin_fmaf((r4[3] - rs2.x) * rs2.y, rg4.w, rb4.w);
                                }
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] += r4[e];
                            } else {
                                v[0] += rr[it].x; v[1] += rr[it].y; v[2] += rr[it].z; v[3] += rr[it].w;
                            }
                            if (p.C) h3_store_f4(p.C + off, make_float4(v[0], v[1], v[2], v[3]));
                        }
                        if constexpr (EPI == H3_GELUGRAD_SPLIT) {
                            v[0] *= gelu_erf_grad(rr[it].x); v[1] *= gelu_erf_grad(rr[it].y);
                            v[2] *= gelu_erf_grad(rr[it].z); v[3] *= gelu_erf_grad(rr[it].w);
                        } else if constexpr (EPI != H3_RESID) {
#if CMDI_AUX_SC1
                            if (p.aux) h3_store_f4(p.aux + off, make_float4(v[0], v[1], v[2], v[3]));
#else
                            if (p.aux) *reinterpret_cast<float4*>(p.aux + off) = make_float4(v[0], v[1], v[2], v[3]);
#endif
                        }
                        if constexpr (EPI == H3_GELU_SPLIT) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                        }
                        if (EPI != H3_RESID || p.Cs) {
                            h4 oh, ol;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                _Float16 a, b;
                                split_f16(v[e], a, b);
                                oh[e] = a; ol[e] = b;
                                overflow |= !(fabsf(v[e]) < 65504.0f);
                            }
                            _Float16* dst = p.Cs + (size_t)m * (p.cs_ld ? p.cs_ld : 2 * p.N) + npos;
                            if constexpr (EPI == H3_PLAIN_SPLIT) {
                                if (p.cs_head_major) dst = p.Cs + ((size_t)(n >> 7) * M + m) * 256 + split_pos(n & 127);
                            }
                            h3_store_h4(dst, oh);
                            h3_store_h4(dst + 32, ol);
                        }
                        if constexpr (EPI == H3_RESID) {
                            if (p.out_part) {
                                float sm = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
                                for (int o = 1; o < 8; o <<= 1) sm += __shfl_xor(sm, o, 64);
                                const float mb = sm * (1.0f / 32.0f);
                                const float d0 = v[0] - mb, d1 = v[1] - mb, d2 = v[2] - mb, d3 = v[3] - mb;
                                float q = __builtin_fmaf(d0, d0, d1 * d1) + __builtin_fmaf(d2, d2, d3 * d3);
#pragma unroll
                                for (int o = 1; o < 8; o <<= 1) q += __shfl_xor(q, o, 64);
                                if ((lane & 7) == 0)
                                    *reinterpret_cast<float2*>(p.out_part + ((size_t)m * 16 + (n >> 5)) * 2) = make_float2(sm, q);
                            }
                        }
                    }
                    }
                    continue;
                }
            }
#pragma unroll
            for (int it = 0; it < 32 / RPI; ++it) {
                const int row = it * RPI + rl;
                int m = m0 + (wm * TM + i) * 32 + row;
                const float4 t = *reinterpret_cast<const float4*>(wl + row * ROWLEN + cl);
                if (m >= M || !nok) continue;
                if (p.c_row_mul) m = m * p.c_row_mul + p.c_row_add;        // convolution: output row
                if (p.tp) {                                                // ... and only frames, not halo
                    const int pos = m % p.tp;
                    if (pos < p.t_lo || pos >= p.t_hi) continue;
                }
                float v[4] = {t.x + bias4.x, t.y + bias4.y, t.z + bias4.z, t.w + bias4.w};
                float2 rst = make_float2(0.f, 1.f);                        // (mean, rstd) of this row's LayerNorm input
                if (p.ln_part) rst = row_stats[(wm * TM + i) * 32 + row];
                if (p.ln_c1) {   // A operand was the raw P: LN(P) W^T + b = rstd (P W'^T - mean c1) + c2
                    v[0] = __builtin_fmaf(rst.y, __builtin_fmaf(-rst.x, c14.x, t.x), bias4.x);
                    v[1] = __builtin_fmaf(rst.y, __builtin_fmaf(-rst.x, c14.y, t.y), bias4.y);
                    v[2] = __builtin_fmaf(rst.y, __builtin_fmaf(-rst.x, c14.z, t.z), bias4.z);
                    v[3] = __builtin_fmaf(rst.y, __builtin_fmaf(-rst.x, c14.w, t.w), bias4.w);
                }
                if constexpr (EPI == H3_TOKENS) {
                    const int b = m / p.tok_T, fr = m - b * p.tok_T;
                    if (p.pe) {
                        const float4 pe4 = *reinterpret_cast<const float4*>(p.pe + (size_t)(1 + fr) * p.N + n);
                        v[0] += pe4.x; v[1] += pe4.y; v[2] += pe4.z; v[3] += pe4.w;
                    }
                    if (p.C) *reinterpret_cast<float4*>(p.C + ((size_t)b * p.tok_S + 1 + fr) * p.N + n) = make_float4(v[0], v[1], v[2], v[3]);
                    if (!p.Cs) continue;
                    h4 oh, ol;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        _Float16 a, b2;
                        split_f16(v[e], a, b2);
                        oh[e] = a; ol[e] = b2;
                        overflow |= !(fabsf(v[e]) < 65504.0f);
                    }
                    _Float16* dst = p.Cs + ((size_t)b * p.tok_S + 1 + fr) * (2 * p.N) + npos;
                    h3_store_h4(dst, oh);
                    h3_store_h4(dst + 32, ol);
                    if (p.tok_dup) {
                        dst += (size_t)p.tok_dup * p.tok_S * (2 * p.N);
                        h3_store_h4(dst, oh);
                        h3_store_h4(dst + 32, ol);
                    }
                    continue;
                }
                if constexpr (EPI == H3_MOTION) {
                    const float bm = p.bias ? p.bias[m] : 0.f;
                    const float unscale = p.gs_bits ? 1.0f / grad_scale_from_bits(*p.gs_bits) : 1.0f;   // exact: a power of two
                    int b = mo_b, sq = mo_s;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (n + e < p.N && sq > 0)
                            p.C[((size_t)b * M + m) * p.tok_T + (sq - 1)] = (v[e] + bm) * unscale;
                        if (++sq == p.tok_S) { sq = 0; ++b; }
                    }
                    continue;
                }
                const size_t off = (size_t)m * p.ldc + n;
                // (non-temporal stores were measured: faster in isolation, slower in the layer chain —
                // the next kernel then finds its input in HBM instead of the memory-side cache)
                if constexpr (EPI == H3_PLAIN) {
                    *reinterpret_cast<float4*>(p.C + (size_t)kslice * p.slice_stride + off) = make_float4(v[0], v[1], v[2], v[3]);
                } else if constexpr (EPI == H3_CONV_GN) {
                    if (p.R) {
                        const float4 rr = *reinterpret_cast<const float4*>(p.R + (p.r_ld ? (size_t)m * p.r_ld + n : off));
                        v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
                    }
                    if (p.C) h3_store_f4(p.C + off, make_float4(v[0], v[1], v[2], v[3]));
                    if (p.Cs) {
                        h4 oh, ol;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            _Float16 a, b;
                            split_f16(v[e], a, b);
                            oh[e] = a; ol[e] = b;
                            overflow |= !(fabsf(v[e]) < 65504.0f);
                        }
                        _Float16* dst = p.Cs + (size_t)m * (p.cs_ld ? p.cs_ld : 2 * p.N) + npos;
                        h3_store_h4(dst, oh);
                        h3_store_h4(dst + 32, ol);
                    }
                } else if constexpr (EPI == H3_RESID) {
                    if (p.Rs) {
                        const _Float16* rs = p.Rs + (size_t)m * (2 * p.N) + npos;
                        const h4 rh = *reinterpret_cast<const h4*>(rs), rl = *reinterpret_cast<const h4*>(rs + 32);
                        float r4[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) r4[e] = (float)rh[e] + (float)rl[e] * kLoInv;
                        if (p.ln_rg) {   // the residual is LayerNorm(P) of the rows read: (x - mean) * rstd * gamma + beta
                            r4[0] = __builtin_fmaf((r4[0] - rst.x) * rst.y, rg4.x, rb4.x);
                            r4[1] = __builtin_fmaf((r4[1] - rst.x) * rst.y, rg4.y, rb4.y);
                            r4[2] = __builtin_fmaf((r4[2] - rst.x) * rst.y, rg4.z, rb4.z);
                            r4[3] = __builtin_fmaf((r4[3] - rst.x) * rst.y, rg4.w, rb4.w);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += r4[e];
                    } else {
                        const float4 rr = *reinterpret_cast<const float4*>(p.R + (p.r_ld ? (size_t)m * p.r_ld + n : off));
                        v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
                    }
                    if (p.C) h3_store_f4(p.C + off, make_float4(v[0], v[1], v[2], v[3]));
                    if (p.Cs) {
                        h4 oh, ol;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            _Float16 a, b;
                            split_f16(v[e], a, b);
                            oh[e] = a; ol[e] = b;
                            overflow |= !(fabsf(v[e]) < 65504.0f);
                        }
                        _Float16* dst = p.Cs + (size_t)m * (p.cs_ld ? p.cs_ld : 2 * p.N) + npos;
                        h3_store_h4(dst, oh);
                        h3_store_h4(dst + 32, ol);
                    }
                    if (p.out_part) {
                        // partial LayerNorm statistics of the row just written, over this lane group's 32 columns (8 lanes
                        // x 4): sum, then the squared deviations from the block mean (two passes, as layernorm_kernel)
                        float sm = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
                        for (int o = 1; o < 8; o <<= 1) sm += __shfl_xor(sm, o, 64);
                        const float mb = sm * (1.0f / 32.0f);
                        const float d0 = v[0] - mb, d1 = v[1] - mb, d2 = v[2] - mb, d3 = v[3] - mb;
                        float q = __builtin_fmaf(d0, d0, d1 * d1) + __builtin_fmaf(d2, d2, d3 * d3);
#pragma unroll
                        for (int o = 1; o < 8; o <<= 1) q += __shfl_xor(q, o, 64);
                        if ((lane & 7) == 0)
                            *reinterpret_cast<float2*>(p.out_part + ((size_t)m * 16 + (n >> 5)) * 2) = make_float2(sm, q);
                    }
                } else {
                    if constexpr (EPI == H3_GELUGRAD_SPLIT) {
                        const float4 ax = *reinterpret_cast<const float4*>(p.aux + off);
                        v[0] *= gelu_erf_grad(ax.x); v[1] *= gelu_erf_grad(ax.y);
                        v[2] *= gelu_erf_grad(ax.z); v[3] *= gelu_erf_grad(ax.w);
                    } else if (p.aux) {
                        *reinterpret_cast<float4*>(p.aux + off) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                    if constexpr (EPI == H3_GELU_SPLIT) {
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
                    _Float16* dst = p.Cs + (size_t)m * (p.cs_ld ? p.cs_ld : 2 * p.N) + npos;
                    if constexpr (EPI == H3_PLAIN_SPLIT) {
                        if (p.cs_head_major) dst = p.Cs + ((size_t)(n >> 7) * M + m) * 256 + split_pos(n & 127);
                    }
                    h3_store_h4(dst, oh);
                    h3_store_h4(dst + 32, ol);
                }
            }
        }
    }
    if constexpr (EPI == H3_GELU_SPLIT || EPI == H3_PLAIN_SPLIT || EPI == H3_GELUGRAD_SPLIT || EPI == H3_RESID ||
                  EPI == H3_TOKENS || EPI == H3_CONV_GN) {
        if (overflow && p.range_flag) atomicOr(p.range_flag, 1);
    }
}

// One output tile of the row range [m_begin, m_end): block `block_id` of the ceil(rows / BM) x ceil(N / BN)
// grid laid over that range.
template <class TC, int EPI>
__device__ __forceinline__ void gemm_h3_body(const H3Params& p, int block_id, int m_begin, int M, char* lds) {
    constexpr int BM = TC::BM, BN = TC::BN, TM = TC::TM, TN = TC::TN, NW = TC::NW;
    constexpr int STAGE = TC::STAGE, NSTAGE = TC::NSTAGE, PW = TC::PW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave / TC::WN, wn = wave % TC::WN;

    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (M - m_begin + BM - 1) / BM;
    int kslice = 0, nslice = 1;                       // split-K (H3_PLAIN): slice of the K loop this block owns
    if constexpr (EPI == H3_PLAIN) {
        if (p.ksplit > 1) { nslice = p.ksplit; kslice = block_id % nslice; block_id /= nslice; }
    }
    const int bid = xcd_remap(block_id, tiles_m * tiles_n);
    const int m0 = m_begin + (p.m_fast ? bid % tiles_m : bid / tiles_n) * BM, n0 = (p.m_fast ? bid / tiles_m : bid % tiles_n) * BN;

    long long t_start = 0, t_loop = 0, t_loop_end = 0, r_start = 0;
    if (CMDI_DBG(p) & 16) { t_start = __builtin_readcyclecounter(); r_start = __builtin_amdgcn_s_memrealtime(); }
    f32x16 acc0[TM][TN], acc1[TM][TN];   // [A fragment (rows m)][W fragment (cols n)]
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }

    // ---- LDS-DMA staging: HBM/L2 -> LDS in lane-linear 1-KiB pieces (8 rows x 128 B).  The bank
    // swizzle sits on the SOURCE address: the 16-B slot c of tile row r is stored at slot position
    // c ^ ((r >> 1) & 7), so the 16 rows of a ds_read_b128 lane group cover all 16 slots of a 256-B
    // bank row.  Slots 0-3 of a row are the hi plane, 4-7 the lo plane.
    // Requests are `buffer_load_dwordx4 ... offen lds` (round 3): operand base in a buffer descriptor, the lane's byte offset
    // in ONE 32-bit register per piece that never changes, the K step (and the convolution tap) in the scalar offset — one M0
    // write + one request per piece.  (Rounds 1-2 used flat global_load_lds with a 64-bit address per lane and piece: two
    // 64-bit vector adds per request, and twice the address registers in a kernel capped at 128.)  The operands are required
    // to span < 2 GiB (launch_gemm_h3 checks): offsets are 32-bit.
    const int prow = lane >> 3, pslot = lane & 7;
    const size_t ldk = 2 * (size_t)p.K;
    const size_t lda = p.a_ld ? (size_t)p.a_ld : ldk;
    const int a_rmul = p.a_row_mul ? p.a_row_mul : 1;
    // (a convolution caller has moved p.A back by the padding rows; base + offset is the address the pointer arithmetic of the
    // flat form produced)
    const auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.A), 0, 0xffffffff, 0x00020000);
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.W), 0, 0xffffffff, 0x00020000);
    unsigned a_voff[BM / 8 / NW], w_voff[BN / 8 / NW];
#pragma unroll
    for (int q = 0; q < BM / 8 / NW; ++q) {
        const int row = (q * NW + wave) * 8 + prow;
        int grow = m0 + row;
        grow = grow < M ? grow : M - 1;
        a_voff[q] = (unsigned)(((size_t)grow * a_rmul * lda + ((pslot ^ ((row >> 1) & 7)) << 3)) * 2);
    }
#pragma unroll
    for (int q = 0; q < BN / 8 / NW; ++q) {
        const int row = (q * NW + wave) * 8 + prow;
        int grow = n0 + row;
        grow = grow < p.N ? grow : p.N - 1;
        w_voff[q] = (unsigned)(((size_t)grow * ldk + ((pslot ^ ((row >> 1) & 7)) << 3)) * 2);
    }
    // plain GEMM: chunk kt of the row.  Convolution (round 5, CHUNK-MAJOR K order): K step kt = chunk * taps + tap reads the
    // 32-channel chunk `kt / taps` of the row `kt % taps` frames further on, so the taps of one chunk are CONSECUTIVE K steps —
    // the shifted re-reads of the same A lines come back within `taps` steps and hit the L2 (tap-major order, rounds 1-4:
    // a tap's re-read came cpt = 32 K steps and a 21-MB weight stream later, and every tap fetched A from HBM again —
    // 682 MB per level-0 launch for 140 MB of distinct bytes).  Weights are packed [Cout][chunk][tap][32] to match
    // (unet.hip pack_conv_w_kernel).  Branch-free for the plain case (taps = 1): the requests sit in the same scheduling
    // region as the MFMAs around them.  CMDI_CONV_KORDER=0 (experiment builds): the tap-major order of rounds 1-4.
#if CMDI_CONV_KORDER
    const int ktaps = p.cpt && p.taps > 0 ? p.taps : 1;
#else
    const int cpt = p.cpt ? p.cpt : 0x40000000;
#endif
    auto issue = [&](int kt, int buf) {
        char* stage = lds + buf * STAGE;
#if CMDI_CONV_KORDER
        const int chunk = kt / ktaps, tap = kt - chunk * ktaps;
        const int a_soff = (int)(((size_t)tap * lda + (size_t)chunk * 64) * 2);
#else
        const int tap = kt / cpt;
        const int a_soff = (int)(((size_t)tap * lda + (size_t)(kt - tap * cpt) * 64) * 2);
#endif
#pragma unroll
        for (int q = 0; q < BM / 8 / NW; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (__attribute__((address_space(3))) void*)(stage + (q * NW + wave) * 1024),
                                                     16, (int)a_voff[q], a_soff, 0, CMDI_A_AUX);
#pragma unroll
        for (int q = 0; q < BN / 8 / NW; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(stage + BM * 128 + (q * NW + wave) * 1024),
                                                     16, (int)w_voff[q], kt * 128, 0, CMDI_W_AUX);
    };

    const int swz = (l31 >> 1) & 7;
    int off_hi[2], off_lo[2];   // byte offset inside a tile row of this lane's slot for k-substep ks
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        off_hi[ks] = ((2 * ks + hi) ^ swz) * 16;
        off_lo[ks] = ((4 + 2 * ks + hi) ^ swz) * 16;
    }
    const int a_row = (wm * TM * 32 + l31) * 128;            // + i * 32 * 128
    const int w_row = BM * 128 + (wn * TN * 32 + l31) * 128;  // + j * 32 * 128

    // folded LayerNorm: (mean, rstd) of the tile's rows from the 16 partials of each, once per block, under the first loads
    float2* row_stats = reinterpret_cast<float2*>(lds + TC::MAIN_BYTES);
    if (p.ln_part && tid < BM) {
        int grow = m0 + tid;
        grow = grow < M ? grow : M - 1;
        const float4* pp = reinterpret_cast<const float4*>(p.ln_part + (size_t)grow * 32);
        // (every product-sum below is an explicit fmaf: the same bits in every instantiation of this template — samples
        // must not depend on the tile shape their batch size selects)
        float mean_b[16], m2 = 0.f, mean = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 v = pp[q];                    // (sum, M2) of blocks 2q, 2q + 1
            mean_b[2 * q] = v.x * (1.0f / 32.0f); mean_b[2 * q + 1] = v.z * (1.0f / 32.0f);
            m2 += v.y + v.w;
            mean += v.x + v.z;
        }
        mean *= (1.0f / 512.0f);
#pragma unroll
        for (int q = 0; q < 16; ++q) { const float dq = mean_b[q] - mean; m2 = __builtin_fmaf(32.0f * dq, dq, m2); }
        const float2 ms = make_float2(mean, 1.0f / sqrtf(__builtin_fmaf(m2, 1.0f / 512.0f, 1e-5f)));
        row_stats[tid] = ms;
        // a stashing forward pass (reconstruction guidance) keeps (mean, rstd) of every LayerNorm for the backward: the blocks
        // of the first column panel write them out
        if (p.ln_stats && n0 == 0 && m0 + tid < M) *reinterpret_cast<float2*>(p.ln_stats + 2 * (size_t)(m0 + tid)) = ms;
    }

    const int nk_all = p.K / 32;
    const int kt0 = (int)((long)nk_all * kslice / nslice), nk = (int)((long)nk_all * (kslice + 1) / nslice);
    if constexpr (TC::LPS > 1) {
        // ---- fat K steps: group g of the ring = LPS mini-stages of the ordinary layout (A rows | W rows, one line each) -------
        constexpr int LPS = TC::LPS;
        auto mini = [&](const char* st) __attribute__((always_inline)) {
            h8 ah[2][TM], al[2][TM], wh[2][TN], wl[2][TN];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[ks][i] = *reinterpret_cast<const h8*>(st + a_row + i * 4096 + off_hi[ks]);
                    al[ks][i] = *reinterpret_cast<const h8*>(st + a_row + i * 4096 + off_lo[ks]);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    wh[ks][j] = *reinterpret_cast<const h8*>(st + w_row + j * 4096 + off_hi[ks]);
                    wl[ks][j] = *reinterpret_cast<const h8*>(st + w_row + j * 4096 + off_lo[ks]);
                }
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks][i], wh[ks][j], acc0[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks][i], wl[ks][j], acc1[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks][i], wh[ks][j], acc1[i][j], 0, 0, 0);
            }
        };
#pragma unroll
        for (int l = 0; l < LPS; ++l) issue(kt0 + l, l);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        int g = 0;
        for (int kt = kt0; kt < nk; kt += LPS) {
            if (kt + LPS < nk) {     // the next fat step into the other group (every wave left it at the last barrier)
#pragma unroll
                for (int l = 0; l < LPS; ++l) issue(kt + LPS + l, (g ^ 1) * LPS + l);
            }
#pragma unroll
            for (int l = 0; l < LPS; ++l) mini(lds + (g * LPS + l) * STAGE);
            wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            g ^= 1;
        }
        h3_epilogue<TC, EPI>(p, acc0, acc1, m0, n0, M, kslice, lds);
        return;
    }
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (kt0 + s < nk) issue(kt0 + s, s);
    if (NSTAGE == 3 && nk - kt0 > 1) wait_vmcnt<PW>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();

    if (CMDI_DBG(p) & 16) t_loop = __builtin_readcyclecounter();
#if CMDI_KABL
    h8 kabl_frag;
#pragma unroll
    for (int e = 0; e < 8; ++e) kabl_frag[e] = (_Float16)(0.001f * (float)(lane + e));
    unsigned kabl_sink = 0;
#endif
    {
    int cur = 0, nxt = NSTAGE - 1;   // stage being multiplied / stage being filled
    for (int kt = kt0; kt < nk; ++kt) {
        const bool more = kt + NSTAGE - 1 < nk;
        if (!(CMDI_KABL & (1 | 32)) && more && !(CMDI_DBG(p) & 1)) issue(kt + NSTAGE - 1, nxt);
        const char* st = lds + cur * STAGE;
        h8 ah[2][TM], al[2][TM], wh[2][TN], wl[2][TN];
#if CMDI_KABL & 4
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i) { ah[ks][i] = kabl_frag; al[ks][i] = kabl_frag; }
#pragma unroll
            for (int j = 0; j < TN; ++j) { wh[ks][j] = kabl_frag; wl[ks][j] = kabl_frag; }
        }
#else
#if CMDI_H3_SETPRIO
        __builtin_amdgcn_s_setprio(CMDI_H3_SETPRIO);
#endif
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[ks][i] = *reinterpret_cast<const h8*>(st + a_row + i * 4096 + off_hi[ks]);
                al[ks][i] = *reinterpret_cast<const h8*>(st + a_row + i * 4096 + off_lo[ks]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                wh[ks][j] = *reinterpret_cast<const h8*>(st + w_row + j * 4096 + off_hi[ks]);
                wl[ks][j] = *reinterpret_cast<const h8*>(st + w_row + j * 4096 + off_lo[ks]);
            }
        }
#endif
#if CMDI_KABL & 2
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i) kabl_sink ^= __builtin_bit_cast(uint4, ah[ks][i]).x ^ __builtin_bit_cast(uint4, al[ks][i]).w;
#pragma unroll
            for (int j = 0; j < TN; ++j) kabl_sink ^= __builtin_bit_cast(uint4, wh[ks][j]).x ^ __builtin_bit_cast(uint4, wl[ks][j]).w;
        }
#else
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks][i], wh[ks][j], acc0[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks][i], wl[ks][j], acc1[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks][i], wh[ks][j], acc1[i][j], 0, 0, 0);
        }
        // issue order inside the K step: the 8 fragment reads of k-substep 0 first, then one MFMA per
        // remaining read (k-substep 1's fragments land under k-substep 0's products), then the rest
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TM + TN), 0);
#pragma unroll
        for (int q = 0; q < 2 * (TM + TN); ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 6 * TM * TN - 2 * (TM + TN), 0);
#endif
#if CMDI_H3_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        if ((CMDI_KABL & 32) && more) issue(kt + NSTAGE - 1, nxt);
#if !(CMDI_KABL & 8)
        // stage kt+1 must have landed (this wave's pieces; the barrier extends it to every wave's);
        // with 3 stages the pieces of stage kt+2, issued above, stay in flight across the barrier
        if (NSTAGE == 3 && more) wait_vmcnt<PW>(); else wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#endif
        cur = cur + 1 == NSTAGE ? 0 : cur + 1;
        nxt = nxt + 1 == NSTAGE ? 0 : nxt + 1;
    }
    }

    if (CMDI_DBG(p) & 16) t_loop_end = __builtin_readcyclecounter();
#if CMDI_KABL
    if (kabl_sink == 0x12345678u) acc0[0][0][0] += 1.0f;
#endif
    h3_epilogue<TC, EPI>(p, acc0, acc1, m0, n0, M, kslice, lds);
    if ((CMDI_DBG(p) & 16) && p.dbg_buf && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long* o = p.dbg_buf + (size_t)blockIdx.x * 6;
        o[0] = t_start; o[1] = t_loop; o[2] = t_loop_end; o[3] = __builtin_readcyclecounter();
        o[4] = r_start; o[5] = __builtin_amdgcn_s_memrealtime();   // 100 MHz constant clock
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        o[2] = (o[2] - o[1]);   // loop duration
        o[1] = (long long)(((unsigned long long)xcc << 32) | hwid);
    }
}

template <class TC, int EPI>
__global__ __launch_bounds__(TC::NT, TC::MINW) void gemm_h3_kernel(const H3Params p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    gemm_h3_body<TC, EPI>(p, blockIdx.x, 0, p.M, lds);
}

// Mixed-granularity grid against wave quantization: the first n_big blocks cover rows [0, m_split) with the
// big tile — a whole number of rounds over the chip's block slots — and the remaining rows get the small
// tile, so the last, partial round is made of short blocks instead of full-length ones.
template <class TBig, class TSmall, int EPI>
__global__ __launch_bounds__(TBig::NT, TBig::MINW) void gemm_h3_mixed_kernel(const H3Params p) {
    static_assert(TBig::NT == TSmall::NT, "one block size");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if ((int)blockIdx.x < p.n_big) gemm_h3_body<TBig, EPI>(p, blockIdx.x, 0, p.m_split, lds);
    else gemm_h3_body<TSmall, EPI>(p, blockIdx.x - p.n_big, p.m_split, p.M, lds);
}

}  // namespace cmdi
