// Epilogue of one 32 x 32 output block of the split-f16 GEMM family with 16-byte global accesses (shared by gemm_h3.hpp's
// interior path and gemm_h3p.hpp).  Included by gemm_h3.hpp after its own helpers (split_f16, kLoInv, h4 / h8, split_pos).
#pragma once

#ifndef CMDI_H3P_OUT_SC1
#define CMDI_H3P_OUT_SC1 0   // 1: the persistent kernel's split-row stores carry sc1 (experiment builds)
#endif
namespace cmdi {

// Epilogue geometry: a lane owns EIGHT consecutive columns of a row (4 lanes per 32-column row segment, 16 rows per
// wave-instruction, two passes per 32 x 32 block), so every global access is 16 bytes per lane: the hi and the lo half of a
// split row are one dwordx4 store each.  (With 4 columns per lane — gemm_h3's geometry — the split stores are dwordx2 and the
// epilogue was store-ISSUE-bound: 32 stores per wave and tile, ~10 k cycles per tile with the matrix pipe idle; see
// MI355X_MICROARCH.md "epilogue store tail".)  The partial LayerNorm statistics reproduce gemm_h3's reduction tree exactly:
// its lanes 2t, 2t + 1 are this lane's two column quads.
//
// Per-column operands of one 32-column fragment (this lane's 8 columns): loaded for every fragment of the wave BEFORE its first
// store — with stores in flight hipcc waits vmcnt(0) in front of the next use of a loaded value (loads and stores share the
// counter and may complete out of order), i.e. for the write latency of everything stored so far.
struct H3PCols {
    float4 bias[2], x1[2], x2[2];      // x1 = c1 (A operand is a folded LayerNorm) or the residual LayerNorm's gamma; x2 = its beta
};
template <int EPI>
__device__ __forceinline__ H3PCols h3p_load_cols(const H3Params& p, int n) {
    H3PCols c;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        c.bias[h] = c.x1[h] = c.x2[h] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) c.bias[h] = *reinterpret_cast<const float4*>(p.bias + n + 4 * h);
        if constexpr (EPI == H3_RESID) {
            if (p.ln_rg) {
                c.x1[h] = *reinterpret_cast<const float4*>(p.ln_rg + n + 4 * h);
                c.x2[h] = *reinterpret_cast<const float4*>(p.ln_rb + n + 4 * h);
            }
        } else {
            if (p.ln_c1) c.x1[h] = *reinterpret_cast<const float4*>(p.ln_c1 + n + 4 * h);
        }
    }
    return c;
}
// Residual / GELU-gradient operand rows of one 32 x 32 block: 2 rows per lane, 32 bytes each (hi and lo half8 of split rows,
// or two float4)
struct H3PRows {
    uint4 a[2], b[2];
};
template <int EPI, bool EDGE>
__device__ __forceinline__ void h3p_load_rows(const H3Params& p, int m_blk, int n, int lane, H3PRows& o) {
    const int rl = lane >> 2;
    const int npos = split_pos(n);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        int m = m_blk + it * 16 + rl;
        if (EDGE) m = m < p.M ? m : p.M - 1;
        if constexpr (EPI == H3_RESID) {
            if (p.Rs) {
                const _Float16* rsrc = p.Rs + (size_t)m * (2 * p.N) + npos;
                o.a[it] = *reinterpret_cast<const uint4*>(rsrc);
                o.b[it] = *reinterpret_cast<const uint4*>(rsrc + 32);
            } else {
                const float* rsrc = p.R + (p.r_ld ? (size_t)m * p.r_ld + n : (size_t)m * p.ldc + n);
                o.a[it] = *reinterpret_cast<const uint4*>(rsrc);
                o.b[it] = *reinterpret_cast<const uint4*>(rsrc + 4);
            }
        }
        if constexpr (EPI == H3_GELUGRAD_SPLIT) {
            const float* rsrc = p.aux + (size_t)m * p.ldc + n;
            o.a[it] = *reinterpret_cast<const uint4*>(rsrc);
            o.b[it] = *reinterpret_cast<const uint4*>(rsrc + 4);
        }
    }
}

// One 32 x 32 block of a wave's outputs: acc (= acc0 + acc1 2^-11, combined by the caller) -> transposed through `wl` (the
// wave's private 4-KiB LDS slice) -> row-major.  (m_blk, n_blk) = global position of the block, rs = (mean, rstd) of its 32
// rows (folded LayerNorm) or null.  EDGE: rows >= M exist (their loads are clamped, their stores skipped).  Per-element
// arithmetic copied from gemm_h3's interior path line by line (same bits).
// LOWREG (gemm_h3's 128-register kernels): the transposed rows are read from LDS one row pass at a time and the accumulator
// pair is combined while it is written out — 24 fewer live registers; the other form reads both passes before the first store.
// CONV (H3_PLAIN of the persistent kernel): the row goes to m * c_row_mul + c_row_add and only if its position inside the
// tp-row frame of its sequence lies in [t_lo, t_hi) — gemm_h3's convolution rule.
template <int EPI, bool EDGE, bool NO_STORE = false, bool LOWREG = false, bool CONV = false>
__device__ __forceinline__ void h3p_epi_block(const H3Params& p, const f32x16& acc, const f32x16& acc_lo, int m_blk, int n_blk,
                                              const float2* rs, float* wl, int lane, const H3PCols& cols, const H3PRows& rows,
                                              bool& overflow) {
    const int l31 = lane & 31;
    const int rl = lane >> 2, cl = (lane & 3) * 8;
    const int n = n_blk + cl;
    const int M = p.M;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        wl[mfma32_row(r, lane) * 32 + l31] = LOWREG ? acc[r] + acc_lo[r] * kLoInv : acc[r];   // (else combined by the caller)
    const int npos = split_pos(n);
    float4 tt[2][2];
    float2 rst[2];
    if constexpr (!LOWREG) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            tt[it][0] = *reinterpret_cast<const float4*>(wl + (it * 16 + rl) * 32 + cl);
            tt[it][1] = *reinterpret_cast<const float4*>(wl + (it * 16 + rl) * 32 + cl + 4);
            rst[it] = rs ? rs[it * 16 + rl] : make_float2(0.f, 1.f);
        }
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        if constexpr (LOWREG) {
            tt[it][0] = *reinterpret_cast<const float4*>(wl + (it * 16 + rl) * 32 + cl);
            tt[it][1] = *reinterpret_cast<const float4*>(wl + (it * 16 + rl) * 32 + cl + 4);
            rst[it] = rs ? rs[it * 16 + rl] : make_float2(0.f, 1.f);
        }
        int m = m_blk + it * 16 + rl;
        bool live = (!EDGE || m < M) && !(NO_STORE && rs != nullptr && rst[0].x != 12345.678f);
        if constexpr (CONV) {
            if (p.rc_tv) {      // logical row -> physical row of the framed layout (always a frame: nothing to mask)
                const int sq = m / p.rc_tv;
                m = sq * p.tp + p.t_lo + (m - sq * p.rc_tv);
            } else if (p.c_row_mul) m = m * p.c_row_mul + p.c_row_add;
            if (p.tp && !p.rc_tv) {
                const int pos = m % p.tp;
                live = live && pos >= p.t_lo && pos < p.t_hi;
            }
        }
        const float2 rs2 = rst[it];
        float v[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 t = tt[it][h], bias4 = cols.bias[h];
            v[4 * h + 0] = t.x + bias4.x; v[4 * h + 1] = t.y + bias4.y; v[4 * h + 2] = t.z + bias4.z; v[4 * h + 3] = t.w + bias4.w;
            if constexpr (EPI != H3_RESID) {
                if (p.ln_c1) {   // A operand was the raw P: LN(P) W^T + b = rstd (P W'^T - mean c1) + c2
                    const float4 c14 = cols.x1[h];
                    v[4 * h + 0] = __builtin_fmaf(rs2.y, __builtin_fmaf(-rs2.x, c14.x, t.x), bias4.x);
                    v[4 * h + 1] = __builtin_fmaf(rs2.y, __builtin_fmaf(-rs2.x, c14.y, t.y), bias4.y);
                    v[4 * h + 2] = __builtin_fmaf(rs2.y, __builtin_fmaf(-rs2.x, c14.z, t.z), bias4.z);
                    v[4 * h + 3] = __builtin_fmaf(rs2.y, __builtin_fmaf(-rs2.x, c14.w, t.w), bias4.w);
                }
            }
        }
        const size_t off = (size_t)m * p.ldc + n;
        if constexpr (EPI == H3_PLAIN) {
            if (live) {
                *reinterpret_cast<float4*>(p.C + off) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(p.C + off + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
            continue;
        }
        if constexpr (EPI == H3_RESID) {
            if (p.Rs) {
                const h8 rh = *reinterpret_cast<const h8*>(&rows.a[it]), rlo = *reinterpret_cast<const h8*>(&rows.b[it]);
                float r8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) r8[e] = (float)rh[e] + (float)rlo[e] * kLoInv;
                if (p.ln_rg) {   // the residual is LayerNorm(P) of the rows read
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float4 rg4 = cols.x1[h], rb4 = cols.x2[h];
                        r8[4 * h + 0] = __builtin_fmaf((r8[4 * h + 0] - rs2.x) * rs2.y, rg4.x, rb4.x);
                        r8[4 * h + 1] = __builtin_fmaf((r8[4 * h + 1] - rs2.x) * rs2.y, rg4.y, rb4.y);
                        r8[4 * h + 2] = __builtin_fmaf((r8[4 * h + 2] - rs2.x) * rs2.y, rg4.z, rb4.z);
                        r8[4 * h + 3] = __builtin_fmaf((r8[4 * h + 3] - rs2.x) * rs2.y, rg4.w, rb4.w);
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += r8[e];
            } else {
                const float4 ra = *reinterpret_cast<const float4*>(&rows.a[it]), rb = *reinterpret_cast<const float4*>(&rows.b[it]);
                v[0] += ra.x; v[1] += ra.y; v[2] += ra.z; v[3] += ra.w;
                v[4] += rb.x; v[5] += rb.y; v[6] += rb.z; v[7] += rb.w;
            }
            if (p.C && live) {
                *reinterpret_cast<float4*>(p.C + off) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(p.C + off + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
        }
        if constexpr (EPI == H3_GELUGRAD_SPLIT) {
            const float4 ra = *reinterpret_cast<const float4*>(&rows.a[it]), rb = *reinterpret_cast<const float4*>(&rows.b[it]);
            v[0] *= gelu_erf_grad(ra.x); v[1] *= gelu_erf_grad(ra.y); v[2] *= gelu_erf_grad(ra.z); v[3] *= gelu_erf_grad(ra.w);
            v[4] *= gelu_erf_grad(rb.x); v[5] *= gelu_erf_grad(rb.y); v[6] *= gelu_erf_grad(rb.z); v[7] *= gelu_erf_grad(rb.w);
        } else if constexpr (EPI != H3_RESID) {
            if (p.aux && live) {
                *reinterpret_cast<float4*>(p.aux + off) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(p.aux + off + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
        }
        if constexpr (EPI == H3_GELU_SPLIT) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
        }
        if (EPI != H3_RESID || p.Cs) {
            h8 oh, ol;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                _Float16 a, b;
                split_f16(v[e], a, b);
                oh[e] = a; ol[e] = b;
                overflow |= !(fabsf(v[e]) < 65504.0f);
            }
            if (live) {
                _Float16* dst = p.Cs + (size_t)m * (p.cs_ld ? p.cs_ld : 2 * p.N) + npos;
#if CMDI_H3P_OUT_SC1
                // experiment builds only (profiles/r05_inproj_l2_counters.txt): split rows written through the L2 (`sc1`) instead of
                // parked there as dirty lines — fabric reads of the C4 in_proj 718 -> see the file; costs the kernel time
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(oh) : "memory");
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst + 32), "v"(ol) : "memory");
#else
                *reinterpret_cast<h8*>(dst) = oh;
                *reinterpret_cast<h8*>(dst + 32) = ol;
#endif
            }
        }
        if constexpr (EPI == H3_RESID) {
            if (p.out_part) {
                // partial LayerNorm statistics of the row just written over its 32-column block: gemm_h3's tree — per column quad
                // (v0 + v1) + (v2 + v3), then pairs of quads, pairs of pairs, ... (its lanes 2t, 2t + 1 = this lane's two quads)
                const float sa = (v[0] + v[1]) + (v[2] + v[3]), sb = (v[4] + v[5]) + (v[6] + v[7]);
                float sm = sa + sb;
                sm += __shfl_xor(sm, 1, 64);
                sm += __shfl_xor(sm, 2, 64);
                const float mb = sm * (1.0f / 32.0f);
                float d[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) d[e] = v[e] - mb;
                const float qa = __builtin_fmaf(d[0], d[0], d[1] * d[1]) + __builtin_fmaf(d[2], d[2], d[3] * d[3]);
                const float qb = __builtin_fmaf(d[4], d[4], d[5] * d[5]) + __builtin_fmaf(d[6], d[6], d[7] * d[7]);
                float q = qa + qb;
                q += __shfl_xor(q, 1, 64);
                q += __shfl_xor(q, 2, 64);
                if ((lane & 3) == 0 && live)
                    *reinterpret_cast<float2*>(p.out_part + ((size_t)m * 16 + (n >> 5)) * 2) = make_float2(sm, q);
            }
        }
    }
}

}  // namespace cmdi
