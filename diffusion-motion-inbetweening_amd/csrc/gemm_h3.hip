// Instantiations + host dispatcher of the split-f16 (fp32-equivalent) NT GEMM family (gemm_h3.hpp)
// and the fp32 -> split-rows conversion kernel.
#include <cstdlib>

#include "gemm_h3.hpp"
#include "kernels.hpp"

namespace cmdi {

using H128x128s2 = H3Tile<128, 128, 2, 2, 2, 2>;    // 4 waves, 64x64 per wave, 64 KiB LDS, 2 blocks/CU
using H256x128s3 = H3Tile<256, 128, 4, 2, 3, 2>;    // 8 waves, 64x64 per wave, 3 stages = 144 KiB
using H256x128s2 = H3Tile<256, 128, 4, 2, 2, 2>;    // 8 waves, 96 KiB
using H128x64s2 = H3Tile<128, 64, 2, 2, 2, 2>;      // 64x32 per wave, 48 KiB, 3 blocks/CU
using H128x64s3 = H3Tile<128, 64, 2, 2, 3, 2>;      // 72 KiB, 2 blocks/CU
using H64x128s2 = H3Tile<64, 128, 2, 2, 2, 2>;      // 32x64 per wave
using H128x128w8s3 = H3Tile<128, 128, 4, 2, 3, 2>;  // 8 waves, 32x64 per wave, 96 KiB
using H128x128w8s2 = H3Tile<128, 128, 4, 2, 2, 4>;  // 8 waves, 64 KiB, 2 blocks/CU (16 waves = 4 per SIMD: at most 128 VGPRs)
using H128x128w8s2E = H3Tile<128, 128, 4, 2, 2, 4, 1>;     // the same, 8-column-per-lane epilogue (dwordx4 split stores) on interior tiles
using H128x256s2 = H3Tile<128, 256, 2, 4, 2, 2>;    // 8 waves, 64x64 per wave, 96 KiB
using H64x128w8s2 = H3Tile<64, 128, 2, 4, 2, 2>;    // 8 waves, 32x32 per wave: the tail tile of the mixed grid
using H64x512ln = H3Tile<64, 512, 2, 4, 2, 2>;      // 8 waves, 32x128 per wave: full rows of d = 512 (LN fused)
using H256x128w16 = H3Tile<256, 128, 8, 2, 3, 4>;   // 16 waves, 32x64 per wave, 3 stages = 144 KiB
// fat K steps for small M (round 5; H3Tile::LPS): one request / wait / barrier per LPS * 32 columns
using H64x64f4 = H3Tile<64, 64, 2, 2, 2, 2, 0, 4>;      // 4 waves of 32x32, 128 columns per step, 128 KiB: one block per CU
using H64x64f2 = H3Tile<64, 64, 2, 2, 2, 2, 0, 2>;      // 4 waves, 64 columns per step, 64 KiB
using H64x128f2 = H3Tile<64, 128, 2, 4, 2, 2, 0, 2>;    // 8 waves of 32x32, 64 columns per step, 96 KiB
using H128x256w16 = H3Tile<128, 256, 4, 4, 3, 4>;   // 16 waves, 32x64 per wave

template <class TC, int EPI>
static hipError_t launch_h3_one(const H3Params& p, hipStream_t stream) {
    auto kern = gemm_h3_kernel<TC, EPI>;
    static PerDevice<bool> attr_done_dev;  // benign race: the attribute call is idempotent
    bool& attr_done = attr_done_dev.get();
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)TC::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    int tiles = ((p.M + TC::BM - 1) / TC::BM) * ((p.N + TC::BN - 1) / TC::BN);
    if (EPI == H3_PLAIN && p.ksplit > 1) tiles *= p.ksplit;
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(TC::NT), TC::LDS_BYTES, stream, p);
    return hipGetLastError();
}

// 128x128 (8 waves) over a whole number of rounds + 64x128 tiles over the remaining rows
template <int EPI>
static hipError_t launch_h3_mixed(const H3Params& p0, hipStream_t stream) {
    using TB = H128x128w8s2;
    using TS = H64x128w8s2;
    static int slots = 0;
    if (!slots) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        slots = 2 * cus;   // two 64-KiB blocks per CU
    }
    const int tiles_n = (p0.N + TB::BN - 1) / TB::BN;
    const int tiles = ((p0.M + TB::BM - 1) / TB::BM) * tiles_n;
    const int panels = (tiles / slots) * slots / tiles_n;   // big row panels = whole rounds
    if (panels <= 0 || panels * TB::BM >= p0.M) return launch_h3_one<TB, EPI>(p0, stream);
    H3Params p = p0;
    p.m_split = panels * TB::BM;
    p.n_big = panels * tiles_n;
    const int n_small = ((p.M - p.m_split + TS::BM - 1) / TS::BM) * ((p.N + TS::BN - 1) / TS::BN);
    auto kern = gemm_h3_mixed_kernel<TB, TS, EPI>;
    constexpr size_t lds = TB::LDS_BYTES > TS::LDS_BYTES ? TB::LDS_BYTES : TS::LDS_BYTES;
    static PerDevice<bool> attr_done_dev;
    bool& attr_done = attr_done_dev.get();
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.n_big + n_small), dim3(TB::NT), lds, stream, p);
    return hipGetLastError();
}

template <int EPI>
static hipError_t launch_h3_tiles(const H3Params& p, int tile, hipStream_t s) {
    switch (tile) {
        case 1: return launch_h3_one<H128x128s2, EPI>(p, s);
        case 2: return launch_h3_one<H256x128s3, EPI>(p, s);
        case 3: return launch_h3_one<H256x128s2, EPI>(p, s);
        case 4: return launch_h3_one<H128x64s2, EPI>(p, s);
        case 5: return launch_h3_one<H128x64s3, EPI>(p, s);
        case 6: return launch_h3_one<H64x128s2, EPI>(p, s);
        case 7: return launch_h3_one<H128x128w8s3, EPI>(p, s);
        case 8: return launch_h3_one<H128x128w8s2, EPI>(p, s);
        case 88:
            if constexpr (EPI == H3_PLAIN_SPLIT || EPI == H3_GELU_SPLIT) return launch_h3_one<H128x128w8s2E, EPI>(p, s);
            else return launch_h3_one<H128x128w8s2, EPI>(p, s);
        case 9: return launch_h3_one<H128x256s2, EPI>(p, s);
        case 10: return launch_h3_one<H256x128w16, EPI>(p, s);
        case 11: return launch_h3_one<H128x256w16, EPI>(p, s);
        case 20: return launch_h3_mixed<EPI>(p, s);
        case 21: return launch_h3_one<H64x128w8s2, EPI>(p, s);
        // fat K steps: the K loop must be a whole number of them, no split-K
        case 24: return p.K % 128 == 0 && p.ksplit <= 1 ? launch_h3_one<H64x64f4, EPI>(p, s) : hipErrorInvalidValue;
        case 25: return p.K % 64 == 0 && p.ksplit <= 1 ? launch_h3_one<H64x128f2, EPI>(p, s) : hipErrorInvalidValue;
        case 26: return p.K % 64 == 0 && p.ksplit <= 1 ? launch_h3_one<H64x64f2, EPI>(p, s) : hipErrorInvalidValue;
        default: return hipErrorInvalidValue;
    }
}

// Measured on MI355X at the denoiser's shapes (M = 12,608; tools/gemm_bench.py): 128x128 with 8 waves
// (32x64 per wave), 2 stages, 2 blocks = 16 waves per CU wins every projection.
int gemm_h3_auto_tile(int M, int N) {
    // fewer than ~3/4 of the chip's 512 block slots with 128x128 tiles (the coarse levels of the U-Net):
    // halve the tile height so twice as many blocks share the work
    const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
    return tiles < 384 ? 21 : 8;
}

// Small batches (round 5; plain GEMMs only): when a launch cannot fill the chip the blocks run alone on their CUs and a K step is
// a fixed ~1,000-1,400 cycles of request round trip whatever the tile — so take fewer, fatter K steps (H3Tile::LPS) on the
// smallest tile that still fits ONE round of block slots.  Same bits as every other tile (profiles/r05_small_batch_tiles.txt:
// M = 788: linear1 12.2 -> 8.8 us, out_proj 12.1 -> 8.9, linear2 19.1 -> 13.1, in_proj 11.8 -> 10.7; no gain once the launch
// needs a second round).  One round = the device's CUs (256 on an MI355X; one 128-KiB block each, two 64-KiB blocks each for
// tile 26) — queried, not assumed, so that a CPX partition does not take the fat tiles into a second round.
static int gemm_h3_small_m_tile(const H3Params& p) {
    if (p.cpt || p.ksplit > 1 || p.N % 64 != 0 || p.K % 64 != 0) return 0;
    const long cus = device_cu_count();
    const long t64 = (long)((p.M + 63) / 64) * (p.N / 64);
    if (t64 <= cus && p.K % 128 == 0) return 24;                 // 64x64, 128 columns per K step
    if (t64 <= 2 * cus) return 26;                                // 64x64, 64 columns per K step, two blocks per CU
    if (p.N % 128 == 0 && (long)((p.M + 63) / 64) * (p.N / 128) <= cus) return 25;   // 64x128, 64 columns per K step
    return 0;
}

// The persistent kernel (gemm_h3p.hpp) computes the same bits as the tiles above; which one runs is a pure speed choice.
// Measured on MI355X (tools/h3p_check.py): it wins from M ~ 50,000 rows up (in_proj at B=256: 536 vs 568 us) and loses below
// (M = 12,608: 84-88 vs 77-80 us — 2.3 tiles per CU, one epilogue per tile with the matrix pipe idle).  CMDI_H3_PERSIST=0 / 1
// forces never / whenever supported.
bool gemm_h3_persistent_for(int M) {
    static int mode = -2;
    if (mode == -2) {
        const char* v = std::getenv("CMDI_H3_PERSIST");
        mode = v ? std::atoi(v) : -1;
    }
    return mode == 1 || (mode == -1 && M >= 32768);
}

// Name of the kernel family the most recent launch_gemm_h3 of this thread dispatched to (bench.py names the kernel that RAN,
// VERDICT r4 weak #13): the tiled gemm_h3_kernel or the persistent gemm_h3p_kernel.
static thread_local const char* g_h3_route = "";
const char* gemm_h3_last_route() { return g_h3_route; }

static hipError_t launch_gemm_h3_routed(int epi, const H3Params& p, int tile, hipStream_t s, const char** route);

hipError_t launch_gemm_h3(int epi, const H3Params& p, int tile, hipStream_t s) {
    const char* route = "gemm_h3_kernel";
    const hipError_t err = launch_gemm_h3_routed(epi, p, tile, s, &route);
    g_h3_route = route;
    return err;
}

static hipError_t launch_gemm_h3_routed(int epi, const H3Params& p, int tile, hipStream_t s, const char** route) {
    if (p.K % 32 != 0 || (p.N % 8 != 0 && epi != H3_MOTION) || p.M <= 0 || p.N <= 0) return hipErrorInvalidValue;
    {   // the LDS-DMA requests address both operands with 32-bit byte offsets from their base (buffer descriptors)
        const size_t a_row = 2 * (p.a_ld ? (size_t)p.a_ld : 2 * (size_t)p.K) * (size_t)(p.a_row_mul ? p.a_row_mul : 1);
        const size_t a_bytes = (size_t)p.M * a_row + (size_t)(p.taps > 0 ? p.taps : 1) * a_row;
        if (a_bytes >= (1ull << 32) || (size_t)p.N * 4 * (size_t)p.K >= (1ull << 32)) return hipErrorInvalidValue;
    }
    if (epi == H3_RESID_LN) {
        if (p.N != 512 || !p.R || !p.ln_g || !p.ln_b || !p.C) return hipErrorInvalidValue;
        return launch_h3_one<H64x512ln, H3_RESID_LN>(p, s);
    }
    if (p.ksplit > 1 && (epi != H3_PLAIN || tile == 20 || p.K / 32 < p.ksplit)) return hipErrorInvalidValue;
    // folded LayerNorm: statistics rows are 16 blocks of 32 columns = d_model 512; residual-LN and the emitted partials
    // belong to the residual epilogue; row statistics are per 128-row... any BM, but not the mixed-granularity grid
    if ((p.ln_part || p.out_part) && tile == 20) return hipErrorInvalidValue;
    if ((p.out_part && (epi != H3_RESID || p.N != 512)) || (p.ln_rg && (epi != H3_RESID || !p.Rs || !p.ln_part || p.N != 512)) ||
        (p.ln_c1 && (!p.ln_part || p.K != 512)))
        return hipErrorInvalidValue;
    if (tile == 50) { *route = "gemm_h3p_kernel"; return launch_gemm_h3p(epi, p, s, 0); }
    if (tile == 60) { *route = "gemm_h3w_kernel"; return launch_gemm_h3w(epi, p, s); }      // weight-stationary (K = 512), gemm_h3w.hpp
    if (p.rc_tv) return hipErrorInvalidValue;      // logical-row GEMMs exist on the persistent kernel only
    if (tile == 0 && gemm_h3_persistent_for(p.M) && gemm_h3p_supports(epi, p)) {
        *route = "gemm_h3p_kernel";
        return launch_gemm_h3p(epi, p, s, 0);
    }
    // the weight-stationary kernel (K = 512; gemm_h3w.hpp) where the caller passes a fragment-ordered copy of W (the engine does
    // for launches of at least CMDI_H3W_MIN_M rows when CMDI_H3W=1); same bits as the tiles below — a pure speed choice
    if (tile == 0 && p.Wp && gemm_h3w_supports(epi, p)) {
        *route = "gemm_h3w_kernel";
        return launch_gemm_h3w(epi, p, s);
    }
    if (tile >= 1000) { *route = "gemm_h3p_kernel"; return launch_gemm_h3p(epi, p, s, tile - 1000); }   // structure variants / ablations (probes library only)
    if (tile == 0 && (epi == H3_PLAIN || epi == H3_GELU_SPLIT || epi == H3_RESID || epi == H3_PLAIN_SPLIT || epi == H3_GELUGRAD_SPLIT))
        tile = gemm_h3_small_m_tile(p);
    if (tile == 0) {
        tile = gemm_h3_auto_tile(p.M, p.N);
        static int epi8 = -1;
        if (epi8 < 0) { const char* v = std::getenv("CMDI_H3_EPI8"); epi8 = v ? std::atoi(v) : 0; }
        if (tile == 8 && epi8) tile = 88;
    }
    if (epi == H3_CONV_GN) {   // tile rows = one framed sequence: 256 (level 0) or 128 (level 1)
        if (!p.ln_g || !p.ln_b || (!p.C && !p.Cs) || p.N % 128 != 0 || (p.gn_cg != 128 && p.gn_cg != 64) || p.M % p.tp != 0 ||
            p.c_row_mul || p.ksplit > 1)
            return hipErrorInvalidValue;
        if (p.tp == 256) return launch_h3_one<H256x128s2, H3_CONV_GN>(p, s);
        if (p.tp == 128) return launch_h3_one<H128x128w8s2, H3_CONV_GN>(p, s);
        return hipErrorInvalidValue;
    }
    if (epi == H3_TOKENS || epi == H3_MOTION) {   // the two I/O projections: two tile shapes only
        if (p.tok_T < 1 || p.tok_S != p.tok_T + 1) return hipErrorInvalidValue;
        if (epi == H3_TOKENS) {
            if (!p.Cs && !p.C) return hipErrorInvalidValue;
            return tile == 8 ? launch_h3_one<H128x128w8s2, H3_TOKENS>(p, s) : launch_h3_one<H64x128w8s2, H3_TOKENS>(p, s);
        }
        if (!p.C) return hipErrorInvalidValue;
        return tile == 8 ? launch_h3_one<H128x128w8s2, H3_MOTION>(p, s) : launch_h3_one<H64x128w8s2, H3_MOTION>(p, s);
    }
    switch (epi) {
        case H3_PLAIN: return launch_h3_tiles<H3_PLAIN>(p, tile, s);
        case H3_GELU_SPLIT: return launch_h3_tiles<H3_GELU_SPLIT>(p, tile, s);
        case H3_RESID: return launch_h3_tiles<H3_RESID>(p, tile, s);
        case H3_PLAIN_SPLIT: return launch_h3_tiles<H3_PLAIN_SPLIT>(p, tile, s);
        case H3_GELUGRAD_SPLIT: return launch_h3_tiles<H3_GELUGRAD_SPLIT>(p, tile, s);
    }
    return hipErrorInvalidValue;
}

// fp32 rows [rows][cols] (row stride ld_src floats) -> split rows [rows][2*cols] halves (32-column
// chunks, hi then lo, see gemm_h3.hpp).  One thread
// converts 8 consecutive elements: two float4 loads, one 16-B store per plane.  HBM-bound.
__global__ __launch_bounds__(256) void split_f16_kernel(const float* __restrict__ src,
                                                        _Float16* __restrict__ dst, int64_t rows,
                                                        int cols, int64_t ld_src,
                                                        int* __restrict__ range_flag) {
    const int chunks = cols >> 3;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * chunks) return;
    const int64_t r = idx / chunks;
    const int c = (int)(idx - r * chunks) * 8;
    const float4 a = *reinterpret_cast<const float4*>(src + r * ld_src + c);
    const float4 b = *reinterpret_cast<const float4*>(src + r * ld_src + c + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    h8 oh, ol;
    bool overflow = false;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        _Float16 h, l;
        split_f16(v[e], h, l);
        oh[e] = h; ol[e] = l;
        overflow |= !(fabsf(v[e]) < 65504.0f);
    }
    _Float16* d = dst + r * (2 * (int64_t)cols) + split_pos(c);
    *reinterpret_cast<h8*>(d) = oh;
    *reinterpret_cast<h8*>(d + 32) = ol;
    if (overflow && range_flag) atomicOr(range_flag, 1);
}

hipError_t launch_split_f16(const float* src, _Float16* dst, int64_t rows, int cols, int64_t ld_src,
                            int* range_flag, hipStream_t stream) {
    if (cols % 32 != 0 || ld_src % 4 != 0) return hipErrorInvalidValue;
    const int64_t n = rows * (cols >> 3);
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(split_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src,
                       dst, rows, cols, ld_src, range_flag);
    return hipGetLastError();
}

}  // namespace cmdi
