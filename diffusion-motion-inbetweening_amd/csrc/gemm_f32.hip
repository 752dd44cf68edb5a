// Instantiations + host dispatcher of the fp32 MFMA NT GEMM family (see gemm_f32.hpp).
#include "gemm_f32.hpp"
#include "kernels.hpp"

namespace cmdi {

using T128x128 = Tile<128, 128, 32, 2, 2>;  // 4 waves, 64x64 per wave, 72 KiB LDS
using T64x128 = Tile<64, 128, 32, 2, 2>;    // 32x64 per wave
using T128x64 = Tile<128, 64, 32, 2, 2>;    // 64x32 per wave
using T64x64 = Tile<64, 64, 32, 2, 2>;      // 32x32 per wave
using T256x128 = Tile<256, 128, 32, 4, 2>;  // 8 waves, 64x64 per wave, 108 KiB LDS
using T128x128k16 = Tile<128, 128, 16, 2, 2>;
using T64x64k16 = Tile<64, 64, 16, 2, 2>;
using T128x128w8 = Tile<128, 128, 32, 4, 2>;  // 8 waves, 32x64 per wave
using T128x256w8 = Tile<128, 256, 32, 2, 4>;  // 8 waves, 64x64 per wave

template <class TC, int AM, int BMD, int EPI, int PIPE = 0>
static hipError_t launch_one(const GemmParams& p, hipStream_t stream) {
    auto kern = gemm_nt_kernel<TC, AM, BMD, EPI, PIPE>;
    static PerDevice<bool> attr_done_dev;  // benign race: the attribute call is idempotent
    bool& attr_done = attr_done_dev.get();
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(PIPE == 2 ? TC::LDS_BYTES_DMA : TC::LDS_BYTES));
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int tiles = ((p.M + TC::BM - 1) / TC::BM) * ((p.N + TC::BN - 1) / TC::BN);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(TC::NT),
                       PIPE == 2 ? TC::LDS_BYTES_DMA : TC::LDS_BYTES, stream, p);
    return hipGetLastError();
}

template <int EPI>
static hipError_t launch_plain_tiles(const GemmParams& p, int tile, hipStream_t s) {
    switch (tile) {
        case 1: return launch_one<T128x128, ROWS_PLAIN, ROWS_PLAIN, EPI>(p, s);
        case 2: return launch_one<T64x128, ROWS_PLAIN, ROWS_PLAIN, EPI>(p, s);
        case 3: return launch_one<T128x64, ROWS_PLAIN, ROWS_PLAIN, EPI>(p, s);
        case 4: return launch_one<T64x64, ROWS_PLAIN, ROWS_PLAIN, EPI>(p, s);
        case 5: return launch_one<T256x128, ROWS_PLAIN, ROWS_PLAIN, EPI>(p, s);
        case 11: return launch_one<T128x128, ROWS_PLAIN, ROWS_PLAIN, EPI, 1>(p, s);
        case 12: return launch_one<T64x128, ROWS_PLAIN, ROWS_PLAIN, EPI, 1>(p, s);
        case 13: return launch_one<T128x64, ROWS_PLAIN, ROWS_PLAIN, EPI, 1>(p, s);
        case 14: return launch_one<T64x64, ROWS_PLAIN, ROWS_PLAIN, EPI, 1>(p, s);
        case 15: return launch_one<T256x128, ROWS_PLAIN, ROWS_PLAIN, EPI, 1>(p, s);
        case 31: return launch_one<T128x128k16, ROWS_PLAIN, ROWS_PLAIN, EPI, 1>(p, s);
        case 34: return launch_one<T64x64k16, ROWS_PLAIN, ROWS_PLAIN, EPI, 1>(p, s);
        case 36: return launch_one<T128x128w8, ROWS_PLAIN, ROWS_PLAIN, EPI, 1>(p, s);
        case 46: return launch_one<T128x128w8, ROWS_PLAIN, ROWS_PLAIN, EPI, 2>(p, s);
        case 37: return launch_one<T128x256w8, ROWS_PLAIN, ROWS_PLAIN, EPI, 1>(p, s);
        case 47: return launch_one<T128x256w8, ROWS_PLAIN, ROWS_PLAIN, EPI, 2>(p, s);
        case 21: return launch_one<T128x128, ROWS_PLAIN, ROWS_PLAIN, EPI, 2>(p, s);
        case 22: return launch_one<T64x128, ROWS_PLAIN, ROWS_PLAIN, EPI, 2>(p, s);
        case 23: return launch_one<T128x64, ROWS_PLAIN, ROWS_PLAIN, EPI, 2>(p, s);
        case 24: return launch_one<T64x64, ROWS_PLAIN, ROWS_PLAIN, EPI, 2>(p, s);
        case 25: return launch_one<T256x128, ROWS_PLAIN, ROWS_PLAIN, EPI, 2>(p, s);
        default: return hipErrorInvalidValue;
    }
}

// Tile / pipeline choice, from the GEMM sweep on MI355X (profiles/r01_gemm_tile_sweep.md):
//   wide outputs (N >= 1536: self_attn.in_proj)  -> 128x128, 8 waves (2 per SIMD), register pipeline
//   everything else                              -> 64x64, BK = 16 (8 co-resident blocks per CU):
// at M = 2B*197 rows the grids are only a few blocks per CU deep, so small tiles (fine-grained
// balance across 256 CUs) and many co-resident waves (to cover each block's prologue / epilogue)
// beat the higher arithmetic intensity of big tiles; fp32 MFMA is slow enough (64 cycles per
// 32x32x2) that the extra LDS / L2 traffic of a 64x64 tile stays hidden.
int gemm_auto_tile(int M, int N) {
    if (N >= 1536 && M >= 2048) return 36;
    return 34;
}

hipError_t launch_gemm(GemmKind kind, const GemmParams& p, int tile, hipStream_t s) {
    if (p.K % 32 != 0 || p.M <= 0 || p.N <= 0) return hipErrorInvalidValue;
    if (tile == 0) tile = gemm_auto_tile(p.M, p.N);
    switch (kind) {
        case GK_PLAIN: return launch_plain_tiles<EPI_PLAIN>(p, tile, s);
        case GK_GELU: return launch_plain_tiles<EPI_GELU>(p, tile, s);
        case GK_RESID: return launch_plain_tiles<EPI_RESID>(p, tile, s);
        case GK_ACCUM: return launch_plain_tiles<EPI_ACCUM>(p, tile, s);
        case GK_GELUGRAD: return launch_plain_tiles<EPI_GELUGRAD>(p, tile, s);
        case GK_SILU: return launch_one<T64x64, ROWS_PLAIN, ROWS_PLAIN, EPI_SILU>(p, s);
        case GK_INPROJ:
            return tile == 1 ? launch_one<T64x128, ROWS_MOTION, ROWS_PLAIN, EPI_INPROJ, 1>(p, s)
                             : launch_one<T64x128, ROWS_MOTION, ROWS_PLAIN, EPI_INPROJ>(p, s);
        case GK_OUTPROJ:
            return tile == 1 ? launch_one<T64x128, ROWS_PLAIN, ROWS_TOK, EPI_MOTION, 1>(p, s)
                             : launch_one<T64x128, ROWS_PLAIN, ROWS_TOK, EPI_MOTION>(p, s);
        case GK_OUTPROJ_BWD: return launch_one<T64x128, ROWS_MOTION, ROWS_PLAIN, EPI_TOKOUT>(p, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace cmdi
