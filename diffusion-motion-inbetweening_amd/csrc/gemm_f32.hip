// Instantiations + host dispatcher of the fp32 MFMA NT GEMM family (see gemm_f32.hpp).
#include "gemm_f32.hpp"
#include "kernels.hpp"

namespace cmdi {

using T128x128 = Tile<128, 128, 32, 2, 2>;  // 4 waves, 64x64 per wave, 72 KiB LDS
using T64x128 = Tile<64, 128, 32, 2, 2>;    // 32x64 per wave
using T128x64 = Tile<128, 64, 32, 2, 2>;    // 64x32 per wave
using T64x64 = Tile<64, 64, 32, 2, 2>;      // 32x32 per wave
using T256x128 = Tile<256, 128, 32, 4, 2>;  // 8 waves, 64x64 per wave, 108 KiB LDS

template <class TC, int AM, int BMD, int EPI>
static hipError_t launch_one(const GemmParams& p, hipStream_t stream) {
    auto kern = gemm_nt_kernel<TC, AM, BMD, EPI>;
    static bool attr_done = false;  // benign race: the attribute call is idempotent
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)TC::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int tiles = ((p.M + TC::BM - 1) / TC::BM) * ((p.N + TC::BN - 1) / TC::BN);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(TC::NT), TC::LDS_BYTES, stream, p);
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
        default: return hipErrorInvalidValue;
    }
}

// Pick the tile that minimises (blocks per CU, rounded up) x (tile area): co-resident blocks on a
// CU share its four matrix pipes, so wall time ~ max blocks per CU x per-block MFMA count.
int gemm_auto_tile(int M, int N) {
    struct Cand { int id, bm, bn; };
    static const Cand cands[] = {{1, 128, 128}, {2, 64, 128}, {3, 128, 64}, {4, 64, 64}};
    const int n_cu = 256;
    long best_cost = -1;
    int best = 1;
    for (const Cand& c : cands) {
        const long blocks = (long)((M + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn);
        const long rounds = (blocks + n_cu - 1) / n_cu;
        const long cost = rounds * c.bm * c.bn;
        if (best_cost < 0 || cost < best_cost) {  // ties keep the larger tile (listed first)
            best_cost = cost;
            best = c.id;
        }
    }
    return best;
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
        case GK_INPROJ: return launch_one<T64x128, ROWS_MOTION, ROWS_PLAIN, EPI_INPROJ>(p, s);
        case GK_OUTPROJ: return launch_one<T64x128, ROWS_PLAIN, ROWS_TOK, EPI_MOTION>(p, s);
        case GK_OUTPROJ_BWD: return launch_one<T64x128, ROWS_MOTION, ROWS_PLAIN, EPI_TOKOUT>(p, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace cmdi
