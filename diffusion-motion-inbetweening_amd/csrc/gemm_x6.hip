// Instantiations + host dispatcher of the bf16x6 exact-operand GEMM (gemm_x6.hpp) and the weight pre-split kernel.
#include "gemm_x6.hpp"
#include "kernels.hpp"

namespace cmdi {

// fp32 W [rows][cols] (row stride ld) -> [rows][cols/32][3 planes][32] bf16: W = p0 + p1 + p2 exactly (gemm_x6.hpp).
// One thread splits 8 consecutive elements: two float4 loads, three 16-B stores.
__global__ __launch_bounds__(256) void pack_x6_kernel(const float* __restrict__ src, __bf16* __restrict__ dst,
                                                      int64_t rows, int cols, int64_t ld) {
    const int chunks = cols >> 3;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * chunks) return;
    const int64_t r = idx / chunks;
    const int c = (int)(idx - r * chunks) * 8;
    const float4 a = *reinterpret_cast<const float4*>(src + r * ld + c);
    const float4 b = *reinterpret_cast<const float4*>(src + r * ld + c + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    bf16x8 p0, p1, p2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        __bf16 b0, b1, b2;
        split_bf16x3(v[e], b0, b1, b2);
        p0[e] = b0; p1[e] = b1; p2[e] = b2;
    }
    __bf16* d = dst + (r * (cols >> 5) + (c >> 5)) * 96 + (c & 31);
    *reinterpret_cast<bf16x8*>(d) = p0;
    *reinterpret_cast<bf16x8*>(d + 32) = p1;
    *reinterpret_cast<bf16x8*>(d + 64) = p2;
}

hipError_t launch_pack_x6(const float* src, void* dst, int64_t rows, int cols, int64_t ld, hipStream_t stream) {
    if (cols % 32 != 0 || ld % 4 != 0) return hipErrorInvalidValue;
    const int64_t n = rows * (cols >> 3);
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(pack_x6_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src,
                       static_cast<__bf16*>(dst), rows, cols, ld);
    return hipGetLastError();
}

static int x6_slots() {
    static int slots = 0;
    if (!slots) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        slots = cus;   // one 104-KiB block per CU
    }
    return slots;
}

template <int EPI, int VAR>
static hipError_t launch_x6_one(const GemmParams& p, hipStream_t stream) {
    auto kern = gemm_x6_kernel<EPI, VAR>;
    static PerDevice<bool> attr_done_dev;  // benign race: the attribute call is idempotent
    bool& attr_done = attr_done_dev.get();
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)X6Tile::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int tiles = ((p.M + X6Tile::BM - 1) / X6Tile::BM) * ((p.N + X6Tile::BN - 1) / X6Tile::BN);
    const int grid = tiles < x6_slots() ? tiles : x6_slots();
    hipLaunchKernelGGL(kern, dim3(grid), dim3(X6Tile::NT), X6Tile::LDS_BYTES, stream, p);
    return hipGetLastError();
}

bool gemm_x6_supports(GemmKind kind, const GemmParams& p) {
    if (!p.Wx || p.K % 32 != 0 || p.lda % 4 != 0 || p.ldc % 4 != 0 || p.N % 4 != 0 || p.M <= 0 || p.N <= 0) return false;
    return kind == GK_PLAIN || kind == GK_GELU || kind == GK_RESID || kind == GK_ACCUM || kind == GK_GELUGRAD;
}

template <int VAR>
static hipError_t launch_x6_var(GemmKind kind, const GemmParams& p, hipStream_t s) {
    switch (kind) {
        case GK_PLAIN: return launch_x6_one<EPI_PLAIN, VAR>(p, s);
        case GK_GELU: return launch_x6_one<EPI_GELU, VAR>(p, s);
        case GK_RESID: return launch_x6_one<EPI_RESID, VAR>(p, s);
        case GK_ACCUM: return launch_x6_one<EPI_ACCUM, VAR>(p, s);
        case GK_GELUGRAD: return launch_x6_one<EPI_GELUGRAD, VAR>(p, s);
        default: return hipErrorInvalidValue;
    }
}

// ---- convolution rows on exact operands (the U-Net's bf16x6 mode, round 5) ---------------------------------------------
bool gemm_x6_conv_supports(GemmKind kind, const GemmParams& p) {
    if (kind != GK_PLAIN && kind != GK_RESID) return false;
    if (!p.Wx || !p.A || (!p.C && !p.C2) || p.M <= 0 || p.N <= 0 || p.N % 4 != 0 || p.taps < 1 || p.K % (32 * p.taps) != 0) return false;
    if (p.lda % 4 != 0 || (p.C && p.ldc % 4 != 0) || (p.C2 && p.ldc2 % 4 != 0) || p.a_row_mul < 0 || p.c_row_mul < 0) return false;
    if (kind == GK_RESID && (!p.R || (p.r_ld ? p.r_ld : p.ldc) % 4 != 0)) return false;
    if (p.tp && (p.t_lo < 0 || p.t_hi > p.tp || p.t_lo >= p.t_hi)) return false;
    return true;
}

template <int EPI, int VAR>
static hipError_t launch_x6_conv_one(const GemmParams& p, hipStream_t stream) {
    auto kern = gemm_x6_kernel<EPI, VAR, true>;
    static PerDevice<bool> attr_done_dev;
    bool& attr_done = attr_done_dev.get();
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)X6Tile::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int tiles = ((p.M + X6Tile::BM - 1) / X6Tile::BM) * ((p.N + X6Tile::BN - 1) / X6Tile::BN);
    const int grid = tiles < x6_slots() ? tiles : x6_slots();
    hipLaunchKernelGGL(kern, dim3(grid), dim3(X6Tile::NT), X6Tile::LDS_BYTES, stream, p);
    return hipGetLastError();
}

hipError_t launch_gemm_x6_conv(GemmKind kind, const GemmParams& p, hipStream_t s, int variant) {
    if (!gemm_x6_conv_supports(kind, p)) return hipErrorInvalidValue;
    if (kind == GK_RESID) return variant == 2 ? launch_x6_conv_one<EPI_RESID, 2>(p, s) : launch_x6_conv_one<EPI_RESID, 0>(p, s);
    return variant == 2 ? launch_x6_conv_one<EPI_PLAIN, 2>(p, s) : launch_x6_conv_one<EPI_PLAIN, 0>(p, s);
}

hipError_t launch_gemm_x6(GemmKind kind, const GemmParams& p, hipStream_t s, int variant) {
    if (!gemm_x6_supports(kind, p)) return hipErrorInvalidValue;
    return variant == 2 ? launch_x6_var<2>(kind, p, s) : variant == 1 ? launch_x6_var<1>(kind, p, s) : launch_x6_var<0>(kind, p, s);
}

}  // namespace cmdi
