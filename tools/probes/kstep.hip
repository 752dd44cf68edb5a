// The K step of the split-f16 GEMM (csrc/gemm_h3.hpp gemm_h3_body) under compile-time ablations (CMDI_KABL), as ONE block alone
// on the chip and as the full in_proj grid: which part of the step is the time?
//   for v in 0 1 2 4 8 ...; do hipcc --offload-arch=gfx950 -O3 -DCMDI_PROBES -DCMDI_KABL=$v -I ../../diffusion-motion-inbetweening_amd/csrc \
//       -o kstep_$v kstep.hip; done
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "gemm_h3.hpp"
using namespace cmdi;

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#ifndef KTILE
#define KTILE 8
#endif
#if KTILE == 8
using TC = H3Tile<128, 128, 4, 2, 2, 4>;
#elif KTILE == 7
using TC = H3Tile<128, 128, 4, 2, 3, 2>;
#elif KTILE == 21
using TC = H3Tile<64, 128, 2, 4, 2, 2>;
#elif KTILE == 6
using TC = H3Tile<64, 128, 2, 2, 2, 2>;
#endif

template <int EPI>
static void run(const char* what, int M, int N, int K, _Float16* A, _Float16* W, float* bias, _Float16* Cs, long long* dbg) {
    H3Params p{};
    p.A = A; p.W = W; p.bias = bias; p.Cs = Cs; p.M = M; p.N = N; p.K = K; p.ldc = N;
    p.dbg = 16; p.dbg_buf = dbg;
    auto kern = gemm_h3_kernel<TC, EPI>;
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)TC::LDS_BYTES));
    const int tiles = ((M + TC::BM - 1) / TC::BM) * ((N + TC::BN - 1) / TC::BN);
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(tiles), dim3(TC::NT), TC::LDS_BYTES, 0, p);
    CHK(hipDeviceSynchronize());
    const int reps = 20;
    CHK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(tiles), dim3(TC::NT), TC::LDS_BYTES, 0, p);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h((size_t)tiles * 6);
    CHK(hipMemcpy(h.data(), dbg, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    double life = 0, loop = 0, wall = 0;
    for (int b = 0; b < tiles; ++b) { life += h[6 * b + 3] - h[6 * b]; loop += h[6 * b + 2]; wall += h[6 * b + 5] - h[6 * b + 4]; }
    life /= tiles; loop /= tiles; wall /= tiles;
    printf("KABL %2d tile %2d %-10s M %5d N %4d K %4d: %4d blocks, kernel %6.1f us; block life %6.0f cycles, K loop %6.0f = %5.0f / step, rest %5.0f; clock %.2f GHz\n",
           CMDI_KABL, KTILE, what, M, N, K, tiles, ms * 1e3 / reps, life, loop, loop / (K / 32), life - loop, life / (wall * 10.0));
    fflush(stdout);
}

int main() {
    const int M = 12608, N = 1536, K = 1024;
    _Float16 *A, *W, *Cs; float* bias; long long* dbg;
    CHK(hipMalloc(&A, (size_t)M * 2 * K * 2)); CHK(hipMalloc(&W, (size_t)N * 2 * K * 2)); CHK(hipMalloc(&Cs, (size_t)M * 2 * N * 2));
    CHK(hipMalloc(&bias, N * 4)); CHK(hipMalloc(&dbg, (size_t)4096 * 6 * 8));
    std::vector<_Float16> h((size_t)M * 2 * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)(((int)((i * 2654435761u) >> 20) % 2001 - 1000) * 1e-3f);
    CHK(hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CHK(hipMemcpy(W, h.data(), (size_t)N * 2 * K * 2, hipMemcpyHostToDevice));
    CHK(hipMemset(bias, 0, N * 4));
    run<H3_PLAIN_SPLIT>("lone", TC::BM, TC::BN, 512, A, W, bias, Cs, dbg);
    run<H3_PLAIN_SPLIT>("lone", TC::BM, TC::BN, 1024, A, W, bias, Cs, dbg);
    run<H3_PLAIN_SPLIT>("1/CU", TC::BM * 16, TC::BN * 16, 512, A, W, bias, Cs, dbg);
    run<H3_PLAIN_SPLIT>("2/CU", TC::BM * 32, TC::BN * 16, 512, A, W, bias, Cs, dbg);
    run<H3_PLAIN_SPLIT>("in_proj", 12608, 1536, 512, A, W, bias, Cs, dbg);
    return 0;
}
