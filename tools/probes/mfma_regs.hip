// Issue rate of v_mfma_f32_32x32x16_f16 from ONE wave per SIMD by register class of its operands (gemm_h3w.hpp question: the
// weight-stationary stream measured 42 cycles per MFMA with C/D in vector registers and B in accumulation registers).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_regs mfma_regs.hip && ./mfma_regs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// CD: 0 = accumulation registers, 1 = vector registers; B: 0 = vector, 1 = accumulation; PAT: 0 = c1 c0 c1 (the split product's
// order: two dependent MFMAs meet at every step boundary), 1 = four accumulators round robin
template <int CD, int B, int PAT>
__global__ __launch_bounds__(256, 1) void k(const uint4* __restrict__ src, float* out, long long* cyc, int iters) {
    const int tid = threadIdx.x + blockIdx.x * blockDim.x;
    h8 a0 = *(const h8*)&src[(tid * 4) & 0xFFFF], a1 = *(const h8*)&src[(tid * 4 + 1) & 0xFFFF];
    h8 b0 = *(const h8*)&src[(tid * 4 + 2) & 0xFFFF], b1 = *(const h8*)&src[(tid * 4 + 3) & 0xFFFF];
    f16v c0, c1, c2, c3;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#define M(c, a, b)                                                                                                          \
    if constexpr (CD == 0 && B == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));     \
    else if constexpr (CD == 0 && B == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "a"(b)); \
    else if constexpr (CD == 1 && B == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); \
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
            if constexpr (PAT == 0) { M(c1, a0, b1) M(c0, a0, b0) M(c1, a1, b0) }
            else { M(c0, a0, b1) M(c1, a0, b0) M(c2, a1, b0) M(c3, a1, b1) }
#undef M
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 12345.678f) out[tid] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CD, int B, int PAT>
static void run(const uint4* d, float* o, long long* c, const char* name) {
    const int iters = 2000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<CD, B, PAT>), dim3(blocks), dim3(256), 0, 0, d, o, c, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<CD, B, PAT>), dim3(blocks), dim3(256), 0, 0, d, o, c, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), c, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
    const double n = (double)iters * 8 * (PAT == 0 ? 3 : 4);
    printf("%-46s %6.2f cycles/MFMA  (%.3f ms, %.0f TF)\n", name, mean / n, ms, n * 4 * blocks * 32768.0 * 2 / 2 / (ms * 1e-3) / 1e12);
}

int main() {
    const int n = 1 << 16;
    std::vector<uint16_t> h(n * 8);
    srand(1);
    for (auto& v : h) { _Float16 f = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f); v = *(uint16_t*)&f; }
    uint4* d; float* o; long long* c;
    hipMalloc(&d, n * 16); hipMalloc(&o, 1 << 22); hipMalloc(&c, 4096);
    hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
    run<0, 0, 0>(d, o, c, "C/D acc, B vec, c1 c0 c1");
    run<0, 1, 0>(d, o, c, "C/D acc, B acc, c1 c0 c1");
    run<1, 0, 0>(d, o, c, "C/D vec, B vec, c1 c0 c1");
    run<1, 1, 0>(d, o, c, "C/D vec, B acc, c1 c0 c1   (gemm_h3w round 6)");
    run<0, 0, 1>(d, o, c, "C/D acc, B vec, 4 independent");
    run<0, 1, 1>(d, o, c, "C/D acc, B acc, 4 independent");
    run<1, 0, 1>(d, o, c, "C/D vec, B vec, 4 independent");
    run<1, 1, 1>(d, o, c, "C/D vec, B acc, 4 independent");
    return 0;
}
