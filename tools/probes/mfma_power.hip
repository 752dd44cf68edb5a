// Is the f16 matrix pipe power-limited on random data?  Pure-MFMA loops (no memory traffic inside the loop) on register-resident
// operands: v_mfma_f32_32x32x16_{f16,bf16}, random vs zero operands, 1 / 2 / 4 waves per SIMD on every CU.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip && ./mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int BF>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ src, float* out, int iters) {
    const int tid = threadIdx.x + blockIdx.x * blockDim.x;
    uint4 ra[4], rw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { ra[i] = src[(tid * 8 + i) & 0xFFFF]; rw[i] = src[(tid * 8 + 4 + i) & 0xFFFF]; }
    f16v acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if constexpr (BF) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(b8*)&ra[i], *(b8*)&rw[(i + j) & 3], acc[i * 2 + j], 0, 0, 0);
                else acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(h8*)&ra[i], *(h8*)&rw[(i + j) & 3], acc[i * 2 + j], 0, 0, 0);
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[tid] = s;
}

int main() {
    const int n = 1 << 16;
    std::vector<uint16_t> h(n * 8);
    uint4* d; float* o;
    hipMalloc(&d, n * 16); hipMalloc(&o, 1 << 22);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int fill = 0; fill < 3; ++fill) {
        // 0: zeros; 1: random f16/bf16 in [-1, 1) full mantissa; 2: random small magnitudes with random signs (like lo planes)
        srand(1);
        for (auto& v : h) {
            if (fill == 0) v = 0;
            else {
                float x = (rand() / (float)RAND_MAX) * 2.f - 1.f;
                _Float16 f = (_Float16)x; v = *(uint16_t*)&f;     // as f16 bits; as bf16 bits it is still a "random" pattern
            }
        }
        hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
        for (int bf = 0; bf < 2; ++bf)
            for (int wps = 1; wps <= 4; wps *= 2) {
                const int blocks = 256 * wps, iters = 20000;
                auto run = [&]() { if (bf) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
                                   else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, o, iters); };
                run(); hipDeviceSynchronize();
                hipEventRecord(e0); run(); run(); run(); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
                const double flops = (double)blocks * 4 * iters * 8 * 2.0 * 32 * 32 * 16;
                printf("fill=%d %s waves/SIMD=%d  %.2f ms  %.0f TF (%.1f%% of 2500)  eff clock if 100%% busy = %.2f GHz\n", fill,
                       bf ? "bf16" : "f16 ", wps, ms, flops / ms / 1e9, flops / ms / 1e9 / 25.0,
                       flops / ms / 1e9 / 2500.0 * 2.4);
            }
    }
    return 0;
}
