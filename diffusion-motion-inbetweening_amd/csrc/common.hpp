// Shared device helpers for the CondMDI gfx950 engine (fp32 everywhere, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cmdi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;

// Function attributes (dynamic LDS size) and CU counts belong to a DEVICE: a launcher's one-time setup is keyed by the
// current device, so a process that drives several GPUs sets every one of them up.  64 slots cover a node in
// CPX mode (8 sockets x 8 partitions).  A device id beyond the table — or a failing hipGetDevice — gets NO cache: get()
// hands out a fresh zero-initialised scratch value, so the attribute call / CU query is repeated on every launch there
// instead of being skipped because some other device had done it.
constexpr int kMaxDevices = 64;
template <class T>
struct PerDevice {
    T slot[kMaxDevices] = {};
    T& get() {
        static thread_local T scratch;
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) {
            scratch = T{};
            return scratch;
        }
        return slot[d];
    }
};

// Compute units of the current device (256 on an MI355X in SPX mode; fewer in CPX partitions), queried once per device.
inline int device_cu_count() {
    static PerDevice<int> cus_dev;
    int& cus = cus_dev.get();
    if (!cus) {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        cus = n > 0 ? n : 256;
    }
    return cus;
}

// Row r (0..15) of a 32x32 MFMA C/D fragment held by lane `lane`:
// col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
__device__ __forceinline__ int mfma32_row(int r, int lane) {
    return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// erf(x) = sign(x) (1 - exp(-t P(t))), t = min(|x|, 3.95): P is a degree-7 fit of -log(erfc(t)) / t on (0, 3.95] (beyond it
// erf rounds to 1 in fp32).  Branch-free — the library erff takes one of two paths per lane and a wave usually pays for
// both: the GELU epilogue of linear1 spent as long in it as the plain epilogue spends in total.  Max abs error against
// float64 over [0, 4.2], evaluated in fp32: 8.8e-8 (a correctly rounded erf: 3.0e-8); oracle/test: test_gelu_epilogue_accuracy.
__device__ __forceinline__ float erf_poly(float x) {
    const float t = fminf(fabsf(x), 3.95f);
    float p = 3.1441086321137846e-05f;
    p = __builtin_fmaf(p, t, -0.0003088079974986613f);
    p = __builtin_fmaf(p, t, 0.0010324155446141958f);
    p = __builtin_fmaf(p, t, 0.0005369179998524487f);
    p = __builtin_fmaf(p, t, -0.01958395168185234f);
    p = __builtin_fmaf(p, t, 0.10291960835456848f);
    p = __builtin_fmaf(p, t, 0.636597752571106f);
    p = __builtin_fmaf(p, t, 1.128380298614502f);
    return copysignf(1.0f - __expf(-t * p), x);
}
__device__ __forceinline__ float gelu_erf(float x) {
    // F.gelu default (exact erf form), torch/nn/functional.py; used by
    // nn.TransformerEncoderLayer(activation="gelu") at model/mdm.py:107-112.
    return 0.5f * x * (1.0f + erf_poly(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erf_poly(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }

// Mish(x) = x tanh(softplus(x)) (nn.Mish, softplus threshold 20).  With w = e^x: tanh(ln(1 + w)) = n / (n + 2), n = w (w + 2)
// — ONE exponential and one division instead of expf + log1pf + tanhf (the libm trio made the GroupNorm kernels
// VALU-bound); same fp32 accuracy (max rel. error 3.4e-7 vs 2.6e-7 for the three-call form, against float64).
__device__ __forceinline__ float mish_f(float x) {
    if (x > 20.f) return x;
    const float w = expf(x), n = w * (w + 2.f);
    return x * (n / (n + 2.f));
}
// d Mish / dx = t + x (1 - t^2) sigmoid(x), t = n / (n + 2); 1 - t^2 = 2 (1 + t) / (n + 2) avoids the cancellation
__device__ __forceinline__ float mish_grad_f(float x) {
    if (x > 20.f) return 1.f;
    const float w = expf(x), n = w * (w + 2.f);
    const float u = 1.f / (n + 2.f), t = n * u;
    return t + x * (2.f * u * (1.f + t)) * (w / (1.f + w));
}

// Power-of-two "loss scale" of the reconstruction-guidance backward pass on the f16 matrix pipe:
// bits = float bits of max|g| over the output gradient; the scale moves that maximum to [2^6, 2^7)
// so the whole gradient chain sits well inside the f16 range of the split operands (gemm_h3.hpp).
// Exact (a power of two) and undone at the end of the chain, which is linear in g.
__device__ __forceinline__ float grad_scale_from_bits(unsigned bits) {
    const int e = (int)(bits >> 23) - 127;          // floor(log2(max)); bits == 0 -> -127
    if (bits == 0u || e > 120) return 1.0f;          // zero / inf / nan gradient: leave as is
    int sh = 6 - e;
    sh = sh < -100 ? -100 : (sh > 100 ? 100 : sh);
    return __uint_as_float((unsigned)(127 + sh) << 23);
}

// XCD-aware bijective remap of a linear block id (guide T1): blocks that the dispatcher places on
// the same XCD (id % 8) get a contiguous chunk of the tile space, so tiles sharing an operand
// panel hit the same private L2.  Speed only; any mapping is correct.
__device__ __forceinline__ int xcd_remap(int id, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7;
    const int xcd = id & 7, idx = id >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

}  // namespace cmdi
