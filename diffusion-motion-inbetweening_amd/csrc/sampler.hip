// Sampler arithmetic of the CondMDI hot loop, one fused HBM-bound pass per denoising step:
//   classifier-free combine      model/cfg_sampler.py:35
//   imputation / recon guidance  diffusion/gaussian_diffusion.py:405-435
//   eps -> x0                    diffusion/gaussian_diffusion.py:536-541
//   posterior mean + noise       diffusion/gaussian_diffusion.py:330-349,696-711   (DDPM)
//   DDIM update                  diffusion/gaussian_diffusion.py:1395-1412
// plus q_sample (:311-328) and the counter-based N(0,1) generator that replaces th.randn_like.
//
// This file is compiled with -ffp-contract=off: every product and sum below is rounded separately,
// in the reference's evaluation order, so given identical inputs the result is bit-identical to the
// reference's chain of elementwise fp32 torch ops (and to oracle/diffusion_oracle.py).
#include "common.hpp"
#include "kernels.hpp"

namespace cmdi {

// ---- Philox4x32-10 (Salmon et al., SC'11; Random123 reference constants) -------------------------
struct Philox {
    static constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    __host__ __device__ static inline void run(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const uint64_t p0 = (uint64_t)M0 * c[0], p1 = (uint64_t)M1 * c[2];
            const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
            const uint32_t n1 = (uint32_t)p1;
            const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
            const uint32_t n3 = (uint32_t)p0;
            c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
            k0 += W0; k1 += W1;
        }
    }
};

void philox4x32_10_host(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    Philox::run(c, key[0], key[1]);
    for (int i = 0; i < 4; ++i) out[i] = c[i];
}

// Four N(0,1) draws for elements 4q..4q+3 of sample `sample` at loop step `step` (-1 = x_T).
// counter = (q, step+1, sample_lo, sample_hi), key = (seed_lo, seed_hi): a function of the GLOBAL
// sample index only, so any sharding of the batch reproduces the same noise.
__device__ __forceinline__ void normal4(float z[4], uint32_t q, int step, int64_t sample,
                                        uint64_t seed) {
    uint32_t c[4] = {q, (uint32_t)(step + 1), (uint32_t)sample, (uint32_t)((uint64_t)sample >> 32)};
    Philox::run(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float k24 = 5.9604644775390625e-08f;  // 2^-24
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const float u0 = ((float)(c[2 * p] >> 8) + 0.5f) * k24;
        const float u1 = ((float)(c[2 * p + 1] >> 8) + 0.5f) * k24;
        const float r = sqrtf(-2.0f * logf(u0));
        float sn, cs;
        sincosf(6.283185307179586f * u1, &sn, &cs);
        z[2 * p] = r * cs;
        z[2 * p + 1] = r * sn;
    }
}

template <bool VEC>
__device__ __forceinline__ void load4(float v[4], const float* p, int n) {
    if constexpr (VEC) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = i < n ? p[i] : 0.f;
    }
}
template <bool VEC>
__device__ __forceinline__ void store4(float* p, const float v[4], int n) {
    if constexpr (VEC) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < n) p[i] = v[i];
    }
}
template <bool VEC>
__device__ __forceinline__ void loadmask4(bool m[4], const uint8_t* p, int n) {
    if constexpr (VEC) {
        const uint32_t w = *reinterpret_cast<const uint32_t*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) m[i] = ((w >> (8 * i)) & 0xffu) != 0;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) m[i] = i < n ? p[i] != 0 : false;
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void sampler_step_kernel(const SamplerIO io, const StepCoef k_arg,
                                                           int64_t per_sample, uint64_t seed,
                                                           int64_t first_sample, int step_arg,
                                                           const StepCoef* __restrict__ ktab,
                                                           const int* __restrict__ cursor) {
    // graph replay: the step index and its coefficients come from device memory (same values, same
    // arithmetic), so one captured launch sequence serves every step of the chain
    const int step = cursor ? *cursor : step_arg;
    const StepCoef k = ktab ? ktab[step] : k_arg;
    const int b = blockIdx.y;
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t e0 = 4 * q;
    if (e0 >= per_sample) return;
    const int n = (per_sample - e0) < 4 ? (int)(per_sample - e0) : 4;
    const int64_t base = (int64_t)b * per_sample + e0;

    float x[4], oc[4], hat[4], x0[4], nz[4];
    load4<VEC>(x, io.x + base, n);
    load4<VEC>(oc, io.out_c + base, n);
    if (io.out_u) {
        float ou[4];
        load4<VEC>(ou, io.out_u + base, n);
        const float s = io.text_scale[b];
#pragma unroll
        for (int i = 0; i < 4; ++i) hat[i] = ou[i] + (s * (oc[i] - ou[i]));
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) hat[i] = oc[i];
    }

    if (k.mean_eps) {
#pragma unroll
        for (int i = 0; i < 4; ++i) x0[i] = (k.sra * x[i]) - (k.srm1a * hat[i]);
        if (k.clip > 0.f) {   // process_xstart: x.clamp(-clip_range, clip_range) (:499)
#pragma unroll
            for (int i = 0; i < 4; ++i) x0[i] = fminf(fmaxf(x0[i], -k.clip), k.clip);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) x0[i] = hat[i];
    }

    if (k.recon || k.impute) {
        bool m[4];
        float inp[4];
        loadmask4<VEC>(m, io.mask + base, n);
        load4<VEC>(inp, io.inpaint + base, n);
        if (k.recon) {
            float g[4];
            load4<VEC>(g, io.grad_c + base, n);
            if (io.grad_u) {
                float gu[4];
                load4<VEC>(gu, io.grad_u + base, n);
#pragma unroll
                for (int i = 0; i < 4; ++i) g[i] = g[i] + gu[i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float tilde = hat[i] - (k.gcoef * g[i]);
                x0[i] = m[i] ? (k.impute ? inp[i] : hat[i]) : tilde;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) x0[i] = m[i] ? inp[i] : hat[i];
        }
    }

    if (io.pred_xstart) store4<VEC>(io.pred_xstart + base, x0, n);

    if (io.noise) {
        load4<VEC>(nz, io.noise + base, n);
    } else {
        normal4(nz, (uint32_t)q, step, first_sample + b, seed);
    }

    float o[4];
    if (!k.ddim) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float mean = (k.c1 * x0[i]) + (k.c2 * x[i]);
            o[i] = mean + (k.sig_nz * nz[i]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float eps = ((k.sra * x[i]) - x0[i]) / k.srm1a;
            const float mean = (x0[i] * k.sqrt_abp) + (k.dir * eps);
            o[i] = mean + (k.sig_nz * nz[i]);
        }
    }
    store4<VEC>(io.x + base, o, n);
}

hipError_t launch_sampler_step(const SamplerIO& io, const StepCoef& k, int batch, int64_t per_sample,
                               uint64_t seed, int64_t first_sample, int step, hipStream_t stream,
                               const StepCoef* ktab, const int* cursor) {
    const int64_t quads = (per_sample + 3) / 4;
    const dim3 grid((unsigned)((quads + 255) / 256), batch), block(256);
    if (per_sample % 4 == 0)
        hipLaunchKernelGGL(sampler_step_kernel<true>, grid, block, 0, stream, io, k, per_sample, seed,
                           first_sample, step, ktab, cursor);
    else
        hipLaunchKernelGGL(sampler_step_kernel<false>, grid, block, 0, stream, io, k, per_sample,
                           seed, first_sample, step, ktab, cursor);
    return hipGetLastError();
}

__global__ void cursor_add_kernel(int* cursor, int delta) { *cursor += delta; }
hipError_t launch_cursor_add(int* cursor, int delta, hipStream_t stream) {
    hipLaunchKernelGGL(cursor_add_kernel, dim3(1), dim3(1), 0, stream, cursor, delta);
    return hipGetLastError();
}

// Seed of the reconstruction-guidance VJP: d loss / d hat_x with loss = sum(m * (inp - hat)^2)
// (gaussian_diffusion.py:415), split over the two CFG passes of cfg_sampler.py:35
// (hat = out_u + s*(out_c - out_u)  =>  d out_c = s*g,  d out_u = g - s*g).
__global__ void recon_gout_kernel(const float* __restrict__ out_c, const float* __restrict__ out_u,
                                  const float* __restrict__ text_scale,
                                  const uint8_t* __restrict__ mask, const float* __restrict__ inp,
                                  float* __restrict__ gc, float* __restrict__ gu,
                                  int64_t per_sample) {
    const int b = blockIdx.y;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < per_sample;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = (int64_t)b * per_sample + e;
        float hat = out_c[i];
        float s = 1.f;
        if (out_u) {
            s = text_scale[b];
            hat = out_u[i] + (s * (out_c[i] - out_u[i]));
        }
        const float g = mask[i] ? 2.0f * (hat - inp[i]) : 0.f;
        if (out_u) {
            const float sg = s * g;
            gc[i] = sg;
            gu[i] = g - sg;
        } else {
            gc[i] = g;
        }
    }
}

hipError_t launch_recon_gout(const float* out_c, const float* out_u, const float* text_scale,
                             const uint8_t* mask, const float* inpaint, float* gout_c, float* gout_u,
                             int batch, int64_t per_sample, hipStream_t stream) {
    int64_t bx = (per_sample + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(recon_gout_kernel, dim3((unsigned)bx, batch), dim3(256), 0, stream, out_c,
                       out_u, text_scale, mask, inpaint, gout_c, gout_u, per_sample);
    return hipGetLastError();
}

__global__ void cfg_combine_kernel(const float* __restrict__ out_c, const float* __restrict__ out_u,
                                   const float* __restrict__ text_scale, float* __restrict__ out,
                                   int64_t per_sample) {
    const int b = blockIdx.y;
    const float s = text_scale[b];
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < per_sample;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = (int64_t)b * per_sample + e;
        out[i] = out_u[i] + (s * (out_c[i] - out_u[i]));
    }
}
hipError_t launch_cfg_combine(const float* out_c, const float* out_u, const float* text_scale,
                              float* out, int batch, int64_t per_sample, hipStream_t stream) {
    int64_t bx = (per_sample + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(cfg_combine_kernel, dim3((unsigned)bx, batch), dim3(256), 0, stream, out_c,
                       out_u, text_scale, out, per_sample);
    return hipGetLastError();
}

// Backward of the CFG combine: d out_c = s*g, d out_u = g - s*g.
__global__ void cfg_split_kernel(const float* __restrict__ g, const float* __restrict__ text_scale,
                                 float* __restrict__ gc, float* __restrict__ gu, int64_t per_sample) {
    const int b = blockIdx.y;
    const float s = text_scale[b];
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < per_sample;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = (int64_t)b * per_sample + e;
        const float sg = s * g[i];
        gc[i] = sg;
        gu[i] = g[i] - sg;
    }
}
hipError_t launch_cfg_split(const float* g, const float* text_scale, float* gc, float* gu, int batch,
                            int64_t per_sample, hipStream_t stream) {
    int64_t bx = (per_sample + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(cfg_split_kernel, dim3((unsigned)bx, batch), dim3(256), 0, stream, g,
                       text_scale, gc, gu, per_sample);
    return hipGetLastError();
}

__global__ void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                float* __restrict__ out, float a, float b, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (a * x0[i]) + (b * noise[i]);
}
hipError_t launch_q_sample(const float* x0, const float* noise, float* out, float a, float b,
                           int64_t n, hipStream_t stream) {
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(q_sample_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x0, noise, out,
                       a, b, n);
    return hipGetLastError();
}

__global__ void randn_kernel(float* __restrict__ out, int64_t per_sample, uint64_t seed,
                             int64_t first_sample, int step) {
    const int b = blockIdx.y;
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t e0 = 4 * q;
    if (e0 >= per_sample) return;
    float z[4];
    normal4(z, (uint32_t)q, step, first_sample + b, seed);
    float* p = out + (int64_t)b * per_sample + e0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (e0 + i < per_sample) p[i] = z[i];
}
hipError_t launch_randn(float* out, int batch, int64_t per_sample, uint64_t seed,
                        int64_t first_sample, int step, hipStream_t stream) {
    const int64_t quads = (per_sample + 3) / 4;
    hipLaunchKernelGGL(randn_kernel, dim3((unsigned)((quads + 255) / 256), batch), dim3(256), 0,
                       stream, out, per_sample, seed, first_sample, step);
    return hipGetLastError();
}

}  // namespace cmdi
