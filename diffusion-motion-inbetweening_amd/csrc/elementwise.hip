// HBM-bound row kernels of the MDM denoiser: LayerNorm (post-norm, eps 1e-5, biased variance —
// nn.TransformerEncoderLayer.norm1/norm2 built at model/mdm.py:107-114), its backward, and the
// conditioning-token assembly of MDM.forward (model/mdm.py:245-251,279-280).
#include "common.hpp"
#include "gemm_h3.hpp"
#include "kernels.hpp"

namespace cmdi {

// One wave per row; a lane owns NV float4 at columns lane*4 + v*256 -> every wave-instruction is a
// contiguous 1 KiB access.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        float* __restrict__ y,
                                                        _Float16* __restrict__ ys,
                                                        int* __restrict__ range_flag,
                                                        float* __restrict__ stats, int rows,
                                                        const _Float16* __restrict__ xs_in) {
    constexpr int D = NV * 256;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * D;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (xs_in) {   // the input arrives as split rows (hi + lo * 2^-11): the pre-LayerNorm stream of the folded path
            const _Float16* sp = xs_in + (size_t)row * (2 * D) + split_pos(i * 256 + lane * 4);
            const h4 a = *reinterpret_cast<const h4*>(sp), b = *reinterpret_cast<const h4*>(sp + 32);
            v[i] = make_float4((float)a[0] + (float)b[0] * kLoInv, (float)a[1] + (float)b[1] * kLoInv,
                               (float)a[2] + (float)b[2] * kLoInv, (float)a[3] + (float)b[3] * kLoInv);
        } else {
            v[i] = *reinterpret_cast<const float4*>(xr + i * 256 + lane * 4);
        }
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
    }
    const float var = wave_sum(q) * (1.0f / D);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    if (stats && lane == 0) {
        stats[2 * row] = mean;
        stats[2 * row + 1] = rstd;
    }
    float* yr = y + (size_t)row * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float4 g = *reinterpret_cast<const float4*>(gamma + i * 256 + lane * 4);
        const float4 bb = *reinterpret_cast<const float4*>(beta + i * 256 + lane * 4);
        float4 o;
        o.x = (v[i].x - mean) * rstd * g.x + bb.x;
        o.y = (v[i].y - mean) * rstd * g.y + bb.y;
        o.z = (v[i].z - mean) * rstd * g.z + bb.z;
        o.w = (v[i].w - mean) * rstd * g.w + bb.w;
        if (y) *reinterpret_cast<float4*>(yr + i * 256 + lane * 4) = o;
        if (ys) {
            const float ov[4] = {o.x, o.y, o.z, o.w};
            h4 oh, ol;
            bool overflow = false;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                _Float16 a, c;
                split_f16(ov[e], a, c);
                oh[e] = a; ol[e] = c;
                overflow |= !(fabsf(ov[e]) < 65504.0f);
            }
            _Float16* d = ys + (size_t)row * (2 * D) + split_pos(i * 256 + lane * 4);
            *reinterpret_cast<h4*>(d) = oh;
            *reinterpret_cast<h4*>(d + 32) = ol;
            if (overflow && range_flag) atomicOr(range_flag, 1);
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = gamma * dy,  xhat = (x - mean) * rstd
template <int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ stats,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ dy,
                                                            float* __restrict__ dx,
                                                            _Float16* __restrict__ dxs, int rows,
                                                            const _Float16* __restrict__ xs_in) {
    constexpr int D = NV * 256;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    const float* xr = x + (size_t)row * D;
    const float* dyr = dy + (size_t)row * D;
    float4 xh[NV], g[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float4 xv;
        if (xs_in) {   // x as the split rows the forward pass multiplied (a stash of the folded schedule)
            const _Float16* sp = xs_in + (size_t)row * (2 * D) + split_pos(i * 256 + lane * 4);
            const h4 a = *reinterpret_cast<const h4*>(sp), b = *reinterpret_cast<const h4*>(sp + 32);
            xv = make_float4((float)a[0] + (float)b[0] * kLoInv, (float)a[1] + (float)b[1] * kLoInv,
                             (float)a[2] + (float)b[2] * kLoInv, (float)a[3] + (float)b[3] * kLoInv);
        } else {
            xv = *reinterpret_cast<const float4*>(xr + i * 256 + lane * 4);
        }
        const float4 dv = *reinterpret_cast<const float4*>(dyr + i * 256 + lane * 4);
        const float4 gm = *reinterpret_cast<const float4*>(gamma + i * 256 + lane * 4);
        xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd,
                            (xv.w - mean) * rstd);
        g[i] = make_float4(dv.x * gm.x, dv.y * gm.y, dv.z * gm.z, dv.w * gm.w);
        s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
        s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
    }
    const float m1 = wave_sum(s1) * (1.0f / D);
    const float m2 = wave_sum(s2) * (1.0f / D);
    // dx (fp32) is optional (round 4): the split-f16 chain adds the residual gradient from the split rows it multiplies anyway
    // (H3Params::Rs), so its LayerNorm backward writes one tensor instead of two
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float4 o;
        o.x = rstd * (g[i].x - m1 - xh[i].x * m2);
        o.y = rstd * (g[i].y - m1 - xh[i].y * m2);
        o.z = rstd * (g[i].z - m1 - xh[i].z * m2);
        o.w = rstd * (g[i].w - m1 - xh[i].w * m2);
        if (dx) *reinterpret_cast<float4*>(dx + (size_t)row * D + i * 256 + lane * 4) = o;
        if (dxs) {   // split rows for the f16-pipe dX GEMMs (gradients: no range flag, see api_denoiser.hip)
            const float ov[4] = {o.x, o.y, o.z, o.w};
            h4 oh, ol;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                _Float16 a, c;
                split_f16(ov[e], a, c);
                oh[e] = a; ol[e] = c;
            }
            _Float16* d = dxs + (size_t)row * (2 * D) + split_pos(i * 256 + lane * 4);
            *reinterpret_cast<h4*>(d) = oh;
            *reinterpret_cast<h4*>(d + 32) = ol;
        }
    }
}

hipError_t launch_layernorm(const float* x, const float* gamma, const float* beta, float* y,
                            _Float16* y_split, int* range_flag, float* stats, int rows, int d,
                            hipStream_t stream, const _Float16* x_split_in) {
    const dim3 grid((rows + 3) / 4), block(256);
    switch (d) {
        case 256: hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, stream, x, gamma, beta, y, y_split, range_flag, stats, rows, x_split_in); break;
        case 512: hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, stream, x, gamma, beta, y, y_split, range_flag, stats, rows, x_split_in); break;
        case 768: hipLaunchKernelGGL(layernorm_kernel<3>, grid, block, 0, stream, x, gamma, beta, y, y_split, range_flag, stats, rows, x_split_in); break;
        case 1024: hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, stream, x, gamma, beta, y, y_split, range_flag, stats, rows, x_split_in); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// LayerNorm folded into the consuming GEMM (gemm_params.hpp): Wf = W diag(gamma), c1[n] = sum_k Wf[n,k],
// c2[n] = sum_k W[n,k] beta[k] + bias[n].  One block per output row n; sums in double.
__global__ __launch_bounds__(256) void fold_ln_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ bias,
                                                      float* __restrict__ Wf, float* __restrict__ c1,
                                                      float* __restrict__ c2, int K) {
    __shared__ double red[2][4];
    const int n = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float w = W[(size_t)n * K + k];
        const float wf = w * gamma[k];
        Wf[(size_t)n * K + k] = wf;
        s1 += (double)wf;
        s2 += (double)w * (double)beta[k];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        c1[n] = (float)((red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
        c2[n] = (float)((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]) + (double)(bias ? bias[n] : 0.f));
    }
}
hipError_t launch_fold_ln(const float* W, const float* gamma, const float* beta, const float* bias, float* Wf, float* c1,
                          float* c2, int N, int K, hipStream_t stream) {
    hipLaunchKernelGGL(fold_ln_kernel, dim3(N), dim3(256), 0, stream, W, gamma, beta, bias, Wf, c1, c2, K);
    return hipGetLastError();
}

hipError_t launch_layernorm_bwd(const float* x, const float* stats, const float* gamma,
                                const float* dy, float* dx, _Float16* dx_split, int rows, int d,
                                hipStream_t stream, const _Float16* x_split) {
    const dim3 grid((rows + 3) / 4), block(256);
    switch (d) {
        case 256: hipLaunchKernelGGL(layernorm_bwd_kernel<1>, grid, block, 0, stream, x, stats, gamma, dy, dx, dx_split, rows, x_split); break;
        case 512: hipLaunchKernelGGL(layernorm_bwd_kernel<2>, grid, block, 0, stream, x, stats, gamma, dy, dx, dx_split, rows, x_split); break;
        case 768: hipLaunchKernelGGL(layernorm_bwd_kernel<3>, grid, block, 0, stream, x, stats, gamma, dy, dx, dx_split, rows, x_split); break;
        case 1024: hipLaunchKernelGGL(layernorm_bwd_kernel<4>, grid, block, 0, stream, x, stats, gamma, dy, dx, dx_split, rows, x_split); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// tok[b*S][:] = (time_table[t_b] + text_term[b]) + pe[0]   — emb = embed_timestep(t); emb +=
// embed_text(mask_cond(enc_text)); xseq = cat(emb, x) + pe[:S]   (model/mdm.py:245-251,279-280,334)
__global__ void token0_kernel(float* __restrict__ tok, const float* __restrict__ time_table,
                              const float* __restrict__ text_term, const float* __restrict__ pe,
                              const int64_t* __restrict__ t_dev, int64_t t_scalar, int n_per_pass,
                              int S, int d, int n_time_rows, const int64_t* __restrict__ tmap_dev,
                              const int* __restrict__ cursor, _Float16* __restrict__ tok_split,
                              int* __restrict__ range_flag) {
    const int b = blockIdx.x;  // sequence index in [0, n_seq); timesteps repeat per CFG pass
    int64_t t = cursor ? tmap_dev[*cursor] : (t_dev ? t_dev[b % n_per_pass] : t_scalar);
    if (t < 0 || t >= n_time_rows) {
        // the reference indexes pe[timesteps] and raises; here the row is clamped (no fault) and bit 1 of the status
        // flag reports it (cmdi_range_status)
        if (range_flag && threadIdx.x == 0) atomicOr(range_flag, 2);
        t = t < 0 ? 0 : n_time_rows - 1;
    }
    for (int n = threadIdx.x; n < d; n += blockDim.x) {
        float e = time_table[(size_t)t * d + n];
        if (text_term) e += text_term[(size_t)b * d + n];
        const float v = e + pe[n];
        if (tok_split) {   // split rows (gemm_h3.hpp): the f16-pipe layers read nothing else
            _Float16 h, l;
            split_f16(v, h, l);
            _Float16* dst = tok_split + (size_t)b * S * (2 * d) + split_pos(n);
            dst[0] = h; dst[32] = l;
            if (!(fabsf(v) < 65504.0f) && range_flag) atomicOr(range_flag, 1);
        } else {
            tok[(size_t)b * S * d + n] = v;
        }
    }
}

// Frame rows for the input projection on the f16 pipe: x [nb][C][T] (T contiguous) -> split rows
// [nb*T][2*Kp] (Kp = C rounded up to 32, zero padded).  A block transposes 32 frames x 64 features of one sequence
// through LDS: reads run along T, writes along the features.  (Round 4: the feature axis is part of the grid — one block
// per (32 frames, sequence) looped 36 times over its 288 feature rows and took 13-15 us at EVERY batch size.)
constexpr int POSE_FC = 64;   // features per block = two 32-column chunks of the split row
__global__ __launch_bounds__(256) void pose_rows_split_kernel(const float* __restrict__ x, _Float16* __restrict__ xs,
                                                             int C, int T, int Kp, int* __restrict__ range_flag,
                                                             const unsigned* __restrict__ gs_bits) {
    __shared__ float tile[POSE_FC * 33];
    const float gscale = gs_bits ? grad_scale_from_bits(*gs_bits) : 1.0f;   // output gradients: the chain's power-of-two scale
    const int b = blockIdx.y, t0 = blockIdx.x * 32, c0 = blockIdx.z * POSE_FC;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 frames x 8 feature lanes
    const int nc = Kp - c0 < POSE_FC ? Kp - c0 : POSE_FC;     // 64, or 32 in the last block of a 288-column row
    const float* xb = x + (size_t)b * C * T;
    bool overflow = false;
    for (int c = ty; c < nc; c += 8) {
        float v = 0.f;
        if (c0 + c < C && t0 + tx < T) v = xb[(size_t)(c0 + c) * T + t0 + tx] * gscale;
        overflow |= !(fabsf(v) < 65504.0f);
        tile[c * 33 + tx] = v;
    }
    __syncthreads();
    const int chunks = nc >> 3;   // 8 consecutive features per thread-item
    for (int it = threadIdx.x; it < 32 * chunks; it += 256) {
        const int fr = it / chunks, c = (it - fr * chunks) * 8;
        if (t0 + fr >= T) continue;
        h8 oh, ol;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            _Float16 h, l;
            split_f16(tile[(c + e) * 33 + fr], h, l);
            oh[e] = h; ol[e] = l;
        }
        _Float16* dst = xs + ((size_t)b * T + t0 + fr) * (2 * Kp) + split_pos(c0 + c);
        *reinterpret_cast<h8*>(dst) = oh;
        *reinterpret_cast<h8*>(dst + 32) = ol;
    }
    if (overflow && range_flag) atomicOr(range_flag, 1);
}

hipError_t launch_pose_rows_split(const float* x, _Float16* xs, int nb, int C, int T, int Kp, int* range_flag,
                                  hipStream_t stream, const unsigned* gs_bits) {
    if (Kp % 32 != 0 || Kp < C) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pose_rows_split_kernel, dim3((T + 31) / 32, nb, (Kp + POSE_FC - 1) / POSE_FC), dim3(256), 0,
                       stream, x, xs, C, T, Kp, range_flag, gs_bits);
    return hipGetLastError();
}

hipError_t launch_token0(float* tok, const float* time_table, const float* text_term,
                         const float* pe, const int64_t* t_dev, int64_t t_scalar, int n_seq,
                         int n_per_pass, int S, int d, int n_time_rows, hipStream_t stream,
                         const int64_t* tmap_dev, const int* cursor, _Float16* tok_split, int* range_flag) {
    // n_seq = B (single pass) or 2B (CFG: conditional rows then unconditional rows); t_dev, when
    // given, holds n_per_pass = B entries shared by both passes.
    hipLaunchKernelGGL(token0_kernel, dim3(n_seq), dim3(256), 0, stream, tok, time_table, text_term,
                       pe, t_dev, t_scalar, n_per_pass, S, d, n_time_rows, tmap_dev, cursor, tok_split, range_flag);
    return hipGetLastError();
}

__global__ void fill_rows_kernel(float* __restrict__ dst, const float* __restrict__ row, int rows,
                                 int d) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int64_t)rows * d) dst[i] = row[i % d];
}
hipError_t launch_fill_rows(float* dst, const float* row, int rows, int d, hipStream_t stream) {
    const int64_t n = (int64_t)rows * d;
    hipLaunchKernelGGL(fill_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, dst, row, rows, d);
    return hipGetLastError();
}

__global__ void add2_kernel(float* __restrict__ dst, const float* __restrict__ a,
                            const float* __restrict__ b, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = a[i] + b[i];
}
// *out = max(*out, bits of max|x|) (non-negative floats order like their bit patterns); *out is
// zeroed by the caller.  NaN / inf propagate as large bit patterns (grad_scale_from_bits -> 1).
__global__ __launch_bounds__(256) void absmax_bits_kernel(const float* __restrict__ x, int64_t n,
                                                          unsigned* __restrict__ out) {
    __shared__ unsigned part[4];
    unsigned m = 0u;
    const int64_t n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = x4[i];
        m = max(max(m, __float_as_uint(fabsf(v.x))), __float_as_uint(fabsf(v.y)));
        m = max(max(m, __float_as_uint(fabsf(v.z))), __float_as_uint(fabsf(v.w)));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3))
        m = max(m, __float_as_uint(fabsf(x[(n4 << 2) + threadIdx.x])));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(part[0], part[1]), max(part[2], part[3]));
        if (m) atomicMax(out, m);   // one atomic per block
    }
}

hipError_t launch_absmax_bits(const float* x, int64_t n, unsigned* out, hipStream_t stream) {
    int64_t blocks = ((n >> 2) + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks);
    hipLaunchKernelGGL(absmax_bits_kernel, dim3(blocks), dim3(256), 0, stream, x, n, out);
    return hipGetLastError();
}

hipError_t launch_add2(float* dst, const float* a, const float* b, int64_t n, hipStream_t stream) {
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(add2_kernel, dim3(blocks), dim3(256), 0, stream, dst, a, b, n);
    return hipGetLastError();
}

__global__ void pad_copy_kernel(float* __restrict__ dst, const float* __restrict__ src, int rows,
                                int cols, int ldd) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * ldd) return;
    const int r = (int)(i / ldd), c = (int)(i % ldd);
    dst[i] = c < cols ? src[(size_t)r * cols + c] : 0.f;
}
hipError_t launch_pad_copy(float* dst, const float* src, int rows, int cols, int ldd,
                           hipStream_t stream) {
    const int64_t n = (int64_t)rows * ldd;
    hipLaunchKernelGGL(pad_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dst,
                       src, rows, cols, ldd);
    return hipGetLastError();
}

__global__ void transpose_pad_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                     int rows, int cols, int ldd) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)cols * ldd) return;
    const int c = (int)(i / ldd), r = (int)(i % ldd);
    dst[i] = r < rows ? src[(size_t)r * cols + c] : 0.f;
}
hipError_t launch_transpose_pad(float* dst, const float* src, int rows, int cols, int ldd,
                                hipStream_t stream) {
    const int64_t n = (int64_t)cols * ldd;
    hipLaunchKernelGGL(transpose_pad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       dst, src, rows, cols, ldd);
    return hipGetLastError();
}

}  // namespace cmdi
