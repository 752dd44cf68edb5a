// The step right after the sampling loop in every caller of the reference, on the
// device: un-normalise the HumanML3D vectors and recover XYZ joint positions, so the 263-dim motion never
// has to visit the host (and a multi-GPU gather moves 66 instead of 263 floats per frame).
//   caller        sample/conditional_synthesis.py:229-235
//   inv_transform data_loaders/humanml/data/dataset.py:378-382            data * std + mean
//   root          data_loaders/humanml/scripts/motion_process.py:402-441  recover_root_rot_pos
//   joints        data_loaders/humanml/scripts/motion_process.py:474-491  recover_from_ric
//   qinv / qrot   data_loaders/humanml/common/quaternion.py:16-20,54-73
// Built with -ffp-contract=off: every multiply / add is rounded separately, in the reference's order
// (torch.cumsum is a sequential fp32 sum); only cos / sin may differ from torch's by an ulp.
#include "common.hpp"
#include "kernels.hpp"

namespace cmdi {

// qrot(qinv(q), v) with q = (c, 0, s, 0): qv = (0, -s, 0); uv = qv x v; uuv = qv x uv; v + 2 (c uv + uuv)
__device__ __forceinline__ void rot_y_inv(float c, float s, float vx, float vy, float vz, float& ox,
                                          float& oy, float& oz) {
    const float ux = (-s) * vz, uz = s * vx;            // uv = (ux, 0, uz)   (uy = 0 exactly)
    const float wx = (-s) * uz, wz = s * ux;            // uuv = (wx, 0, wz)
    ox = vx + 2.0f * (c * ux + wx);
    oy = vy;                                            // + 2 * (c * 0 + 0)
    oz = vz + 2.0f * (c * uz + wz);
}

// One block per sample.  x [B, J, 1, T] (T contiguous), out [B, n_joints, 3, T].
__global__ __launch_bounds__(256) void recover_xyz_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ std,
                                                          float* __restrict__ out, int J, int T,
                                                          int n_joints, int abs_3d) {
    extern __shared__ float sm[];   // ang / cos [T], sin [T], rx [T], rz [T]
    float* cs = sm;
    float* sn = sm + T;
    float* rx = sm + 2 * T;
    float* rz = sm + 3 * T;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* xb = x + (size_t)b * J * T;
    auto feat = [&](int c, int t) {
        const float v = xb[(size_t)c * T + t];
        return mean ? v * std[c] + mean[c] : v;
    };
    // root yaw: absolute, or the running sum of the rotation velocity shifted by one frame
    if (abs_3d) {
        for (int t = tid; t < T; t += 256) cs[t] = feat(0, t);
    } else {
        for (int t = tid; t < T; t += 256) sn[t] = t == 0 ? 0.f : feat(0, t - 1);
        __syncthreads();
        if (tid == 0) {
            float a = 0.f;
            for (int t = 0; t < T; ++t) { a += sn[t]; cs[t] = a; }   // cumsum of [0, v0, v1, ...]
        }
    }
    __syncthreads();
    for (int t = tid; t < T; t += 256) {
        const float a = cs[t];
        cs[t] = cosf(a);
        sn[t] = sinf(a);
    }
    __syncthreads();
    // root XZ: absolute, or the running sum of the yaw-corrected velocities (shifted by one frame)
    if (abs_3d) {
        for (int t = tid; t < T; t += 256) { rx[t] = feat(1, t); rz[t] = feat(2, t); }
    } else {
        for (int t = tid; t < T; t += 256) {
            float vx = 0.f, vz = 0.f;
            if (t > 0) { vx = feat(1, t - 1); vz = feat(2, t - 1); }
            float ox, oy, oz;
            rot_y_inv(cs[t], sn[t], vx, 0.f, vz, ox, oy, oz);
            rx[t] = ox; rz[t] = oz;
        }
        __syncthreads();
        if (tid < 2) {   // two independent sequential sums
            float* r = tid == 0 ? rx : rz;
            float a = 0.f;
            for (int t = 0; t < T; ++t) { a += r[t]; r[t] = a; }
        }
    }
    __syncthreads();
    float* ob = out + (size_t)b * n_joints * 3 * T;
    for (int idx = tid; idx < n_joints * T; idx += 256) {
        const int j = idx / T, t = idx - j * T;
        float px, py, pz;
        if (j == 0) {
            px = rx[t]; py = feat(3, t); pz = rz[t];
        } else {
            const int c = 4 + 3 * (j - 1);
            rot_y_inv(cs[t], sn[t], feat(c, t), feat(c + 1, t), feat(c + 2, t), px, py, pz);
            px += rx[t];
            pz += rz[t];
        }
        ob[((size_t)j * 3 + 0) * T + t] = px;
        ob[((size_t)j * 3 + 1) * T + t] = py;
        ob[((size_t)j * 3 + 2) * T + t] = pz;
    }
}

hipError_t launch_recover_xyz(const float* x, const float* mean, const float* std, float* out, int batch,
                              int n_feats, int n_frames, int n_joints, int abs_3d, hipStream_t stream) {
    if (batch < 1 || n_frames < 1 || n_joints < 1 || n_feats < 4 + 3 * (n_joints - 1) || (mean == nullptr) != (std == nullptr))
        return hipErrorInvalidValue;
    const size_t lds = 4ull * n_frames * sizeof(float);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(recover_xyz_kernel, dim3(batch), dim3(256), lds, stream, x, mean, std, out, n_feats,
                       n_frames, n_joints, abs_3d);
    return hipGetLastError();
}

}  // namespace cmdi
