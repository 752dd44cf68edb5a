// Instantiations + host launcher of the persistent split-f16 GEMM (gemm_h3p.hpp).
#include "gemm_h3p.hpp"
#include "kernels.hpp"

namespace cmdi {

static bool is_conv(const H3Params& p) { return p.cpt || p.a_ld || p.a_row_mul || p.c_row_mul || p.tp; }

bool gemm_h3p_supports(int epi, const H3Params& p) {
    if (epi != H3_PLAIN && epi != H3_PLAIN_SPLIT && epi != H3_GELU_SPLIT && epi != H3_GELUGRAD_SPLIT && epi != H3_RESID) return false;
    if (p.ksplit > 1 || p.cs_head_major || p.m_fast) return false;
    if (p.M <= 0 || p.N % H3PTile::BN != 0 || p.K % 32 != 0 || p.K < 64) return false;
    if (is_conv(p)) {   // convolution rows (U-Net): the plain fp32 epilogue only; K = taps * cpt * 32
        if (epi != H3_PLAIN || p.ln_part || p.ln_c1 || p.aux) return false;
        if (p.cpt < 1 || p.taps < 1 || p.K != p.taps * p.cpt * 32 || (p.a_ld && p.a_ld < 2 * p.cpt * 32)) return false;
        if (p.rc_tv && (p.rc_tv < 1 || p.tp < 1 || p.a_row_mul > 1 || p.c_row_mul || p.t_lo < 0 || p.t_lo + p.rc_tv > p.tp ||
                        p.M % p.rc_tv != 0))
            return false;
        const size_t a_row = 2 * (p.a_ld ? (size_t)p.a_ld : 2 * (size_t)p.K) * (size_t)(p.a_row_mul ? p.a_row_mul : 1);
        const size_t phys_rows = p.rc_tv ? (size_t)(p.M / p.rc_tv) * p.tp : (size_t)p.M;
        if (phys_rows * a_row + (size_t)p.taps * a_row >= (1ull << 31)) return false;
    } else if ((size_t)p.M * 4 * (size_t)p.K >= (1ull << 31)) {
        return false;
    }
    if ((size_t)p.N * 4 * (size_t)p.K >= (1ull << 31)) return false;   // 32-bit request offsets
    if (epi == H3_RESID && p.ln_c1) return false;        // (a folded-LayerNorm A operand together with a residual: not used, not built)
    return true;
}

template <int EPI, int ABL = 0, bool CONV = false>
static hipError_t launch_h3p(const H3Params& p, hipStream_t stream) {
    auto kern = gemm_h3p_kernel<EPI, ABL, CONV>;
    static PerDevice<bool> attr_done_dev;   // benign race: the attribute call is idempotent
    bool& attr_done = attr_done_dev.get();
    static PerDevice<int> blocks_dev;
    int& blocks = blocks_dev.get();
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)H3PTile::LDS_BYTES);
        if (e != hipSuccess) return e;
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        blocks = cus / 8 * 8;        // one block per CU; block id % 8 = XCD
        if (blocks < 8) blocks = 8;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(H3PTile::NT), H3PTile::LDS_BYTES, stream, p);
    return hipGetLastError();
}

hipError_t launch_gemm_h3p(int epi, const H3Params& p, hipStream_t s, int ablation) {
    if (!gemm_h3p_supports(epi, p)) return hipErrorInvalidValue;
#ifdef CMDI_PROBES
    if (ablation && epi == H3_PLAIN_SPLIT) {
        switch (ablation) {
#define H3P_CASE(A) case A: return launch_h3p<H3_PLAIN_SPLIT, A>(p, s);
            H3P_CASE(1) H3P_CASE(2) H3P_CASE(3) H3P_CASE(4) H3P_CASE(5) H3P_CASE(6) H3P_CASE(7) H3P_CASE(8)
            H3P_CASE(32) H3P_CASE(36) H3P_CASE(37) H3P_CASE(38) H3P_CASE(39)          // two-interval K step (+ ablations)
            H3P_CASE(64) H3P_CASE(68) H3P_CASE(71)                                   // 4 tail MFMAs
            H3P_CASE(96) H3P_CASE(100) H3P_CASE(103)                                 // both
            H3P_CASE(132) H3P_CASE(260) H3P_CASE(388) H3P_CASE(512)                  // no waits / half the requests / both / no stores
#undef H3P_CASE
        }
        return hipErrorInvalidValue;
    }
#endif
    if (ablation) return hipErrorInvalidValue;
    switch (epi) {
        case H3_PLAIN: return is_conv(p) ? launch_h3p<H3_PLAIN, 0, true>(p, s) : launch_h3p<H3_PLAIN>(p, s);
        case H3_PLAIN_SPLIT: return launch_h3p<H3_PLAIN_SPLIT>(p, s);
        case H3_GELU_SPLIT: return launch_h3p<H3_GELU_SPLIT>(p, s);
        case H3_GELUGRAD_SPLIT: return launch_h3p<H3_GELUGRAD_SPLIT>(p, s);
        case H3_RESID: return launch_h3p<H3_RESID>(p, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace cmdi
