"""bf16x6 GEMM (gemm_x6.hpp) on the denoiser's shapes: accuracy vs float64 beside the fp32-MFMA and f16x3 kernels, and
time per launch of each schedule variant.   python tools/x6_bench.py [iters]"""
import importlib
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
eng = importlib.import_module("diffusion-motion-inbetweening_amd.engine")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


def main():
    dev = torch.device("cuda:0")
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    M = 2 * 32 * 197
    shapes = [(M, 1536, 512, 0, "in_proj"), (M, 512, 512, 3, "out_proj+res"), (M, 1024, 512, 1, "linear1+gelu"),
              (M, 512, 1024, 3, "linear2+res"), (M // 2, 1536, 512, 0, "in_proj half"),
              (2 * 256 * 197, 1536, 512, 0, "in_proj B=256"), (333, 512, 512, 3, "ragged small")]
    g = torch.Generator(device="cpu").manual_seed(1)
    for (m, n, k, epi, name) in shapes:
        a = torch.randn(m, k, generator=g).to(dev)
        w = (torch.randn(n, k, generator=g) * (torch.arange(n).float()[:, None] % 7 + 1) * 0.05).to(dev)
        b = torch.randn(n, generator=g).to(dev)
        r = torch.randn(m, n, generator=g).to(dev)
        ref = a.double() @ w.double().T + b.double()
        if epi == 3:
            ref = ref + r.double()
        if epi == 1:
            ref = torch.nn.functional.gelu(ref)
        wx = eng.pack_x6(w)
        c = torch.empty(m, n, device=dev)
        row = [f"{name:14s} M={m:6d} N={n:5d} K={k:5d}"]
        out32 = eng.gemm_nt(a, w, b, epi=epi, resid=r)
        row.append(f"err f32 {rel(out32, ref):.2e}")
        for var in (0, 1, 2):
            out = eng.gemm_x6(a, wx, b, epi=epi, resid=r, variant=var)
            row.append(f"x6v{var} {rel(out, ref):.2e}")
        a_s, w_s = eng.split_f16(a), eng.split_f16(w)
        o3 = eng.gemm_h3(a_s, w_s, b, epi=epi, resid=r)
        o3 = eng.unsplit_f16(o3) if epi == 1 else o3
        row.append(f"h3 {rel(o3, ref):.2e}")
        if epi == 1:   # where does the f16x3 GELU + split output lose accuracy?
            pre = eng.gemm_h3(a_s, w_s, b, epi=0)
            pre_ref = a.double() @ w.double().T + b.double()
            gpre = torch.nn.functional.gelu(pre)
            d = (o3.double() - ref).abs()
            i = int(d.argmax())
            row.append(f"[h3 pre {rel(pre, pre_ref):.2e} torch-gelu(h3 pre) {rel(gpre, ref):.2e} unsplit-vs-that {rel(o3, gpre.double()):.2e} "
                       f"worst |d|={float(d.flatten()[i]):.3e} at ref={float(ref.flatten()[i]):.4e} pre={float(pre_ref.flatten()[i]):.4e} got={float(o3.flatten()[i]):.6e}]")
        t32 = timeit(lambda: eng.gemm_nt(a, w, b, epi=epi, resid=r, out=c), iters=iters)
        row.append(f"| us f32 {t32 * 1e6:7.1f}")
        for var in (0, 1, 2):
            t = timeit(lambda: eng.gemm_x6(a, wx, b, epi=epi, resid=r, variant=var, out=c), iters=iters)
            row.append(f"x6v{var} {t * 1e6:7.1f} ({2.0 * m * n * k / t / 1e12:5.1f} TF)")
        cs = torch.empty(m, 2 * n, device=dev, dtype=torch.float16)
        t3 = timeit(lambda: eng.gemm_h3(a_s, w_s, b, epi=epi, resid=r, out=cs if epi == 1 else c), iters=iters)
        row.append(f"h3 {t3 * 1e6:7.1f}")
        print("  ".join(row), flush=True)


if __name__ == "__main__":
    main()
