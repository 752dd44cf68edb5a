"""GEMM microbenchmark on the denoiser's shapes: TFLOP/s per tile / pipeline variant and epilogue."""
import os as _os; _os.environ.setdefault("CMDI_PROBES_LIB", "1")   # instrumented library (build.py --probes)
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


def main():
    dev = torch.device("cuda:0")
    tiles = [int(t) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 3, 4, 5, 11, 12, 13, 14, 15]
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    M = 2 * 32 * 197
    shapes = [(M, 1536, 512, 0, "in_proj"), (M, 512, 512, 3, "out_proj+res"),
              (M, 1024, 512, 1, "linear1+gelu"), (M, 512, 1024, 3, "linear2+res"),
              (2 * 256 * 197, 1536, 512, 0, "in_proj B=256"), (4096, 4096, 4096, 0, "4096^3")]
    h3_tiles = [int(t) for t in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 3, 4, 5, 6, 7]
    print("tiles: 1=128x128 2=64x128 3=128x64 4=64x64 5=256x128(8w); +10 = pipelined loop; "
          "hN = split-f16 family (fp32-equivalent TFLOP/s = 2MNK/t; executed f16 flops are 3x)")
    for (m, n, k, epi, name) in shapes:
        a = torch.randn(m, k, device=dev)
        w = torch.randn(n, k, device=dev)
        b = torch.randn(n, device=dev)
        r = torch.randn(m, n, device=dev)
        c = torch.empty(m, n, device=dev)
        row = []
        for tile in tiles:
            dt = timeit(lambda: eng.gemm_nt(a, w, b, tile=tile, epi=epi, resid=r, out=c), iters=iters)
            row.append(f"t{tile}:{2.0 * m * n * k / dt / 1e12:6.1f}")
        a_s, w_s = eng.split_f16(a), eng.split_f16(w)
        cs = torch.empty(m, 2 * n, device=dev, dtype=torch.float16)
        for tile in h3_tiles:
            o = cs if epi == 1 else c
            dt = timeit(lambda: eng.gemm_h3(a_s, w_s, b, tile=tile, epi=epi, resid=r, out=o), iters=iters)
            row.append(f"h{tile}:{2.0 * m * n * k / dt / 1e12:6.1f}")
        print(f"{name:14s} M={m:6d} N={n:5d} K={k:5d}  " + " ".join(row), flush=True)


if __name__ == "__main__":
    main()
