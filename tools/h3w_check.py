"""Weight-stationary GEMM (gemm_h3w.hpp, tile id 60) against the tiled kernel (tile 8): bitwise equality on every hook epilogue
and a range of row counts, then time per launch of both at the denoiser's shapes.   python tools/h3w_check.py [iters]"""
import importlib
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
eng = importlib.import_module("diffusion-motion-inbetweening_amd.engine")
from tools.x6_bench import timeit  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = 0
for m in (1, 31, 32, 33, 333, 788, 3940, 6304, 12608, 100864):
    for n in (128, 512, 1024, 1536):
        if m > 20000 and n != 1536:
            continue
        a = torch.randn(m, 512, generator=g).to(dev)
        w = (torch.randn(n, 512, generator=g) * 0.05).to(dev)
        b = torch.randn(n, generator=g).to(dev)
        r = torch.randn(m, n, generator=g).to(dev)
        a_s, w_s, r_s = eng.split_f16(a), eng.split_f16(w), eng.split_f16(r)
        row = [f"M={m:6d} N={n:4d}"]
        for (epi, split, resid, name) in ((0, False, None, "plain"), (0, True, None, "plain_split"), (1, True, None, "gelu_split"),
                                          (3, False, r, "resid"), (4, False, r_s, "resid_split_rows")):
            ref = eng.gemm_h3(a_s, w_s, b, tile=8, epi=epi, resid=resid, split_out=split)
            out = eng.gemm_h3(a_s, w_s, b, tile=60, epi=epi, resid=resid, split_out=split)
            same = torch.equal(out, ref)
            bad += not same
            row.append(f"{name}:{'=' if same else 'DIFF %.3g' % float((out.float() - ref.float()).abs().max())}")
        print("  ".join(row), flush=True)
print("bitwise:", "ALL EQUAL" if not bad else f"{bad} DIFFER")
M = 2 * 32 * 197
for (m, n, epi, name) in [(M, 1536, 0, "in_proj"), (M, 1024, 1, "linear1"), (M, 512, 3, "out_proj"), (M // 2, 1536, 0, "in_proj/2"),
                          (M // 2, 1024, 1, "linear1/2"), (M // 2, 512, 3, "out_proj/2"), (8 * M, 1536, 0, "in_proj B=256")]:
    a = torch.randn(m, 512, generator=g).to(dev); w = (torch.randn(n, 512, generator=g) * 0.05).to(dev)
    b = torch.randn(n, generator=g).to(dev); r = torch.randn(m, n, generator=g).to(dev)
    a_s, w_s = eng.split_f16(a), eng.split_f16(w)
    cs = torch.empty(m, 2 * n, device=dev, dtype=torch.float16); c = torch.empty(m, n, device=dev)
    row = [f"{name:14s}"]
    for tile in (8, 60, 0):
        for rep in range(2):
            t = timeit(lambda: eng.gemm_h3(a_s, w_s, b, tile=tile, epi=epi, resid=r, split_out=(epi == 0), out=(cs if epi in (0, 1) else c)), iters=iters)
            row.append(f"t{tile}: {t * 1e6:6.1f}us")
    print("  ".join(row), flush=True)
sys.exit(1 if bad else 0)
