"""Small-M GEMMs (latency-bound regime): tile 21 (64x128) vs 8 (128x128); rings of 3 and 4 stages were measured with it in round 3: no gain."""
import importlib, sys, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
eng = importlib.import_module("diffusion-motion-inbetweening_amd.engine")
from tools.x6_bench import timeit
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
TILES = tuple(int(t) for t in sys.argv[1].split(",")) if len(sys.argv) > 1 else (21, 8)
MS = tuple(int(m) for m in sys.argv[2].split(",")) if len(sys.argv) > 2 else (244, 788, 3152)
for M in MS:
    for (n, k, epi, name) in [(1536, 512, 0, "in_proj"), (1024, 512, 1, "linear1"), (512, 512, 4, "out_proj"), (512, 1024, 4, "linear2")]:
        a = torch.randn(M, k, generator=g).to(dev); w = (torch.randn(n, k, generator=g) * 0.05).to(dev); b = torch.randn(n, generator=g).to(dev); r = torch.randn(M, n, generator=g).to(dev)
        a_s, w_s, r_s = eng.split_f16(a), eng.split_f16(w), eng.split_f16(r)
        cs = torch.empty(M, 2 * n, device=dev, dtype=torch.float16); c = torch.empty(M, n, device=dev)
        ref = eng.gemm_h3(a_s, w_s, b, tile=21, epi=epi, resid=r_s, split_out=(epi == 0))
        row = [f"M={M:5d} {name:9s}"]
        for tile in TILES:
            out = eng.gemm_h3(a_s, w_s, b, tile=tile, epi=epi, resid=r_s, split_out=(epi == 0))
            same = torch.equal(out, ref)
            t = timeit(lambda: eng.gemm_h3(a_s, w_s, b, tile=tile, epi=epi, resid=r_s, split_out=(epi == 0), out=(cs if epi in (0, 1) else c)), iters=50)
            row.append(f"t{tile}: {t*1e6:6.1f}us {'=' if same else 'DIFF'}")
        print("  ".join(row), flush=True)
