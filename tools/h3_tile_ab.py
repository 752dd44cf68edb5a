import importlib, sys, torch
sys.path.insert(0, "/root/repo")
eng = importlib.import_module("diffusion-motion-inbetweening_amd.engine")
from tools.x6_bench import timeit, rel
dev = torch.device("cuda:0")
M = 2 * 32 * 197
g = torch.Generator().manual_seed(1)
TILES = tuple(int(t) for t in sys.argv[1].split(",")) if len(sys.argv) > 1 else (8, 1, 88)
for (m, n, k, epi, name) in [(M, 1536, 512, 0, "in_proj"), (M, 1024, 512, 1, "linear1"), (M, 512, 512, 3, "out_proj"), (M, 512, 1024, 3, "linear2"),
                            (M // 2, 1536, 512, 0, "in_proj/2"), (M // 2, 512, 1024, 3, "linear2/2")]:
    a = torch.randn(m, k, generator=g).to(dev); w = (torch.randn(n, k, generator=g) * 0.05).to(dev); b = torch.randn(n, generator=g).to(dev); r = torch.randn(m, n, generator=g).to(dev)
    a_s, w_s = eng.split_f16(a), eng.split_f16(w)
    cs = torch.empty(m, 2 * n, device=dev, dtype=torch.float16); c = torch.empty(m, n, device=dev)
    ref = eng.gemm_h3(a_s, w_s, b, tile=8, epi=epi, resid=r, split_out=(epi == 0))
    row = [name]
    for tile in TILES:
        out = eng.gemm_h3(a_s, w_s, b, tile=tile, epi=epi, resid=r, split_out=(epi == 0))
        same = torch.equal(out, ref)
        t = timeit(lambda: eng.gemm_h3(a_s, w_s, b, tile=tile, epi=epi, resid=r, split_out=(epi == 0), out=(cs if epi in (0, 1) else c)), iters=30)
        row.append(f"t{tile}: {t*1e6:6.1f}us {'=' if same else 'DIFF'}")
    print("  ".join(row), flush=True)
