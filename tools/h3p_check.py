"""Persistent GEMM (tile 50) vs tile 8: bitwise equality over epilogues / ragged shapes, then timings on the denoiser's shapes."""
import importlib, sys, torch
sys.path.insert(0, "/root/repo")
eng = importlib.import_module("diffusion-motion-inbetweening_amd.engine")
from tools.x6_bench import timeit
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
ok = True
for (m, n, k) in [(333, 512, 512), (128, 256, 64), (129, 256, 96), (197 * 4, 1536, 512), (1000, 512, 1024), (12608, 1536, 64), (12608, 1536, 512),
                  (6304, 1024, 512), (12608, 512, 1024), (40000, 256, 512), (33000, 768, 128)]:
    a = torch.randn(m, k, generator=g).to(dev); w = (torch.randn(n, k, generator=g) * 0.05).to(dev)
    b = torch.randn(n, generator=g).to(dev); r = torch.randn(m, n, generator=g).to(dev)
    a_s, w_s, r_s = eng.split_f16(a), eng.split_f16(w), eng.split_f16(r)
    for (epi, kw, name) in [(0, dict(split_out=True), "plain_split"), (0, {}, "plain"), (1, {}, "gelu"), (3, dict(resid=r), "resid"), (4, dict(resid=r_s), "resid_split")]:
        ref = eng.gemm_h3(a_s, w_s, b, tile=8, epi=epi, **kw)
        out = eng.gemm_h3(a_s, w_s, b, tile=50, epi=epi, **kw)
        torch.cuda.synchronize()
        same = torch.equal(out, ref)
        ok &= same
        if not same:
            d = (out.float() - ref.float()).abs()
            bad = (out != ref).nonzero()
            print(f"DIFF m={m} n={n} k={k} {name}: {bad.shape[0]} elements, max abs {d.max().item():.3e}, first {bad[:3].tolist()}, nan {torch.isnan(out.float()).sum().item()}", flush=True)
print("bitwise:", "ALL EQUAL" if ok else "MISMATCH", flush=True)
M = 2 * 32 * 197
for (m, n, k, epi, name) in [(M, 1536, 512, 0, "in_proj"), (M, 1024, 512, 1, "linear1"), (M, 512, 512, 4, "out_proj"), (M, 512, 1024, 4, "linear2"),
                            (M // 2, 1536, 512, 0, "in_proj/2"), (M // 2, 512, 1024, 4, "linear2/2"), (2 * 256 * 197, 1536, 512, 0, "in_proj B=256")]:
    a = torch.randn(m, k, generator=g).to(dev); w = (torch.randn(n, k, generator=g) * 0.05).to(dev); b = torch.randn(n, generator=g).to(dev); r = torch.randn(m, n, generator=g).to(dev)
    a_s, w_s, r_s = eng.split_f16(a), eng.split_f16(w), eng.split_f16(r)
    cs = torch.empty(m, 2 * n, device=dev, dtype=torch.float16); c = torch.empty(m, n, device=dev)
    row = [f"{name:14s}"]
    for rep in range(2):
        for tile in (8, 50):
            t = timeit(lambda: eng.gemm_h3(a_s, w_s, b, tile=tile, epi=epi, resid=r_s, split_out=(epi == 0), out=(cs if epi in (0, 1) else c)), iters=30)
            row.append(f"t{tile}: {t*1e6:6.1f}us {3 * 2.0 * m * n * k / t / 1e12:6.0f}TF")
    print("  ".join(row), flush=True)
