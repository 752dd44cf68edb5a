"""Ablations of the persistent GEMM (probes build): CMDI_H3_DBG tile 50 + bits, bits 1 = no in-loop DMA, 2 = no fragment reads, 4 = no epilogue (compile-time variants)."""
import os as _os; _os.environ.setdefault("CMDI_PROBES_LIB", "1")
import importlib, os, sys, subprocess
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
if len(sys.argv) > 1:
    abl = int(sys.argv[1])
    import torch
    sys.path.insert(0, str(REPO))
    eng = importlib.import_module("diffusion-motion-inbetweening_amd.engine")
    from tools.x6_bench import timeit
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    row = [f"dbg={sys.argv[1]:>2s}"]
    for (m, n, k) in [(100864, 1536, 512), (12608, 1536, 512)]:
        a = torch.randn(m, k, generator=g).to(dev); w = (torch.randn(n, k, generator=g) * 0.05).to(dev); b = torch.randn(n, generator=g).to(dev)
        a_s, w_s = eng.split_f16(a), eng.split_f16(w)
        cs = torch.empty(m, 2 * n, device=dev, dtype=torch.float16)
        for tile in (8, 1000 + abl if abl else 50):
            t = timeit(lambda: eng.gemm_h3(a_s, w_s, b, tile=tile, epi=0, split_out=True, out=cs), iters=20)
            row.append(f"M={m} t{tile}: {t*1e6:7.1f}us {3 * 2.0 * m * n * k / t / 1e12:6.0f}TF")
    print("  ".join(row), flush=True)
else:
    for bits in ("0", "4", "132", "260", "388", "512", "5", "6", "7"):
        subprocess.run([sys.executable, __file__, bits])
