"""How much of a GEMM's time is wave quantization?  Times the in_proj / linear1 shapes at row counts that
fill an integer number of rounds (512 block slots) and at the real M."""
import os as _os; _os.environ.setdefault("CMDI_PROBES_LIB", "1")   # instrumented library (build.py --probes)
import importlib, sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
eng = importlib.import_module("diffusion-motion-inbetweening_amd.engine")
from tools.gemm_bench import timeit
dev = torch.device("cuda:0")
for (n, k, epi, name) in [(1536, 512, 0, "in_proj"), (1024, 512, 1, "linear1"), (512, 512, 3, "out_proj"), (512, 1024, 3, "linear2")]:
    tn = n // 128
    for m in [12608, (1024 // tn) * 128, (512 // tn) * 128, 1728, 12608 - (1024 // tn) * 128 if (1024 // tn) * 128 < 12608 else 12608 - (512 // tn) * 128]:
        if m <= 0: continue
        a = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev); b = torch.randn(n, device=dev)
        r = torch.randn(m, n, device=dev)
        a_s, w_s = eng.split_f16(a), eng.split_f16(w)
        row = []
        for tile in (8, 4, 6):
            o = torch.empty(m, 2 * n, device=dev, dtype=torch.float16) if epi in (0, 1) else torch.empty(m, n, device=dev)
            dt = timeit(lambda: eng.gemm_h3(a_s, w_s, b, tile=tile, epi=epi, resid=r, split_out=(epi == 0), out=o), iters=20)
            row.append(f"t{tile}: {dt*1e6:6.1f} us ({2.0*m*n*k/dt/1e12:5.1f} TF)")
        print(f"{name:9s} M={m:6d} tiles128={((m+127)//128)*tn:5d}  " + "  ".join(row), flush=True)
