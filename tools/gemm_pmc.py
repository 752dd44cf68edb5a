"""Launch a few GEMM variants once each (after warm-up) for rocprofv3 --pmc."""
import os as _os; _os.environ.setdefault("CMDI_PROBES_LIB", "1")   # instrumented library (build.py --probes)
import importlib, sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
eng = importlib.import_module("diffusion-motion-inbetweening_amd.engine")
dev = torch.device("cuda:0")
M = 2 * 32 * 197
for (m, n, k) in [(M, 1536, 512), (M, 512, 512), (4096, 4096, 4096)]:
    a = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev); b = torch.randn(n, device=dev)
    for tile in (11, 14):
        for _ in range(3):
            eng.gemm_nt(a, w, b, tile=tile)
torch.cuda.synchronize()
