"""Launch the split-f16 GEMMs of one encoder layer (and the attention core) a few times each, for
rocprofv3 --pmc / --kernel-trace runs:  rocprofv3 --pmc <counters> --output-format csv -d out -- python tools/h3_pmc.py"""
import os as _os; _os.environ.setdefault("CMDI_PROBES_LIB", "1")   # instrumented library (build.py --probes)
import importlib, os, sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
eng = importlib.import_module("diffusion-motion-inbetweening_amd.engine")
dev = torch.device("cuda:0")
M = 2 * 32 * 197
tile = int(os.environ.get("PMC_TILE", "0"))
reps = int(os.environ.get("PMC_REPS", "3"))
for (m, n, k, epi) in [(M, 1536, 512, 0), (M, 512, 512, 3), (M, 1024, 512, 1), (M, 512, 1024, 3)]:
    a = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) * 0.05; b = torch.randn(n, device=dev)
    r = torch.randn(m, n, device=dev)
    a_s, w_s = eng.split_f16(a), eng.split_f16(w)
    for _ in range(reps):
        eng.gemm_h3(a_s, w_s, b, tile=tile, epi=epi, resid=r, split_out=(epi == 0))
qkv = torch.randn(M, 1536, device=dev)
for _ in range(reps):
    eng.attention_fwd_h3(qkv, 64, 197, 4)
torch.cuda.synchronize()
