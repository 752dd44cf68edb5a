import os as _os; _os.environ.setdefault("CMDI_PROBES_LIB", "1")   # instrumented library (build.py --probes)
import importlib, os, sys
from pathlib import Path
import numpy as np
import torch
os.environ["CMDI_H3_DBG"] = "16"
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
eng = importlib.import_module("diffusion-motion-inbetweening_amd.engine")
dev = torch.device("cuda:0")
m, n, k = 12608, 1536, 512
a = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev); b = torch.randn(n, device=dev)
a_s, w_s = eng.split_f16(a), eng.split_f16(w)
nblk = 99 * 12
buf = torch.zeros(nblk * 6 + 1024, dtype=torch.int64, device=dev)
out = torch.empty(m, 2 * n, device=dev, dtype=torch.float16)
for _ in range(2):
    eng.gemm_h3(a_s, w_s, b, tile=8, epi=0, resid=buf, split_out=True, out=out)
torch.cuda.synchronize()
raw = buf[:nblk * 6].cpu().numpy().reshape(nblk, 6)
hw = raw[:, 1]
xcc = (hw >> 32) & 0xF
hwid = hw & 0xFFFFFFFF
cu = (hwid >> 8) & 0xF; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 0x7
rstart = raw[:, 4] - raw[:, 4].min()
for bidx in list(range(0, 40)) + [256, 257, 264, 512, 513, 520]:
    print(bidx, "xcc", int(xcc[bidx]), "se", int(se[bidx]), "sh", int(sh[bidx]), "cu", int(cu[bidx]), "start(10ns)", int(rstart[bidx]), "hwid", hex(int(hwid[bidx])))
key = xcc * 10000 + se * 100 + sh * 50 + cu
first = {}
pairs = []
for bidx in range(nblk):
    kk = int(key[bidx])
    first.setdefault(kk, []).append(bidx)
print("distinct CUs", len(first))
print("example CU block lists:", list(first.values())[:6])
