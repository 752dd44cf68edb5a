"""Launch the U-Net's level-0 k=5 convolution GEMM (64 framed sequences x 256 rows, 1024 -> 1024 channels) a few times
for rocprofv3 --pmc passes:  rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- python tools/conv_pmc.py"""
import os as _os; _os.environ.setdefault("CMDI_PROBES_LIB", "1")   # instrumented library (build.py --probes)
import importlib, os, sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
eng = importlib.import_module("diffusion-motion-inbetweening_amd.engine")
N = importlib.import_module("diffusion-motion-inbetweening_amd._native")
lib = N.load()
dev = torch.device("cuda:0")
nseq, tp, h, tv, C, taps = 64, 256, 16, 224, 1024, 5
guard = 16
rows = torch.zeros(guard + nseq * tp + guard, C, device=dev)
rows[guard:guard + nseq * tp].view(nseq, tp, C)[:, h:h + tv] = torch.randn(nseq, tv, C, device=dev)
a_s = eng.split_f16(rows)
w_s = eng.split_f16(torch.randn(C, taps * C, device=dev) * 0.02)
bias = torch.randn(C, device=dev)
out = torch.zeros(nseq * tp, C, device=dev)
a_ptr = a_s.data_ptr() + guard * (2 * C) * 2
for _ in range(int(os.environ.get("PMC_REPS", "3"))):
    N.check(lib.cmdi_conv_rows_h3(a_ptr, 2 * C, N.ptr(w_s), N.ptr(bias), 0, N.ptr(out), 0, nseq * tp, C, C, taps, 2, 1, 0, 0,
                                  tp, h, h + tv, 0, N.current_stream(dev)))
torch.cuda.synchronize()
