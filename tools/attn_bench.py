"""Attention core microbenchmark (B=32 CFG shape): fp32-MFMA kernel vs the split-f16 kernel."""
import os as _os; _os.environ.setdefault("CMDI_PROBES_LIB", "1")   # instrumented library (build.py --probes)
import importlib, sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
eng = importlib.import_module("diffusion-motion-inbetweening_amd.engine")
N = importlib.import_module("diffusion-motion-inbetweening_amd._native")
from tools.gemm_bench import timeit
dev = torch.device("cuda:0")
lib = N.load()
for n_seq in (64, 512):
    S, H = 197, 4
    qkv = torch.randn(n_seq * S, 3 * H * 128, device=dev)
    qs = eng.split_f16(qkv)
    out = torch.empty(n_seq * S, H * 128, device=dev)
    flops = 4.0 * S * S * 128 * H * n_seq
    dt = timeit(lambda: eng.attention_fwd(qkv, n_seq, S, H), iters=20)
    st = N.current_stream(dev)
    dh = timeit(lambda: N.check(lib.cmdi_attention_fwd_h3(N.ptr(qs), N.ptr(out), n_seq, S, H, st)), iters=20)
    print(f"n_seq={n_seq}: f32 {dt*1e6:7.1f} us ({flops/dt/1e12:5.1f} TF)   h3 {dh*1e6:7.1f} us ({flops/dh/1e12:5.1f} TF fp32-equiv)", flush=True)
import os
if int(os.environ.get("CMDI_ATTN_DBG", "0")) & 16:
    import numpy as np
    n_seq = 64
    qkv = torch.randn(n_seq * 197, 1536, device=dev); qs = eng.split_f16(qkv)
    out = torch.zeros(n_seq * 197 * 512 + n_seq * 4 * 8, device=dev)
    N.check(lib.cmdi_attention_fwd_h3(N.ptr(qs), N.ptr(out), n_seq, 197, 4, N.current_stream(dev)))
    torch.cuda.synchronize()
    st = out[n_seq * 197 * 512:].view(torch.int64).cpu().numpy().reshape(-1, 4)
    print("per-block cycles: prologue %.0f  loop %.0f  epilogue %.0f" % tuple(st[:, :3].mean(0)))
