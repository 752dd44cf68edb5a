"""Attention input-VJP microbenchmark (C3 shape: 64 sequences x 197 tokens x 4 heads) + the probes build's cycle stamps.

    CMDI_PROBES_LIB=1 python tools/attn_bwd_bench.py        # on the GPU box
"""
import os as _os; _os.environ.setdefault("CMDI_PROBES_LIB", "1")
import ctypes, importlib, sys
from pathlib import Path
import numpy as np
import torch
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
eng = importlib.import_module("diffusion-motion-inbetweening_amd.engine")
N = importlib.import_module("diffusion-motion-inbetweening_amd._native")
from tools.gemm_bench import timeit
dev = torch.device("cuda:0")
lib = N.load()
names = {0: "dQ", 1: "dK (kv<0>)", 2: "dV (kv<1>)"}
for n_seq in (64,):
    S, H = 197, 4
    d = H * 128
    M = n_seq * S
    qkv = torch.randn(M, 3 * d, device=dev)
    dout = torch.randn(M, d, device=dev)
    qs = eng.split_f16(qkv)
    dqs = torch.zeros(M, 6 * d, dtype=torch.float16, device=dev)
    work = torch.empty(2 * M * d + n_seq * H * (2 * S + 96 * ((S + 31) // 32)), device=dev)
    st = N.current_stream(dev)
    call = lambda: N.check(lib.cmdi_attention_vjp_h3(N.ptr(qs), N.ptr(dout), N.ptr(dqs), N.ptr(work), n_seq, S, H, st))
    t_all = timeit(call, iters=20)
    t_fwd = timeit(lambda: N.check(lib.cmdi_attention_fwd_h3(N.ptr(qs), N.ptr(work), n_seq, S, H, st)), iters=20)
    print(f"n_seq={n_seq}: forward(stash) + split + backward {t_all*1e6:.1f} us; forward alone {t_fwd*1e6:.1f} us", flush=True)
    if hasattr(lib, "cmdi_probe_bwd_stamps"):
        buf = np.zeros((3, 1024, 24), dtype=np.int64)
        lib.cmdi_probe_bwd_stamps.argtypes = [ctypes.c_void_p]
        lib.cmdi_probe_bwd_stamps.restype = ctypes.c_int
        call(); torch.cuda.synchronize()
        N.check(lib.cmdi_probe_bwd_stamps(buf.ctypes.data))
        nb = 2 * n_seq * H
        for k in range(3):
            s = buf[k, :nb]
            ok = s[:, 11] > 0
            s = s[ok]
            if not len(s):
                continue
            f = lambda a: float(np.mean(a))
            it = [f(s[:, 2 + t] - (s[:, 1 + t])) for t in range(7)]
            t0 = s[:, 0].min()
            print(f"{names.get(k, k)}: blocks {len(s)}  prologue {f(s[:,1]-s[:,0]):.0f}  iters {' '.join('%.0f' % v for v in it)}  "
                  f"epilogue(issue) {f(s[:,10]-s[:,8]):.0f}  store drain {f(s[:,11]-s[:,10]):.0f}  block total {f(s[:,11]-s[:,0]):.0f}  "
                  f"kernel span {(s[:,11].max()-t0)} cycles", flush=True)
            print(f"    inside iteration 1: first product {f(s[:,12]-s[:,2]):.0f}  second {f(s[:,13]-s[:,12]):.0f}  "
                  f"valu {f(s[:,14]-s[:,13]):.0f}  accumulate {f(s[:,15]-s[:,14]):.0f}  wait+barrier {f(s[:,3]-s[:,15]):.0f}", flush=True)
            if s[:, 16].max() > 0:
                print(f"    accumulate detail: to first wait {f(s[:,16]-s[:,14]):.0f}  unit0 issue {f(s[:,17]-s[:,16]):.0f}  wait1 {f(s[:,18]-s[:,17]):.0f}  "
                      f"unit1 issue {f(s[:,19]-s[:,18]):.0f}  wait2+unit2 {f(s[:,20]-s[:,19]):.0f}  wait3+unit3 {f(s[:,15]-s[:,20]):.0f}", flush=True)
            # start-time distribution: how many blocks started in the first wave
            st0 = np.sort(s[:, 0] - t0)
            print(f"    block starts (cycles after the first): median {st0[len(st0)//2]}  90% {st0[int(len(st0)*0.9)]}  max {st0[-1]}", flush=True)
