"""Cycle stamps of the weight-stationary GEMM (gemm_h3w.hpp; probes library, CMDI_H3_DBG=16): per block the prologue (W loads +
first tile's requests up to the first barrier) and per tile stream / barrier wait / epilogue.   python tools/h3w_timeline.py"""
import os as _os; _os.environ.setdefault("CMDI_PROBES_LIB", "1")
import importlib, os, sys
from pathlib import Path
import numpy as np
import torch
os.environ["CMDI_H3_DBG"] = "16"
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
eng = importlib.import_module("diffusion-motion-inbetweening_amd.engine")
dev = torch.device("cuda:0")
for (m, n, epi, name) in [(12608, 512, 0, "out_proj shape"), (12608, 1536, 0, "in_proj"), (12608, 1024, 1, "linear1")]:
    a = torch.randn(m, 512, device=dev); w = torch.randn(n, 512, device=dev) * 0.05; b = torch.randn(n, device=dev)
    a_s, w_s = eng.split_f16(a), eng.split_f16(w)
    nblk = 256
    buf = torch.zeros(nblk * 64 + 1024, dtype=torch.int64, device=dev)
    out = torch.empty(m, 2 * n, device=dev, dtype=torch.float16)
    for _ in range(3):
        eng.gemm_h3(a_s, w_s, b, tile=60, epi=epi, resid=buf, split_out=True, out=out)
    torch.cuda.synchronize()
    raw = buf[:nblk * 64].cpu().numpy().reshape(nblk, 64)
    cnt = raw[:, 62]
    ok = cnt > 2
    r = raw[ok].astype(np.float64)
    c = int(cnt[ok].min())
    n_t = (c - 2) // 3
    t0 = r[:, 0]
    print(f"{name}: {ok.sum()} blocks with work, stamps per block {c} ({n_t} tiles incl. all passes); start skew {((raw[ok, 63] - raw[ok, 63].min()) / 100.0).max():.2f} us")
    print(f"  prologue (start -> first barrier passed): mean {np.mean(r[:, 1] - t0):.0f} cycles, max {np.max(r[:, 1] - t0):.0f}")
    stream, wait, epi_c = [], [], []
    prev = r[:, 1]
    for k in range(n_t):
        s_end, b_end, e_end = r[:, 2 + 3 * k], r[:, 3 + 3 * k], r[:, 4 + 3 * k]
        stream.append(np.mean(s_end - prev)); wait.append(np.mean(b_end - s_end)); epi_c.append(np.mean(e_end - b_end))
        prev = e_end
    print("  per tile  stream:", " ".join(f"{x:.0f}" for x in stream))
    print("            barrier:", " ".join(f"{x:.0f}" for x in wait))
    print("            epilogue:", " ".join(f"{x:.0f}" for x in epi_c))
    if raw[ok, 48].any():
        tk = raw[ok, 48:55].astype(np.float64)
        print("  inside tile 2's stream (begin -> step 0 / 7 / 15 / 23 / 31 issued -> end):", " ".join(f"{x:.0f}" for x in np.mean(tk[:, 1:] - tk[:, :-1], axis=0)))
    print(f"  block life mean {np.mean(prev - t0):.0f} cycles; sum stream {sum(stream):.0f}, barrier {sum(wait):.0f}, epilogue {sum(epi_c):.0f}")
