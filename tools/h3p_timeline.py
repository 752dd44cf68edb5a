"""Cycle stamps of the persistent split-f16 GEMM (gemm_h3p.hpp, probes build, CMDI_H3_DBG=16): one K step (the 9th of each block's
first tile) cut into its four barrier intervals, per wave, + the tile's K loop and the block's life."""
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
for (m, n, k, epi, name) in [(12608, 1536, 512, 0, "in_proj"), (100864, 1536, 512, 0, "in_proj B=256")]:
    a = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) * 0.05; b = torch.randn(n, device=dev)
    a_s, w_s = eng.split_f16(a), eng.split_f16(w)
    buf = torch.zeros(256 * 8 * 16 + 64, dtype=torch.int64, device=dev)
    out = torch.empty(m, 2 * n, device=dev, dtype=torch.float16)
    for _ in range(3):
        eng.gemm_h3(a_s, w_s, b, tile=1008, epi=epi, resid=buf, split_out=True, out=out)
    torch.cuda.synchronize()
    raw = buf[:256 * 8 * 16].cpu().numpy().reshape(256, 8, 16).astype(np.int64)
    ts = raw[:, :, :10].astype(np.float64)
    ok = ts[:, 0, 0] > 0
    print(f"== {name}: {ok.sum()} blocks stamped")
    # stamps per k-substep ks (offset 5 ks): 0 interval start, 1 reads done (before barrier), 2 after barrier, 3 MFMAs issued, 4 after barrier
    names = ["read0", "bar_wait0", "mfma0", "bar_wait0b", "read1", "bar_wait1", "mfma1", "bar_wait1b"]
    d = np.stack([ts[..., 1] - ts[..., 0], ts[..., 2] - ts[..., 1], ts[..., 3] - ts[..., 2], ts[..., 4] - ts[..., 3],
                  ts[..., 6] - ts[..., 5], ts[..., 7] - ts[..., 6], ts[..., 8] - ts[..., 7], ts[..., 9] - ts[..., 8]], -1)[ok]
    for g, sel in (("group 0 (waves 0-3)", slice(0, 4)), ("group 1 (waves 4-7)", slice(4, 8))):
        mean = d[:, sel].mean((0, 1))
        print(f"  {g}: " + "  ".join(f"{nm} {v:6.0f}" for nm, v in zip(names, mean)) + f"   | K step total {mean.sum():6.0f}")
    kloop = (raw[:, :, 11] - raw[:, :, 10])[ok].astype(np.float64)
    stats = (raw[:, :, 12] - raw[:, :, 11])[ok].astype(np.float64)
    epi = (raw[:, :, 13] - raw[:, :, 12])[ok].astype(np.float64)
    life = (raw[:, :, 14] - raw[:, :, 10])[ok].astype(np.float64)
    print(f"  first tile: K loop {kloop.mean():.0f} cycles ({kloop.mean() / (k // 32):.0f} per K step), next tile's row statistics {stats.mean():.0f}, "
          f"epilogue group 0 {epi[:, :4].mean():.0f} / group 1 {epi[:, 4:].mean():.0f}; block life from its first K loop {life.mean():.0f}")
    t0 = raw[:, :, 10][ok].astype(np.float64)
    rel = lambda i: (raw[:, :, i][ok] - t0[:, :1]).mean(0)
    print("  per wave, relative to wave 0's K-loop start: K-loop end", np.round(rel(11)).tolist(), " epilogue end", np.round(rel(13)).tolist())
    simd = (raw[:, :, 15] >> 4) & 3     # HW_ID: wave_id[3:0], simd_id[5:4]
    print("  SIMD of waves 0..7 (block 0):", simd[0].tolist())
