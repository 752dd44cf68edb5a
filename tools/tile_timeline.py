"""Per-block timeline of one split-f16 GEMM launch (CMDI_H3_DBG=16: s_memtime stamps written by the kernel)."""
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
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for (m, n, k, epi, name) in [(12608, 1536, 512, 0, "in_proj"), (12608, 1024, 512, 1, "linear1"), (5376, 1536, 512, 0, "in_proj 1 round")]:
    a = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev); b = torch.randn(n, device=dev)
    a_s, w_s = eng.split_f16(a), eng.split_f16(w)
    bm, bn = {2: (256, 128), 3: (256, 128), 10: (256, 128), 11: (128, 256), 9: (128, 256), 4: (128, 64), 5: (128, 64),
              6: (64, 128)}.get(tile, (128, 128))
    nblk = ((m + bm - 1) // bm) * (n // bn)
    buf = torch.zeros(nblk * 6 + 1024, dtype=torch.int64, device=dev)
    out = torch.empty(m, 2 * n, device=dev, dtype=torch.float16)
    for _ in range(3):
        eng.gemm_h3(a_s, w_s, b, tile=tile, epi=epi, resid=buf, split_out=True, out=out)
    torch.cuda.synchronize()
    raw = buf[:nblk * 6].cpu().numpy().reshape(nblk, 6).astype(np.float64)
    # per block: [0] start, [1] (xcc << 32 | hw_id), [2] K-loop cycles, [3] end (shader clock); [4], [5] start / end (100 MHz)
    life = raw[:, 3] - raw[:, 0]
    loop = raw[:, 2]
    ghz = (life / ((raw[:, 5] - raw[:, 4]) * 10.0)).mean()   # ticks per ns
    span_us = (raw[:, 5].max() - raw[:, 4].min()) / 100.0
    start_us = (raw[:, 4] - raw[:, 4].min()) / 100.0
    first = start_us < 1.0
    print(f"{name}: {nblk} blocks, span {span_us:.1f} us, shader clock {ghz:.2f} GHz; per block mean: life {life.mean():.0f} cycles "
          f"({life.mean() / ghz / 1e3:.1f} us), k-loop {loop.mean():.0f} ({loop.mean() / (k // 32):.0f}/step, {100 * loop.mean() / life.mean():.0f}%), "
          f"rest {(life - loop).mean():.0f}; first-round blocks ({first.sum()}): life {life[first].mean():.0f}, k-loop {loop[first].mean():.0f}; "
          f"later: life {life[~first].mean():.0f}, k-loop {loop[~first].mean():.0f}")
