"""Small-M GEMMs (B <= 10): where a block's life goes (CMDI_H3_DBG=16 stamps of the probes library), per tile shape.
   python tools/small_m_timeline.py [tiles, comma separated]"""
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
TILES = tuple(int(t) for t in sys.argv[1].split(",")) if len(sys.argv) > 1 else (21, 8, 7, 6, 4, 5)
SHAPE = {2: (256, 128), 3: (256, 128), 10: (256, 128), 11: (128, 256), 9: (128, 256), 4: (128, 64), 5: (128, 64), 6: (64, 128), 21: (64, 128)}
for m in (788, 3940):
    for (n, k, name) in [(1536, 512, "in_proj"), (512, 512, "out_proj-like"), (512, 1024, "linear2-like")]:
        a = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev); b = torch.randn(n, device=dev)
        a_s, w_s = eng.split_f16(a), eng.split_f16(w)
        for tile in TILES:
            bm, bn = SHAPE.get(tile, (128, 128))
            nblk = ((m + bm - 1) // bm) * ((n + bn - 1) // bn)
            buf = torch.zeros(nblk * 6 + 1024, dtype=torch.int64, device=dev)
            out = torch.empty(m, 2 * n, device=dev, dtype=torch.float16)
            for _ in range(3):
                eng.gemm_h3(a_s, w_s, b, tile=tile, epi=0, resid=buf, split_out=True, out=out)
            torch.cuda.synchronize()
            raw = buf[:nblk * 6].cpu().numpy().reshape(nblk, 6).astype(np.float64)
            life = raw[:, 3] - raw[:, 0]
            loop = raw[:, 2]
            ghz = (life / ((raw[:, 5] - raw[:, 4]) * 10.0)).mean()
            span_us = (raw[:, 5].max() - raw[:, 4].min()) / 100.0
            print(f"M={m} {name:13s} tile {tile:2d}: {nblk:4d} blocks, span {span_us:5.1f} us, clock {ghz:.2f} GHz; block life {life.mean() / ghz / 1e3:5.1f} us, "
                  f"k-loop {loop.mean() / ghz / 1e3:5.1f} us ({loop.mean() / (k // 32):5.0f} cycles/step), rest {(life - loop).mean() / ghz / 1e3:4.1f} us", flush=True)
