"""Run in a subprocess by test_attention_split_schedule_is_bitwise_identical (tests/test_gpu_parity.py) with CMDI_ATTN_SPLIT
= 0 / 1 / unset in the environment (read once per process): the split-f16 attention core on a small batch (where the unset
default picks the two-block schedule) and on one big enough to fill the chip -> <out>.npz."""
import importlib
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO))
eng = importlib.import_module("diffusion-motion-inbetweening_amd.engine")


def main(out_path):
    g = torch.Generator().manual_seed(11)
    res = {}
    for name, n_seq, S in (("small", 3, 197), ("edge", 2, 129), ("big", 40, 197)):
        qkv = (torch.randn(n_seq * S, 3 * 512, generator=g) * 1.5).to("cuda:0")
        res[name] = eng.attention_fwd_h3(qkv, n_seq, S, 4).cpu().numpy()
    np.savez(out_path, **res)


if __name__ == "__main__":
    main(sys.argv[1])
