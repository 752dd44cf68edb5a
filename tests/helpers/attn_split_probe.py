"""Run in a subprocess by test_attention_split_schedule_is_bitwise_identical (tests/test_gpu_parity.py) with CMDI_ATTN_SPLIT
= 0 / 1 / unset (and by test_attention_persistent_schedule_is_bitwise_identical with CMDI_ATTN_PERSIST = 0 / 1) in the environment
(read once per process): the split-f16 attention core on a small batch (where the unset default picks the two-block schedule),
on one big enough to fill the chip and on batches with several (sequence, head) pairs per CU -> <out>.npz."""
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
    # "many": more than two (sequence, head) pairs per CU — the persistent schedule (CMDI_ATTN_PERSIST) when it is on: uneven
    # shares (133 x 4 = 532 pairs on 256 blocks), S = 197 and a shorter sequence
    for name, n_seq, S in (("small", 3, 197), ("edge", 2, 129), ("big", 40, 197), ("many", 133, 197), ("many_short", 140, 150)):
        qkv = (torch.randn(n_seq * S, 3 * 512, generator=g) * 1.5).to("cuda:0")
        res[name] = eng.attention_fwd_h3(qkv, n_seq, S, 4).cpu().numpy()
    np.savez(out_path, **res)


if __name__ == "__main__":
    main(sys.argv[1])
