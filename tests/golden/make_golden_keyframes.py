"""Golden keyframe masks from the REAL reference's get_keyframes_mask (utils/editing_util.py:56-229), bit-packed.
python tests/golden/make_golden_keyframes.py"""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(HERE))
import cases  # noqa: E402
from oracle import ref_shims  # noqa: E402

ref_shims.import_reference()
from utils.editing_util import get_keyframes_mask  # noqa: E402  (the reference's)

out = {}
kc = cases.KEYFRAME_CASE
data = torch.zeros(kc["B"], 263, 1, kc["T"])
lengths = torch.tensor(kc["lengths"])
for i, (mode, trans, feat, nk) in enumerate(cases.KEYFRAME_MODES):
    np.random.seed(kc["seed"] + i)
    full, joint = get_keyframes_mask(data, lengths, edit_mode=mode, trans_length=trans, feature_mode=feat,
                                     get_joint_mask=True, n_keyframes=nk)
    assert full.shape == (kc["B"], 263, 1, kc["T"]) and joint.shape == (kc["B"], 22, 1, kc["T"])
    out[f"full.{i}"] = np.packbits(full.numpy())
    out[f"joint.{i}"] = np.packbits(joint.numpy())
    print(mode, trans, feat, int(full.sum()), int(joint.sum()))
rc = cases.KEYFRAME_RANDOM_FRAMES
np.random.seed(rc["seed"])
full = get_keyframes_mask(torch.zeros(rc["B"], 263, 1, rc["T"]), torch.tensor(rc["lengths"]), edit_mode="random_frames")
out["full.random_frames"] = np.packbits(full.numpy())
np.savez_compressed(HERE / "keyframe_masks.npz", **out)
