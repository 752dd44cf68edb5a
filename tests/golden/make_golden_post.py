"""Golden vectors of the post-sampling step, produced by the REAL reference on CPU:

    sample.cpu().permute(0, 2, 3, 1) -> dataset.inv_transform (data * std + mean)
    -> recover_from_ric(sample, 22, abs_3d) -> view(-1, T, 22, 3).permute(0, 2, 3, 1)
(sample/conditional_synthesis.py:229-235; data_loaders/humanml/data/dataset.py:378-382;
 data_loaders/humanml/scripts/motion_process.py:402-441,474-491).   python tests/golden/make_golden_post.py
"""
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
from data_loaders.humanml.scripts.motion_process import recover_from_ric  # noqa: E402  (the reference's)

inp = cases.make_post_inputs()
sample = torch.from_numpy(inp["sample"]).permute(0, 2, 3, 1)               # [B, 1, T, 263]
data = sample * torch.from_numpy(inp["std"]) + torch.from_numpy(inp["mean"])  # inv_transform
out = {}
for abs_3d in (False, True):
    xyz = recover_from_ric(data.float(), cases.POST_CASE["n_joints"], abs_3d=abs_3d)  # [B, 1, T, 22, 3]
    xyz = xyz.view(-1, *xyz.shape[2:]).permute(0, 2, 3, 1)                               # [B, 22, 3, T]
    out[f"xyz_abs{int(abs_3d)}"] = xyz.numpy().astype(np.float32)
out["fingerprint"] = cases.fingerprint(inp)
np.savez_compressed(HERE / "post_ric.npz", **out)
print({k: v.shape for k, v in out.items()})
