"""Device-side replacements for the two functions the reference's sample scripts call right after the
sampling loop (data_loaders/humanml/scripts/motion_process.py:474-491 ``recover_from_ric`` and, fused with it,
``t2m_dataset.inv_transform``, data_loaders/humanml/data/dataset.py:378-382).  Only these: the rest of the
reference's motion_process.py (feature extraction from raw mocap) is data preparation, out of scope.
"""
from __future__ import annotations

import torch

from .... import _native as N


def sample_to_xyz(sample: torch.Tensor, mean=None, std=None, n_joints: int = 22, abs_3d: bool = False):
    """The block of sample/conditional_synthesis.py:229-235 in one kernel: sample [B, 263, 1, T] on the GPU
    (z-scored; pass mean / std [263] to un-normalise) -> XYZ joint positions [B, n_joints, 3, T]."""
    if not sample.is_cuda:
        raise N.NativeError("sample_to_xyz runs on a HIP device only (no CPU path)")
    lib = N.load()
    x = sample.detach().to(torch.float32).contiguous()
    B, J, Fd, T = x.shape
    assert Fd == 1
    dev = x.device
    m = s = None
    if mean is not None:
        m = torch.as_tensor(mean, dtype=torch.float32).to(dev).contiguous()
        s = torch.as_tensor(std, dtype=torch.float32).to(dev).contiguous()
        assert m.numel() == J and s.numel() == J
    out = torch.empty((B, n_joints, 3, T), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        N.check(lib.cmdi_recover_xyz(N.ptr(x), N.ptr(m), N.ptr(s), N.ptr(out), B, J, T, n_joints, int(abs_3d),
                                     N.current_stream(dev)))
    return out


def recover_from_ric(data: torch.Tensor, joints_num: int, abs_3d: bool = False) -> torch.Tensor:
    """Same signature and layout as the reference: data [..., T, 263] (already un-normalised) ->
    [..., T, joints_num, 3].  Runs on the tensor's HIP device."""
    lead, T, J = data.shape[:-2], data.shape[-2], data.shape[-1]
    x = data.reshape(-1, T, J).permute(0, 2, 1).unsqueeze(2)            # [B, 263, 1, T]
    xyz = sample_to_xyz(x, None, None, joints_num, abs_3d)               # [B, joints, 3, T]
    return xyz.permute(0, 3, 1, 2).reshape(*lead, T, joints_num, 3)
