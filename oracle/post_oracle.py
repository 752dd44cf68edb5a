"""numpy fp32 restatement of the step that follows the sampling loop in every caller of the reference:
un-normalise the HumanML3D vectors and recover XYZ joint positions.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows, line by line:
    caller        sample/conditional_synthesis.py:229-235   sample.cpu().permute(0,2,3,1) -> inv_transform
                  -> recover_from_ric(sample, 22, abs_3d) -> view(-1, T, 22, 3).permute(0, 2, 3, 1)
    inv_transform data_loaders/humanml/data/dataset.py:378-382   data * std + mean
    root          data_loaders/humanml/scripts/motion_process.py:402-441  recover_root_rot_pos
    joints        data_loaders/humanml/scripts/motion_process.py:474-491  recover_from_ric
    qinv / qrot   data_loaders/humanml/common/quaternion.py:16-20,54-73
Pinned by tests/golden/post_ric.npz (outputs of the real reference, tests/golden/make_golden_post.py).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def _qrot(q, v):
    """v + 2 * (q0 * (qv x v) + qv x (qv x v)), all fp32 (quaternion.py:70-73)."""
    qv = q[..., 1:]
    uv = np.cross(qv, v).astype(F32)
    uuv = np.cross(qv, uv).astype(F32)
    return (v + F32(2) * (q[..., :1] * uv + uuv)).astype(F32)


def recover_xyz(sample, mean, std, n_joints: int = 22, abs_3d: bool = False):
    """sample [B, 263, 1, T] (z-scored) -> joint positions [B, n_joints, 3, T]."""
    sample = np.asarray(sample, dtype=F32)
    B, J, Fd, T = sample.shape
    data = sample.transpose(0, 2, 3, 1).reshape(B, T, J) * np.asarray(std, F32) + np.asarray(mean, F32)
    data = data.astype(F32)
    if abs_3d:
        ang = data[..., 0]
    else:
        ang = np.zeros((B, T), dtype=F32)
        ang[:, 1:] = data[:, :-1, 0]
        ang = np.cumsum(ang, axis=-1, dtype=F32)
    quat = np.zeros((B, T, 4), dtype=F32)
    quat[..., 0] = np.cos(ang)
    quat[..., 2] = np.sin(ang)
    qinv = quat * np.asarray([1, -1, -1, -1], dtype=F32)
    r_pos = np.zeros((B, T, 3), dtype=F32)
    if abs_3d:
        r_pos[..., [0, 2]] = data[..., 1:3]
    else:
        r_pos[:, 1:, [0, 2]] = data[:, :-1, 1:3]
        r_pos = _qrot(qinv, r_pos)
        r_pos = np.cumsum(r_pos, axis=-2, dtype=F32)
    r_pos[..., 1] = data[..., 3]
    pos = data[..., 4:(n_joints - 1) * 3 + 4].reshape(B, T, n_joints - 1, 3)
    pos = _qrot(np.broadcast_to(qinv[:, :, None, :], pos.shape[:-1] + (4,)), pos)
    pos[..., 0] += r_pos[..., None, 0]
    pos[..., 2] += r_pos[..., None, 2]
    xyz = np.concatenate([r_pos[:, :, None, :], pos], axis=2)       # [B, T, n_joints, 3]
    return np.ascontiguousarray(xyz.transpose(0, 2, 3, 1))          # [B, n_joints, 3, T]
