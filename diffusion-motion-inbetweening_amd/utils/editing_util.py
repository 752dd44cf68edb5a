"""Host-side gates and schedules of imputation / reconstruction guidance.

Mirror of the part of the reference's ``utils/editing_util.py`` that the sampler calls EVERY step
(:299-346).  The reference evaluates ``(t >= stop_at).all()`` on a device tensor (a host sync per
step); all samples of a batch share the step index, so here the gates are plain integers handed to
the engine once (``cmdi_condition``), and the per-step weights are a precomputed table.

Keyframe-mask CONSTRUCTION (``get_keyframes_mask`` :56-229) runs once per batch on the host and
stays with the caller; `joint_to_full_mask` and the HumanML3D joint→feature matrices are provided
because mask semantics are part of the path's contract.
"""
from __future__ import annotations

import numpy as np

N_JOINTS, N_FEATS = 22, 263


def _joint_feature_matrices():
    """HumanML3D 263-vector layout (reference data_loaders/humanml_utils.py:68-92):
    [root rot-vel(1), root xz-vel(2), root y(1) | 21x3 ric pos | 21x6 rot | 22x3 vel | 4 contacts]."""
    pos = np.zeros((N_JOINTS, N_FEATS), dtype=bool)
    rot = np.zeros_like(pos)
    vel = np.zeros_like(pos)
    cnt = np.zeros_like(pos)
    pos[0, 1:4] = True
    rot[0, 0] = True
    for j in range(1, N_JOINTS):
        pos[j, 4 + 3 * (j - 1):4 + 3 * j] = True
        rot[j, 4 + 63 + 6 * (j - 1):4 + 63 + 6 * j] = True
    for j in range(N_JOINTS):
        vel[j, 4 + 63 + 126 + 3 * j:4 + 63 + 126 + 3 * (j + 1)] = True
    for joint, col in ((7, -4), (10, -3), (8, -2), (11, -1)):
        cnt[joint, col] = True
    return pos, rot, vel, cnt


MAT_POS, MAT_ROT, MAT_VEL, MAT_CNT = _joint_feature_matrices()


def joint_to_full_mask(joint_mask, mode='pos_rot_vel'):
    """[B, 22, 1, T] joint mask -> [B, 263, 1, T] feature mask (reference :30-44)."""
    import torch
    assert mode in ('pos', 'pos_rot', 'pos_rot_vel')
    mats = [MAT_POS, MAT_CNT]
    if mode in ('pos_rot', 'pos_rot_vel'):
        mats.append(MAT_ROT)
    if mode == 'pos_rot_vel':
        mats.append(MAT_VEL)
    sel = torch.from_numpy(np.any(np.stack(mats), axis=0)).to(joint_mask.device)  # [22, 263]
    jm = joint_mask.bool().permute(0, 2, 3, 1)                                     # [B, 1, T, 22]
    full = (jm.unsqueeze(-1) & sel).any(dim=-2)                                    # [B, 1, T, 263]
    return full.permute(0, 3, 1, 2)


def get_gradient_schedule(schedule_name=None, num_diffusion_steps=1000, scale=.05):
    """Reconstruction-guidance weight per step (reference :299-322), float64 numpy."""
    n = num_diffusion_steps
    if schedule_name is None:
        return np.ones(n)
    if schedule_name == 'first-half':
        return np.concatenate((np.ones(n // 2), np.zeros(n - n // 2)))
    if schedule_name == 'last-half':
        return np.concatenate((np.zeros(n // 2), np.ones(n // 2)))
    if schedule_name == 'exponential':
        return np.exp(-scale * np.arange(n)[::-1])
    if schedule_name == 'sigmoid':
        return 1 / (1 + np.exp((scale / 5) * (-np.arange(n) + n / 2)))
    if schedule_name == 'half-sigmoid':
        return 1 / (1 + np.exp((scale / 5) * (-np.arange(n))))
    raise NotImplementedError(
        f"unknown guidance schedule for reconstruction guidance: {schedule_name}")


def uses_reconstruction_guidance(y) -> bool:
    """Step-independent half of requires_reconstruction_guidance (reference :325-333)."""
    if not y.get('reconstruction_guidance', False):
        return False
    assert 'stop_recguidance_at' in y
    assert 'inpainting_mask' in y and 'inpainted_motion' in y
    return True


def uses_imputation(y) -> bool:
    """Step-independent half of requires_imputation (reference :336-346)."""
    if not y.get('imputate', False):
        return False
    assert 'stop_imputation_at' in y
    assert 'inpainting_mask' in y and 'inpainted_motion' in y
    return True


def requires_reconstruction_guidance(model_kwargs, denoising_step) -> bool:
    y = model_kwargs['y']
    return uses_reconstruction_guidance(y) and int(_min_step(denoising_step)) >= int(y['stop_recguidance_at'])


def requires_imputation(model_kwargs, denoising_step) -> bool:
    y = model_kwargs['y']
    return uses_imputation(y) and int(_min_step(denoising_step)) >= int(y['stop_imputation_at'])


def _min_step(step):
    return step.min().item() if hasattr(step, 'min') else step
