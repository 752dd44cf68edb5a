"""Host-side gates and schedules of imputation / reconstruction guidance.

Mirror of the part of the reference's ``utils/editing_util.py`` that the sampler calls EVERY step
(:299-346).  The reference evaluates ``(t >= stop_at).all()`` on a device tensor (a host sync per
step); all samples of a batch share the step index, so here the gates are plain integers handed to
the engine once (``cmdi_condition``), and the per-step weights are a precomputed table.

Keyframe-mask CONSTRUCTION (``get_keyframes_mask`` :56-229) is the step right before
the loop: the reference fills the mask sample by sample after ``lengths.cpu()`` (a host sync and B small
device writes); here the inference modes are one broadcast expression on the data's device.  The modes that
draw from ``np.random`` consume it in the reference's order, so a seeded call gives the same mask.
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


# reference data_loaders/humanml_utils.py:1-37 (joint order of HML_JOINT_NAMES)
HML_LOWER_BODY_JOINTS = (0, 1, 2, 4, 5, 7, 8, 10, 11)   # pelvis, hips, knees, ankles, feet
HML_PELVIS_FEET = (0, 10, 11)
HML_PELVIS_VR = (0, 20, 21, 15)                         # pelvis, wrists, head
_JOINT_MODES = {'right_wrist': (0, 21), 'lower_body': HML_LOWER_BODY_JOINTS, 'pelvis_feet': HML_PELVIS_FEET,
                'pelvis_vr': HML_PELVIS_VR, 'pelvis': (0,)}


def get_keyframes_mask(data, lengths, edit_mode='benchmark_sparse', trans_length=10, feature_mode='pos_rot_vel',
                       get_joint_mask=False, n_keyframes=5):
    """Observation mask [B, 263, 1, T] (and the joint mask [B, 22, 1, T] if ``get_joint_mask``) of the
    reference's ``get_keyframes_mask`` (utils/editing_util.py:56-229) for HumanML3D data, built on
    ``data.device`` without a host round trip for the deterministic modes:
      benchmark_sparse  every trans_length-th frame below the sequence length          (:85-91)
      benchmark_clip    everything but the middle trans_length frames                  (:93-100)
      uncond            nothing                                                        (:102-105)
      right_wrist / lower_body / pelvis_feet / pelvis_vr / pelvis   joint trajectories (:107-148)
      gmd_keyframes / random_frames   n_keyframes / 20 frames drawn with np.random.choice per sample (:150-167)
    The training-only modes ('random_joints', 'random') and AMASS data (764 features) are not part of the path."""
    import torch
    B, n_joints, n_features, T = data.shape
    if n_joints != N_FEATS:
        raise ValueError('Unknown number of joints: {}'.format(n_joints)) if n_joints != 764 else \
            NotImplementedError('AMASS keyframe masks are outside the sampling path')
    dev = data.device
    lengths = torch.as_tensor(lengths, device=dev).long().view(B, 1)
    frame = torch.arange(T, device=dev).view(1, T)
    valid = frame < lengths                                                   # [B, T]
    joints = torch.ones(N_JOINTS, dtype=torch.bool, device=dev)
    if edit_mode == 'benchmark_sparse':
        frames = valid & (frame % int(trans_length) == 0)
    elif edit_mode == 'benchmark_clip':
        end = torch.div(lengths - int(trans_length), 2, rounding_mode='floor')
        frames = valid & ((frame < end) | (frame >= end + int(trans_length)))
    elif edit_mode == 'uncond':
        frames = torch.zeros_like(valid)
    elif edit_mode in _JOINT_MODES:
        frames = valid
        joints = torch.zeros_like(joints)
        joints[list(_JOINT_MODES[edit_mode])] = True
    elif edit_mode in ('gmd_keyframes', 'random_frames'):
        n = int(n_keyframes) if edit_mode == 'gmd_keyframes' else 20
        picks = np.zeros((B, T), dtype=bool)
        for i, length in enumerate(lengths.view(-1).cpu().numpy()):           # same np.random stream as the reference
            picks[i, np.random.choice(range(int(length)), n, replace=False)] = True
        frames = torch.from_numpy(picks).to(dev)
    elif edit_mode in ('random_joints', 'random'):
        raise NotImplementedError(f"edit_mode '{edit_mode}' is a training-time augmentation, not part of the sampling path")
    else:
        raise ValueError(f'unknown edit_mode: {edit_mode}')   # (the reference silently returns an empty mask)
    joint_mask = (joints.view(1, N_JOINTS, 1, 1) & frames.view(B, 1, 1, T)).expand(B, N_JOINTS, n_features, T)
    assert feature_mode in ('pos', 'pos_rot', 'pos_rot_vel')
    mats = [MAT_POS, MAT_CNT] + ([MAT_ROT] if feature_mode != 'pos' else []) + ([MAT_VEL] if feature_mode == 'pos_rot_vel' else [])
    sel = torch.from_numpy(np.any(np.stack(mats), axis=0)).to(dev)            # [22, 263]
    feats = (sel & joints.view(N_JOINTS, 1)).any(dim=0)                       # [263]
    full = (feats.view(1, N_FEATS, 1, 1) & frames.view(B, 1, 1, T)).expand(B, N_FEATS, n_features, T).contiguous()
    if get_joint_mask:
        return full, joint_mask.contiguous()
    return full


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
