"""`Rotation2xyz` stand-in.

The reference's MDM instantiates an SMPL-backed ``Rotation2xyz`` (model/rotation2xyz.py:11-15,
model/mdm.py:165) although, for the HumanML3D ``hml_vec`` representation this path serves, callers
only ever use ``pose_rep='xyz'`` for which it is the identity (:20-21).  SMPL forward kinematics is
post-processing outside the sampling hot path, so only the identity is provided.
"""
import torch.nn as nn


class Rotation2xyz:
    def __init__(self, device='cpu', dataset='amass'):
        self.device = device
        self.dataset = dataset
        self.smpl_model = nn.Identity()  # callers do .smpl_model._apply / .train on it

    def __call__(self, x, mask=None, pose_rep='xyz', translation=True, glob=True,
                 jointstype='smpl', vertstrans=False, **kwargs):
        if pose_rep == 'xyz':
            return x
        raise NotImplementedError(
            "SMPL forward kinematics is outside the MI355X sampling path (pose_rep must be 'xyz')")
