"""Classifier-free guidance wrapper, mirror of the reference's ``model/cfg_sampler.py:5-35``.

``forward`` = out_uncond + text_scale * (out_cond - out_uncond).  The reference runs the denoiser
twice on a deep-copied ``y``; here the two passes are ONE native forward over a 2B batch (they share
every frame token and differ only in the conditioning token), combined by a fused kernel.
"""
import torch
import torch.nn as nn


class ClassifierFreeSampleModel(nn.Module):

    def __init__(self, model):
        super().__init__()
        self.model = model
        assert self.model.cond_mask_prob > 0, \
            'Cannot run a guided diffusion on a model that has not been trained with no conditions'
        # attributes the sample scripts read through the wrapper
        self.rot2xyz = self.model.rot2xyz
        self.translation = self.model.translation
        self.njoints = self.model.njoints
        self.nfeats = self.model.nfeats
        self.data_rep = self.model.data_rep
        self.cond_mode = self.model.cond_mode
        self.keyframe_conditioned = self.model.keyframe_conditioned
        self.mask_value = -2.0

    def forward(self, x, timesteps, y=None, obs_x0=None, obs_mask=None, **kwargs):
        assert self.model.cond_mode in ['text', 'action']
        if getattr(self.model, 'arch', '') == 'unet':   # MDM_UNET consumes the keyframe observations
            return self.model._forward_native(x, timesteps, y, cfg=True, obs_x0=obs_x0, obs_mask=obs_mask)
        return self.model._forward_native(x, timesteps, y, cfg=True)
