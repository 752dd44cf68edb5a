"""Factory = the drop-in boundary.  Mirror of the reference's ``utils/model_util.py``:
``create_model_and_diffusion`` (:26-37), ``get_model_args`` (:40-119), ``create_gaussian_diffusion``
(:122-165), ``load_model_wo_clip`` / ``load_saved_model`` (:19-23,168-182).  `args` is the same
options object the reference builds (utils/parser_util.py dataclasses or a namespace with the same
attributes); only the attributes the transformer path reads are required.
"""
from __future__ import annotations

import torch

from ..diffusion import gaussian_diffusion as gd
from ..diffusion.respace import DiffusionConfig, SpacedDiffusion, space_timesteps
from ..model.mdm import MDM
from ..model.mdm_unet import MDM_UNET


def load_model_wo_clip(model, state_dict):
    missing_keys, unexpected_keys = model.load_state_dict(state_dict, strict=False)
    assert len(unexpected_keys) == 0, f'unexpected keys: {unexpected_keys}'
    assert all(k.startswith('clip_model.') for k in missing_keys), missing_keys


def create_model_and_diffusion(args, data=None):
    arch = getattr(args, 'arch', 'trans_enc')
    if arch.startswith('dit') or arch == 'unet_large':
        raise NotImplementedError(
            f"arch={arch!r}: MDM_DiT / TemporalUnetLarge are outside the MI355X hot path (SURVEY.md §8f); "
            "any torch denoiser can still be sampled through diffusion.p_sample_loop")
    if arch == 'unet':   # reference utils/model_util.py:31-33
        model = MDM_UNET(**get_model_args(args, data))
    else:
        model = MDM(**get_model_args(args, data))
    diffusion = create_gaussian_diffusion(args)
    return model, diffusion


def get_model_args(args, data=None):
    dataset = getattr(args, 'dataset', 'humanml')
    if getattr(args, 'unconstrained', False) or dataset == 'amass':
        cond_mode = 'no_cond'
    elif dataset in ('kit', 'humanml'):
        cond_mode = 'text'
    else:
        cond_mode = 'action'
    ds = getattr(data, 'dataset', None)
    num_actions = getattr(ds, 'num_actions', 1)

    data_rep, njoints, nfeats = 'rot6d', 25, 6
    if dataset == 'humanml':
        data_rep, nfeats = 'hml_vec', 1
        njoints = 67 if getattr(args, 'drop_redundant', False) else 263
    elif dataset == 'kit':
        data_rep, njoints, nfeats = 'hml_vec', 251, 1
    elif dataset == 'amass':
        data_rep, njoints, nfeats = 'hml_vec', 764, 1
    if getattr(args, 'traj_only', False):
        njoints, nfeats = 4, 1

    return {
        'modeltype': '', 'njoints': njoints, 'nfeats': nfeats, 'num_actions': num_actions,
        'translation': True, 'pose_rep': 'rot6d', 'glob': True, 'glob_rot': True,
        'latent_dim': getattr(args, 'latent_dim', 512), 'ff_size': getattr(args, 'ff_size', 1024),
        'num_layers': getattr(args, 'layers', 8), 'num_heads': 4, 'dropout': 0.1,
        'activation': "gelu", 'data_rep': data_rep, 'cond_mode': cond_mode,
        'cond_mask_prob': getattr(args, 'cond_mask_prob', .1), 'action_emb': 'tensor',
        'arch': getattr(args, 'arch', 'trans_enc'),
        'emb_trans_dec': getattr(args, 'emb_trans_dec', False),
        'clip_version': 'ViT-B/32', 'dataset': dataset,
        'keyframe_conditioned': getattr(args, 'keyframe_conditioned', False),
        # UNET-only (reference utils/model_util.py:96-117)
        'dim_mults': tuple(getattr(args, 'dim_mults', (2, 2, 2, 2))), 'adagn': getattr(args, 'unet_adagn', True),
        'zero': getattr(args, 'unet_zero', True), 'xz_only': getattr(args, 'xz_only', False),
    }


def create_gaussian_diffusion(args):
    steps = 1000
    timestep_respacing = 'ddim100' if getattr(args, 'use_ddim', False) else ''
    betas = gd.get_named_beta_schedule(getattr(args, 'noise_schedule', 'cosine'), steps, 1.)
    if not timestep_respacing:
        timestep_respacing = [steps]
    g = lambda name, default: getattr(args, name, default)
    return SpacedDiffusion(
        use_timesteps=space_timesteps(steps, timestep_respacing),
        conf=DiffusionConfig(
            betas=betas,
            model_mean_type=(gd.ModelMeanType.START_X if g('predict_xstart', True)
                             else gd.ModelMeanType.EPSILON),
            model_var_type=(gd.ModelVarType.FIXED_SMALL if g('sigma_small', True)
                            else gd.ModelVarType.FIXED_LARGE),
            loss_type=gd.LossType.MSE,
            rescale_timesteps=False,
            lambda_vel=g('lambda_vel', 0.), lambda_rcxyz=g('lambda_rcxyz', 0.),
            lambda_fc=g('lambda_fc', 0.), clip_range=g('clip_range', 6.0),
            train_trajectory_only_xz=g('xz_only', False), use_random_proj=g('use_random_proj', False),
            fp16=g('use_fp16', False), traj_only=g('traj_only', False), abs_3d=g('abs_3d', False),
            apply_zero_mask=g('apply_zero_mask', False), traj_extra_weight=g('traj_extra_weight', 1.),
            time_weighted_loss=g('time_weighted_loss', False),
            train_x0_as_eps=g('train_x0_as_eps', False),
        ),
    )


def load_saved_model(model, model_path, use_avg: bool = True):
    state_dict = torch.load(model_path, map_location='cpu')
    if use_avg and 'model_avg' in state_dict:
        print('loading avg model')
        state_dict = state_dict['model_avg']
    elif 'model' in state_dict:
        print('loading model without avg')
        state_dict = state_dict['model']
    else:
        print('checkpoint has no avg model')
    load_model_wo_clip(model, state_dict)
    return model
