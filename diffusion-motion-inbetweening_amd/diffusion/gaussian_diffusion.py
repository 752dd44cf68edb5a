"""Gaussian diffusion sampler, MI355X-native.

Host-side mirror of the SAMPLING half of the reference's ``diffusion/gaussian_diffusion.py``
(:24-71 schedules, :139-241 tables, :311-349 q_sample / posterior, :352-534 p_mean_variance,
:656-713 p_sample, :1149-1297 p_sample_loop[_progressive], :1300-1587 DDIM).  Same class / method /
keyword names, so ``sample.conditional_synthesis`` / ``sample.edit`` / ``sample.synthesize`` call
it unchanged — but no tensor arithmetic happens here: the float64 schedule tables are built once
on the host, cast to fp32 exactly like the reference's ``_extract_into_tensor`` (:2215-2229), and
every denoising step (denoiser + classifier-free combine + imputation / reconstruction guidance +
posterior update + noise) runs inside libcondmdi_hip.so on the model's HIP device.

``cond_fn`` guidance (the GMD legacy the reference keeps: ``p_sample_with_grad`` :715-800 /
``condition_mean_with_grad`` :579-603, ``ddim_sample_with_grad`` / ``condition_score_with_grad`` :636-660,1358-1416) is
served too: the denoiser and its input-VJP stay native (torch.autograd reaches them through ``_NativeDenoise``), only the
few elementwise lines that combine the caller's gradient with the posterior run as torch ops on the device.

Out of scope here (reference-only): training losses / VLB (:1805-2212), PLMS (:1589-1803) and the GMD post-sampling
imputation of ``p_sample_with_grad`` (:800-1105: needs the GMD data transforms).
"""
from __future__ import annotations

import enum
import math
from copy import deepcopy
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch

from .. import _native as N
from ..utils import editing_util


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps, scale_betas=1.):
    """Named beta schedules (reference :24-51): 'linear' (Ho et al., rescaled) and 'cosine'."""
    if schedule_name == "linear":
        scale = scale_betas * 1000 / num_diffusion_timesteps
        return np.linspace(scale * 0.0001, scale * 0.02, num_diffusion_timesteps, dtype=np.float64)
    if schedule_name == "cosine":
        return betas_for_alpha_bar(
            num_diffusion_timesteps,
            lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2)
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


def betas_for_alpha_bar(num_diffusion_timesteps, alpha_bar, max_beta=0.999):
    """beta_i = min(1 - abar((i+1)/N) / abar(i/N), max_beta)  (reference :54-71)."""
    n = num_diffusion_timesteps
    return np.array([min(1 - alpha_bar((i + 1) / n) / alpha_bar(i / n), max_beta)
                     for i in range(n)])


class ModelMeanType(enum.Enum):
    PREVIOUS_X = enum.auto()
    START_X = enum.auto()
    EPSILON = enum.auto()


class ModelVarType(enum.Enum):
    LEARNED = enum.auto()
    FIXED_SMALL = enum.auto()
    FIXED_LARGE = enum.auto()
    LEARNED_RANGE = enum.auto()


class LossType(enum.Enum):
    MSE = enum.auto()
    RESCALED_MSE = enum.auto()
    KL = enum.auto()
    RESCALED_KL = enum.auto()

    def is_vb(self):
        return self in (LossType.KL, LossType.RESCALED_KL)


@dataclass
class DiffusionConfig:
    """Same fields and defaults as the reference dataclass (:110-136); the sampler reads only
    betas / model_mean_type / model_var_type / rescale_timesteps, the rest is carried for callers."""
    betas: List
    model_mean_type: ModelMeanType = ModelMeanType.START_X
    model_var_type: ModelVarType = ModelVarType.FIXED_SMALL
    loss_type: LossType = LossType.MSE
    rescale_timesteps: bool = False
    lambda_rcxyz: float = 0.
    lambda_vel: float = 0.
    lambda_pose: float = 1.
    lambda_orient: float = 1.
    lambda_loc: float = 1.
    data_rep: str = 'rot6d'
    lambda_root_vel: float = 0.
    lambda_vel_rcxyz: float = 0.
    lambda_fc: float = 0.
    clip_range: float = None
    train_trajectory_only_xz: bool = False
    use_random_proj: bool = False
    fp16: bool = False
    traj_only: bool = False
    abs_3d: bool = False
    apply_zero_mask: bool = False
    traj_extra_weight: float = 1.
    time_weighted_loss: bool = False
    train_x0_as_eps: bool = False
    train_keypoint_mask: str = 'none'


def _unwrap_model(model):
    """(mdm, cfg_wrapper_or_None) if `model` is the native denoiser, else (None, None)."""
    from ..model.cfg_sampler import ClassifierFreeSampleModel
    from ..model.mdm import MDM
    from ..model.mdm_unet import MDM_UNET
    native = (MDM, MDM_UNET)
    if isinstance(model, ClassifierFreeSampleModel) and isinstance(model.model, native):
        return model.model, model
    if isinstance(model, native):
        return model, None
    return None, None


def _add_observations(cond, mdm, model_kwargs, B, n_feats, T):
    """MDM_UNET with keyframe conditioning consumes model_kwargs['obs_x0'] / ['obs_mask'], which
    sample/conditional_synthesis.py:159-162 passes next to 'y' (reference mdm_unet.py:766-782)."""
    if mdm is not None and getattr(mdm, 'arch', '') == 'unet' and getattr(mdm, 'keyframe_conditioned', False):
        if 'obs_x0' not in model_kwargs or 'obs_mask' not in model_kwargs:
            raise KeyError("a keyframe-conditioned MDM_UNET needs model_kwargs['obs_x0'] and ['obs_mask']")
        cond['obs_x0'] = model_kwargs['obs_x0'].reshape(B, n_feats, 1, T)
        cond['obs_mask'] = model_kwargs['obs_mask'].reshape(B, n_feats, 1, T)


class GaussianDiffusion:
    """Schedules + samplers.  See the module docstring for the split host / device."""

    def __init__(self, conf: DiffusionConfig):
        self.conf = conf
        self.model_mean_type = conf.model_mean_type
        self.model_var_type = conf.model_var_type
        self.loss_type = conf.loss_type
        self.rescale_timesteps = conf.rescale_timesteps
        self.data_rep = conf.data_rep
        self.clip_range = conf.clip_range
        if conf.data_rep != 'rot_vel' and conf.lambda_pose != 1.:
            raise ValueError('lambda_pose is relevant only when training on velocities!')

        betas = np.array(conf.betas, dtype=np.float64)
        assert betas.ndim == 1, "betas must be 1-D"
        assert (betas > 0).all() and (betas <= 1).all()
        self.betas = betas
        self.num_timesteps = int(betas.shape[0])

        alphas = 1.0 - betas
        ab = np.cumprod(alphas, axis=0)
        self.alphas_cumprod = ab
        self.alphas_cumprod_prev = np.append(1.0, ab[:-1])
        self.alphas_cumprod_next = np.append(ab[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(ab)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - ab)
        self.log_one_minus_alphas_cumprod = np.log(1.0 - ab)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / ab)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / ab - 1)
        abp = self.alphas_cumprod_prev
        self.posterior_variance = betas * (1.0 - abp) / (1.0 - ab)
        # the variance is 0 at t=0, so the log is taken of the t=1 value there
        self.posterior_log_variance_clipped = np.log(
            np.append(self.posterior_variance[1], self.posterior_variance[1:])) \
            if self.num_timesteps > 1 else np.log(np.maximum(self.posterior_variance, 1e-20))
        self.posterior_mean_coef1 = betas * np.sqrt(abp) / (1.0 - ab)
        self.posterior_mean_coef2 = (1.0 - abp) * np.sqrt(alphas) / (1.0 - ab)

        self.data_transform_fn = None
        self.data_inv_transform_fn = None
        self.data_get_mean_fn = None
        self.log_trajectory_fn = None
        # Test hook: a tensor [n_draws, B, J, F, T]; when set, x_T (if `noise` is None) and the
        # per-step draws are taken from it in loop order instead of the engine's Philox generator
        # (the shared-noise parity mode of SURVEY.md §8c).
        self.injected_noise: Optional[torch.Tensor] = None
        # Multi-GPU batch sharding (utils/dist_util.py): GLOBAL index of this rank's first sample.  The engine's
        # Philox noise is keyed by (seed, first_sample + b, step, element), so with the same torch seed on every
        # rank (utils.fixseed) a sharded run draws exactly the single-device batch's x_T and per-step noise.
        # model_kwargs['y']['first_sample'] (set by dist_util.shard_call) overrides it per call.
        self.first_sample = 0
        self._sampler_engines = {}

    # ---- schedule hand-off to the engine --------------------------------------------------------
    def _timestep_map(self):
        return list(range(self.num_timesteps))

    def _original_num_steps(self):
        return self.num_timesteps

    def _model_variance(self):
        """Reference :466-480: FIXED_SMALL -> posterior_variance, FIXED_LARGE -> append(posterior_variance[1], betas[1:])."""
        if self.model_var_type == ModelVarType.FIXED_SMALL:
            return self.posterior_variance
        if self.model_var_type == ModelVarType.FIXED_LARGE:
            return np.append(self.posterior_variance[1], self.betas[1:])
        raise NotImplementedError(f"learned variances are not supported by the sampling engine ({self.model_var_type})")

    def _model_log_variance(self):
        if self.model_var_type == ModelVarType.FIXED_SMALL:
            return self.posterior_log_variance_clipped
        if self.model_var_type == ModelVarType.FIXED_LARGE:
            return np.log(np.append(self.posterior_variance[1], self.betas[1:]))
        raise NotImplementedError("learned variances are not supported by the sampling engine "
                                  f"({self.model_var_type})")

    def _clip_x0(self, clip_denoised) -> float:
        """process_xstart (reference :489-505): START_X never clips; an EPSILON model clamps the derived x0 to
        +-clip_range for abs_3d trajectory models and is NotImplemented otherwise."""
        if not clip_denoised or self.model_mean_type == ModelMeanType.START_X:
            return 0.0
        if getattr(self.conf, 'abs_3d', False) and getattr(self.conf, 'traj_only', False):
            if not self.clip_range or self.clip_range <= 0:
                raise ValueError("clip_denoised needs conf.clip_range > 0")
            return float(self.clip_range)
        raise NotImplementedError()

    def engine_tables(self, clip_x0: float = 0.0) -> dict:
        """fp32 per-step tables for cmdi_set_schedule: float64 numpy -> .float(), like
        _extract_into_tensor (reference :2225); sigma = exp(0.5 * log_variance) in fp32 (:710)."""
        if self.model_mean_type == ModelMeanType.START_X:
            mean_type = N.CMDI_MEAN_START_X
        elif self.model_mean_type == ModelMeanType.EPSILON:
            mean_type = N.CMDI_MEAN_EPSILON
        else:
            raise NotImplementedError(self.model_mean_type)
        f32 = lambda a: np.asarray(a, dtype=np.float64).astype(np.float32)
        logvar = f32(self._model_log_variance())
        return {
            "n_steps": self.num_timesteps,
            "mean_type": mean_type,
            "post_coef1": f32(self.posterior_mean_coef1),
            "post_coef2": f32(self.posterior_mean_coef2),
            "sigma": np.exp(np.float32(0.5) * logvar).astype(np.float32),
            "sqrt_ab": f32(self.sqrt_alphas_cumprod),
            "sqrt_1mab": f32(self.sqrt_one_minus_alphas_cumprod),
            "sqrt_recip_ab": f32(self.sqrt_recip_alphas_cumprod),
            "sqrt_recipm1_ab": f32(self.sqrt_recipm1_alphas_cumprod),
            "ab": f32(self.alphas_cumprod),
            "ab_prev": f32(self.alphas_cumprod_prev),
            "timestep_map": np.asarray(self._timestep_map(), dtype=np.int64),
            "clip_x0": float(clip_x0),
        }

    @staticmethod
    def _tables_key(tables: dict):
        """Content key of a schedule hand-off: the engine lives on the (long-lived) model while diffusion objects
        come and go, so identity (id()) is not a safe cache key."""
        import hashlib
        h = hashlib.blake2b(digest_size=16)
        for name in sorted(tables):
            v = tables[name]
            h.update(name.encode())
            h.update(np.ascontiguousarray(v).tobytes() if isinstance(v, np.ndarray) else repr(v).encode())
        return h.hexdigest()

    # ---- small tensor helpers kept for API compatibility ----------------------------------------
    def _engine_for(self, model, device, batch, n_feats, n_frames, want_grad, clip_denoised=False):
        mdm, _ = _unwrap_model(model)
        if mdm is not None:
            eng = mdm.engine(device, max_batch=batch, max_frames=n_frames, want_grad=want_grad,
                             n_time_rows=self._original_num_steps())
        else:
            key = (str(device), n_feats)
            eng = self._sampler_engines.get(key)
            if eng is None or eng.max_batch < batch or eng.max_frames < n_frames:
                from ..engine import Engine
                eng = Engine(n_layers=0, d_model=0, d_ff=0, n_heads=0, n_feats=n_feats,
                             max_frames=max(n_frames, 1), max_batch=max(batch, 1), device=device)
                self._sampler_engines[key] = eng
        tables = self.engine_tables(self._clip_x0(clip_denoised))
        eng.set_schedule(tables, key=self._tables_key(tables))
        return eng

    def q_sample(self, x_start, t, noise=None):
        """x_t ~ q(x_t | x_0) (reference :311-328); all entries of t must be equal."""
        t_host = int(t.reshape(-1)[0].item())
        assert bool((t == t_host).all()), "q_sample: per-sample timesteps are not supported"
        eng = self._engine_for(None, x_start.device, x_start.shape[0], x_start.shape[1] * x_start.shape[2],
                               x_start.shape[-1], False)
        if noise is None:
            noise = eng.randn(x_start.shape, seed=_fresh_seed())
        assert noise.shape == x_start.shape
        return eng.q_sample(x_start.float().contiguous(), noise.float().contiguous(), t_host)

    def _scale_timesteps(self, t):
        if self.rescale_timesteps:
            return t.float() * (1000.0 / self.num_timesteps)
        return t

    # ---- the loop -------------------------------------------------------------------------------
    def _sample_loop_progressive(self, sampler, model, shape, noise, clip_denoised, denoised_fn,
                                 cond_fn, model_kwargs, device, progress, eta, skip_timesteps,
                                 init_image, randomize_class, fast, cond_fn_with_grad=False):
        if randomize_class:
            raise NotImplementedError("randomize_class is not supported")
        if model_kwargs is None or 'y' not in model_kwargs:
            # the reference dereferences model_kwargs['y'] unconditionally (:1280)
            raise KeyError("model_kwargs['y'] is required")
        if device is None:
            device = next(model.parameters()).device
        device = torch.device(device)
        if device.type != "cuda":
            raise N.NativeError("p_sample_loop runs on a HIP device only (no CPU path); move the "
                                f"model to cuda first (got {device})")
        assert isinstance(shape, (tuple, list)) and len(shape) == 4
        B, J, F, T = (int(v) for v in shape)
        y = model_kwargs['y']
        mdm, cfg = _unwrap_model(model)

        use_recon = bool(y.get('reconstruction_guidance', False))
        # (cond_fn differentiates through the denoiser: the engine needs its activation stash from the start, or the first
        # guided model call would rebuild it)
        eng = self._engine_for(model, device, B, J * F, T,
                               want_grad=(use_recon or cond_fn is not None) and mdm is not None,
                               clip_denoised=clip_denoised)
        cond = self._condition_from_kwargs(y, mdm, cfg, B, J * F, T, device)
        _add_observations(cond, mdm, model_kwargs, B, J * F, T)
        eng.set_condition(**cond)
        eng.clear_range()      # a chain reports its own range / timestep events only

        first = int(y.get('first_sample', self.first_sample))
        seed = _fresh_seed()
        draws = _NoiseSource(self.injected_noise, shape, device)
        if noise is not None:
            img = noise.to(device=device, dtype=torch.float32).contiguous().clone()
        elif draws.active:
            img = draws.next().clone()
        else:
            img = eng.randn(shape, seed=seed, step=-1, first_sample=first)

        if skip_timesteps and init_image is None:
            init_image = torch.zeros_like(img)
        indices = list(range(self.num_timesteps - skip_timesteps))[::-1]
        if init_image is not None:
            init_image = init_image.to(device=device, dtype=torch.float32).contiguous()
            img = eng.q_sample(init_image, img, indices[0])

        self._range_probe(eng, mdm, img, indices)
        sampler_id = N.CMDI_SAMPLER_DDIM if sampler == "ddim" else N.CMDI_SAMPLER_DDPM
        if cond_fn is not None:
            # p_sample asserts cond_fn is None (:685): the ancestral loop honours it only through p_sample_with_grad, i.e.
            # with cond_fn_with_grad=True (or the 'gmd' key, :1280-1282); the DDIM loop always takes the with-grad form (:1572)
            if sampler == "ddpm" and not (cond_fn_with_grad or 'gmd' in y):
                raise AssertionError("only support the case where cond_fn is None")
            if progress:
                from tqdm.auto import tqdm
                indices = tqdm(indices)
            for i in indices:
                nz = draws.next() if draws.active else eng.randn(shape, seed=seed, first_sample=first, step=i)
                out = self._guided_step(sampler, model, img, i, cond_fn, model_kwargs, eta, nz)
                img = out["sample"]
                yield out
            if mdm is not None and mdm._engine is not None:
                mdm._engine.check_range()
            return
        if mdm is not None and fast and not progress:
            # whole loop in one native call, nothing materialised per step
            stream = draws.take(len(indices)) if draws.active else None
            eng.sample_loop(img, indices[0], indices[-1], sampler=sampler_id, eta=eta,
                            noise_stream=stream, seed=seed, first_sample=first)
            eng.check_range()
            yield {"sample": img, "pred_xstart": None}
            return

        if progress:
            from tqdm.auto import tqdm
            indices = tqdm(indices)
        for i in indices:
            nz = draws.next() if draws.active else None
            pred = torch.empty_like(img)
            if mdm is not None:
                eng.step(img, i, sampler=sampler_id, eta=eta, noise=nz, pred_xstart=pred, seed=seed,
                         first_sample=first)
            else:
                self._generic_step(eng, model, img, i, sampler_id, eta, nz, pred, seed, model_kwargs, first)
            yield {"sample": img.clone(), "pred_xstart": pred}
        if mdm is not None:
            eng.check_range()

    RANGE_PROBE_MIN_STEPS = 50

    def _range_probe(self, eng, mdm, img, indices):
        """A long chain on the range-limited default precision (f16x3) is not started blind: two denoiser evaluations of the
        chain's first x_t — at its first and at its last timestep — are checked for the f16 range first, so weights / conditions
        that overflow systematically send the chain to the bf16x6 engine BEFORE its (up to 1000) steps are spent, not after
        (VERDICT r2 task 7; the end-of-chain check stays: an overflow that only a later x_t provokes is still caught).  Costs 2
        of >= 50 evaluations and one 4-byte read-back; skipped for short chains.  MDM_UNET: the same probe and, since round 5,
        the same fallback (its convolutions on gemm_x6: exact operands at any activation scale, as the reference's plain fp32
        U-Net, model/mdm_unet.py:561-849).  An engine PINNED to f16x3 — or MDM_UNET(attention=True), which has no wider mode — is
        probed as well: its RangeError, which names the cause, then reaches the caller before any step is spent."""
        if mdm is None or eng.precision != "f16x3" or len(indices) < self.RANGE_PROBE_MIN_STEPS:
            return
        if getattr(mdm, "_range_fallback", False):
            return
        tmap = self._timestep_map()
        for i in (indices[0], indices[-1]):
            eng.mdm_forward(img, torch.full((img.shape[0],), int(tmap[i]), dtype=torch.int64, device=img.device))
        eng.check_range()        # RangeError -> _with_range_fallback re-runs the call on bf16x6

    def _guided_step(self, sampler, model, img, i, cond_fn, model_kwargs, eta, noise):
        """One step of p_sample_with_grad (:715-800, without its GMD post-imputation) or ddim_sample_with_grad
        (:1358-1416) with a caller-supplied cond_fn(x, t, p_mean_var, **model_kwargs) -> gradient.  The model call
        (and, through torch.autograd inside cond_fn, its input-VJP) is native; the lines below restate the reference's
        elementwise fp32 arithmetic in its evaluation order."""
        y = model_kwargs['y']
        if editing_util.uses_imputation(y) or editing_util.uses_reconstruction_guidance(y):
            raise NotImplementedError("cond_fn together with imputation / reconstruction guidance is not supported")
        if 'cond_until_second_stage' in y:
            # the reference switches cond_until AND the inpainting targets of its GMD post-imputation here (:771-781)
            raise NotImplementedError("cond_until_second_stage (two-stage GMD guidance) is reference-only")
        if 'inpainting_mask' in y and 'inpainted_motion' in y:
            # the reference would run the GMD post-sampling imputation here (:800-1105), which needs data_transform_fn
            raise NotImplementedError("GMD post-sampling imputation (inpainting_mask with cond_fn) is reference-only")
        if self.model_mean_type != ModelMeanType.START_X:
            raise NotImplementedError("cond_fn guidance is implemented for START_X models")
        dev, B = img.device, img.shape[0]
        t = torch.full((B,), i, device=dev, dtype=torch.long)
        t_model = torch.as_tensor(self._timestep_map(), device=dev, dtype=torch.long)[t]
        if self.rescale_timesteps:
            t_model = t_model.float() * (1000.0 / self._original_num_steps())
        f32 = lambda table: torch.tensor(float(np.float32(np.asarray(table, dtype=np.float64)[i])), device=dev)
        nonzero = 0.0 if i == 0 else 1.0
        with torch.enable_grad():
            x = img.detach().requires_grad_()
            model_output = model(x, t_model, **model_kwargs)
            if isinstance(model_output, tuple):
                model_output = model_output[0]
            pred_xstart = model_output                                   # START_X, clip_denoised is a no-op (:489-492)
            mean = f32(self.posterior_mean_coef1) * pred_xstart + f32(self.posterior_mean_coef2) * x
            variance = f32(self._model_variance())          # the TABLE (reference :466-480), not exp(log table): FIXED_SMALL's
            log_variance = f32(self._model_log_variance())  # variance is 0 at i = 0 while its clipped log is that of i = 1
            p_mean_var = {"mean": mean, "variance": variance.expand_as(x), "log_variance": log_variance.expand_as(x),
                          "pred_xstart": pred_xstart, "model_output": model_output}
            if sampler == "ddpm":
                cond_until = y.get('cond_until', 1)
                if i >= cond_until:                                      # condition_mean_with_grad, var_scale=True
                    gradient = cond_fn(x, t, p_mean_var, **model_kwargs)
                    mean = mean.float() + variance * gradient.float()
                sample = mean + (nonzero * torch.exp(0.5 * log_variance)) * noise
                return {"sample": sample.detach(), "pred_xstart": pred_xstart.detach()}
            # ddim_sample_with_grad -> condition_score_with_grad (:636-660)
            sra, srm1a = f32(self.sqrt_recip_alphas_cumprod), f32(self.sqrt_recipm1_alphas_cumprod)
            alpha_bar, alpha_bar_prev = f32(self.alphas_cumprod), f32(self.alphas_cumprod_prev)
            eps = (sra * x - pred_xstart) / srm1a
            gradient = cond_fn(x, t, p_mean_var, **model_kwargs)
            eps = eps - (1 - alpha_bar).sqrt() * gradient
            new_xstart = (sra * x - srm1a * eps).detach()                # _predict_xstart_from_eps
        eps = (sra * x.detach() - new_xstart) / srm1a
        sigma = (eta * torch.sqrt((1 - alpha_bar_prev) / (1 - alpha_bar))) * torch.sqrt(1 - alpha_bar / alpha_bar_prev)
        mean_pred = new_xstart * torch.sqrt(alpha_bar_prev) + torch.sqrt(1 - alpha_bar_prev - sigma ** 2) * eps
        sample = mean_pred + (nonzero * sigma) * noise
        return {"sample": sample, "pred_xstart": pred_xstart.detach()}   # the UNconditioned prediction (:1413-1416)

    def _generic_step(self, eng, model, img, i, sampler_id, eta, nz, pred, seed, model_kwargs, first=0):
        """Any callable denoiser: the model runs in torch, the sampler arithmetic in the engine."""
        y = model_kwargs['y']
        B = img.shape[0]
        t = torch.full((B,), i, device=img.device, dtype=torch.long)
        t_model = torch.as_tensor(self._timestep_map(), device=img.device, dtype=torch.long)[t]
        if self.rescale_timesteps:
            t_model = t_model.float() * (1000.0 / self._original_num_steps())
        recon = bool(y.get('reconstruction_guidance', False)) and i >= int(y['stop_recguidance_at'])
        grad = None
        if recon:
            mask = (y['inpainting_mask'].to(img.device) * y['mask'].float().to(img.device)).bool()
            with torch.enable_grad():
                z = img.detach().requires_grad_(True)
                out = model(z, t_model, **model_kwargs)
                loss = ((y['inpainted_motion'].to(img.device) - out).square() * mask).sum()
                grad = torch.autograd.grad(loss, z)[0].contiguous()
            out = out.detach()
        else:
            with torch.no_grad():
                out = model(img, t_model, **model_kwargs)
        if isinstance(out, tuple):
            out = out[0]
        eng.sampler_update(img, out.float().contiguous(), i, sampler=sampler_id, eta=eta,
                           recon_grad=grad, noise=nz, pred_xstart=pred, seed=seed, first_sample=first)

    def _condition_from_kwargs(self, y, mdm, cfg, B, n_feats, T, device):
        """model_kwargs['y'] -> cmdi_condition (gates of utils/editing_util.py)."""
        cond = dict(batch=B, n_frames=T, cfg=cfg is not None)
        if mdm is not None and 'text' in mdm.cond_mode:
            if not y.get('uncond', False):
                cond['enc_text'] = mdm.text_embedding(y, B, device)
        if cfg is not None:
            cond['text_scale'] = torch.as_tensor(y['text_scale'], dtype=torch.float32).reshape(-1)
        imputate = editing_util.uses_imputation(y)
        recon = editing_util.uses_reconstruction_guidance(y)
        impute_mode = 1 if imputate else 0
        if imputate and y.get('replacement_distribution', 'conditional') == 'marginal':
            # the reference's 'marginal' branch is a no-op (:437-439) — but it is only reached when reconstruction
            # guidance is NOT active at that step: inside the guidance branch (:424) imputation happens whatever the
            # replacement distribution.  Mode 2 = impute only at steps where reconstruction guidance runs.
            impute_mode = 2 if recon else 0
            imputate = impute_mode != 0
        elif imputate and y.get('replacement_distribution', 'conditional') != 'conditional' and not recon:
            raise NotImplementedError
        if imputate or recon:
            if self.model_mean_type != ModelMeanType.START_X:
                raise AssertionError('This feature supports only X_start pred for now!')
            mask = y['inpainting_mask'].to(device)
            seq_mask = y['mask'].to(device)
            # inpainting_mask = (inpainting_mask * y['mask'].float()).bool()  (:408-409,432-433)
            cond['inpaint_mask'] = (mask.bool() & seq_mask.bool()).expand(B, n_feats, 1, T) \
                .reshape(B, n_feats, 1, T)
            cond['inpaint_motion'] = y['inpainted_motion'].to(device).float().reshape(B, n_feats, 1, T)
        cond['imputate'] = impute_mode
        cond['stop_imputation_at'] = int(y.get('stop_imputation_at', 0)) if imputate else 0
        cond['recon_guidance'] = recon
        cond['stop_recguidance_at'] = int(y.get('stop_recguidance_at', 0)) if recon else 0
        if recon:
            ws = editing_util.get_gradient_schedule(y.get('gradient_schedule'),
                                                    num_diffusion_steps=y['diffusion_steps'])
            # w_r = _extract_into_tensor(grad_ws, t).float() * reconstruction_weight  (:418-420)
            ws = np.asarray(ws, dtype=np.float64)[:self.num_timesteps].astype(np.float32)
            if ws.shape[0] != self.num_timesteps:
                raise ValueError("gradient schedule shorter than the respaced chain")
            cond['recon_w'] = ws * np.float32(y['reconstruction_weight'])
        return cond

    # ---- public samplers (reference signatures) -------------------------------------------------
    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None,
                      cond_fn=None, model_kwargs=None, device=None, progress=False,
                      skip_timesteps=0, init_image=None, randomize_class=False,
                      cond_fn_with_grad=False, dump_steps=None, const_noise=False):
        """Reference :1149-1214.  Returns the final sample, or — if `dump_steps` is given — the
        list of deep-copied ``pred_xstart`` at those loop counters."""
        if const_noise:
            raise NotImplementedError()
        return self._loop("ddpm", model, shape, noise, clip_denoised, denoised_fn, cond_fn,
                          model_kwargs, device, progress, 0.0, skip_timesteps, init_image,
                          randomize_class, dump_steps, cond_fn_with_grad)

    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True,
                                  denoised_fn=None, cond_fn=None, model_kwargs=None, device=None,
                                  progress=False, skip_timesteps=0, init_image=None,
                                  randomize_class=False, cond_fn_with_grad=False, const_noise=False):
        """Reference :1217-1297: yields {"sample", "pred_xstart"} after every step."""
        if const_noise:
            raise NotImplementedError()
        yield from self._sample_loop_progressive(
            "ddpm", model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs, device,
            progress, 0.0, skip_timesteps, init_image, randomize_class, fast=False,
            cond_fn_with_grad=cond_fn_with_grad)

    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None,
                         cond_fn=None, model_kwargs=None, device=None, progress=False, eta=0.0,
                         skip_timesteps=0, init_image=None, randomize_class=False,
                         cond_fn_with_grad=False, dump_steps=None, const_noise=False):
        """Reference :1454-1512."""
        if const_noise:
            raise NotImplementedError()
        return self._loop("ddim", model, shape, noise, clip_denoised, denoised_fn, cond_fn,
                          model_kwargs, device, progress, eta, skip_timesteps, init_image,
                          randomize_class, dump_steps)

    def ddim_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True,
                                     denoised_fn=None, cond_fn=None, model_kwargs=None, device=None,
                                     progress=False, eta=0.0, skip_timesteps=0, init_image=None,
                                     randomize_class=False, cond_fn_with_grad=False):
        """Reference :1514-1587."""
        yield from self._sample_loop_progressive(
            "ddim", model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs, device,
            progress, eta, skip_timesteps, init_image, randomize_class, fast=False)

    @staticmethod
    def _with_range_fallback(model, run):
        """The default precision (f16x3) is range-limited; if an activation leaves the f16 range, `run` is repeated — same
        torch RNG state, hence same engine seed and noise — on a bf16x6 engine (exact operands, fp32 range).  Used by the
        sampling loops AND by the single-step entry points (p_sample, ddim_sample, *_with_grad)."""
        mdm, _ = _unwrap_model(model)
        rng_state = torch.random.get_rng_state()
        try:
            return run()
        except N.RangeError:
            if mdm is None or not hasattr(mdm, "range_fallback") or not mdm.range_fallback():
                raise
            torch.random.set_rng_state(rng_state)
            return run()

    def _loop(self, sampler, model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs,
              device, progress, eta, skip_timesteps, init_image, randomize_class, dump_steps,
              cond_fn_with_grad=False):
        return self._with_range_fallback(model, lambda: self._loop_once(
            sampler, model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs, device, progress, eta,
            skip_timesteps, init_image, randomize_class, dump_steps, cond_fn_with_grad))

    def _loop_once(self, sampler, model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs,
                   device, progress, eta, skip_timesteps, init_image, randomize_class, dump_steps,
                   cond_fn_with_grad=False):
        final, dump = None, []
        gen = self._sample_loop_progressive(
            sampler, model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs, device,
            progress, eta, skip_timesteps, init_image, randomize_class, fast=dump_steps is None,
            cond_fn_with_grad=cond_fn_with_grad)
        for i, out in enumerate(gen):
            if dump_steps is not None and i in dump_steps:
                dump.append(deepcopy(out["pred_xstart"]))
            final = out
        if dump_steps is not None:
            return dump
        return final["sample"]

    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None,
                 model_kwargs=None, const_noise=False, previous_xstart=None):
        """One ancestral step (reference :656-713) on a fresh copy of x."""
        return self._with_range_fallback(model, lambda: self._single_step(
            "ddpm", model, x, t, cond_fn, model_kwargs, const_noise, 0.0, clip_denoised))

    def ddim_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None,
                    model_kwargs=None, eta=0.0, previous_xstart=None):
        """One DDIM step (reference :1300-1356)."""
        return self._with_range_fallback(model, lambda: self._single_step(
            "ddim", model, x, t, cond_fn, model_kwargs, False, eta, clip_denoised))

    def p_sample_with_grad(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None,
                           model_kwargs=None, const_noise=False, previous_xstart=None):
        """Reference :715-800 (cond_fn(x, t, p_mean_var, **model_kwargs) -> gradient, added as variance * gradient)."""
        return self._with_range_fallback(model, lambda: self._single_guided("ddpm", model, x, t, cond_fn, model_kwargs, 0.0))

    def ddim_sample_with_grad(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None,
                              model_kwargs=None, eta=0.0, previous_xstart=None):
        """Reference :1358-1416 (condition_score_with_grad)."""
        return self._with_range_fallback(model, lambda: self._single_guided("ddim", model, x, t, cond_fn, model_kwargs, eta))

    def _single_guided(self, sampler, model, x, t, cond_fn, model_kwargs, eta):
        i = int(t.reshape(-1)[0].item())
        assert bool((t == i).all()), "all samples of a batch share the denoising step"
        if cond_fn is None:
            return self._single_step(sampler, model, x, t, None, model_kwargs, False, eta)
        draws = _NoiseSource(self.injected_noise, x.shape, x.device)
        if draws.active:
            nz = draws.next()
        else:
            eng = self._engine_for(None, x.device, x.shape[0], x.shape[1] * x.shape[2], x.shape[-1], False)
            nz = eng.randn(x.shape, seed=_fresh_seed(), step=i,
                           first_sample=int(model_kwargs['y'].get('first_sample', self.first_sample)))
        mdm, _ = _unwrap_model(model)
        if mdm is not None and mdm._engine is not None:
            mdm._engine.clear_range()
        out = self._guided_step(sampler, model, x.detach().float(), i, cond_fn, model_kwargs, eta, nz)
        if mdm is not None and mdm._engine is not None:
            mdm._engine.check_range()     # (the step synchronised already: t.item() above)
        return out

    def _single_step(self, sampler, model, x, t, cond_fn, model_kwargs, const_noise, eta, clip_denoised=False):
        assert cond_fn is None, "only support the case where cond_fn is None"
        if const_noise:
            raise NotImplementedError()
        i = int(t.reshape(-1)[0].item())
        assert bool((t == i).all()), "all samples of a batch share the denoising step"
        B, J, F, T = x.shape
        y = model_kwargs['y']
        mdm, cfg = _unwrap_model(model)
        use_recon = bool(y.get('reconstruction_guidance', False))
        eng = self._engine_for(model, x.device, B, J * F, T, want_grad=use_recon and mdm is not None,
                               clip_denoised=clip_denoised)
        cond = self._condition_from_kwargs(y, mdm, cfg, B, J * F, T, x.device)
        _add_observations(cond, mdm, model_kwargs, B, J * F, T)
        eng.set_condition(**cond)
        eng.clear_range()
        first = int(y.get('first_sample', self.first_sample))
        out = x.detach().float().contiguous().clone()
        pred = torch.empty_like(out)
        sid = N.CMDI_SAMPLER_DDIM if sampler == "ddim" else N.CMDI_SAMPLER_DDPM
        draws = _NoiseSource(self.injected_noise, x.shape, x.device)
        nz = draws.next() if draws.active else None
        if mdm is not None:
            eng.step(out, i, sampler=sid, eta=eta, noise=nz, pred_xstart=pred, seed=_fresh_seed(), first_sample=first)
        else:
            self._generic_step(eng, model, out, i, sid, eta, nz, pred, _fresh_seed(), model_kwargs, first)
        if mdm is not None:
            eng.check_range()      # one 4-byte read-back; a public single step must not hand back clamped / overflowed values
        return {"sample": out, "pred_xstart": pred}


class _NoiseSource:
    """Consumes GaussianDiffusion.injected_noise in order."""

    def __init__(self, stream, shape, device):
        self.stream = stream
        self.pos = 0
        self.active = stream is not None
        if self.active:
            if tuple(stream.shape[1:]) != tuple(shape):
                raise ValueError(f"injected_noise must be [n, {tuple(shape)}], got {tuple(stream.shape)}")
            self.stream = stream.to(device=device, dtype=torch.float32).contiguous()

    def next(self):
        if self.pos >= self.stream.shape[0]:
            raise IndexError("injected_noise exhausted")
        self.pos += 1
        return self.stream[self.pos - 1]

    def take(self, n):
        if self.pos + n > self.stream.shape[0]:
            raise IndexError("injected_noise exhausted")
        out = self.stream[self.pos:self.pos + n]
        self.pos += n
        return out


def _fresh_seed() -> int:
    """Seed of the engine's Philox stream, drawn from torch's global generator so that
    utils.fixseed / torch.manual_seed keep making runs reproducible."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


def _extract_into_tensor(arr, timesteps, broadcast_shape):
    """Reference :2215-2229 (kept for callers that import it)."""
    res = torch.from_numpy(np.asarray(arr))[timesteps.cpu()].float().to(timesteps.device)
    while len(res.shape) < len(broadcast_shape):
        res = res[..., None]
    return res.expand(broadcast_shape)
