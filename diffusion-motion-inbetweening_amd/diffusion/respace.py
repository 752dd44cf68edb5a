"""Timestep respacing, mirror of the reference's ``diffusion/respace.py``.

``space_timesteps`` (:9-62) picks the retained original timesteps ('ddimN' = fixed integer stride,
otherwise per-section fractional strides); ``SpacedDiffusion`` (:65-118) rebuilds the betas of the
shortened chain from ratios of the base chain's alpha-bar and remembers ``timestep_map`` so that the
denoiser is always queried at ORIGINAL timesteps (the job of ``_WrappedModel`` :121-133 — done here
by handing ``timestep_map`` to the engine, which indexes its time-embedding table with it).
"""
from __future__ import annotations

from copy import deepcopy

import numpy as np

from .gaussian_diffusion import DiffusionConfig, GaussianDiffusion


def space_timesteps(num_timesteps, section_counts):
    """Set of original timesteps to keep.

    "ddimN": the unique integer stride s with len(range(0, num_timesteps, s)) == N.
    "a,b,c" / [a, b, c]: split the chain into equal sections and take a, b, c evenly spaced
    (first and last included) steps from them.
    """
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[len("ddim"):])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    n_sections = len(section_counts)
    base, extra = divmod(num_timesteps, n_sections)
    kept, start = [], 0
    for idx, count in enumerate(section_counts):
        size = base + (1 if idx < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        pos = 0.0
        for _ in range(count):
            kept.append(start + round(pos))
            pos += stride
        start += size
    return set(kept)


class SpacedDiffusion(GaussianDiffusion):
    """A diffusion process that visits only `use_timesteps` of a base process."""

    def __init__(self, use_timesteps, conf: DiffusionConfig):
        self.use_timesteps = set(use_timesteps)
        self.original_num_steps = len(conf.betas)
        base = GaussianDiffusion(conf)
        self.timestep_map = []
        new_betas, last_ab = [], 1.0
        for t, ab in enumerate(base.alphas_cumprod):
            if t in self.use_timesteps:
                new_betas.append(1 - ab / last_ab)
                last_ab = ab
                self.timestep_map.append(t)
        new_conf = deepcopy(conf)
        new_conf.betas = np.array(new_betas)
        super().__init__(new_conf)

    def _timestep_map(self):
        return self.timestep_map

    def _original_num_steps(self):
        return self.original_num_steps

    def _scale_timesteps(self, t):
        return t  # the mapping to original timesteps happens inside the engine

    def _wrap_model(self, model):
        if isinstance(model, _WrappedModel):
            return model
        return _WrappedModel(model, self.timestep_map, self.rescale_timesteps, self.original_num_steps)


class _WrappedModel:
    """Callable view of a model at respaced timesteps (for callers that want the torch path)."""

    def __init__(self, model, timestep_map, rescale_timesteps, original_num_steps):
        self.model = model
        self.timestep_map = timestep_map
        self.rescale_timesteps = rescale_timesteps
        self.original_num_steps = original_num_steps

    def __call__(self, x, ts, **kwargs):
        import torch
        new_ts = torch.as_tensor(self.timestep_map, device=ts.device, dtype=ts.dtype)[ts]
        if self.rescale_timesteps:
            new_ts = new_ts.float() * (1000.0 / self.original_num_steps)
        return self.model(x, new_ts, **kwargs)
