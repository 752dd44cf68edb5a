"""numpy restatement of the CondMDI sampler — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows the reference file:line by file:line:
    schedules        diffusion/gaussian_diffusion.py:24-71 (named betas), :184-217 (tables)
    respacing        diffusion/respace.py:9-62 (space_timesteps), :74-91 (SpacedDiffusion betas/map)
    fp32 extraction  diffusion/gaussian_diffusion.py:2215-2229 (float64 table -> [t] -> .float())
    p_mean_variance  :352-534 incl. reconstruction guidance :405-425 and imputation :427-435
    p_sample         :656-713          ddim_sample :1300-1356 / 1358-1416
    loops            :1217-1297, :1514-1587 (x_T, init_image/skip_timesteps via q_sample :311-328)
    guidance sched.  utils/editing_util.py:299-322 ; gates :325-346
    CFG              model/cfg_sampler.py:25-35 (through oracle/mdm_oracle.py)
All elementwise math is numpy fp32 with the reference's evaluation order (separate roundings).
Also restates the engine's counter-based generator (Philox4x32-10, Salmon et al. SC'11 + Box-Muller)
so that its integer stream can be checked bit-for-bit.
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32


# ---- schedules ---------------------------------------------------------------------------------
def named_betas(name: str, n: int, scale_betas: float = 1.0) -> np.ndarray:
    if name == "linear":
        scale = scale_betas * 1000 / n
        return np.linspace(scale * 0.0001, scale * 0.02, n, dtype=np.float64)
    if name == "cosine":
        abar = lambda u: math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2
        return np.array([min(1 - abar((i + 1) / n) / abar(i / n), 0.999) for i in range(n)])
    raise NotImplementedError(name)


def space_timesteps(num_timesteps: int, section_counts) -> set:
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[4:])
            for i in range(1, num_timesteps):
                if len(range(0, num_timesteps, i)) == want:
                    return set(range(0, num_timesteps, i))
            raise ValueError("no integer stride")
        section_counts = [int(v) for v in section_counts.split(",")]
    size_per, extra = divmod(num_timesteps, len(section_counts))
    start, steps = 0, []
    for i, cnt in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < cnt:
            raise ValueError("section too small")
        stride = 1 if cnt <= 1 else (size - 1) / (cnt - 1)
        cur = 0.0
        for _ in range(cnt):
            steps.append(start + round(cur))
            cur += stride
        start += size
    return set(steps)


class Schedule:
    """float64 tables of GaussianDiffusion.__init__ for (optionally respaced) betas."""

    def __init__(self, betas, use_timesteps=None):
        betas = np.asarray(betas, dtype=np.float64)
        self.original_num_steps = len(betas)
        self.timestep_map = list(range(len(betas)))
        if use_timesteps is not None:
            base_ab = np.cumprod(1.0 - betas)
            new_betas, last, tmap = [], 1.0, []
            for i, ab in enumerate(base_ab):
                if i in use_timesteps:
                    new_betas.append(1 - ab / last)
                    last = ab
                    tmap.append(i)
            betas, self.timestep_map = np.array(new_betas), tmap
        self.betas = betas
        self.n = len(betas)
        alphas = 1.0 - betas
        self.ab = np.cumprod(alphas)
        self.ab_prev = np.append(1.0, self.ab[:-1])
        self.sqrt_ab = np.sqrt(self.ab)
        self.sqrt_1mab = np.sqrt(1.0 - self.ab)
        self.sqrt_recip_ab = np.sqrt(1.0 / self.ab)
        self.sqrt_recipm1_ab = np.sqrt(1.0 / self.ab - 1)
        self.post_var = betas * (1.0 - self.ab_prev) / (1.0 - self.ab)
        self.post_logvar_clipped = np.log(np.append(self.post_var[1], self.post_var[1:]))
        self.coef1 = betas * np.sqrt(self.ab_prev) / (1.0 - self.ab)
        self.coef2 = (1.0 - self.ab_prev) * np.sqrt(alphas) / (1.0 - self.ab)

    def f32(self, table, i) -> np.float32:
        """_extract_into_tensor(...): index the float64 table, then cast."""
        return F32(np.asarray(table, dtype=np.float64)[i])


def gradient_schedule(name, n, scale=.05) -> np.ndarray:
    if name is None:
        return np.ones(n)
    if name == 'first-half':
        return np.concatenate((np.ones(n // 2), np.zeros(n - n // 2)))
    if name == 'last-half':
        return np.concatenate((np.zeros(n // 2), np.ones(n // 2)))
    if name == 'exponential':
        return np.exp(-scale * np.arange(n)[::-1])
    if name == 'sigmoid':
        s = scale / 5
        return 1 / (1 + np.exp(s * (-np.arange(n) + n / 2)))
    if name == 'half-sigmoid':
        s = scale / 5
        return 1 / (1 + np.exp(s * (-np.arange(n))))
    raise NotImplementedError(name)


# ---- one denoising step given the (CFG-combined) model output ----------------------------------
def step_update(sch: Schedule, i: int, x, hat, noise, *, sampler="ddpm", eta=0.0, mean_eps=False,
                mask=None, inpaint=None, impute=False, recon=False, grad=None, recon_w=None, clip=0.0):
    """Returns (x_{i-1}, pred_xstart).  `hat` is the model output; `mask` is already AND-ed with
    y['mask']; `grad` is autograd.grad(loss, z)[0] (unmasked); recon_w = grad_ws[i]*weight (fp32)."""
    x = np.asarray(x, dtype=F32)
    hat = np.asarray(hat, dtype=F32)
    noise = np.asarray(noise, dtype=F32)
    if mean_eps:
        # _predict_xstart_from_eps (:536-541), then process_xstart's clamp (:497-499) when clip_denoised applies
        x0 = (sch.f32(sch.sqrt_recip_ab, i) * x) - (sch.f32(sch.sqrt_recipm1_ab, i) * hat)
        if clip > 0:
            x0 = np.clip(x0, F32(-clip), F32(clip))
    else:
        x0 = hat
    if recon:
        m = np.asarray(mask, dtype=bool)
        cond_grad = (np.asarray(grad, dtype=F32) * (~m).astype(F32)).astype(F32)
        w_r = F32(recon_w)
        tilde = hat - ((w_r * sch.f32(sch.sqrt_ab, i)) / F32(2)) * cond_grad
        other = np.asarray(inpaint, dtype=F32) if impute else hat
        x0 = (tilde * (~m)) + (other * m)
    elif impute:
        m = np.asarray(mask, dtype=bool)
        x0 = (hat * (~m)) + (np.asarray(inpaint, dtype=F32) * m)
    x0 = x0.astype(F32)
    nz = F32(0.0 if i == 0 else 1.0)
    if sampler == "ddpm":
        mean = (sch.f32(sch.coef1, i) * x0) + (sch.f32(sch.coef2, i) * x)
        sigma = np.exp(F32(0.5) * sch.f32(sch.post_logvar_clipped, i))
        out = mean + ((nz * sigma) * noise)
    else:
        eps = ((sch.f32(sch.sqrt_recip_ab, i) * x) - x0) / sch.f32(sch.sqrt_recipm1_ab, i)
        ab, abp = sch.f32(sch.ab, i), sch.f32(sch.ab_prev, i)
        sigma = (F32(eta) * np.sqrt((F32(1) - abp) / (F32(1) - ab))) * np.sqrt(F32(1) - ab / abp)
        mean = (x0 * np.sqrt(abp)) + (np.sqrt((F32(1) - abp) - sigma * sigma) * eps)
        out = mean + ((nz * sigma) * noise)
    return out.astype(F32), x0


def recon_loss_grad_seed(hat, mask, inpaint):
    """d/d hat of sum(mask * (inpaint - hat)^2) = 2 * (hat - inpaint) * mask."""
    return (F32(2.0) * (np.asarray(hat, F32) - np.asarray(inpaint, F32))
            * np.asarray(mask, bool)).astype(F32)


def sample_loop(sch: Schedule, model, x_T, noise_stream, *, sampler="ddpm", eta=0.0,
                enc_text=None, text_scale=None, cfg=False, mask=None, inpaint=None,
                imputate=False, stop_imputation_at=0, recon_guidance=False, stop_recguidance_at=0,
                recon_weight=0.0, grad_schedule=None, diffusion_steps=1000, first_step=None,
                last_step=0, collect=False, mean_eps=False, clip=0.0, collect_samples=False, marginal=False):
    """p_sample_loop / ddim_sample_loop with injected noise.  `model` is an MDMOracle;
    noise_stream[k] is the k-th th.randn_like draw (loop order)."""
    x = np.asarray(x_T, dtype=F32)
    B = x.shape[0]
    first = sch.n - 1 if first_step is None else first_step
    grad_ws = gradient_schedule(grad_schedule, diffusion_steps) if recon_guidance else None
    preds = []
    for k, i in enumerate(range(first, last_step - 1, -1)):
        t = np.full((B,), sch.timestep_map[i], dtype=np.int64)
        if cfg:
            hat, _, _ = model.forward_cfg(x, t, enc_text, text_scale)
        else:
            hat = model.forward(x, t, enc_text)
        do_rec = recon_guidance and i >= stop_recguidance_at
        # replacement_distribution 'marginal': the plain imputation branch is a no-op (:437-439), the guidance branch
        # imputes whatever the distribution (:424)
        do_imp = imputate and i >= stop_imputation_at and (do_rec or not marginal)
        grad, w = None, None
        if do_rec:
            seed = recon_loss_grad_seed(hat, mask, inpaint)
            grad = model.vjp_cfg(x, t, seed, enc_text, text_scale) if cfg else \
                model.vjp(x, t, seed, enc_text)
            w = F32(grad_ws[i]) * F32(recon_weight)
        x, x0 = step_update(sch, i, x, hat, noise_stream[k], sampler=sampler, eta=eta, mask=mask,
                            inpaint=inpaint, impute=do_imp, recon=do_rec, grad=grad, recon_w=w,
                            mean_eps=mean_eps, clip=clip)
        if collect:
            preds.append(x if collect_samples else x0)
    return (x, preds) if collect else x


def q_sample(sch: Schedule, i, x0, noise):
    return ((sch.f32(sch.sqrt_ab, i) * np.asarray(x0, F32))
            + (sch.f32(sch.sqrt_1mab, i) * np.asarray(noise, F32))).astype(F32)


# ---- Philox4x32-10 + Box-Muller (the engine's generator) -----------------------------------------
_M0, _M1, _W0, _W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def philox4x32_10(counter, key):
    """Vectorised over leading dims: counter [..., 4], key [..., 2] (uint32) -> [..., 4] uint32."""
    c = np.array(counter, dtype=np.uint64) & 0xFFFFFFFF
    k = np.array(key, dtype=np.uint64) & 0xFFFFFFFF
    c0, c1, c2, c3 = (c[..., j].copy() for j in range(4))
    k0, k1 = k[..., 0].copy(), k[..., 1].copy()
    for _ in range(10):
        p0, p1 = _M0 * c0, _M1 * c2
        n0 = ((p1 >> 32) ^ c1 ^ k0) & 0xFFFFFFFF
        n1 = p1 & 0xFFFFFFFF
        n2 = ((p0 >> 32) ^ c3 ^ k1) & 0xFFFFFFFF
        n3 = p0 & 0xFFFFFFFF
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def engine_uint32(batch, per_sample, seed, first_sample=0, step=-1):
    """Raw Philox words the engine draws for a [batch, per_sample] tensor: word j of quad q of
    sample b = philox((q, step+1, sample_lo, sample_hi), (seed_lo, seed_hi))[j]."""
    quads = (per_sample + 3) // 4
    ctr = np.zeros((batch, quads, 4), dtype=np.uint64)
    ctr[..., 0] = np.arange(quads, dtype=np.uint64)[None, :]
    ctr[..., 1] = (step + 1) & 0xFFFFFFFF
    samples = np.arange(batch, dtype=np.uint64) + np.uint64(first_sample)
    ctr[..., 2] = (samples & np.uint64(0xFFFFFFFF))[:, None]
    ctr[..., 3] = (samples >> np.uint64(32))[:, None]
    key = np.zeros((batch, quads, 2), dtype=np.uint64)
    key[..., 0] = seed & 0xFFFFFFFF
    key[..., 1] = (seed >> 32) & 0xFFFFFFFF
    return philox4x32_10(ctr, key)


def engine_randn(batch, per_sample, seed, first_sample=0, step=-1):
    """fp32 N(0,1) values of cmdi_randn / the in-kernel draw (Box-Muller on 24-bit uniforms)."""
    w = engine_uint32(batch, per_sample, seed, first_sample, step)
    k24 = F32(2.0 ** -24)
    u = ((w >> np.uint32(8)).astype(F32) + F32(0.5)) * k24
    out = np.empty(w.shape, dtype=F32)
    for p in range(2):
        r = np.sqrt(F32(-2.0) * np.log(u[..., 2 * p]))
        th = F32(6.283185307179586) * u[..., 2 * p + 1]
        out[..., 2 * p] = r * np.cos(th)
        out[..., 2 * p + 1] = r * np.sin(th)
    return out.reshape(batch, -1)[:, :per_sample]
