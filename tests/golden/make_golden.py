"""Generate tests/golden/*.npz by running the REAL reference (/root/reference) on CPU.

    python tests/golden/make_golden.py            # writes the fixtures next to this file

The reference ships no tests or golden vectors for the sampling path, so these
fixtures — outputs of the reference's own code on seeded synthetic inputs — are what pins the numpy
oracle (oracle/) and, through it and directly, the HIP engine.  Inputs are NOT stored: they are
regenerated from seeds by tests/golden/cases.py (shared with the tests); each fixture carries a
fingerprint of its inputs so that a drifted generator is detected instead of silently compared.
Runs only where /root/reference exists (the build container); takes ~2 minutes on 8 cores.
"""
from __future__ import annotations

import os
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(HERE))

import cases  # noqa: E402
from oracle import ref_shims, weights  # noqa: E402


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def save(name, **arrays):
    path = HERE / f"{name}.npz"
    np.savez_compressed(path, **arrays)
    print(f"wrote {path.name}: {os.path.getsize(path) / 1024:.0f} KiB "
          f"({', '.join(f'{k}{tuple(np.shape(v))}' for k, v in arrays.items())})")


def schedules(ref):
    out = {}
    for tag, name, resp in (("cos1000", "cosine", None), ("lin1000", "linear", None),
                            ("cos_ddim100", "cosine", "ddim100"), ("cos_10", "cosine", [10]),
                            ("cos_ddim10", "cosine", "ddim10")):
        betas = ref.gd.get_named_beta_schedule(name, 1000)
        use = ref.respace.space_timesteps(1000, resp if resp is not None else [1000])
        d = ref.respace.SpacedDiffusion(use_timesteps=use, conf=ref.gd.DiffusionConfig(betas=betas))
        for attr in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                     "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
                     "sqrt_recipm1_alphas_cumprod", "posterior_variance",
                     "posterior_log_variance_clipped", "posterior_mean_coef1",
                     "posterior_mean_coef2"):
            out[f"{tag}.{attr}"] = np.asarray(getattr(d, attr), dtype=np.float64)
        out[f"{tag}.timestep_map"] = np.asarray(d.timestep_map, dtype=np.int64)
    for sched in (None, 'first-half', 'last-half', 'exponential', 'sigmoid', 'half-sigmoid'):
        out[f"grad_ws.{sched}"] = np.asarray(
            ref.editing.get_gradient_schedule(sched, num_diffusion_steps=1000), dtype=np.float64)
    save("schedules", **out)


def build(ref, case):
    sd = weights.to_torch(weights.make_state_dict(case["weight_seed"], text=case["text"]))
    args = ref_shims.default_args(unconstrained=not case["text"])
    model, _ = ref_shims.make_reference_model(ref, args, sd, cfg=case.get("cfg", False))
    betas = ref.gd.get_named_beta_schedule("cosine", 1000)
    use = ref.respace.space_timesteps(1000, case.get("respacing") or [1000])
    diffusion = ref.respace.SpacedDiffusion(
        use_timesteps=use, conf=ref.gd.DiffusionConfig(betas=betas))
    return model, diffusion


def forward_cases(ref):
    for name in ("fwd_uncond", "fwd_text"):
        case = cases.CASES[name]
        inp = cases.make_inputs(case)
        model, _ = build(ref, dict(case, cfg=False))
        out = {"fingerprint": cases.fingerprint(inp)}
        x, tt = t(inp["x"]), t(inp["t"])
        with torch.no_grad():
            if not case["text"]:
                out["out"] = model(x, tt, y={}).numpy()
            else:
                ref_shims.set_text_embedding(t(inp["enc_text"]))
                y = {"text": ["a"] * x.shape[0]}
                out["out_cond"] = model(x, tt, y=y).numpy()
                out["out_uncond"] = model(x, tt, y=dict(y, uncond=True)).numpy()
                model.keyframe_conditioned = False
                wrapped = ref.cfg.ClassifierFreeSampleModel(model)
                out["out_cfg"] = wrapped(x, tt, y=dict(y, text_scale=t(inp["text_scale"]))).numpy()
        save(name, **out)


def vjp_case(ref):
    case = cases.CASES["vjp_text_cfg"]
    inp = cases.make_inputs(case)
    model, _ = build(ref, case)
    ref_shims.set_text_embedding(t(inp["enc_text"]))
    y = {"text": ["a"] * inp["x"].shape[0], "text_scale": t(inp["text_scale"])}
    z = t(inp["x"]).requires_grad_(True)
    out = model(z, t(inp["t"]), y=y)
    (gx,) = torch.autograd.grad((out * t(inp["gout"])).sum(), z)
    save("vjp_text_cfg", fingerprint=cases.fingerprint(inp), out=out.detach().numpy(),
         gx=gx.numpy())


def chain_cases(ref):
    for name in ("chain_uncond_ddpm", "chain_edit_recon", "chain_impute_only", "chain_ddim_eta0",
                 "chain_ddim_eta05", "chain_skip_init", "chain_marginal_recon", "chain_condfn_ddpm",
                 "chain_condfn_ddim"):
        case = cases.CASES[name]
        inp = cases.make_inputs(case)
        model, diffusion = build(ref, case)
        B = inp["x_T"].shape[0]
        y = {"mask": t(inp["len_mask"]), "lengths": t(inp["lengths"])}
        if case["text"]:
            ref_shims.set_text_embedding(t(inp["enc_text"]))
            y.update(text=["a"] * B, text_scale=t(inp["text_scale"]))
        if case.get("edit"):
            # the fixture's mask comes from the reference's own generator; cases.py restates it
            ref_mask = ref.editing.get_keyframes_mask(
                data=t(inp["x0"]), lengths=t(inp["lengths"]), edit_mode='benchmark_sparse',
                trans_length=case["trans_length"], feature_mode='pos_rot_vel')
            assert np.array_equal(ref_mask.numpy(), inp["inpaint_mask"]), "keyframe mask restatement drifted"
            y.update(inpainting_mask=t(inp["inpaint_mask"]), inpainted_motion=t(inp["x0"]),
                     imputate=case["imputate"], stop_imputation_at=case["stop_imputation_at"],
                     replacement_distribution=case.get("replacement", "conditional"),
                     reconstruction_guidance=case["recon"],
                     reconstruction_weight=case["recon_weight"],
                     gradient_schedule=case["grad_schedule"], diffusion_steps=1000,
                     stop_recguidance_at=case["stop_recguidance_at"])
        loop = diffusion.ddim_sample_loop if case["sampler"] == "ddim" else diffusion.p_sample_loop
        kw = dict(noise=t(inp["x_T"]), clip_denoised=False, model_kwargs={"y": y},
                  skip_timesteps=case.get("skip", 0), device=torch.device("cpu"),
                  init_image=t(inp["init_image"]) if "init_image" in inp else None)
        if case.get("cond_fn"):
            kw.update(cond_fn=cases.make_cond_fn(t(inp["x0"]), t(inp["inpaint_mask"] & inp["len_mask"]).float(),
                                                 case["cond_weight"]), cond_fn_with_grad=True)
        if case["sampler"] == "ddim":
            kw["eta"] = case["eta"]
        n_steps = diffusion.num_timesteps - case.get("skip", 0)
        stream = [t(inp["noise"][k]) for k in range(n_steps)]
        with ref_shims.injected_noise(stream):
            final = loop(model, inp["x_T"].shape, **kw)
        with ref_shims.injected_noise(stream):
            dumps = loop(model, inp["x_T"].shape, dump_steps=list(cases.DUMP_STEPS), **kw)
        save(name, fingerprint=cases.fingerprint(inp), final=final.detach().numpy(),
             pred_xstart=np.stack([d.detach().numpy() for d in dumps]))


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    ref = ref_shims.import_reference()
    schedules(ref)
    forward_cases(ref)
    vjp_case(ref)
    chain_cases(ref)


if __name__ == "__main__":
    main()
