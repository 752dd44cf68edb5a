"""Golden vectors of the MDM_UNET denoiser, produced by the REAL reference on CPU (model/mdm_unet.py:561-849,
TemporalUnet :214-358, through model/cfg_sampler.py for the guided output).   python tests/golden/make_golden_unet.py
Weights: oracle.weights.fill_like over the reference module's own state-dict shapes (the tests rebuild the same
values from the shapes of OUR module and assert that names and shapes coincide)."""
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(HERE))
import cases  # noqa: E402
from oracle import ref_shims, weights  # noqa: E402

ref = ref_shims.import_reference()
import model.mdm as ref_mdm  # noqa: E402
import model.mdm_unet as ref_unet  # noqa: E402
ref_unet.Rotation2xyz = ref_mdm.Rotation2xyz   # identity shim (SMPL files absent)

case = cases.UNET_CASE
inp = cases.make_unet_inputs()
args = ref_shims.default_args(arch='unet', keyframe_conditioned=True, abs_3d=True, latent_dim=512,
                              dim_mults=case["dim_mults"], cond_mask_prob=0.1)
model, _ = ref.model_util.create_model_and_diffusion(args, SimpleNamespace(dataset=SimpleNamespace()))
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.startswith("clip_model.")}
sd = weights.fill_like(shapes, case["weight_seed"])
missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
assert not unexpected and all(k.startswith("clip_model.") or k.endswith(".pe") for k in missing), (missing, unexpected)
model.eval()

t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
x, ts = t(inp["x"]), t(inp["t"])
obs, m = t(inp["obs_x0"]), t(inp["obs_mask"])
ref_shims.set_text_embedding(t(inp["enc_text"]))
y = {"text": ["a"] * case["B"], "mask": torch.ones(case["B"], 1, 1, case["T"], dtype=torch.bool)}
with torch.no_grad():
    oc = model(x, ts, y=dict(y), obs_x0=obs, obs_mask=m)
    ou = model(x, ts, y=dict(y, uncond=True), obs_x0=obs, obs_mask=m)
    wrapped = ref.cfg.ClassifierFreeSampleModel(model)
    cfg = wrapped(x, ts, y=dict(y, text_scale=t(inp["text_scale"])), obs_x0=obs, obs_mask=m)
out = {"out_cond": oc.numpy(), "out_uncond": ou.numpy(), "out_cfg": cfg.numpy(), "fingerprint": cases.fingerprint(inp),
       "names": np.asarray(sorted(shapes))}
np.savez_compressed(HERE / "unet_fwd.npz", **out)
print({k: getattr(v, "shape", None) for k, v in out.items()}, float(np.abs(out["out_cond"]).mean()))


# ---- a short sampling chain through the reference's own p_sample_loop (the conditional_synthesis.py call) ----
cc = cases.UNET_CHAIN
ci = cases.make_unet_chain_inputs()
betas = ref.gd.get_named_beta_schedule("cosine", 1000)
diffusion = ref.respace.SpacedDiffusion(use_timesteps=ref.respace.space_timesteps(1000, cc["respacing"]),
                                        conf=ref.gd.DiffusionConfig(betas=betas))
ref_shims.set_text_embedding(t(ci["enc_text"]))
obs_mask = t(ci["obs_mask"])
y = {"mask": t(ci["len_mask"]), "lengths": t(ci["lengths"]), "text": ["a"] * cc["B"], "text_scale": t(ci["text_scale"]),
     "inpainting_mask": obs_mask, "inpainted_motion": t(ci["x0"]), "imputate": True,
     "stop_imputation_at": cc["stop_imputation_at"], "replacement_distribution": "conditional",
     "reconstruction_guidance": False, "diffusion_steps": 1000}
kw = dict(noise=t(ci["x_T"]), clip_denoised=False, device=torch.device("cpu"),
          model_kwargs={"y": y, "obs_x0": t(ci["x0"]), "obs_mask": obs_mask})
stream = [t(ci["noise"][k]) for k in range(diffusion.num_timesteps)]
with ref_shims.injected_noise(stream):
    final = diffusion.p_sample_loop(wrapped, ci["x_T"].shape, **kw)
np.savez_compressed(HERE / "unet_chain.npz", final=final.detach().numpy(), fingerprint=cases.fingerprint(ci))
print("chain", final.shape, float(final.abs().mean()))


# ---- input-VJP through ClassifierFreeSampleModel(MDM_UNET): what torch.autograd.grad(loss, z) computes at
# diffusion/gaussian_diffusion.py:411-416 -----------------------------------------------------------------------
vc = cases.UNET_VJP_CASE
vi = cases.make_unet_vjp_inputs()
ref_shims.set_text_embedding(t(vi["enc_text"]))
z = t(vi["x"]).clone().requires_grad_(True)
yv = {"text": ["a"] * vc["B"], "mask": torch.ones(vc["B"], 1, 1, vc["T"], dtype=torch.bool), "text_scale": t(vi["text_scale"])}
with torch.enable_grad():
    outv = wrapped(z, t(vi["t"]), y=yv, obs_x0=t(vi["obs_x0"]), obs_mask=t(vi["obs_mask"]))
    gxv, = torch.autograd.grad((outv * t(vi["gout"])).sum(), z)
np.savez_compressed(HERE / "unet_vjp.npz", out=outv.detach().numpy(), gx=gxv.numpy(), fingerprint=cases.fingerprint(vi))
print("vjp", float(gxv.abs().mean()), float(gxv[t(vi["obs_mask"])].abs().max()))

# ---- the same chain with reconstruction guidance through the UNET (imputation + guidance, ragged lengths) -----
rc = cases.UNET_RECON_CHAIN
ri = cases.make_unet_chain_inputs(rc)
ref_shims.set_text_embedding(t(ri["enc_text"]))
obs_mask_r = t(ri["obs_mask"])
yr = {"mask": t(ri["len_mask"]), "lengths": t(ri["lengths"]), "text": ["a"] * rc["B"], "text_scale": t(ri["text_scale"]),
      "inpainting_mask": obs_mask_r, "inpainted_motion": t(ri["x0"]), "imputate": True,
      "stop_imputation_at": rc["stop_imputation_at"], "replacement_distribution": "conditional",
      "reconstruction_guidance": True, "reconstruction_weight": rc["recon_weight"], "gradient_schedule": None,
      "stop_recguidance_at": rc["stop_recguidance_at"], "diffusion_steps": 1000}
kwr = dict(noise=t(ri["x_T"]), clip_denoised=False, device=torch.device("cpu"),
           model_kwargs={"y": yr, "obs_x0": t(ri["x0"]), "obs_mask": obs_mask_r})
stream = [t(ri["noise"][k]) for k in range(diffusion.num_timesteps)]
with ref_shims.injected_noise(stream):
    final_r = diffusion.p_sample_loop(wrapped, ri["x_T"].shape, **kwr)
np.savez_compressed(HERE / "unet_recon_chain.npz", final=final_r.detach().numpy(), fingerprint=cases.fingerprint(ri))
print("recon chain", float(final_r.abs().mean()))


# ---- attention=True: forward (cond / uncond / CFG) and the input-VJP of the guided output ----------------------
ac = cases.UNET_ATTN_CASE
ai = cases.make_unet_vjp_inputs(ac)
model_a = ref_unet.MDM_UNET(**ref.model_util.get_model_args(args, SimpleNamespace(dataset=SimpleNamespace())), attention=True)
shapes_a = {k: tuple(v.shape) for k, v in model_a.state_dict().items() if not k.startswith("clip_model.")}
assert any(".fn.fn.to_qkv." in k for k in shapes_a)
sd_a = weights.fill_like(shapes_a, ac["weight_seed"])
missing, unexpected = model_a.load_state_dict({k: torch.from_numpy(v) for k, v in sd_a.items()}, strict=False)
assert not unexpected and all(k.startswith("clip_model.") or k.endswith(".pe") for k in missing), (missing, unexpected)
model_a.eval()
wrapped_a = ref.cfg.ClassifierFreeSampleModel(model_a)
ref_shims.set_text_embedding(t(ai["enc_text"]))
ya = {"text": ["a"] * ac["B"], "mask": torch.ones(ac["B"], 1, 1, ac["T"], dtype=torch.bool)}
with torch.no_grad():
    oc_a = model_a(t(ai["x"]), t(ai["t"]), y=dict(ya), obs_x0=t(ai["obs_x0"]), obs_mask=t(ai["obs_mask"]))
    ou_a = model_a(t(ai["x"]), t(ai["t"]), y=dict(ya, uncond=True), obs_x0=t(ai["obs_x0"]), obs_mask=t(ai["obs_mask"]))
za = t(ai["x"]).clone().requires_grad_(True)
with torch.enable_grad():
    out_a = wrapped_a(za, t(ai["t"]), y=dict(ya, text_scale=t(ai["text_scale"])), obs_x0=t(ai["obs_x0"]), obs_mask=t(ai["obs_mask"]))
    gx_a, = torch.autograd.grad((out_a * t(ai["gout"])).sum(), za)
np.savez_compressed(HERE / "unet_attn.npz", out_cond=oc_a.numpy(), out_uncond=ou_a.numpy(), out_cfg=out_a.detach().numpy(),
                    gx=gx_a.numpy(), fingerprint=cases.fingerprint(ai), names=np.asarray(sorted(shapes_a)))
print("attention", float(oc_a.abs().mean()), float(gx_a.abs().mean()), float((oc_a - ou_a).abs().mean()))
