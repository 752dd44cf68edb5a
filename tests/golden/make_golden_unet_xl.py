"""MDM_UNET at the RELEASED geometry (configs/model.py motion_unet_adagn_xl: dim 512 x mults (2,2,2,2) = 1024 channels at all
four levels, 128 channels per GroupNorm group) — forward (cond / uncond / CFG) and the input-VJP of the guided output, by the
REAL reference on CPU (model/mdm_unet.py:561-849 through model/cfg_sampler.py; torch.autograd for the VJP, which is what
diffusion/gaussian_diffusion.py:411-416 calls).  VERDICT r3 task 5c: the xl test compared with the torch port only.

    python tests/golden/make_golden_unet_xl.py        ->  tests/golden/unet_xl.npz   (about a minute on 8 cores)
"""
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

case = cases.UNET_XL_CASE
inp = cases.make_unet_vjp_inputs(case)
args = ref_shims.default_args(arch='unet', keyframe_conditioned=True, abs_3d=True, latent_dim=512,
                              dim_mults=case["dim_mults"], cond_mask_prob=0.1)
model, _ = ref.model_util.create_model_and_diffusion(args, SimpleNamespace(dataset=SimpleNamespace()))
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.startswith("clip_model.")}
sd = weights.fill_like(shapes, case["weight_seed"])
missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
assert not unexpected and all(k.startswith("clip_model.") or k.endswith(".pe") for k in missing), (missing, unexpected)
model.eval()
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
ref_shims.set_text_embedding(t(inp["enc_text"]))
wrapped = ref.cfg.ClassifierFreeSampleModel(model)
y = {"text": ["a"] * case["B"], "mask": torch.ones(case["B"], 1, 1, case["T"], dtype=torch.bool)}
kw = dict(obs_x0=t(inp["obs_x0"]), obs_mask=t(inp["obs_mask"]))
with torch.no_grad():
    oc = model(t(inp["x"]), t(inp["t"]), y=dict(y), **kw)
    ou = model(t(inp["x"]), t(inp["t"]), y=dict(y, uncond=True), **kw)
z = t(inp["x"]).clone().requires_grad_(True)
with torch.enable_grad():
    out = wrapped(z, t(inp["t"]), y=dict(y, text_scale=t(inp["text_scale"])), **kw)
    gx, = torch.autograd.grad((out * t(inp["gout"])).sum(), z)
# the same evaluation and input-VJP in float64 (model.double()): the ground truth the fp32 reference and every native precision
# mode are measured against (VERDICT r5 task 1c: is the U-Net's distance from the reference rounding or a defect?)
model.double()
model.encode_text = lambda raw_text: ref_shims._state["text_embed"].double()
ref_shims.set_text_embedding(t(inp["enc_text"]).double())
z64 = t(inp["x"]).double().clone().requires_grad_(True)
kw64 = dict(obs_x0=t(inp["obs_x0"]).double(), obs_mask=t(inp["obs_mask"]))
with torch.enable_grad():
    out64 = wrapped(z64, t(inp["t"]), y=dict(y, text_scale=t(inp["text_scale"]).double()), **kw64)
    gx64, = torch.autograd.grad((out64 * t(inp["gout"]).double()).sum(), z64)
rel = lambda a, b: float((a.double() - b).norm() / b.norm())
print("reference fp32 vs float64: out_cfg", rel(out.detach(), out64.detach()), "gx", rel(gx, gx64))
np.savez_compressed(HERE / "unet_xl.npz", out_cond=oc.numpy(), out_uncond=ou.numpy(), out_cfg=out.detach().numpy(),
                    gx=gx.numpy(), fingerprint=cases.fingerprint(inp), names=np.asarray(sorted(shapes)),
                    out_cfg_f64=out64.detach().numpy(), gx_f64=gx64.numpy())
print("unet_xl", float(oc.abs().mean()), float(gx.abs().mean()), float(gx[t(inp["obs_mask"])].abs().max()))
