"""Run in a subprocess by test_persistent_gemm_keep_path_is_bitwise_the_tiled_schedule (tests/test_gpu_parity.py) with
CMDI_H3_PERSIST = 0 / 1 in the environment (the switch is read once per process): the stashing forward pass (folded
LayerNorms: ln_part / ln_c1 / ln_rg / out_part / ln_stats, aux under a folded A operand, C + Cs together), the input-VJP and
a reconstruction-guidance chain of the vjp_text_cfg / chain_edit_recon goldens -> <out>.npz."""
import importlib
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests" / "golden"))
import cases  # noqa: E402
from oracle import weights  # noqa: E402

PKG = "diffusion-motion-inbetweening_amd"
sub = lambda n: importlib.import_module(f"{PKG}.{n}")
DEV = "cuda:0"
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def model_for(case):
    mu = sub("utils.model_util")
    model, _ = mu.create_model_and_diffusion(SimpleNamespace(dataset="humanml", unconstrained=not case["text"], layers=8), None)
    mu.load_model_wo_clip(model, weights.to_torch(weights.make_state_dict(case["weight_seed"], text=case["text"], n_layers=8)))
    model.to(DEV).eval()
    model.native_precision = "f16x3"
    return sub("model.cfg_sampler").ClassifierFreeSampleModel(model).eval()


def main(out_path):
    res = {}
    case = cases.CASES["vjp_text_cfg"]
    inp = cases.make_inputs(case)
    model = model_for(case)
    B, _, _, T = inp["x"].shape
    eng = model.model.engine(torch.device(DEV), max_batch=B, max_frames=T, want_grad=True)
    eng.set_condition(batch=B, n_frames=T, cfg=True, enc_text=tt(inp["enc_text"]), text_scale=tt(inp["text_scale"]))
    res["vjp_out"] = eng.mdm_forward(tt(inp["x"]), tt(inp["t"])).cpu().numpy()
    res["vjp_gx"] = eng.mdm_vjp(tt(inp["gout"])).cpu().numpy()

    case = cases.CASES["chain_edit_recon"]
    inp = cases.make_inputs(case)
    model = model_for(case)
    gd, rs = sub("diffusion.gaussian_diffusion"), sub("diffusion.respace")
    diffusion = rs.SpacedDiffusion(rs.space_timesteps(1000, case["respacing"]),
                                   gd.DiffusionConfig(betas=gd.get_named_beta_schedule("cosine", 1000)))
    y = {"mask": tt(inp["len_mask"]), "lengths": tt(inp["lengths"]), "text_embed": tt(inp["enc_text"]),
         "text_scale": tt(inp["text_scale"]), "inpainting_mask": tt(inp["inpaint_mask"]), "inpainted_motion": tt(inp["x0"]),
         "imputate": True, "stop_imputation_at": case["stop_imputation_at"], "replacement_distribution": "conditional",
         "reconstruction_guidance": True, "reconstruction_weight": case["recon_weight"], "gradient_schedule": None,
         "diffusion_steps": 1000, "stop_recguidance_at": case["stop_recguidance_at"]}
    diffusion.injected_noise = tt(inp["noise"])
    res["chain_final"] = diffusion.p_sample_loop(model, inp["x_T"].shape, noise=tt(inp["x_T"]), clip_denoised=False,
                                                 model_kwargs={"y": y}).cpu().numpy()
    np.savez(out_path, **res)


if __name__ == "__main__":
    main(sys.argv[1])
