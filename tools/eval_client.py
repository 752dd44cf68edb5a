"""Throughput client with the call pattern of the reference's evaluation generator
(data_loaders/humanml/motion_loaders/comp_v6_model_dataset_condmdi.py:24-355, driven by eval/eval_humanml_condmdi.py:444-568):
for every test batch (32 motions) and every replication

    fixseed(seed * 100_000 + i * 100 + t)                                  (:209-210)
    model_kwargs <- set_inference_editing_args / set_conditional_synthesis_args (:488-565)
    sample = motion_diffusion.p_sample_loop(motion_model, (B, njoints, nfeats, T), clip_denoised=False,
                                            model_kwargs=model_kwargs, skip_timesteps=0, init_image=None,
                                            progress=False, dump_steps=None, noise=None, const_noise=False)   (:343-355)
    motion = sample_to_motion(sample)     -> here: inv_transform + recover_from_ric on the device (cmdi_recover_xyz)

and reports motions/s end to end (the workload the reference's README quotes at ~20 h).  Data are synthetic (no dataset
offline): z-scored N(0,1) motions, ragged lengths, fake CLIP embeddings, benchmark_sparse keyframes (T=5).

    python tools/eval_client.py [--batches 4] [--reps 1] [--model mdm|unet] [--mode edit|impute|plain] [--steps 1000]
                                [--precision f16x3|bf16x6|f32]
"""
import argparse
import importlib
import json
import sys
import time
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
PKG = "diffusion-motion-inbetweening_amd"
sub = lambda n: importlib.import_module(f"{PKG}.{n}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=4, help="test batches of 32 (the full test split is 32)")
    ap.add_argument("--reps", type=int, default=1, help="replications per batch (the evaluator runs 20 over the split)")
    ap.add_argument("--batch-size", type=int, default=32)
    ap.add_argument("--model", default="mdm", choices=["mdm", "unet"])
    ap.add_argument("--mode", default="edit", choices=["edit", "impute", "plain"],
                    help="edit = imputation + reconstruction guidance (edit.py / --imputate --reconstruction_guidance), "
                         "impute = imputation only, plain = text-conditioned CFG only")
    ap.add_argument("--steps", type=int, default=1000, help="diffusion steps of the chain (1000, or e.g. 100 = 'ddim100' respacing)")
    ap.add_argument("--precision", default=None, choices=["f32", "f16x3", "bf16x6"])
    ap.add_argument("--seed", type=int, default=10)
    args = ap.parse_args()

    from oracle import weights   # synthetic weight recipe only
    dev = torch.device("cuda:0")
    mu, gd, rs, eu = sub("utils.model_util"), sub("diffusion.gaussian_diffusion"), sub("diffusion.respace"), sub("utils.editing_util")
    mp = sub("data_loaders.humanml.scripts.motion_process")
    fixseed = sub("utils.fixseed").fixseed
    B, T, J = args.batch_size, 196, 263
    if args.model == "unet":
        margs = SimpleNamespace(dataset="humanml", arch="unet", keyframe_conditioned=True, dim_mults=(2, 2, 2, 2), cond_mask_prob=0.1)
        model, _ = mu.create_model_and_diffusion(margs, None)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        sd = weights.to_torch(weights.fill_like(shapes, 0))
        sd.update({k: v for k, v in model.state_dict().items() if k.endswith(".pe")})
        mu.load_model_wo_clip(model, sd)
    else:
        model, _ = mu.create_model_and_diffusion(SimpleNamespace(dataset="humanml"), None)
        mu.load_model_wo_clip(model, weights.to_torch(weights.make_state_dict(0, text=True)))
    model.to(dev).eval()
    model.native_precision = args.precision
    motion_model = sub("model.cfg_sampler").ClassifierFreeSampleModel(model)     # args.guidance_param != 1
    resp = [1000] if args.steps == 1000 else f"ddim{args.steps}"
    motion_diffusion = rs.SpacedDiffusion(rs.space_timesteps(1000, resp),
                                          gd.DiffusionConfig(betas=gd.get_named_beta_schedule("cosine", 1000)))
    rng = np.random.default_rng(args.seed)
    mean = torch.from_numpy((0.3 * rng.standard_normal(J)).astype(np.float32)).to(dev)   # stand-ins for Mean/Std_abs_3d.npy
    std = torch.from_numpy((0.05 + rng.random(J)).astype(np.float32)).to(dev)

    n_motions, t_loop, t_post = 0, 0.0, 0.0
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    for i in range(args.batches):
        # the dataloader's batch: (motion, model_kwargs) with y = {mask, lengths, text, tokens}
        motion = torch.from_numpy(rng.standard_normal((B, J, 1, T)).astype(np.float32)).to(dev)
        lengths = torch.from_numpy(rng.integers(40, T + 1, B)).to(dev)
        y = {"mask": (torch.arange(T, device=dev)[None, :] < lengths[:, None]).view(B, 1, 1, T), "lengths": lengths,
             "text": ["a person walks"] * B,
             "text_embed": torch.from_numpy(rng.standard_normal((B, 512)).astype(np.float32)).to(dev),   # CLIP(text), cached
             "text_scale": torch.ones(B, device=dev) * 2.5}
        model_kwargs = {"y": y}
        if args.model == "unet":          # set_conditional_synthesis_args (:522-565)
            model_kwargs["obs_x0"] = motion
            model_kwargs["obs_mask"], _ = eu.get_keyframes_mask(data=motion, lengths=lengths, edit_mode="benchmark_sparse",
                                                                feature_mode="pos_rot_vel", trans_length=5, get_joint_mask=True)
            y["diffusion_steps"] = 1000
            if args.mode != "plain":
                y.update(imputate=1, stop_imputation_at=1, replacement_distribution="conditional",
                         inpainted_motion=motion, inpainting_mask=model_kwargs["obs_mask"])
                if args.mode == "edit":
                    y.update(reconstruction_guidance=True, reconstruction_weight=20.0, gradient_schedule=None, stop_recguidance_at=0)
        elif args.mode != "plain":        # set_inference_editing_args (:488-519)
            y.update(inpainted_motion=motion, imputate=True, replacement_distribution="conditional",
                     reconstruction_guidance=args.mode == "edit", reconstruction_weight=20.0, diffusion_steps=1000,
                     gradient_schedule=None, stop_imputation_at=1, stop_recguidance_at=0)
            y["inpainting_mask"], _ = eu.get_keyframes_mask(data=motion, lengths=lengths, edit_mode="benchmark_sparse",
                                                            trans_length=5, feature_mode="pos_rot_vel", get_joint_mask=True)
        for t in range(args.reps):
            fixseed(args.seed * 100_000 + i * 100 + t)
            t0 = time.perf_counter()
            sample_motion = motion_diffusion.p_sample_loop(
                motion_model, (B, model.njoints, model.nfeats, T), clip_denoised=False, model_kwargs=model_kwargs,
                skip_timesteps=0, init_image=None, progress=False, dump_steps=None, noise=None, const_noise=False)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            cur_motion = mp.sample_to_xyz(sample_motion, mean, std, 22, abs_3d=True)     # sample_to_motion on the device
            skel = cur_motion.cpu()                                                         # what the evaluator consumes
            t2 = time.perf_counter()
            assert torch.isfinite(skel).all() and skel.shape == (B, 22, 3, T)
            n_motions += B
            t_loop += t1 - t0
            t_post += t2 - t1
    total = time.perf_counter() - t_all
    n_steps = motion_diffusion.num_timesteps
    print(json.dumps({
        "client": "CompMDMGeneratedDatasetCondMDI call pattern", "model": args.model, "mode": args.mode,
        "precision": model._engine.precision, "batches": args.batches, "reps": args.reps, "batch_size": B,
        "chain_steps": n_steps, "motions": n_motions, "motions_per_sec_end_to_end": n_motions / total,
        "steps_per_sec_in_loop": args.batches * args.reps * n_steps / t_loop, "seconds_total": total,
        "seconds_in_p_sample_loop": t_loop, "seconds_post_sampling_incl_d2h": t_post,
        "projected_hours_for_the_20_replication_eval": 20 * 32 * (total / (args.batches * args.reps)) / 3600.0}))


if __name__ == "__main__":
    main()
