"""MDM_UNET at the transformer's parity depth (VERDICT r5 task 1b), by the REAL reference on CPU:

    python tests/golden/make_golden_unet_long.py [long_unet] [big_unet]      ->  tests/golden/<case>.npz

long_unet   released geometry (configs: dim 512 x mults (2,2,2,2)), B=2, ALL 1000 ancestral steps of p_sample_loop through
            ClassifierFreeSampleModel(MDM_UNET) with keyframe conditioning (obs_x0 / obs_mask), imputation and reconstruction
            guidance (weight 20) on every step (reference diffusion/gaussian_diffusion.py:405-435, model/mdm_unet.py:561-849) — in
            fp32 and again in float64 (model.double()): the ground truth the drift of every precision mode is measured against.
big_unet    the same at B=32, ragged lengths, on the 'ddim100' respacing through p_sample_loop (what
            sample/conditional_synthesis.py calls): six stored samples + float64 (sum, sum^2) of all 32, sample 0 every 10 steps.
Runs only where /root/reference exists.
"""
import os
import sys
import time
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

t = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def build(case, dtype):
    args = ref_shims.default_args(arch='unet', keyframe_conditioned=True, abs_3d=True, latent_dim=512,
                                  dim_mults=case["dim_mults"], cond_mask_prob=0.1)
    model, _ = ref.model_util.create_model_and_diffusion(args, SimpleNamespace(dataset=SimpleNamespace()))
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.startswith("clip_model.")}
    sd = weights.fill_like(shapes, case["weight_seed"])
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.startswith("clip_model.") or k.endswith(".pe") for k in missing), (missing, unexpected)
    model.eval()
    if dtype == torch.float64:
        model.double()
        # encode_text ends in .float() (the reference's MDM.encode_text): keep the embedding in float64 for the ground truth
        model.encode_text = lambda raw_text: ref_shims._state["text_embed"].double()
    wrapped = ref.cfg.ClassifierFreeSampleModel(model)
    betas = ref.gd.get_named_beta_schedule("cosine", 1000)
    use = ref.respace.space_timesteps(1000, case.get("respacing") or [1000])
    diffusion = ref.respace.SpacedDiffusion(use_timesteps=use, conf=ref.gd.DiffusionConfig(betas=betas))
    return wrapped, diffusion, sorted(shapes)


def run(case, inp, dtype, rows=None):
    """rows: run only these samples of the case's batch (samples are independent — GroupNorm is per sample — and the noise
    draws are sliced from the full-batch draws, so the chain of sample i is the one it has inside the full batch)."""
    wrapped, diffusion, names = build(case, dtype)
    if rows is not None:
        inp = {k: (v[list(rows)] if getattr(v, "shape", ()) and v.shape[0] == case["B"] else v) for k, v in inp.items()}
        case = dict(case, B=len(rows))
    sel = (lambda a: a[list(rows)]) if rows is not None else (lambda a: a)
    cast = (lambda a: t(a).to(dtype)) if dtype != torch.float32 else t
    ref_shims.set_text_embedding(cast(inp["enc_text"]))
    obs_mask = t(inp["obs_mask"])
    y = {"mask": t(inp["len_mask"]), "lengths": t(inp["lengths"]), "text": ["a"] * case["B"], "text_scale": cast(inp["text_scale"]),
         "inpainting_mask": obs_mask, "inpainted_motion": cast(inp["x0"]), "imputate": True,
         "stop_imputation_at": case["stop_imputation_at"], "replacement_distribution": "conditional",
         "reconstruction_guidance": True, "reconstruction_weight": case["recon_weight"], "gradient_schedule": None,
         "stop_recguidance_at": case["stop_recguidance_at"], "diffusion_steps": 1000}
    n = diffusion.num_timesteps
    assert n == cases.unet_long_steps(case)
    full = cases.UNET_LONG_CASES[case["name"]] if "name" in case else case
    stream = (cast(sel(cases.unet_long_draw(full, 1 + k))) for k in range(n))
    every, dumps, final = case["every"], {}, None
    t0 = time.time()
    with ref_shims.injected_noise(stream):
        for i, out in enumerate(diffusion.p_sample_loop_progressive(
                wrapped, inp["draw0"].shape, noise=cast(inp["draw0"]), clip_denoised=False, device=torch.device("cpu"),
                model_kwargs={"y": y, "obs_x0": cast(inp["x0"]), "obs_mask": obs_mask})):
            final = out["sample"]
            if (i + 1) % every == 0:
                dumps[i] = out["sample"].detach().numpy().copy()
            if (i + 1) % max(1, n // 20) == 0:
                print(f"  step {i + 1}/{n}  {time.time() - t0:.0f}s", flush=True)
    return final.detach().numpy(), dumps, names


def main():
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", 0)) or os.cpu_count() or 1)
    only_rows = "--f64-rows-only" in sys.argv       # add the float64 rows to an existing <case>.npz (keeps its fp32 chain)
    for name in [a for a in sys.argv[1:] if not a.startswith("--")] or list(cases.UNET_LONG_CASES):
        case = cases.UNET_LONG_CASES[name]
        inp = cases.make_unet_long_inputs(case)
        print(f"== {name}", flush=True)
        if only_rows:
            old = dict(np.load(HERE / f"{name}.npz"))
            assert np.array_equal(old["fingerprint"], cases.fingerprint(inp))
            rows = list(case["f64_rows"])
            f64, d64, _ = run(dict(case, name=name), inp, torch.float64, rows=rows)
            old["f64_rows"] = np.asarray(rows, dtype=np.int64)
            old["final_f64_rows"] = f64.astype(np.float32)
            old["dumps_f64"] = np.stack([d64[i][:1] for i in sorted(d64)]).astype(np.float32)
            keep = list(case["keep"])
            for j, r in enumerate(rows):
                ref32 = old["final"][keep.index(r)].astype(np.float64)
                print(f"  {name}: sample {r}: reference fp32 vs float64 chain rel-L2 "
                      f"{np.linalg.norm(ref32 - f64[j]) / np.linalg.norm(f64[j]):.3e}", flush=True)
            np.savez_compressed(HERE / f"{name}.npz", **old)
            continue
        final, dumps, names = run(case, inp, torch.float32)
        keep = list(case.get("keep", range(case["B"])))
        out = {"fingerprint": cases.fingerprint(inp), "final": final[keep], "stats": cases.sample_stats(final),
               "dump_at": np.asarray(sorted(dumps), dtype=np.int64), "dumps": np.stack([dumps[i][:1] for i in sorted(dumps)]),
               "names": np.asarray(names)}
        if case.get("f64"):
            f64, d64, _ = run(case, inp, torch.float64)
            out["final_f64"] = f64.astype(np.float32)
            out["dumps_f64"] = np.stack([d64[i][:1] for i in sorted(d64)]).astype(np.float32)
            err = np.linalg.norm(final.astype(np.float64) - f64) / np.linalg.norm(f64)
            print(f"  {name}: reference fp32 vs float64 chain rel-L2 {err:.3e}", flush=True)
        if case.get("f64_rows"):
            # the float64 ground truth of a FEW samples of the batch (the whole batch in float64 would take hours)
            rows = list(case["f64_rows"])
            f64, d64, _ = run(dict(case, name=name), inp, torch.float64, rows=rows)
            out["f64_rows"] = np.asarray(rows, dtype=np.int64)
            out["final_f64_rows"] = f64.astype(np.float32)
            out["dumps_f64"] = np.stack([d64[i][:1] for i in sorted(d64)]).astype(np.float32)     # first of the rows
            for j, r in enumerate(rows):
                err = np.linalg.norm(final[r].astype(np.float64) - f64[j]) / np.linalg.norm(f64[j])
                print(f"  {name}: sample {r}: reference fp32 vs float64 chain rel-L2 {err:.3e}", flush=True)
        path = HERE / f"{name}.npz"
        np.savez_compressed(path, **out)
        print(f"wrote {path.name}: {os.path.getsize(path) / 1024:.0f} KiB", flush=True)


if __name__ == "__main__":
    main()
