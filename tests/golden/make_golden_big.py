"""Golden vectors at the BASELINE shapes and chain lengths, produced by the REAL reference on CPU.

    python tests/golden/make_golden_big.py [case ...]        # default: every case of cases.BIG_CASES

big_c2 / big_c3    B=32 x 196 frames, 20 respaced steps of the 1000-step chain, ragged lengths, text CFG
                   (c3: + keyframe imputation and reconstruction guidance = the edit.py path).  The final sample is
                   stored for six samples (both halves of the engine's two-pipeline split) plus float64 (sum, sum^2)
                   of every sample.
long_ddpm          the full 1000-step ancestral chain (B=2, CFG): x_t every 100 steps, fp32 reference AND the same
                   chain run in float64 (reference model .double()) as the ground truth for the drift table.
long_ddim100       ddim_sample_loop on the 'ddim100' respacing, eta 0.
eps_ddpm/eps_ddim  ModelMeanType.EPSILON (reference gaussian_diffusion.py:536-555).
fwd_b256           one CFG evaluation at B=256 (C4's GEMM height M = 100,864).
Runs only where /root/reference exists; ~15 minutes on 8 cores.
"""
from __future__ import annotations

import os
import sys
import time
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
          f"({', '.join(f'{k}{tuple(np.shape(v))}' for k, v in arrays.items())})", flush=True)


def build(ref, case, dtype=torch.float32):
    sd = weights.to_torch(weights.make_state_dict(case["weight_seed"], text=case["text"]))
    args = ref_shims.default_args(unconstrained=not case["text"])
    model, _ = ref_shims.make_reference_model(ref, args, sd, cfg=case.get("cfg", False))
    if dtype == torch.float64:
        model.double()
        inner = getattr(model, "model", model)
        # MDM.encode_text ends in .float() (model/mdm.py:237): keep the embedding in float64 for the ground-truth chain
        inner.encode_text = lambda raw_text: ref_shims._state["text_embed"].double()
    betas = ref.gd.get_named_beta_schedule("cosine", 1000)
    use = ref.respace.space_timesteps(1000, case.get("respacing") or [1000])
    conf = ref.gd.DiffusionConfig(betas=betas)
    if case.get("mean_type") == "eps":
        conf.model_mean_type = ref.gd.ModelMeanType.EPSILON
    diffusion = ref.respace.SpacedDiffusion(use_timesteps=use, conf=conf)
    return model, diffusion


def run_chain(ref, case, inp, dtype):
    """p_sample_loop_progressive / ddim_sample_loop_progressive of the reference on the case's draws.
    Returns (final sample, {loop counter: sample}) as numpy."""
    model, diffusion = build(ref, case, dtype)
    B = case["B"]
    cast = (lambda a: t(a).to(dtype)) if dtype != torch.float32 else t
    y = {"mask": t(inp["len_mask"]), "lengths": t(inp["lengths"])}
    if case["text"]:
        ref_shims.set_text_embedding(cast(inp["enc_text"]))
        y.update(text=["a"] * B, text_scale=cast(inp["text_scale"]))
    if case.get("edit"):
        ref_mask = ref.editing.get_keyframes_mask(
            data=t(inp["x0"]), lengths=t(inp["lengths"]), edit_mode='benchmark_sparse',
            trans_length=case["trans_length"], feature_mode='pos_rot_vel')
        assert np.array_equal(ref_mask.numpy(), inp["inpaint_mask"]), "keyframe mask restatement drifted"
        y.update(inpainting_mask=t(inp["inpaint_mask"]), inpainted_motion=cast(inp["x0"]),
                 imputate=case["imputate"], stop_imputation_at=case["stop_imputation_at"],
                 replacement_distribution='conditional', reconstruction_guidance=case["recon"],
                 reconstruction_weight=case["recon_weight"], gradient_schedule=case["grad_schedule"],
                 diffusion_steps=1000, stop_recguidance_at=case["stop_recguidance_at"])
    n = diffusion.num_timesteps
    assert n == cases.big_n_steps(case)
    prog = diffusion.ddim_sample_loop_progressive if case["sampler"] == "ddim" else diffusion.p_sample_loop_progressive
    kw = dict(noise=cast(inp["draw0"]), clip_denoised=False, model_kwargs={"y": y}, device=torch.device("cpu"),
              skip_timesteps=case.get("skip", 0),
              init_image=cast(inp["init_image"]) if "init_image" in inp else None)
    if case["sampler"] == "ddim":
        kw["eta"] = case["eta"]
    stream = (cast(cases.big_draw(case, 1 + k)) for k in range(n - case.get("skip", 0)))
    every = case.get("every")
    dumps, final = {}, None
    t0 = time.time()
    with ref_shims.injected_noise(stream):
        for i, out in enumerate(prog(model, inp["draw0"].shape, **kw)):
            final = out["sample"]
            if every and (i + 1) % every == 0:
                dumps[i] = out["sample"].detach().numpy().copy()
            if (i + 1) % max(1, n // 10) == 0:
                print(f"  step {i + 1}/{n}  {time.time() - t0:.0f}s", flush=True)
    return final.detach().numpy(), dumps


def chain_case(ref, name):
    case = cases.BIG_CASES[name]
    inp = cases.make_big_inputs(case)
    out = {"fingerprint": cases.fingerprint(inp)}
    final, dumps = run_chain(ref, case, inp, torch.float32)
    keep = list(case.get("keep", range(case["B"])))
    out["final"] = final[keep]
    out["stats"] = cases.sample_stats(final)
    if dumps:
        out["dump_at"] = np.asarray(sorted(dumps), dtype=np.int64)
        out["dumps"] = np.stack([dumps[i][:1] for i in sorted(dumps)])   # sample 0 only
    if case.get("f64"):
        f64, d64 = run_chain(ref, case, inp, torch.float64)
        out["final_f64"] = f64.astype(np.float32)
        out["dumps_f64"] = np.stack([d64[i][:1] for i in sorted(d64)]).astype(np.float32)
        err = np.linalg.norm(final.astype(np.float64) - f64) / np.linalg.norm(f64)
        print(f"  {name}: reference fp32 vs float64 chain rel-L2 {err:.3e}", flush=True)
    save(name, **out)


def fwd_case(ref, name):
    case = cases.BIG_CASES[name]
    inp = cases.make_big_inputs(case)
    model, _ = build(ref, dict(case, cfg=True))
    ref_shims.set_text_embedding(t(inp["enc_text"]))
    y = {"text": ["a"] * case["B"], "text_scale": t(inp["text_scale"])}
    with torch.no_grad():
        cfg = model(t(inp["x"]), t(inp["t"]), y=y).numpy()
    save(name, fingerprint=cases.fingerprint(inp), out_cfg=cfg[list(case["keep"])], stats=cases.sample_stats(cfg))


def main():
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", 0)) or os.cpu_count() or 1)
    ref = ref_shims.import_reference()
    names = sys.argv[1:] or list(cases.BIG_CASES)
    for name in names:
        print(f"== {name}", flush=True)
        (fwd_case if cases.BIG_CASES[name]["kind"] == "fwd" else chain_case)(ref, name)


if __name__ == "__main__":
    main()
