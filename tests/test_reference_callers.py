"""north_star: 'sample.conditional_synthesis and sample.edit call the new path unchanged'.  These tests EXECUTE the
reference's own ``main()`` of both scripts (from /root/reference, untouched) on top of compat.install_reference_aliases(),
with only the data loader / plotting stubbed (tests/helpers/run_reference_caller.py).  The build container has no GPU, so
the p_sample_loop call itself is recorded after the package's host-side argument translation; the kernels behind it are
covered by the -m gpu tests, which feed p_sample_loop the same model_kwargs these scripts build.
Needs /root/reference: skipped on the GPU box."""
import json
import os
import subprocess
import sys
from types import SimpleNamespace

import pytest
import torch

from conftest import REPO, sub
from oracle import ref_shims

pytestmark = pytest.mark.skipif(not ref_shims.available(), reason="the reference checkout is not on this box")


def run_caller(tmp_path, script, model_args, extra, edit_args=True):
    mu = sub("utils.model_util")
    model, _ = mu.create_model_and_diffusion(SimpleNamespace(**model_args), None)
    ck = tmp_path / "save" / "ckpt"
    ck.mkdir(parents=True)
    torch.save({"model": {k: v for k, v in model.state_dict().items() if not k.startswith("clip_model.")}},
               ck / "model000000010.pt")
    (ck / "args.json").write_text(json.dumps(dict(model_args, abs_3d=True, latent_dim=512)))
    cmd = [sys.executable, str(REPO / "tests" / "helpers" / "run_reference_caller.py"), script, str(tmp_path),
           "--model_path", "save/ckpt/model000000010.pt", "--num_samples", "3", "--num_repetitions", "1",
           "--output_dir", "out"] + (["--edit_mode", "benchmark_sparse", "--transition_length", "5"] if edit_args else []) + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, CUDA_VISIBLE_DEVICES=""))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = next(ln for ln in r.stdout.splitlines() if ln.startswith("CALLER_RESULT "))
    return json.loads(line[len("CALLER_RESULT "):])


def test_sample_edit_main_runs_unchanged(tmp_path):
    """reference sample/edit.py:26-264: imputation + reconstruction guidance (BASELINE config 3's caller)."""
    res = run_caller(tmp_path, "edit", dict(dataset="humanml", arch="trans_enc", cond_mask_prob=0.1,
                                            keyframe_conditioned=False, layers=8),
                     ["--imputate", "--reconstruction_guidance"])
    (call,) = res["calls"]
    assert call["shape"] == [3, 263, 1, 196] and call["native_denoiser"] == "MDM" and call["cfg"] is True
    assert call["diffusion"].endswith("_amd.diffusion.respace.SpacedDiffusion") and call["n_steps"] == 1000
    assert {"imputate", "inpainted_motion", "inpainting_mask", "reconstruction_guidance", "reconstruction_weight",
            "gradient_schedule", "stop_imputation_at", "stop_recguidance_at", "text_scale"} <= set(call["y_keys"])
    c = call["condition"]
    assert c["imputate"] == 1 and c["recon_guidance"] is True and c["recon_w"] == [1000]
    assert c["inpaint_mask"] == [3, 263, 1, 196] and c["text_scale"] == [3]
    assert "out/results.npy" in res["results"]


def test_sample_conditional_synthesis_main_runs_unchanged(tmp_path):
    """reference sample/conditional_synthesis.py:26-330 with a keyframe-conditioned MDM_UNET (obs_x0 / obs_mask)."""
    res = run_caller(tmp_path, "conditional_synthesis",
                     dict(dataset="humanml", arch="unet", cond_mask_prob=0.1, keyframe_conditioned=True,
                          dim_mults=[1, 1, 1, 1], unet_adagn=True, unet_zero=True), ["--imputate"])
    (call,) = res["calls"]
    assert call["native_denoiser"] == "MDM_UNET" and call["extra_model_kwargs"] == ["obs_mask", "obs_x0"]
    c = call["condition"]
    assert c["obs_x0"] == [3, 263, 1, 196] and c["obs_mask"] == [3, 263, 1, 196] and c["imputate"] == 1
    assert "out/results.npy" in res["results"]


def test_sample_synthesize_main_runs_unchanged(tmp_path):
    """reference sample/synthesize.py:39-200 (plain text-to-motion: BASELINE config 2's caller), test-set prompts."""
    res = run_caller(tmp_path, "synthesize", dict(dataset="humanml", arch="trans_enc", cond_mask_prob=0.1,
                                                  keyframe_conditioned=False, layers=8), [], edit_args=False)
    (call,) = res["calls"]
    assert call["shape"] == [3, 263, 1, 196] and call["native_denoiser"] == "MDM" and call["cfg"] is True
    assert call["diffusion"].endswith("_amd.diffusion.respace.SpacedDiffusion") and call["n_steps"] == 1000
    assert {"const_noise", "dump_steps", "init_image", "noise", "progress", "skip_timesteps", "clip_denoised"} <= set(call["kwargs"])
    assert call["condition"]["text_scale"] == [3] and call["condition"].get("imputate", 0) == 0
    assert "out/results.npy" in res["results"]
