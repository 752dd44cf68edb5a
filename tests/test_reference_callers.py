"""north_star: 'sample.conditional_synthesis and sample.edit call the new path unchanged'.  These tests EXECUTE the
reference's own ``main()`` of the three scripts (from /root/reference, untouched) on top of
compat.install_reference_aliases(), with only the data loader / plotting / CLIP stubbed
(tests/helpers/run_reference_caller.py, mode `aliased`).  The build container has no GPU, so the p_sample_loop call itself is
recorded after the package's host-side argument translation — AND its exact arguments (shape, model_kwargs, keyword
arguments) are compared, tensor by tensor, with tests/golden/caller_<script>.npz: the arguments the SAME script builds on the
reference's OWN modules, stored together with the real reference sampler's output for them
(tests/golden/make_golden_callers.py).  On the GPU box tests/test_gpu_parity.py::test_reference_callers_replayed_on_the_gpu
feeds those arguments to the native p_sample_loop and compares with that output — the two halves together take the mock out
of the argument: same call (here), same result for that call (there).
Needs /root/reference: skipped on the GPU box."""
import json
import sys

import numpy as np
import pytest

from conftest import GOLDEN, REPO
from oracle import ref_shims

sys.path.insert(0, str(REPO / "tests" / "helpers"))
import caller_setup  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_shims.available(), reason="the reference checkout is not on this box")


def run_caller(tmp_path, cases, name):
    case = cases.CALLER_CASES[name]
    caller_setup.write_checkpoint(tmp_path, case["model_args"], case["weight_seed"])
    res = caller_setup.run_script(tmp_path, case, "aliased", cases.CALLER_SAMPLES)
    same_call_as_the_reference_run(tmp_path / "recorded_call.npz", GOLDEN / f"caller_{name}.npz")
    return res


def same_call_as_the_reference_run(ours_path, golden_path):
    """Every argument of the recorded p_sample_loop call — the script running on THIS package — equals what the script
    passes when it runs on the reference's own modules (get_keyframes_mask output included: bit-exact)."""
    ours, gold = np.load(ours_path), np.load(golden_path)
    assert json.loads(str(ours["meta"])) == json.loads(str(gold["meta"]))
    arg_keys = lambda z: sorted(k for k in z.files if k.split(".")[0] in ("y", "mk", "kw"))
    assert arg_keys(ours) == arg_keys(gold)
    for k in arg_keys(gold):
        assert ours[k].dtype == gold[k].dtype and np.array_equal(ours[k], gold[k]), k


def test_sample_edit_main_runs_unchanged(tmp_path, cases):
    """reference sample/edit.py:26-264: imputation + reconstruction guidance (BASELINE config 3's caller)."""
    res = run_caller(tmp_path, cases, "edit")
    (call,) = res["calls"]
    assert call["shape"] == [3, 263, 1, 196] and call["native_denoiser"] == "MDM" and call["cfg"] is True
    assert call["diffusion"].endswith("_amd.diffusion.respace.SpacedDiffusion") and call["n_steps"] == 1000
    assert {"imputate", "inpainted_motion", "inpainting_mask", "reconstruction_guidance", "reconstruction_weight",
            "gradient_schedule", "stop_imputation_at", "stop_recguidance_at", "text_scale"} <= set(call["y_keys"])
    c = call["condition"]
    assert c["imputate"] == 1 and c["recon_guidance"] is True and c["recon_w"] == [1000]
    assert c["inpaint_mask"] == [3, 263, 1, 196] and c["text_scale"] == [3]
    assert "out/results.npy" in res["results"]


def test_sample_conditional_synthesis_main_runs_unchanged(tmp_path, cases):
    """reference sample/conditional_synthesis.py:26-330 with a keyframe-conditioned MDM_UNET (obs_x0 / obs_mask)."""
    res = run_caller(tmp_path, cases, "conditional_synthesis")
    (call,) = res["calls"]
    assert call["native_denoiser"] == "MDM_UNET" and call["extra_model_kwargs"] == ["obs_mask", "obs_x0"]
    c = call["condition"]
    assert c["obs_x0"] == [3, 263, 1, 196] and c["obs_mask"] == [3, 263, 1, 196] and c["imputate"] == 1
    assert "out/results.npy" in res["results"]


def test_sample_synthesize_main_runs_unchanged(tmp_path, cases):
    """reference sample/synthesize.py:39-200 (plain text-to-motion: BASELINE config 2's caller), test-set prompts."""
    res = run_caller(tmp_path, cases, "synthesize")
    (call,) = res["calls"]
    assert call["shape"] == [3, 263, 1, 196] and call["native_denoiser"] == "MDM" and call["cfg"] is True
    assert call["diffusion"].endswith("_amd.diffusion.respace.SpacedDiffusion") and call["n_steps"] == 1000
    assert {"const_noise", "dump_steps", "init_image", "noise", "progress", "skip_timesteps", "clip_denoised"} <= set(call["kwargs"])
    assert call["condition"]["text_scale"] == [3] and call["condition"].get("imputate", 0) == 0
    assert "out/results.npy" in res["results"]
