"""Run one of the REFERENCE's own sample scripts (sample/edit.py, sample/conditional_synthesis.py or sample/synthesize.py
``main()``), unchanged,
on top of this package: ``compat.install_reference_aliases()`` first, exactly as INTEGRATION.md recipe A says, then the
reference tree on sys.path.  Only what is NOT on the hot path is stubbed: the HumanML3D data loader (no dataset offline),
the mp4 plotting and ffmpeg.  The build container has no GPU, so the one call into the hot path —
``diffusion.p_sample_loop`` — is recorded: its arguments go through the package's own host-side translation
(GaussianDiffusion._condition_from_kwargs + _add_observations, i.e. everything up to the native call) and a tensor of the
right shape comes back, so the script runs to its end and writes results.npy.   Usage:
    python run_reference_caller.py <edit|conditional_synthesis|synthesize> <workdir> [script args ...]
Prints one JSON line describing the recorded call."""
import importlib
import json
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REFERENCE = os.environ.get("CONDMDI_REFERENCE", "/root/reference")
PKG = "diffusion-motion-inbetweening_amd"


def main():
    script, workdir = sys.argv[1], sys.argv[2]
    sys.path.insert(0, REPO)
    compat = importlib.import_module(f"{PKG}.compat")
    compat.install_reference_aliases()                      # INTEGRATION.md, recipe A, line 1
    sys.path.insert(0, REFERENCE)
    from oracle import ref_shims                            # numpy aliases + clip / spacy / smplx stubs of SURVEY 8c
    ref_shims._install_stubs()
    os.chdir(workdir)

    # ---- the caller's module, imported UNCHANGED from the reference tree ---------------------------------------
    mod = importlib.import_module(f"sample.{script}")
    gd = importlib.import_module("diffusion.gaussian_diffusion")
    assert gd.__name__.startswith(PKG), "diffusion.gaussian_diffusion is not the aliased module"
    assert mod.create_model_and_diffusion.__module__.startswith(PKG)
    assert mod.ClassifierFreeSampleModel.__module__.startswith(PKG)
    if hasattr(mod, "get_keyframes_mask"):
        assert mod.get_keyframes_mask.__module__.startswith(PKG)

    # ---- stubs OUTSIDE the hot path -----------------------------------------------------------------------------
    rng = np.random.default_rng(0)
    n, T = int(os.environ.get("CALLER_SAMPLES", "3")), 196

    class FakeT2M:
        mean, std = np.zeros(263, np.float32), np.ones(263, np.float32)

        def inv_transform(self, data):
            return data * torch.from_numpy(self.std) + torch.from_numpy(self.mean)

    class FakeLoader:
        dataset = SimpleNamespace(t2m_dataset=FakeT2M())

        def __iter__(self):
            lengths = torch.tensor([196, 150, 77][:n])
            motion = torch.from_numpy(rng.standard_normal((n, 263, 1, T)).astype(np.float32))
            y = {"mask": (torch.arange(T)[None, :] < lengths[:, None]).view(n, 1, 1, T), "lengths": lengths,
                 "text": ["a person walks"] * n, "tokens": ["a/DET person/NOUN walks/VERB"] * n}
            yield motion, {"y": y}

    mod.load_dataset = lambda args, max_frames, *a, **k: FakeLoader()
    mod.plot_3d_motion = lambda *a, **k: None
    if hasattr(mod, "plot_conditional_samples"):
        mod.plot_conditional_samples = lambda *a, **k: None
    mod.os.system = lambda cmd: 0                          # ffmpeg hstack of the mp4s

    # ---- record the hot-path call ----------------------------------------------------------------------------------
    calls = []
    real = gd.GaussianDiffusion.p_sample_loop

    def recorded(self, model, shape, **kw):
        mdm, cfg = gd._unwrap_model(model)
        y = kw["model_kwargs"]["y"]
        if mdm is not None and "text" in mdm.cond_mode and "text_embed" not in y:
            y = dict(y, text_embed=torch.zeros(shape[0], 512))          # CLIP is stubbed away on this box
        B, J, F, Tn = shape
        cond = self._condition_from_kwargs(y, mdm, cfg, B, J * F, Tn, torch.device("cpu"))
        gd._add_observations(cond, mdm, kw["model_kwargs"], B, J * F, Tn)
        calls.append({"shape": list(shape), "native_denoiser": type(mdm).__name__, "cfg": cfg is not None,
                      "kwargs": sorted(k for k in kw if k != "model_kwargs"),
                      "y_keys": sorted(kw["model_kwargs"]["y"]), "extra_model_kwargs": sorted(set(kw["model_kwargs"]) - {"y"}),
                      "condition": {k: (list(v.shape) if hasattr(v, "shape") else v) for k, v in cond.items()},
                      "diffusion": type(self).__module__ + "." + type(self).__name__, "n_steps": self.num_timesteps})
        return torch.from_numpy(rng.standard_normal(tuple(shape)).astype(np.float32))

    gd.GaussianDiffusion.p_sample_loop = recorded
    sys.argv = [f"sample/{script}.py"] + sys.argv[3:]
    try:
        mod.main()
    finally:
        gd.GaussianDiffusion.p_sample_loop = real
    print("CALLER_RESULT " + json.dumps({"calls": calls, "results": sorted(
        os.path.relpath(os.path.join(dp, f), workdir) for dp, _, fs in os.walk(workdir) for f in fs if f.startswith("results"))}))


if __name__ == "__main__":
    main()
