"""Shared by tests/test_reference_callers.py and tests/golden/make_golden_callers.py: write the checkpoint directory a
reference sample script loads (``--model_path save/ckpt/model000000010.pt`` + its args.json) and run the script through
tests/helpers/run_reference_caller.py in a subprocess."""
import importlib
import json
import os
import subprocess
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent.parent
PKG = "diffusion-motion-inbetweening_amd"
for p in (str(REPO), str(REPO / "tests" / "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def caller_state_dict(model_args: dict, weight_seed: int) -> dict:
    """Deterministic weights for the module `create_model_and_diffusion(model_args)` builds (numpy, by state-dict name)."""
    from oracle import weights
    mu = importlib.import_module(f"{PKG}.utils.model_util")
    model, _ = mu.create_model_and_diffusion(SimpleNamespace(**model_args), None)
    own = {k: v for k, v in model.state_dict().items() if not k.startswith("clip_model.")}
    if model_args.get("arch", "trans_enc") == "trans_enc":
        # (fill_like would draw self_attn.in_proj_weight — a name that does not end in '.weight' — like a bias, std 0.1:
        # 16x the attention logits of a default-initialised layer, and the sampling chain turns chaotic)
        sd = weights.make_state_dict(weight_seed, n_layers=model_args.get("layers", 8), text=True)
        assert sorted(sd) == sorted(own), sorted(set(sd) ^ set(own))
        return sd
    sd = weights.fill_like({k: tuple(v.shape) for k, v in own.items()}, weight_seed)
    # positional tables are persistent buffers of the reference modules, i.e. part of every real checkpoint
    sd.update({k: v.numpy().copy() for k, v in own.items() if k.endswith(".pe")})
    return sd


def write_checkpoint(workdir: Path, model_args: dict, weight_seed: int) -> dict:
    sd = caller_state_dict(model_args, weight_seed)
    ck = workdir / "save" / "ckpt"
    ck.mkdir(parents=True)
    torch.save({"model": {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}}, ck / "model000000010.pt")
    (ck / "args.json").write_text(json.dumps(dict(model_args, abs_3d=True, latent_dim=512)))
    return sd


def run_script(workdir: Path, case: dict, mode: str, n_samples: int = 3, timeout: int = 1800) -> dict:
    """-> {"calls": [...], "results": [...]} printed by run_reference_caller.py; <workdir>/recorded_call.npz holds the call."""
    cmd = [sys.executable, str(REPO / "tests" / "helpers" / "run_reference_caller.py"), case["script"], str(workdir),
           "--model_path", "save/ckpt/model000000010.pt", "--num_samples", str(n_samples), "--num_repetitions", "1",
           "--output_dir", "out"] + list(case["cli"])
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", CALLER_MODE=mode, CALLER_SAMPLES=str(n_samples))
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = next(ln for ln in r.stdout.splitlines() if ln.startswith("CALLER_RESULT "))
    return json.loads(line[len("CALLER_RESULT "):])
