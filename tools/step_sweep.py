"""Time one CFG denoising step (B=32, T=196) under different engine knobs (env vars read at
cmdi_create): sequence groups / streams and per-GEMM tile variants."""
import os as _os; _os.environ.setdefault("CMDI_PROBES_LIB", "1")   # instrumented library (build.py --probes)
import importlib
import os
import sys
from pathlib import Path
from types import SimpleNamespace

import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
PKG = "diffusion-motion-inbetweening_amd"
sub = lambda n: importlib.import_module(f"{PKG}.{n}")


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    from oracle import weights
    dev = torch.device("cuda:0")
    B = int(os.environ.get("SWEEP_B", "32"))
    T = 196
    mu = sub("utils.model_util")
    model, _ = mu.create_model_and_diffusion(SimpleNamespace(dataset="humanml"), None)
    mu.load_model_wo_clip(model, weights.to_torch(weights.make_state_dict(0, text=True)))
    model.to(dev).eval()
    gd, rs = sub("diffusion.gaussian_diffusion"), sub("diffusion.respace")
    diff = rs.SpacedDiffusion(rs.space_timesteps(1000, [1000]),
                              gd.DiffusionConfig(betas=gd.get_named_beta_schedule("cosine", 1000)))
    enc = torch.randn(B, 512, device=dev)
    scale = torch.full((B,), 2.5, device=dev)
    ref = None
    configs = [dict()]
    for spec in sys.argv[1:]:
        configs.append(dict(kv.split("=") for kv in spec.split(",")))
    for cfg in configs:
        for k in list(os.environ):
            if k.startswith("CMDI_"):
                del os.environ[k]
        os.environ.update({f"CMDI_{k.upper()}": v for k, v in cfg.items()})
        model.invalidate_engine()
        e = model.engine(dev, max_batch=B, max_frames=T)
        e.set_schedule(diff.engine_tables())
        e.set_condition(batch=B, n_frames=T, cfg=True, enc_text=enc, text_scale=scale)
        x0 = e.randn((B, 263, 1, T), seed=1)
        x = x0.clone()
        e.step(x, 500, seed=1)
        if ref is None:
            ref = x.clone()
        same = torch.equal(ref, x)
        dt = timeit(lambda: e.step(x, 500, seed=1))
        print(f"{str(cfg):70s} {dt * 1e3:7.3f} ms  {1 / dt:6.1f} steps/s  bitwise_same={same}", flush=True)


if __name__ == "__main__":
    main()
