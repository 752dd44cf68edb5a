"""GPU-box diagnostics: device properties, GEMM tile sweep on the denoiser's shapes, per-kernel
timings of one denoiser pass.  Writes human-readable text to stdout (redirect into gpurun_out/)."""
import importlib
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


def timeit(fn, iters=20, warm=3):
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
    dev = torch.device("cuda:0")
    p = torch.cuda.get_device_properties(0)
    print("device:", p.name, "CUs:", p.multi_processor_count, "mem GB:", p.total_memory / 2**30,
          "LDS/block:", getattr(p, "shared_memory_per_block", "?"),
          "LDS/block optin:", getattr(p, "shared_memory_per_block_optin", "?"))
    eng = sub("engine")
    maps = open("/proc/self/maps").read()
    print("hip runtimes:", {l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l})

    print("\n== GEMM tile sweep (fp32 MFMA 32x32x2), TFLOP/s ==")
    M = 2 * 32 * 197
    for (m, n, k, name) in [(M, 1536, 512, "in_proj"), (M, 512, 512, "out_proj"),
                            (M, 1024, 512, "linear1"), (M, 512, 1024, "linear2"),
                            (2 * 256 * 197, 1536, 512, "in_proj B=256"), (4096, 4096, 4096, "4096^3")]:
        a = torch.randn(m, k, device=dev)
        w = torch.randn(n, k, device=dev)
        b = torch.randn(n, device=dev)
        row = []
        for tile in (1, 2, 3, 4, 5):
            dt = timeit(lambda: eng.gemm_nt(a, w, b, tile=tile), iters=10)
            row.append(f"t{tile}:{2.0 * m * n * k / dt / 1e12:6.1f}")
        ref = timeit(lambda: torch.nn.functional.linear(a, w, b), iters=10)
        print(f"{name:14s} M={m:6d} N={n:5d} K={k:5d}  " + "  ".join(row)
              + f"   rocBLAS(torch):{2.0 * m * n * k / ref / 1e12:6.1f}")

    print("\n== one CFG denoising step, B=32 T=196 ==")
    from oracle import weights
    mu = sub("utils.model_util")
    model, _ = mu.create_model_and_diffusion(SimpleNamespace(dataset="humanml"), None)
    mu.load_model_wo_clip(model, weights.to_torch(weights.make_state_dict(0, text=True)))
    model.to(dev).eval()
    B, T = 32, 196
    e = model.engine(dev, max_batch=B, max_frames=T)
    gd, rs = sub("diffusion.gaussian_diffusion"), sub("diffusion.respace")
    diff = rs.SpacedDiffusion(rs.space_timesteps(1000, [1000]),
                              gd.DiffusionConfig(betas=gd.get_named_beta_schedule("cosine", 1000)))
    e.set_schedule(diff.engine_tables())
    e.set_condition(batch=B, n_frames=T, cfg=True, enc_text=torch.randn(B, 512, device=dev),
                    text_scale=torch.full((B,), 2.5, device=dev))
    x = e.randn((B, 263, 1, T), seed=1)
    for tile in (0, 1, 2, 3, 4, 5):
        import os
        dt = timeit(lambda: e.step(x, 500, seed=1), iters=10)
        print(f"auto-tile step: {dt * 1e3:.3f} ms  ({1 / dt:.1f} steps/s, "
              f"{470.6e9 / dt / 1e12:.1f} TFLOP/s algorithmic)")
        break
    print("workspace MB:", e.workspace_bytes() / 2**20)


if __name__ == "__main__":
    main()
