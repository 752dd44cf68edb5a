"""Per-call cost of cmdi_sample_loop: wall time of K-step chains, K = 1 .. 64, T(K) = a + b K (least squares); `a` is what a
short chain pays on top of its steps.  Runs the default engine and CMDI_GROUPS=1 (one pipeline) in subprocesses."""
import importlib, os, subprocess, sys, time
from pathlib import Path
import numpy as np
REPO = Path(__file__).resolve().parent.parent
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, str(REPO))
    import bench
    sub = lambda n: importlib.import_module("diffusion-motion-inbetweening_amd." + n)
    gd, rs = sub("diffusion.gaussian_diffusion"), sub("diffusion.respace")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    model, sd = bench.build_model(True, dev)
    diffusion = rs.SpacedDiffusion(rs.space_timesteps(1000, [1000]), gd.DiffusionConfig(betas=gd.get_named_beta_schedule("cosine", 1000)))
    B, T = int(os.environ.get("OVH_B", "32")), 196
    eng = model.engine(dev, max_batch=B, max_frames=T)
    eng.set_schedule(diffusion.engine_tables(), key=None)
    g = torch.Generator().manual_seed(1)
    eng.set_condition(batch=B, n_frames=T, cfg=True, enc_text=torch.randn(B, 512, generator=g).to(dev), text_scale=torch.full((B,), 2.5, device=dev))
    x = eng.randn((B, 263, 1, T), seed=1)
    eng.sample_loop(x, 999, 960, seed=1)      # warm
    torch.cuda.synchronize()
    Ks, Ts = [1, 2, 4, 8, 16, 32, 64], []
    for K in Ks:
        best = []
        for rep in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.sample_loop(x, 900, 900 - K + 1, seed=1)
            torch.cuda.synchronize()
            best.append((time.perf_counter() - t0) * 1e3)
        Ts.append(float(np.median(best)))
    b, a = np.polyfit(Ks, Ts, 1)
    print(f"{os.environ.get('OVH_TAG', ''):28s} B={B} parts={eng.pipeline_parts()}  per-call a = {a:6.3f} ms, per-step b = {b:6.3f} ms   T(K) ms: " +
          " ".join(f"K={k}:{t:.2f}" for k, t in zip(Ks, Ts)), flush=True)
else:
    for tag, env in (("default", {}), ("CMDI_GROUPS=1", {"CMDI_GROUPS": "1"}), ("CMDI_PIPELINES=0", {"CMDI_PIPELINES": "0"}),
                     ("CMDI_GRAPH=1", {"CMDI_GRAPH": "1"})) + tuple((k, v) for k, v in ()):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, OVH_TAG=tag, **env))
