"""How long does the chain take to reach its steady step time?  One engine, consecutive cmdi_sample_loop calls of 5 steps
each right after engine creation (what bench.py --steps 20 --warmup 5 sees), each call timed with a device sync."""
import importlib, sys, time
from pathlib import Path
import numpy as np
import torch
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import bench
sub = lambda n: importlib.import_module("diffusion-motion-inbetweening_amd." + n)
gd, rs, N = sub("diffusion.gaussian_diffusion"), sub("diffusion.respace"), sub("_native")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
model, sd = bench.build_model(True, dev)
diffusion = rs.SpacedDiffusion(rs.space_timesteps(1000, [1000]), gd.DiffusionConfig(betas=gd.get_named_beta_schedule("cosine", 1000)))
B, T = 32, 196
eng = model.engine(dev, max_batch=B, max_frames=T)
eng.set_schedule(diffusion.engine_tables(), key=None)
g = torch.Generator().manual_seed(1)
eng.set_condition(batch=B, n_frames=T, cfg=True, enc_text=torch.randn(B, 512, generator=g).to(dev), text_scale=torch.full((B,), 2.5, device=dev))
x = eng.randn((B, 263, 1, T), seed=1)
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 5
out = []
step = 999
torch.cuda.synchronize()
for i in range(40):
    t0 = time.perf_counter()
    eng.sample_loop(x, step, step - chunk + 1, seed=1)
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / chunk * 1e3)
    step -= chunk
print("ms/step per consecutive %d-step call:" % chunk, " ".join(f"{v:.3f}" for v in out))
