"""How far is bench.py's cpu_baseline ("kind": "port", oracle/torch_cpu_port.py) from the REAL reference on the same cores?
(VERDICT r3 task 8.)  Runs where /root/reference exists (the build container, not the GPU box):

    python tools/cpu_port_vs_reference.py [steps]      ->  profiles/r04_cpu_port_vs_reference.json

Both legs: BASELINE config 2's step (B=32 x 196 x 263, text CFG, ancestral update), same weights, same thread count, 1 warm-up
+ `steps` timed steps each, median step quoted.  Reference leg = diffusion.p_sample_loop_progressive of the imported reference
(model/cfg_sampler.py:25-35 around model/mdm.py, diffusion/gaussian_diffusion.py:1217-1297) with its noise injected.
"""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests" / "golden"))
import bench  # noqa: E402
import cases  # noqa: E402
from oracle import ref_shims, weights  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    B, T = 32, 196
    threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    sd = weights.make_state_dict(41, text=True)
    port = bench.cpu_baseline(sd, B, n_steps=n)

    ref = ref_shims.import_reference()
    args = ref_shims.default_args(unconstrained=False)
    model, _ = ref_shims.make_reference_model(ref, args, weights.to_torch(sd), cfg=True)
    diffusion = ref.respace.SpacedDiffusion(use_timesteps=ref.respace.space_timesteps(1000, [1000]),
                                            conf=ref.gd.DiffusionConfig(betas=ref.gd.get_named_beta_schedule("cosine", 1000)))
    rng = np.random.default_rng(1)
    shape = (B, 263, 1, T)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    ref_shims.set_text_embedding(t(rng.standard_normal((B, 512))))
    y = {"mask": torch.ones(B, 1, 1, T, dtype=torch.bool), "lengths": torch.full((B,), T), "text": ["a"] * B,
         "text_scale": torch.full((B,), 2.5)}
    torch.set_num_threads(port["cores"])
    stream = (t(rng.standard_normal(shape)) for _ in range(n + 2))
    per = []
    with ref_shims.injected_noise(stream):
        gen = diffusion.p_sample_loop_progressive(model, shape, noise=t(rng.standard_normal(shape)), clip_denoised=False,
                                                  model_kwargs={"y": y}, device=torch.device("cpu"))
        t0 = time.perf_counter()
        for i, _ in enumerate(gen):
            t1 = time.perf_counter()
            if i > 0:
                per.append(t1 - t0)
            t0 = t1
            if i == n:
                break
    per.sort()
    med = per[len(per) // 2] if len(per) % 2 else 0.5 * (per[len(per) // 2 - 1] + per[len(per) // 2])
    out = {"host_logical_cpus": threads, "threads_used": port["cores"], "timed_steps": n,
           "port": {k: port[k] for k in ("value", "step_s_min", "step_s_median", "step_s_max")},
           "reference": {"value": 1.0 / med, "step_s_min": per[0], "step_s_median": med, "step_s_max": per[-1]},
           "port_over_reference_steps_per_s": port["value"] * med,
           "note": "port = oracle/torch_cpu_port.py (what bench.py's cpu_baseline times on the GPU box, where /root/reference "
                   "does not exist); reference = the imported /root/reference sampler on the same weights and thread count"}
    (REPO / "profiles" / "r04_cpu_port_vs_reference.json").write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
