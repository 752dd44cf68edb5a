"""Where along the chain does the native U-Net part from the reference?  Error of sample 0's x_t at every stored checkpoint of a
tests/golden UNET_LONG case (big_unet: every 10 of 100 steps; long_unet: every 100 of 1000 — there also the reference's own fp32
chain against its float64 chain), per precision.   python tools/unet_chain_error.py big_unet [long_unet]"""
import importlib
import json
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))
sys.path.insert(0, str(REPO / "tests" / "golden"))
import cases  # noqa: E402
import test_gpu_parity as T  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


for name in sys.argv[1:] or ["big_unet"]:
    for precision in ("f16x3", "bf16x6"):
        case, inp, g, wrapped, diffusion, kw = T.unet_long_setup(cases, name, precision)
        at = {int(i): k for k, i in enumerate(g["dump_at"])}
        rows = []
        for i, out in enumerate(diffusion.p_sample_loop_progressive(wrapped, inp["draw0"].shape, **kw)):
            if i in at:
                x = out["sample"][:1].cpu().numpy()
                row = {"step": i + 1, "vs_fp32": rel(x, g["dumps"][at[i]])}
                if "dumps_f64" in g.files:
                    row["vs_f64"] = rel(x, g["dumps_f64"][at[i]])
                    row["reference_fp32_vs_f64"] = rel(g["dumps"][at[i]], g["dumps_f64"][at[i]])
                rows.append(row)
        print(json.dumps({"case": name, "precision": precision, "sample0": rows}))
        # the chain's own sensitivity: the same engine, x_T moved by ONE fp32 ulp — how far apart do two runs of the SAME arithmetic end?
        final = diffusion.p_sample_loop(wrapped, inp["draw0"].shape, **kw).cpu().numpy()
        kw2 = dict(kw, noise=torch.nextafter(kw["noise"], torch.full_like(kw["noise"], float("inf"))))
        final2 = diffusion.p_sample_loop(wrapped, inp["draw0"].shape, **kw2).cpu().numpy()
        keep = list(case.get("keep", range(case["B"])))
        per = {int(k): {"vs_reference_fp32": rel(final[k], g["final"][i]), "one_ulp_of_x_T": rel(final2[k], final[k])} for i, k in enumerate(keep)}
        if "f64_rows" in g.files:
            for j, r in enumerate(g["f64_rows"]):
                per[int(r)]["vs_f64"] = rel(final[int(r)], g["final_f64_rows"][j])
                per[int(r)]["reference_fp32_vs_f64"] = rel(g["final"][keep.index(int(r))], g["final_f64_rows"][j])
        elif "final_f64" in g.files:
            for k in keep:
                per[int(k)]["vs_f64"] = rel(final[k], g["final_f64"][k])
                per[int(k)]["reference_fp32_vs_f64"] = rel(g["final"][keep.index(k)], g["final_f64"][k])
        print(json.dumps({"case": name, "precision": precision, "final_per_sample": per}))
