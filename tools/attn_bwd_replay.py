"""Isolated replay of the attention input-VJP kernels (cmdi_attention_vjp_h3) on the per-layer (qkv, d out) of the guided chain's
operating point (tools/data/tmp_attn_cases.npz, written by a float64 oracle run on the build container): per-row error of
d q | d k | d v against the float64 VJP of the SAME fp32 inputs.  Localises row-level error events of the f16x3 backward."""
import importlib, sys
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
eng = importlib.import_module("diffusion-motion-inbetweening_amd.engine")
z = np.load(REPO / "tools" / "data" / "tmp_attn_cases.npz")
w = np.load(REPO / "tools" / "data" / "recon_chain_wants.npz")
dev = torch.device("cuda:0")
S, d = 61, 512
gmax = float(np.abs(2.5 * w["op_seed"]).max())
scale = 2.0 ** (6 - int(np.floor(np.log2(gmax))))
print(f"# gradient scale of the chain at this point: 2^{int(np.log2(scale))} (max |gout| {gmax:.3e})")
for use_scale in (True, False):
    print(f"# d out {'x the chain scale' if use_scale else 'unscaled'}")
    for l in range(7, -1, -1):
        for c in "cu":
            tag = c + str(l)
            k = scale if use_scale else 1.0
            qkv = torch.from_numpy(z["qkv_" + tag]).to(dev)
            dout = torch.from_numpy((z["dout_" + tag].astype(np.float64) * k).astype(np.float32)).to(dev)
            got = eng.attention_vjp_h3(qkv, dout, 2, S, 4).cpu().numpy().astype(np.float64) / k
            want = z["want_" + tag]
            parts = []
            for name, sl in (("dq", slice(0, d)), ("dk", slice(d, 2 * d)), ("dv", slice(2 * d, 3 * d))):
                e = np.linalg.norm(got[:, sl] - want[:, sl], axis=1)
                n = np.linalg.norm(want[:, sl], axis=1)
                tot = np.linalg.norm(got[:, sl] - want[:, sl]) / np.linalg.norm(want[:, sl])
                worst = np.argsort(-e)[:3]
                parts.append(f"{name} {tot:.2e} (row abs err median {np.median(e):.1e} max {e.max():.1e} at "
                             f"{[(int(r) // S, int(r) % S) for r in worst]}, |row| there {n[worst[0]]:.1e} median {np.median(n):.1e})")
            dmax = float(np.abs(z["dout_" + tag]).max() * k); dmed = float(np.median(np.abs(z["dout_" + tag])) * k)
            print(f" layer {l} {c}: |d out| max {dmax:.2e} median {dmed:.2e} | " + " | ".join(parts), flush=True)
