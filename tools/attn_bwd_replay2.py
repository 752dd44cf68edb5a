"""Follow-up of tools/attn_bwd_replay.py: raw d qkv of layer 7 (cond pass) for the build container — both sequences, each
sequence alone, and the two in swapped order (is the outlier row tied to the data or to its position in the launch?)."""
import importlib, sys
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
eng = importlib.import_module("diffusion-motion-inbetweening_amd.engine")
z = np.load(REPO / "tools" / "data" / "tmp_attn_cases.npz")
dev = torch.device("cuda:0")
S, d = 61, 512
out = {}
for tag in ("c7", "c0", "u1"):
    qkv, dout = z["qkv_" + tag], z["dout_" + tag]
    run = lambda q, g, n: eng.attention_vjp_h3(torch.from_numpy(np.ascontiguousarray(q)).to(dev), torch.from_numpy(np.ascontiguousarray(g)).to(dev), n, S, 4).cpu().numpy()
    out[tag + "_both"] = run(qkv, dout, 2)
    out[tag + "_seq0"] = run(qkv[:S], dout[:S], 1)
    out[tag + "_seq1"] = run(qkv[S:], dout[S:], 1)
    out[tag + "_swap"] = run(np.concatenate([qkv[S:], qkv[:S]]), np.concatenate([dout[S:], dout[:S]]), 2)
    rep = np.concatenate([qkv, qkv, qkv]), np.concatenate([dout, dout, dout])
    out[tag + "_x3"] = run(rep[0], rep[1], 6)
np.savez_compressed(sys.argv[1], **out)
print("ok", {k: v.shape for k, v in out.items()})
