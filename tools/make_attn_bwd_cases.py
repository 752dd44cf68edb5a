"""Per-layer inputs of the attention backward at the guided chain's operating point (float64 oracle), for an isolated
replay through cmdi_attention_vjp_h3 on the GPU
(tools/attn_bwd_replay.py).  CPU only, ~1 minute; writes tools/data/tmp_attn_cases.npz (37 MB, git-ignored: it travels to the GPU
box with the working tree)."""
import sys
from pathlib import Path
import numpy as np
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "tests" / "golden"))
from oracle import mdm_oracle as mo, weights
mo.F32 = np.float64
F = np.float64
w = np.load(REPO / "tools" / "data" / "recon_chain_wants.npz")
sd = weights.make_state_dict(3, text=True)
rng = np.random.default_rng(0); shape=(2,263,1,60)
_ = rng.standard_normal(shape); _ = rng.standard_normal(shape); _ = rng.standard_normal((3,)+shape)
enc = rng.standard_normal((2, 512)).astype(np.float32).astype(F)
x1, t0, seed = w["op_x1"].astype(F), w["op_t"], w["op_seed"].astype(F)
rec = {}
class Model(mo.MDMOracle):
    def vjp(self, x, t, gout, enc_text=None, uncond=False):
        sd, H, d = self.sd, self.H, self.d
        keep = []
        x = np.asarray(x, dtype=F)
        B, J, Fd, T = x.shape
        self.forward(x, t, enc_text, uncond, keep=keep)
        g = np.asarray(gout, dtype=F).reshape(B, J * Fd, T).transpose(0, 2, 1)
        dh_ = np.zeros((B, T + 1, d), dtype=F)
        dh_[:, 1:, :] = g @ sd["output_process.poseFinal.weight"]
        for l in reversed(range(self.L)):
            st = keep[l]
            p = f"seqTransEncoder.layers.{l}."
            dpre2 = mo._layernorm_bwd(dh_, st["xh2"], st["r2"], sd[p + "norm2.weight"])
            du = (dpre2 @ sd[p + "linear2.weight"]) * mo._gelu_grad(st["u"])
            dh1 = dpre2 + du @ sd[p + "linear1.weight"]
            dpre1 = mo._layernorm_bwd(dh1, st["xh1"], st["r1"], sd[p + "norm1.weight"])
            dout = (dpre1 @ sd[p + "self_attn.out_proj.weight"])
            S = T + 1
            do4 = dout.reshape(B, S, H, d // H).transpose(0, 2, 1, 3)
            P, q, k, v, scale = st["P"], st["q"], st["k"], st["v"], st["scale"]
            dv = P.transpose(0, 1, 3, 2) @ do4
            dP = do4 @ v.transpose(0, 1, 3, 2)
            dS = P * (dP - (dP * P).sum(axis=-1, keepdims=True))
            dq = (dS @ k) * scale
            dk = (dS.transpose(0, 1, 3, 2) @ q) * scale
            dqkv = np.concatenate([m.transpose(0, 2, 1, 3).reshape(B, S, d) for m in (dq, dk, dv)], axis=-1)
            qkv = np.concatenate([m.transpose(0, 2, 1, 3).reshape(B, S, d) for m in (q, k, v)], axis=-1)
            tag = ("u" if uncond else "c") + str(l)
            rec["qkv_" + tag] = qkv.reshape(B * S, 3 * d).astype(np.float32)
            rec["dout_" + tag] = dout.reshape(B * S, d).astype(np.float32)
            rec["dqkv_" + tag] = dqkv.reshape(B * S, 3 * d)           # float64 truth for the float64 inputs
            rec["pmax_" + tag] = P.max(-1).reshape(B, H, S)
            dh_ = dpre1 + dqkv @ sd[p + "self_attn.in_proj_weight"]
        gx = dh_[:, 1:, :] @ sd["input_process.poseEmbedding.weight"]
        return gx.transpose(0, 2, 1).reshape(B, J, Fd, T)
g = Model(sd).vjp_cfg(x1, t0, seed, enc, np.array([2.5, 2.5]))
print("check vs wants:", np.linalg.norm(g - w["op_g_f64"]) / np.linalg.norm(w["op_g_f64"]))
# truth for the ROUNDED (fp32) inputs the GPU will see: recompute the attention VJP in float64 from the fp32-rounded qkv / dout
def attn_vjp64(qkv, dout, B=2, S=61, H=4, d=512):
    qkv = qkv.astype(F).reshape(B, S, 3 * d); dout = dout.astype(F).reshape(B, S, d)
    q, k, v = (qkv[..., i * d:(i + 1) * d].reshape(B, S, H, d // H).transpose(0, 2, 1, 3) for i in range(3))
    scale = 1.0 / np.sqrt(d // H)
    s = (q * scale) @ k.transpose(0, 1, 3, 2)
    P = np.exp(s - s.max(-1, keepdims=True)); P /= P.sum(-1, keepdims=True)
    do4 = dout.reshape(B, S, H, d // H).transpose(0, 2, 1, 3)
    dv = P.transpose(0, 1, 3, 2) @ do4
    dP = do4 @ v.transpose(0, 1, 3, 2)
    dS = P * (dP - (dP * P).sum(-1, keepdims=True))
    dq = (dS @ k) * scale; dk = (dS.transpose(0, 1, 3, 2) @ q) * scale
    return np.concatenate([m.transpose(0, 2, 1, 3).reshape(B, S, d) for m in (dq, dk, dv)], axis=-1).reshape(B * S, 3 * d)
out = {}
for l in range(8):
    for c in "cu":
        tag = c + str(l)
        out["qkv_" + tag] = rec["qkv_" + tag]; out["dout_" + tag] = rec["dout_" + tag]
        out["want_" + tag] = attn_vjp64(rec["qkv_" + tag], rec["dout_" + tag])
        out["pmax_" + tag] = rec["pmax_" + tag]
np.savez_compressed(REPO / "tools" / "data" / "tmp_attn_cases.npz", **out)
pm = np.stack([rec["pmax_c%d" % l] for l in range(8)])
print("max softmax probability per row: median %.3f, 99%% %.3f, max %.3f" % (np.median(pm), np.quantile(pm, .99), pm.max()))
for l in range(8):
    a = rec["pmax_c%d" % l]; i = np.unravel_index(np.argmax(a), a.shape); print(" layer", l, "max P", a.max(), "at (seq, head, token)", i)
