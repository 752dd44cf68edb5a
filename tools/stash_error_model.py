"""CPU model (float64 oracle + injected roundings) of what the split-row stash of the folded schedule can cost the guided VJP
(round 5, profiles/r05_guided_error_attribution.md section 3): the pre-LayerNorm sums, the attention output and the gradient
stream rounded to 22 bits (split f16) or to fp32 at the operating point of the guided chain's last step.  Answer: 4e-8 ... 6e-7 —
not the 1.2e-5 that was measured outside the keyframes.  CPU only, ~2 minutes."""
import sys
from pathlib import Path
import numpy as np
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "tests" / "golden"))
from oracle import mdm_oracle as mo, diffusion_oracle as do, weights
import cases
mo.F32 = np.float64; do.F32 = np.float64
F = np.float64

def q22(x):      # split-f16 round trip: hi = f16(x), lo = f16((x - hi) 2^11)
    x32 = np.asarray(x, np.float32)
    hi = x32.astype(np.float16)
    lo = ((x32 - hi.astype(np.float32)) * np.float32(2048)).astype(np.float16)
    return hi.astype(np.float64) + lo.astype(np.float64) / 2048.0

def q32(x): return np.asarray(x, np.float32).astype(np.float64)

class Model(mo.MDMOracle):
    mode = {}
    def _layer(self, l, h, keep):
        sd, H, d = self.sd, self.H, self.d
        p = f"seqTransEncoder.layers.{l}."
        B, S, _ = h.shape
        dh = d // H
        qkv = mo._linear(h, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"])
        q, k, v = (qkv[..., i * d:(i + 1) * d].reshape(B, S, H, dh).transpose(0, 2, 1, 3) for i in range(3))
        scale = F(1.0 / np.sqrt(dh))
        P = mo._softmax((q * scale) @ k.transpose(0, 1, 3, 2))
        o4 = P @ v
        o = o4.transpose(0, 2, 1, 3).reshape(B, S, d)
        a = mo._linear(o, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        pre1 = h + a
        h1, xh1, r1 = mo._layernorm(pre1, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
        u = mo._linear(h1, sd[p + "linear1.weight"], sd[p + "linear1.bias"])
        ff = mo._linear(mo._gelu(u), sd[p + "linear2.weight"], sd[p + "linear2.bias"])
        pre2 = h1 + ff
        out, xh2, r2 = mo._layernorm(pre2, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
        if keep is not None:
            qq = self.mode.get("pre")
            if qq:      # x-hat of the backward from the ROUNDED pre-LN sums, (mean, rstd) of the unrounded ones
                m1 = pre1.mean(-1, keepdims=True); m2 = pre2.mean(-1, keepdims=True)
                xh1 = (qq(pre1) - m1) * r1; xh2 = (qq(pre2) - m2) * r2
            keep.append(dict(q=q, k=k, v=v, P=P, xh1=xh1, r1=r1, u=u, xh2=xh2, r2=r2, scale=scale, o4=o4))
        return out
    def vjp(self, x, t, gout, enc_text=None, uncond=False):
        sd, H, d = self.sd, self.H, self.d
        keep = []
        x = np.asarray(x, dtype=F)
        B, J, Fd, T = x.shape
        self.forward(x, t, enc_text, uncond, keep=keep)
        g = np.asarray(gout, dtype=F).reshape(B, J * Fd, T).transpose(0, 2, 1)
        dh_ = np.zeros((B, T + 1, d), dtype=F)
        dh_[:, 1:, :] = g @ sd["output_process.poseFinal.weight"]
        qg = self.mode.get("grad")      # rounding of the gradient stream between kernels (both schedules have it)
        for l in reversed(range(self.L)):
            st = keep[l]
            p = f"seqTransEncoder.layers.{l}."
            dpre2 = mo._layernorm_bwd(dh_, st["xh2"], st["r2"], sd[p + "norm2.weight"])
            if qg: dpre2 = qg(dpre2)
            du = (dpre2 @ sd[p + "linear2.weight"]) * mo._gelu_grad(st["u"])
            if qg: du = qg(du)
            dh1 = dpre2 + du @ sd[p + "linear1.weight"]
            dpre1 = mo._layernorm_bwd(dh1, st["xh1"], st["r1"], sd[p + "norm1.weight"])
            if qg: dpre1 = qg(dpre1)
            dout = (dpre1 @ sd[p + "self_attn.out_proj.weight"])
            if qg: dout = qg(dout)
            S = T + 1
            dout = dout.reshape(B, S, H, d // H).transpose(0, 2, 1, 3)
            P, q, k, v, scale = st["P"], st["q"], st["k"], st["v"], st["scale"]
            dv = P.transpose(0, 1, 3, 2) @ dout
            dP = dout @ v.transpose(0, 1, 3, 2)
            qo = self.mode.get("o")
            if qo: D = (dout * qo(st["o4"])).sum(-1, keepdims=True)
            else: D = (dP * P).sum(axis=-1, keepdims=True)
            dS = P * (dP - D)
            dq = (dS @ k) * scale
            dk = (dS.transpose(0, 1, 3, 2) @ q) * scale
            dqkv = np.concatenate([m.transpose(0, 2, 1, 3).reshape(B, S, d) for m in (dq, dk, dv)], axis=-1)
            if qg: dqkv = qg(dqkv)
            dh_ = dpre1 + dqkv @ sd[p + "self_attn.in_proj_weight"]
        gx = dh_[:, 1:, :] @ sd["input_process.poseEmbedding.weight"]
        return gx.transpose(0, 2, 1).reshape(B, J, Fd, T)

w = np.load(REPO / "tools" / "data" / "recon_chain_wants.npz")
sd = weights.make_state_dict(3, text=True)
lengths = np.array([60, 44]); T = 60
len_mask = (np.arange(T)[None] < lengths[:, None]).reshape(2, 1, 1, T)
m = np.broadcast_to(cases.sparse_keyframe_mask(lengths, T, 5) & len_mask, (2, 263, 1, T))
rng = np.random.default_rng(0); shape=(2,263,1,60)
_ = rng.standard_normal(shape); _ = rng.standard_normal(shape); _ = rng.standard_normal((3,)+shape)
enc = rng.standard_normal((2, 512)).astype(np.float32).astype(F); sc = np.array([2.5, 2.5])
x1, t0, seed, g64 = w["op_x1"].astype(F), w["op_t"], w["op_seed"].astype(F), w["op_g_f64"]
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
for name, mode in (("none", {}), ("pre->22 bits", {"pre": q22}), ("pre->fp32", {"pre": q32}), ("O->22 bits", {"o": q22}), ("O->fp32", {"o": q32}),
                   ("grad->22 bits", {"grad": q22}), ("grad->fp32", {"grad": q32})):
    Model.mode = mode
    g = Model(sd).vjp_cfg(x1, t0, seed, enc, sc)
    print(f"{name:>14}: VJP all {rel(g, g64):.3e}  outside keyframes {rel(g[~m], g64[~m]):.3e}  on keyframes {rel(g[m], g64[m]):.3e}", flush=True)
