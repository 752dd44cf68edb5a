"""Where the guided (reconstruction-guidance) path's error comes from, per arithmetic mode (VERDICT r4 task 1b).

Truth = the numpy oracle run in float64.  Compared with it, on the same inputs:
  * the numpy oracle in float32 (what smoke() and the fp32 goldens' arithmetic are),
  * the native engine in each precision mode: f16x3 (default, 22-bit split operands + split stash), bf16x6 (exact operands),
    f32 (fp32 MFMA, fp32 stash).
Stage 1: ONE evaluation — CFG forward and the input-VJP of the reconstruction loss seed (the two model-side ingredients of a
         guided step), each mode vs float64.
Stage 2: guided chains (B=2, T=60, CFG 2.5, 'benchmark_sparse' keyframes, imputation + guidance weight 20; weight 0 as the
         control) of 3 / 10 / 100 steps — the LAST n steps of the respaced chain from a noised init_image.
The float64 / float32 oracle results ("wants") take minutes of CPU per chain, so they are computed ONCE on the build container
and shipped (tools/data/recon_chain_wants.npz); the GPU box only runs the engines:
   python tools/recon_chain_error.py --make-wants            # CPU, here (~30 min)
   python tools/recon_chain_error.py > gpurun_out/recon_chain_error.txt      # GPU box
"""
import argparse
import importlib
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "tests" / "golden"))
import cases
from oracle import diffusion_oracle as do, mdm_oracle as mo, weights

PKG = "diffusion-motion-inbetweening_amd"
sub = lambda n: importlib.import_module(f"{PKG}.{n}")
dev = torch.device("cuda:0")
MODES = ("f16x3", "bf16x6", "f32")
WANTS = REPO / "tools" / "data" / "recon_chain_wants.npz"
ENV_TAG = "".join(f" [{k}={v}]" for k, v in sorted(__import__("os").environ.items()) if k in ("CMDI_LN_FOLD_KEEP", "CMDI_LN_FOLD", "CMDI_GROUPS", "CMDI_STASH_F32"))
PLAN = ((3, 20.0), (3, 0.0), (10, 20.0), (10, 0.0), (100, 20.0))
f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))


def oracle_dtype(dt):
    do.F32 = dt
    mo.F32 = dt


def native_model(sd, precision):
    mu = sub("utils.model_util")
    model, _ = mu.create_model_and_diffusion(SimpleNamespace(dataset="humanml"), None)
    mu.load_model_wo_clip(model, weights.to_torch(sd))
    model.native_precision = precision
    return sub("model.cfg_sampler").ClassifierFreeSampleModel(model.to(dev).eval())


def inputs(seed, n_noise):
    rng = np.random.default_rng(seed)
    B, T = 2, 60
    shape = (B, 263, 1, T)
    d = dict(shape=shape, x_T=f32(rng.standard_normal(shape)), x0=f32(rng.standard_normal(shape)),
             noise=f32(rng.standard_normal((n_noise,) + shape)), enc=f32(rng.standard_normal((B, 512))), scale=f32([2.5, 2.5]),
             lengths=np.array([60, 44]))
    d["len_mask"] = (np.arange(T)[None] < d["lengths"][:, None]).reshape(B, 1, 1, T)
    d["kf_mask"] = cases.sparse_keyframe_mask(d["lengths"], T, 5)
    return d


def eval_inputs():
    d = inputs(1, 1)
    gout = f32(np.random.default_rng(2).standard_normal(d["shape"]) * (d["kf_mask"] & d["len_mask"]))
    return d, d["x_T"], np.array([500, 500]), gout


def eval_wants(sd):
    d, x, t, gout = eval_inputs()
    out = {}
    for name, dt in (("f32", np.float32), ("f64", np.float64)):
        oracle_dtype(dt)
        orc = mo.MDMOracle(sd)
        out[f"eval_fwd_{name}"] = np.asarray(orc.forward_cfg(x.astype(dt), t, d["enc"].astype(dt), d["scale"].astype(dt))[0], np.float64)
        out[f"eval_vjp_{name}"] = np.asarray(orc.vjp_cfg(x.astype(dt), t, gout.astype(dt), d["enc"].astype(dt), d["scale"].astype(dt)), np.float64)
    oracle_dtype(np.float32)
    return out


def one_evaluation(sd, want, modes=MODES):
    """CFG forward + input-VJP of a keyframe-masked output gradient at t = 500."""
    d, x, t, gout = eval_inputs()
    B, T = 2, 60
    print("# stage 1: one CFG evaluation + input-VJP, rel-L2 vs the float64 oracle")
    print(f"{'numpy fp32 oracle':>18}: forward {rel(want['eval_fwd_f32'], want['eval_fwd_f64']):.3e} | "
          f"VJP {rel(want['eval_vjp_f32'], want['eval_vjp_f64']):.3e}")
    for mode in modes:
        model = native_model(sd, mode)
        eng = model.model.engine(dev, max_batch=B, max_frames=T, want_grad=True)
        eng.set_condition(batch=B, n_frames=T, cfg=True, enc_text=tt(d["enc"]), text_scale=tt(d["scale"]))
        out = eng.mdm_forward(tt(x), tt(t)).cpu().numpy()
        gx = eng.mdm_vjp(tt(gout)).cpu().numpy()
        assert eng.precision == mode
        print(f"{'engine ' + mode:>18}: forward {rel(out, want['eval_fwd_f64']):.3e} | VJP {rel(gx, want['eval_vjp_f64']):.3e}", flush=True)


def opoint_wants(sd, n=3, w=20.0):
    """The operating point of the guided chain's LAST step (t = 0) on the float64 chain: state x_1, the model output there,
    the reconstruction-loss seed and its input-VJP — float64 (truth) and float32 oracle arithmetic at the SAME state and seed."""
    d = inputs(0, n)
    sch = do.Schedule(do.named_betas("cosine", 1000), do.space_timesteps(1000, [10]))
    mask = d["kf_mask"] & d["len_mask"]
    oracle_dtype(np.float64)
    orc = mo.MDMOracle(sd)
    x = do.q_sample(sch, n - 1, d["x0"], d["x_T"])
    for k, i in enumerate(range(n - 1, 0, -1)):          # every step but the last
        t = np.full((2,), sch.timestep_map[i], dtype=np.int64)
        hat = orc.forward_cfg(x, t, d["enc"], d["scale"])[0]
        seed = do.recon_loss_grad_seed(hat, mask, d["x0"])
        grad = orc.vjp_cfg(x, t, seed, d["enc"], d["scale"])
        x, _ = do.step_update(sch, i, x, hat, d["noise"][k], mask=mask, inpaint=d["x0"], impute=True, recon=True, grad=grad,
                              recon_w=np.float64(w))
    t0 = np.full((2,), sch.timestep_map[0], dtype=np.int64)
    x1 = np.asarray(x, np.float32)                        # the state every engine is handed
    out = {"op_x1": x1, "op_t": t0, "op_k": np.float64(w) * sch.sqrt_ab[0] / 2.0}
    for name, dt in (("f64", np.float64), ("f32", np.float32)):
        oracle_dtype(dt)
        orc = mo.MDMOracle(sd)
        hat = orc.forward_cfg(x1.astype(dt), t0, d["enc"].astype(dt), d["scale"].astype(dt))[0]
        if name == "f64":
            out["op_seed"] = np.asarray(do.recon_loss_grad_seed(hat, mask, d["x0"]), np.float32)   # ONE seed for every mode
        out[f"op_hat_{name}"] = np.asarray(hat, np.float64)
        out[f"op_g_{name}"] = np.asarray(orc.vjp_cfg(x1.astype(dt), t0, out["op_seed"].astype(dt), d["enc"].astype(dt),
                                                     d["scale"].astype(dt)), np.float64)
    oracle_dtype(np.float32)
    return out


def operating_point(sd, want, modes):
    """Stage 1b: the two model-side ingredients of the LAST guided step at the chain's own state — and the VJP error where the
    sampler uses it: the guidance term is g * (1 - mask), i.e. ONLY the entries outside the keyframes, where g is small."""
    d = inputs(0, 3)
    B, T = 2, 60
    m = np.broadcast_to(d["kf_mask"] & d["len_mask"], d["shape"])
    x1, t0, seed, k = want["op_x1"], want["op_t"], want["op_seed"], float(want["op_k"])
    g64, h64 = want["op_g_f64"], want["op_hat_f64"]
    print(f"# stage 1b: last guided step (t = {int(t0[0])}) at the float64 chain's state.  |k g (1-m)| / |x_hat| = "
          f"{k * np.linalg.norm(g64[~m]) / np.linalg.norm(h64):.3f}; |g| on keyframe entries / elsewhere = "
          f"{np.sqrt(np.mean(g64[m] ** 2)) / np.sqrt(np.mean(g64[~m] ** 2)):.1f}")

    def row(name, hat, g):
        tilde = (hat - k * g * (~m)) * (~m) + hat * m
        tilde64 = (h64 - k * g64 * (~m)) * (~m) + h64 * m
        print(f"{name:>26}: forward {rel(hat, h64):.3e} | VJP all {rel(g, g64):.3e}  outside keyframes {rel(g[~m], g64[~m]):.3e}  "
              f"on keyframes {rel(g[m], g64[m]):.3e} | x0 estimate of the step {rel(tilde, tilde64):.3e}", flush=True)

    row("numpy fp32 oracle", want["op_hat_f32"], want["op_g_f32"])
    for mode in modes:
        model = native_model(sd, mode)
        eng = model.model.engine(dev, max_batch=B, max_frames=T, want_grad=True)
        eng.set_condition(batch=B, n_frames=T, cfg=True, enc_text=tt(d["enc"]), text_scale=tt(d["scale"]))
        hat = eng.mdm_forward(tt(x1), tt(t0)).cpu().numpy().astype(np.float64)
        g = eng.mdm_vjp(tt(seed)).cpu().numpy().astype(np.float64)
        row("engine " + mode + ENV_TAG, hat, g)
        dump = __import__("os").environ.get("RECON_DUMP")
        if dump:      # the raw tensors, for an error map on the build container
            np.savez_compressed(f"{dump}_{mode}.npz", hat=hat, g=g)


def stash_audit(sd, want):
    """Probes library + CMDI_STASH_F32=6: read back, per layer, the (mean, rstd) the stashing forward handed to the LayerNorm
    backward and the fp32 pre-LayerNorm rows, recompute the statistics in float64 on the host, list the rows that disagree."""
    import ctypes
    N = sub("_native")
    lib = N.load()
    if not hasattr(lib, "cmdi_probe_read_stash"):
        print("# stash audit needs the probes library (CMDI_PROBES_LIB=1)")
        return
    lib.cmdi_probe_read_stash.restype = ctypes.c_int64
    lib.cmdi_probe_read_stash.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64]
    d = inputs(0, 3)
    B, T, S, dm = 2, 60, 61, 512
    M = 2 * B * S
    model = native_model(sd, "f16x3")
    eng = model.model.engine(dev, max_batch=B, max_frames=T, want_grad=True)
    eng.set_condition(batch=B, n_frames=T, cfg=True, enc_text=tt(d["enc"]), text_scale=tt(d["scale"]))
    eng.mdm_forward(tt(want["op_x1"]), tt(want["op_t"]))

    def read(layer, which, n):
        buf = np.empty(n, np.float32)
        got = lib.cmdi_probe_read_stash(eng._h, layer, which, buf.ctypes.data, n)
        return buf if got == n else None

    print(f"# stash audit{ENV_TAG}: (mean, rstd) handed to the LayerNorm backward vs float64 statistics of the stashed fp32 rows")
    for layer in range(8):
        for which, pre_id, name in ((0, 2, "norm1"), (1, 3, "norm2")):
            st, pre = read(layer, which, 2 * M), read(layer, pre_id, M * dm)
            if st is None or pre is None:
                print(f" layer {layer} {name}: not available (CMDI_STASH_F32 & {2 if which == 0 else 4}?)")
                continue
            st = st.reshape(M, 2).astype(np.float64)
            x = pre.reshape(M, dm).astype(np.float64)
            mean, rstd = x.mean(1), 1.0 / np.sqrt(x.var(1) + 1e-5)
            em = np.abs(st[:, 0] - mean) * rstd             # mean error in units of sigma
            er = np.abs(st[:, 1] / rstd - 1.0)
            worst = np.argsort(-np.maximum(em, er))[:4]
            print(f" layer {layer} {name}: |d mean|/sigma median {np.median(em):.2e} max {em.max():.2e} | |d rstd|/rstd median {np.median(er):.2e} "
                  f"max {er.max():.2e} | worst rows (seq, token) {[(int(r) // S, int(r) % S) for r in worst]}", flush=True)


def stash_audit2(sd, want):
    """Second audit (probes library, CMDI_STASH_F32=6): the stash tensors the forward pass itself does not read back — the FFN
    pre-activation (aux), the split qkv rows and the softmax row statistics — against float64 values recomputed on the host
    from the stashed fp32 pre-LayerNorm rows.  Per layer: rel-L2 and the rows with the largest error."""
    import ctypes
    N = sub("_native")
    lib = N.load()
    if not hasattr(lib, "cmdi_probe_read_stash"):
        print("# stash audit needs the probes library (CMDI_PROBES_LIB=1)")
        return
    lib.cmdi_probe_read_stash.restype = ctypes.c_int64
    lib.cmdi_probe_read_stash.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64]
    d = inputs(0, 3)
    B, T, S, dm, f, H = 2, 60, 61, 512, 1024, 4
    nseq = 2 * B
    M = nseq * S
    model = native_model(sd, "f16x3")
    eng = model.model.engine(dev, max_batch=B, max_frames=T, want_grad=True)
    eng.set_condition(batch=B, n_frames=T, cfg=True, enc_text=tt(d["enc"]), text_scale=tt(d["scale"]))
    eng.mdm_forward(tt(want["op_x1"]), tt(want["op_t"]))

    def read(layer, which, n):
        buf = np.empty(n, np.float32)
        got = lib.cmdi_probe_read_stash(eng._h, layer, which, buf.ctypes.data, n)
        return buf if got == n else None

    def ln(x, g, b):
        mu = x.mean(1, keepdims=True)
        return (x - mu) / np.sqrt(x.var(1, keepdims=True) + 1e-5) * g + b

    def unsplit(raw, cols):          # [M][2 cols] halves in chunks of 32 (hi | lo) -> float64 [M][cols]
        h = raw.view(np.float16).reshape(M, cols // 32, 2, 32).astype(np.float64)
        return (h[:, :, 0, :] + h[:, :, 1, :] / 2048.0).reshape(M, cols)

    def report(name, got, ref):
        err = np.linalg.norm(got - ref, axis=1) / np.linalg.norm(ref, axis=1)
        worst = np.argsort(-err)[:4]
        print(f"   {name}: rel-L2 {np.linalg.norm(got - ref) / np.linalg.norm(ref):.2e} | per-row median {np.median(err):.2e} max {err.max():.2e} at "
              f"(seq, token) {[(int(r) // S, int(r) % S) for r in worst]}", flush=True)

    g64 = lambda k: np.asarray(sd[k], np.float64)
    print(f"# stash audit 2{ENV_TAG}: aux / qkv / softmax statistics of the stashing forward vs float64 recomputation from the stashed rows")
    for layer in range(8):
        p = f"seqTransEncoder.layers.{layer}."
        pre1, aux = read(layer, 2, M * dm), read(layer, 5, M * f)
        print(f" layer {layer}:")
        if pre1 is not None and aux is not None:
            h1 = ln(pre1.reshape(M, dm).astype(np.float64), g64(p + "norm1.weight"), g64(p + "norm1.bias"))
            report("aux (FFN pre-activation)", aux.reshape(M, f).astype(np.float64), h1 @ g64(p + "linear1.weight").T + g64(p + "linear1.bias"))
        qraw, rs = read(layer, 6, M * 3 * dm), read(layer, 7, nseq * H * S * 2)
        if layer > 0 and qraw is not None:
            pre2 = read(layer - 1, 3, M * dm)
            pp = f"seqTransEncoder.layers.{layer - 1}."
            hin = ln(pre2.reshape(M, dm).astype(np.float64), g64(pp + "norm2.weight"), g64(pp + "norm2.bias"))
            qkv_ref = hin @ g64(p + "self_attn.in_proj_weight").T + g64(p + "self_attn.in_proj_bias")
            qkv = unsplit(qraw, 3 * dm)
            report("qkv (split rows)", qkv, qkv_ref)
            if rs is not None:      # P rows from the stashed statistics must sum to one: sum_j exp(s_ij - m_i) * inv_i
                rs = rs.reshape(nseq, H, S, 2).astype(np.float64)
                q = qkv[:, :dm].reshape(nseq, S, H, 128).transpose(0, 2, 1, 3)
                k = qkv[:, dm:2 * dm].reshape(nseq, S, H, 128).transpose(0, 2, 1, 3)
                s_ = (q @ k.transpose(0, 1, 3, 2)) / np.sqrt(128.0)
                psum = (np.exp(s_ - rs[..., 0:1]) * rs[..., 1:2]).sum(-1)             # [nseq, H, S]
                dev_ = np.abs(psum - 1.0)
                w = np.argsort(-dev_.max(1).reshape(-1))[:4]
                print(f"   softmax statistics: |sum_j P_ij - 1| median {np.median(dev_):.2e} max {dev_.max():.2e} at (seq, token) "
                      f"{[(int(r) // S, int(r) % S) for r in w]}; max (true max - stashed reference max) {float((s_.max(-1) - rs[..., 0]).max()):.2f}", flush=True)


def chain_wants(sd, n, w):
    n_resp = max(10, n)
    d = inputs(0, n)
    sch = do.Schedule(do.named_betas("cosine", 1000), do.space_timesteps(1000, [n_resp]))
    out = {}
    for name, dt in (("f32", np.float32), ("f64", np.float64)):
        oracle_dtype(dt)
        x = do.q_sample(sch, n - 1, d["x0"], d["x_T"])
        out[f"chain_{n}_{w:g}_{name}"] = np.asarray(
            do.sample_loop(sch, mo.MDMOracle(sd), x, d["noise"], enc_text=d["enc"], text_scale=d["scale"], cfg=True,
                           mask=d["kf_mask"] & d["len_mask"], inpaint=d["x0"], imputate=True, stop_imputation_at=1,
                           recon_guidance=True, recon_weight=w, first_step=n - 1), dtype=np.float64)
    oracle_dtype(np.float32)
    return out


def chain(sd, n, w, want, modes=MODES):
    n_resp = max(10, n)
    d = inputs(0, n)
    rs, gd = sub("diffusion.respace"), sub("diffusion.gaussian_diffusion")
    wants = {k: want[f"chain_{n}_{w:g}_{k}"] for k in ("f32", "f64")}
    row = [f"steps {n:3d} weight {w:4.1f}: fp32 oracle {rel(wants['f32'], wants['f64']):.3e}"]
    for mode in modes:
        model = native_model(sd, mode)
        diffusion = rs.SpacedDiffusion(rs.space_timesteps(1000, [n_resp]), gd.DiffusionConfig(betas=gd.get_named_beta_schedule("cosine", 1000)))
        y = dict(mask=tt(d["len_mask"]), lengths=tt(d["lengths"]), text_embed=tt(d["enc"]), text_scale=tt(d["scale"]),
                 inpainting_mask=tt(d["kf_mask"]), inpainted_motion=tt(d["x0"]), imputate=True, stop_imputation_at=1,
                 replacement_distribution='conditional', reconstruction_guidance=True, reconstruction_weight=w, gradient_schedule=None,
                 diffusion_steps=1000, stop_recguidance_at=0)
        diffusion.injected_noise = tt(d["noise"])
        out = diffusion.p_sample_loop(model, d["shape"], noise=tt(d["x_T"]), clip_denoised=False, model_kwargs={"y": y},
                                      skip_timesteps=n_resp - n, init_image=tt(d["x0"])).cpu().numpy()
        assert model.model._engine.precision == mode, model.model._engine.precision
        row.append(f"{mode}{ENV_TAG} {rel(out, wants['f64']):.3e} (vs fp32 oracle {rel(out, wants['f32']):.3e})")
    print(" | ".join(row), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--make-wants", action="store_true")
    ap.add_argument("--weight-seed", type=int, default=3)
    ap.add_argument("--modes", default=",".join(MODES))
    ap.add_argument("--stages", default="1,1b,2")
    args = ap.parse_args()
    modes = tuple(args.modes.split(","))
    sd = weights.make_state_dict(args.weight_seed, text=True)
    if args.make_wants:
        import time
        out, t0 = eval_wants(sd), time.time()
        out.update(opoint_wants(sd))
        for n, w in PLAN:
            out.update(chain_wants(sd, n, w))
            print(f"oracle chains steps {n} weight {w}: {time.time() - t0:.0f}s", flush=True)
            WANTS.parent.mkdir(exist_ok=True)
            np.savez_compressed(WANTS, weight_seed=args.weight_seed, **out)
        return
    want = np.load(WANTS)
    assert int(want["weight_seed"]) == args.weight_seed
    stages = args.stages.split(",")
    if "1" in stages:
        one_evaluation(sd, want, modes)
    if "1b" in stages:
        operating_point(sd, want, modes)
    if "audit" in stages:
        stash_audit(sd, want)
    if "audit2" in stages:
        stash_audit2(sd, want)
    if "2" in stages:
        print("# stage 2: guided chains, rel-L2 of the final sample vs the float64 oracle chain (in brackets: vs the fp32 oracle chain)")
        for n, w in PLAN:
            if f"chain_{n}_{w:g}_f64" in want.files:
                chain(sd, n, w, want, modes)


if __name__ == "__main__":
    main()
