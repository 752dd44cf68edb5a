"""Throughput of the CondMDI sampling hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4|c5|unet|unet_recon] [--no-cpu] [--no-pmc]

Workload (BASELINE.json configs[1], "c2"): HumanML3D shape B=32 x 263 feats x 196 frames per GPU,
1000-step DDPM chain, text-conditioned classifier-free guidance (2 denoiser passes per step),
random-init MDM trans_enc (8 layers, d=512, ff=1024, 4 heads), synthetic z-scored inputs, fp32.
One "step" = one denoising step x_t -> x_{t-1} of the whole per-GPU batch (both CFG passes, the
sampler update and the per-step Philox noise); the K timed steps are the LAST K steps of the chain
entered at step K-1 (t = K-1 .. 0), bracketed by barrier + torch.cuda.synchronize on both sides;
the maximum over ranks is the time.  value = N * K / time (weak scaling: every rank runs its own
B=32 slice; no collective inside the loop, one RCCL all-gather of the samples afterwards).

Prints ONE JSON line on rank 0 (see the task contract) with these extra objects:
  roofline      the dominant kernel = the self-attention in_proj GEMM, timed with HIP events on its own stream inside
                a second, instrumented pass over the same K steps; `traffic` (HBM bytes per launch), `mfma_busy` and
                `hbm_gbps` come from rocprofv3 --pmc sub-runs of THIS config (separate passes: FETCH_SIZE doubled per
                the gfx950 correction of MI355X_MICROARCH.md | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE)
  f32_exact     the same K steps on the exact-fp32 engine (CMDI_PREC_F32: v_mfma_f32_32x32x2_f32 products) with its own
                roofline against the 157.3 TFLOP/s fp32-matrix peak (transformer configs, N=1)
  bf16x6        the same on the CMDI_PREC_BF16X6 engine (exact three-plane bf16 operands, six bf16 MFMA products per
                fp32 product: no operand truncation, no range limit)
  cpu_baseline  the reference's CPU path restated on torch CPU kernels (oracle/torch_cpu_port.py) on the
                host cores, a bounded sample of the same workload (1 warm-up + 10 full CFG steps at B=32, quoted on the
                median step with min / max; intra-op threads calibrated on a full step)

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches ITSELF as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py ...`
(one rank per GPU over RCCL; rank 0 prints the line), so the bare command line and the driver's torchrun form are the same job.
"""
from __future__ import annotations

import argparse
import csv
import importlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch
import torch.distributed as dist

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))
PKG = "diffusion-motion-inbetweening_amd"
sub = lambda n: importlib.import_module(f"{PKG}.{n}")

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
F16_MFMA_PEAK_TFLOPS = 2500.0   # same guide: dense f16/bf16 MFMA (v_mfma_f32_32x32x16_f16)
HBM_PEAK_GBPS = 8000.0          # same guide: HBM3E
N_FEATS, T_FRAMES = 263, 196
CONFIGS = {
    # name: (batch per GPU, respacing, sampler, cfg, edit)
    "c2": dict(B=32, respacing=[1000], sampler="ddpm", cfg=True, edit=False,
               desc="HumanML3D 196x263, 1000-step DDPM, B=32/GPU, text CFG"),
    "c3": dict(B=32, respacing=[1000], sampler="ddpm", cfg=True, edit=True,
               desc="benchmark_sparse imputation + reconstruction guidance, 1000-step DDPM, B=32/GPU, CFG"),
    "c4": dict(B=256, respacing="ddim100", sampler="ddim", cfg=True, edit=False,
               desc="DDIM-100 respaced, B=256/GPU, text CFG"),
    # BASELINE configs[4]: batch 1024 = 8 x 128 sharded over the node.  STRONG scaling: the global batch stays 1024, every
    # rank samples 1024 / N of it (one GPU holds all 1024: 19 GB of workspace), one all-gather at the end
    "c5": dict(B=1024, respacing=[1000], sampler="ddpm", cfg=True, edit=False, strong=True,
               desc="conditional_synthesis batch 1024 sharded over the GPUs (1024/N per GPU), 1000-step DDPM, text CFG"),
    # the denoiser CondMDI trains / releases (configs/model.py motion_unet_adagn_xl)
    "unet": dict(B=32, respacing=[1000], sampler="ddpm", cfg=True, edit=False, unet=True,
                 desc="MDM_UNET (dim_mults 2,2,2,2, keyframe-conditioned), HumanML3D 196x263, 1000-step DDPM, "
                      "B=32/GPU, text CFG, sparse keyframes observed + imputed"),
    "unet_recon": dict(B=32, respacing=[1000], sampler="ddpm", cfg=True, edit=True, unet=True,
                       desc="MDM_UNET as above + reconstruction guidance through the U-Net (native input-VJP), "
                            "1000-step DDPM, B=32/GPU, text CFG"),
}


def flops_per_sample_eval(T=T_FRAMES, J=N_FEATS, d=512, f=1024, L=8):
    S = T + 1
    layer = 2 * S * d * 3 * d + 4 * S * S * d + 2 * S * d * d + 4 * S * d * f
    return 2 * T * J * d + L * layer + 2 * T * d * J + 4 * d * d + 2 * 512 * d


def unet_flops_per_sample_eval(J=N_FEATS, dim=512, mult=2, keyframe=True):
    """2*MACs of one MDM_UNET evaluation (reference model/mdm_unet.py TemporalUnet, 224 padded frames)."""
    C, Tl = dim * mult, [224, 112, 56, 28]
    conv = lambda T, k, cin, cout: 2.0 * T * k * cin * cout
    rb = lambda T, cin, cout: conv(T, 5, cin, cout) + conv(T, 5, cout, cout) + (conv(T, 1, cin, cout) if cin != cout else 0) \
        + 2.0 * dim * 2 * cout
    cin0 = J * (2 if keyframe else 1)
    f = 2.0 * dim * 4 * dim * 2
    for l, T in enumerate(Tl):
        f += rb(T, cin0 if l == 0 else C, C) + rb(T, C, C)
        if l < 3:
            f += conv(Tl[l + 1], 3, C, C)
    f += 2 * rb(Tl[3], C, C)
    for l in (3, 2, 1):
        f += rb(Tl[l], 2 * C, C) + rb(Tl[l], C, C) + conv(Tl[l - 1], 2, C, C)   # transposed conv: 2 taps per output row
    return f + conv(224, 5, C, C) + conv(224, 1, C, J)


def build_unet(dev, seed=0):
    from oracle import weights
    mu = sub("utils.model_util")
    args = SimpleNamespace(dataset="humanml", arch="unet", keyframe_conditioned=True, dim_mults=(2, 2, 2, 2),
                           cond_mask_prob=0.1)
    model, _ = mu.create_model_and_diffusion(args, None)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = weights.to_torch(weights.fill_like(shapes, seed))
    sd.update({k: v for k, v in model.state_dict().items() if k.endswith(".pe")})
    mu.load_model_wo_clip(model, sd)
    return model.to(dev).eval(), sd


def build_model(cfg_on, dev, seed=0):
    from oracle import weights  # synthetic weights recipe only (no oracle compute)
    mu = sub("utils.model_util")
    model, _ = mu.create_model_and_diffusion(SimpleNamespace(dataset="humanml"), None)
    sd = weights.make_state_dict(seed, text=True)
    mu.load_model_wo_clip(model, weights.to_torch(sd))
    model.to(dev).eval()
    return model, sd


def _timed_steps(one, x, n_steps):
    """1 warm-up + n_steps timed steps of `one`; per-step wall times (VERDICT r3 task 8: three timed steps gave a 35 % spread
    between boxes — the line now carries min / median / max per step and is quoted on the median)."""
    xc = one(x, 999, 0)
    per = []
    for k in range(n_steps):
        t0 = time.perf_counter()
        xc = one(xc, 998 - k, k + 1)
        per.append(time.perf_counter() - t0)
    per.sort()
    med = per[len(per) // 2] if len(per) % 2 else 0.5 * (per[len(per) // 2 - 1] + per[len(per) // 2])
    return {"value": 1.0 / med, "steps_per_s_mean": n_steps / sum(per), "step_s_min": per[0], "step_s_median": med,
            "step_s_max": per[-1], "timed_steps": n_steps}


def _calibrate_threads(step_fn):
    """Intra-op thread count of the CPU leg: the fastest of {all, 64, 32, 16} logical CPUs on ONE FULL step of the workload
    (round 3 calibrated on a single conditional forward), after a warm-up at that count."""
    host = os.cpu_count() or 1
    best, best_dt = host, None
    for n in sorted({host, min(host, 64), min(host, 32), min(host, 16)}, reverse=True):
        torch.set_num_threads(n)
        step_fn()
        t0 = time.perf_counter()
        step_fn()
        dt = time.perf_counter() - t0
        if best_dt is None or dt < best_dt:
            best, best_dt = n, dt
    torch.set_num_threads(best)
    return best, host


def cpu_baseline(sd, B, n_steps=10):
    """The reference's CPU path restated on torch CPU kernels (oracle/torch_cpu_port.py: the reference's
    denoiser IS torch's nn.TransformerEncoder, two sequential CFG passes, then the posterior update) at
    the bench shape.  The intra-op thread count is calibrated first (one full CFG step per candidate)
    and reported as `cores`.  profiles/r04_cpu_port_vs_reference.json holds the port / real-reference ratio measured
    where /root/reference exists (tools/cpu_port_vs_reference.py)."""
    from oracle import diffusion_oracle as do
    from oracle.torch_cpu_port import TorchCpuMDM
    rng = np.random.default_rng(1)
    m = TorchCpuMDM(sd)
    sch = do.Schedule(do.named_betas("cosine", 1000), do.space_timesteps(1000, [1000]))
    x = rng.standard_normal((B, N_FEATS, 1, T_FRAMES)).astype(np.float32)
    enc = torch.from_numpy(rng.standard_normal((B, 512)).astype(np.float32))
    scale = torch.full((B,), 2.5)
    nz = rng.standard_normal((n_steps + 1,) + x.shape).astype(np.float32)

    def one(xc, i, k):
        t = torch.full((B,), i, dtype=torch.long)
        hat, _, _ = m.forward_cfg(torch.from_numpy(xc), t, enc, scale)
        return do.step_update(sch, i, xc, hat.numpy(), nz[k])[0]

    prev = torch.get_num_threads()
    best, host = _calibrate_threads(lambda: one(x, 999, 0))
    res = _timed_steps(one, x, n_steps)
    torch.set_num_threads(prev)
    res.update({"unit": "denoising steps/s", "cores": best, "kind": "port",
                "sample": f"{n_steps} full CFG DDPM steps at B={B}x{T_FRAMES}x{N_FEATS} after 1 warm-up, quoted on the median step "
                          f"(oracle/torch_cpu_port.py: torch {torch.__version__} CPU nn.TransformerEncoder, the "
                          f"reference's own denoiser arithmetic; {best} intra-op threads picked from a calibration "
                          f"of one full step per candidate on a host with {host} logical CPUs)"})
    return res


def cpu_baseline_unet(sd, B, n_steps=5):
    """The same for --config unet: MDM_UNET on torch's CPU conv1d / group_norm / mish kernels
    (oracle/torch_cpu_port.py::TorchCpuUNET), two sequential CFG passes + the posterior update."""
    from oracle import diffusion_oracle as do
    from oracle.torch_cpu_port import TorchCpuUNET
    rng = np.random.default_rng(1)
    m = TorchCpuUNET(sd)
    sch = do.Schedule(do.named_betas("cosine", 1000), do.space_timesteps(1000, [1000]))
    shape = (B, N_FEATS, 1, T_FRAMES)
    x = rng.standard_normal(shape).astype(np.float32)
    obs = torch.from_numpy(rng.standard_normal(shape).astype(np.float32))
    mask = torch.zeros(shape, dtype=torch.bool)
    mask[..., ::5] = True
    enc = torch.from_numpy(rng.standard_normal((B, 512)).astype(np.float32))
    scale = torch.full((B,), 2.5)
    nz = rng.standard_normal((n_steps + 1,) + shape).astype(np.float32)
    t999 = torch.full((B,), 999, dtype=torch.long)
    xs = torch.from_numpy(x[:4])
    prev = torch.get_num_threads()
    # (calibrated on a 4-sequence conditional pass: a full U-Net step is 5-10 s of CPU per candidate)
    best, host = _calibrate_threads(lambda: m.forward(xs, t999[:4], enc[:4], False, obs[:4], mask[:4]))

    def one(xc, i, k):
        t = torch.full((B,), i, dtype=torch.long)
        hat, _, _ = m.forward_cfg(torch.from_numpy(xc), t, enc, scale, obs, mask)
        return do.step_update(sch, i, xc, hat.numpy(), nz[k])[0]

    res = _timed_steps(one, x, n_steps)
    torch.set_num_threads(prev)
    res.update({"unit": "denoising steps/s", "cores": best, "kind": "port",
                "sample": f"{n_steps} full CFG DDPM steps of MDM_UNET at B={B}x{T_FRAMES}x{N_FEATS} after 1 warm-up, quoted on the "
                          f"median step (oracle/torch_cpu_port.py::TorchCpuUNET: torch {torch.__version__} CPU conv1d / group_norm / "
                          f"mish, the reference's own layer kernels; {best} intra-op threads picked from a calibration "
                          f"on a host with {host} logical CPUs)"})
    return res

N_SIMD = 1024            # 256 CUs x 4 SIMDs
N_XCD = 8


def expected_mfma_busy(precision: str, m: int, n: int, k: int):
    """SQ_VALU_MFMA_BUSY_CYCLES of ONE launch of the split GEMMs, an exact function of its MFMA count: 32 cycles per
    v_mfma_f32_32x32x16_{f16,bf16} (MI355X_MICROARCH.md), 3 (f16x3) or 6 (bf16x6) of them per 32 x 32 x 16 block of the
    algorithmic product; None for the fp32-MFMA engine (another instruction, not calibrated)."""
    per = {"f16x3": 3, "bf16x6": 6}.get(precision)
    return None if per is None else per * 32.0 * (m * n * k / (32.0 * 32.0 * 16.0))


def pmc_counters(config: str, precision: str, batch: int, timeout_s: float = 150.0, timed=None):
    """rocprofv3 --pmc sub-runs of `bench.py --pmc-child` for THIS config and precision (counters serialise kernels, so
    they never run inside the timed region): per launch of the kernel the HIP events timed — FETCH_SIZE, WRITE_SIZE,
    SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE, each in its own pass (MI355X_MICROARCH.md: FETCH_SIZE takes 3 of the 4 TCC
    slots).  `timed` = {"kernel": family name cmdi_profile_kernel recorded for the timed launches, "mnk": their (M, N, K)}:
    the launches are picked BY THAT NAME and by the MFMA count (M, N, K) imply (round 6, VERDICT r5 weak #8: "largest MFMA
    count of the step" picked a backward GEMM of another kernel family for unet_recon).  Returns {} if rocprofv3 is unavailable."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"pmc_error": "rocprofv3 not found"}
    res, t_end = {}, time.time() + timeout_s
    tmp = tempfile.mkdtemp(prefix="cmdi_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    env = dict(os.environ, TMPDIR=tmp, CMDI_GROUPS="1")
    env.pop("CMDI_PROBES_LIB", None)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "MASTER_ADDR", "MASTER_PORT",
              "TORCHELASTIC_RUN_ID"):   # the counter passes are plain single-process runs
        env.pop(k, None)
    try:
        # pass 0 identifies the dominant kernel: SQ_VALU_MFMA_BUSY_CYCLES is an exact function of a launch's MFMA count,
        # and the in_proj projection (largest M*N*K of the step) has the largest — its Dispatch_Ids (the launch order is
        # deterministic) select the same launches in the FETCH_SIZE / WRITE_SIZE passes
        dominant = None
        for i, ctrs in enumerate((["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"], ["FETCH_SIZE"], ["WRITE_SIZE"])):
            left = t_end - time.time()
            if left < 20:
                res["pmc_error"] = "time budget exhausted"
                break
            out_dir = os.path.join(tmp, f"p{i}")
            cmd = [exe, "--pmc", *ctrs, "--output-format", "csv", "-d", out_dir, "-o", "p", "--", sys.executable,
                   str(REPO / "bench.py"), "--pmc-child", "--config", config, "--precision", precision,
                   "--batch", str(batch)]
            try:
                r = subprocess.run(cmd, cwd=tmp, env=env, capture_output=True, text=True, timeout=left)
            except subprocess.TimeoutExpired:
                res["pmc_error"] = "rocprofv3 pass timed out"
                break
            if r.returncode != 0:
                res["pmc_error"] = f"rocprofv3 rc={r.returncode}: {r.stderr[-300:]}"
                break
            files = [os.path.join(dp, f) for dp, _, fs in os.walk(out_dir) for f in fs if f.endswith("counter_collection.csv")]
            # the GEMM family of this precision only: with reconstruction guidance the step also launches two fp32-MFMA boundary
            # GEMMs whose busy-cycle count (1/16 of the f16 rate per flop) would otherwise win the "largest count" rule below
            family = {"f16x3": "gemm_h3", "bf16x6": "gemm_x6", "f32": "gemm_nt"}.get(precision, "gemm")
            if timed and timed.get("kernel"):
                family = timed["kernel"]          # "gemm_h3_kernel" / "gemm_h3p_kernel" / "gemm_h3w_kernel" / ...: exact family
            rows = []
            for fcsv in files:
                with open(fcsv, newline="") as fh:
                    rows += [r_ for r_ in csv.DictReader(fh) if family in r_["Kernel_Name"]]
            if not rows:
                res["pmc_error"] = "no gemm dispatches in the counter file"
                break
            if dominant is None:
                busy = {int(r_["Dispatch_Id"]): float(r_["Counter_Value"]) for r_ in rows
                        if r_["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES"}
                want = expected_mfma_busy(precision, *timed["mnk"]) if timed and timed.get("mnk") else None
                if want:
                    # the launches whose MFMA count is that of the timed (M, N, K): padded rows of the last tile add < 2 %
                    top = min(busy.values(), key=lambda v: abs(v / want - 1.0))
                    if abs(top / want - 1.0) > 0.05:
                        res["pmc_error"] = f"no {family} launch with the timed launch's MFMA count ({want:.4g} busy cycles; nearest {top:.4g})"
                        break
                    res["pmc_expected_busy"] = want
                else:
                    top = max(busy.values())
                dominant = {d for d, v in busy.items() if v == top}
                # (with reconstruction guidance the backward dX GEMM of in_proj has the same M*N*K: keep the launches of
                # the kernel that was dispatched first — the forward one, the launch the HIP events time)
                first = min((r_ for r_ in rows if int(r_["Dispatch_Id"]) in dominant), key=lambda r_: int(r_["Dispatch_Id"]))
                dominant = {int(r_["Dispatch_Id"]) for r_ in rows
                            if int(r_["Dispatch_Id"]) in dominant and r_["Kernel_Name"] == first["Kernel_Name"]}
                res["pmc_kernel"], res["pmc_grid"] = first["Kernel_Name"][:160], int(first["Grid_Size"])
                res["pmc_launches"] = len(dominant)
            if i == 0 and rows and "Start_Timestamp" in rows[0] and "End_Timestamp" in rows[0]:
                # wall time of the same launches IN THIS counter pass (ns): effective clock = GRBM_GUI_ACTIVE / 8 / that
                seen, dur = set(), []
                for r_ in rows:
                    d = int(r_["Dispatch_Id"])
                    if d in dominant and d not in seen:
                        seen.add(d)
                        dur.append(float(r_["End_Timestamp"]) - float(r_["Start_Timestamp"]))
                if dur and min(dur) > 0:
                    res["pmc_pass_kernel_ns"] = sum(dur) / len(dur)
            for c in ctrs:
                vals = [float(r_["Counter_Value"]) for r_ in rows if int(r_["Dispatch_Id"]) in dominant and r_["Counter_Name"] == c]
                if len(vals) != len(dominant):
                    res["pmc_error"] = f"dispatch ids moved between passes ({c}: {len(vals)} of {len(dominant)})"
                    break
                res[c] = sum(vals) / len(vals)
            if "pmc_error" in res:
                break
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return res


def timed_launches(eng, run_loop):
    """Instrumented pass: HIP events around every in_proj GEMM launch (or one level-0 convolution GEMM per U-Net evaluation)."""
    eng.profile_enable(True)
    run_loop()
    torch.cuda.synchronize()
    ms, launches, mnk = eng.profile_read()
    ran = eng.profile_kernel()          # the kernel family the bracketed launches dispatched to, recorded by the library
    eng.profile_enable(False)
    return {"ms": ms, "launches": launches, "mnk": tuple(int(v) for v in mnk), "kernel": ran}


def roofline_from(eng, timed, split, is_unet, pmc):
    """The roofline object of the timed launches + the PMC numbers of the same launches (pmc_counters(timed=...))."""
    ms, launches, (m, n, k), ran = timed["ms"], timed["launches"], timed["mnk"], timed["kernel"]
    avg_s = ms / max(launches, 1) * 1e-3
    # the counters must describe the kernel the events timed: same family by name, and the same duration within 15 % (a counter
    # pass serialises and slows a kernel by a few per cent); otherwise they are dropped, not reported beside another kernel's time
    if pmc.get("pmc_kernel") and ran and ran not in pmc["pmc_kernel"]:
        pmc = {"pmc_error": f"counters of {pmc['pmc_kernel'][:60]} do not belong to the timed {ran}"}
    elif pmc.get("pmc_pass_kernel_ns") and abs(pmc["pmc_pass_kernel_ns"] * 1e-9 / avg_s - 1.0) > 0.15:
        pmc = {"pmc_error": f"counter-pass launch {pmc['pmc_pass_kernel_ns'] * 1e-3:.1f} us vs timed {avg_s * 1e6:.1f} us: not the same launches",
               "pmc_kernel": pmc.get("pmc_kernel")}
    flop_launch = 2.0 * m * n * k
    ach = flop_launch / avg_s / 1e12
    traffic = mfma_busy = clock = None
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        # rocprofv3 reports KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM) -> x2
        traffic = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
    if "SQ_VALU_MFMA_BUSY_CYCLES" in pmc and "GRBM_GUI_ACTIVE" in pmc and pmc["GRBM_GUI_ACTIVE"] > 0:
        cyc = pmc["GRBM_GUI_ACTIVE"] / N_XCD              # the counter is summed over the 8 XCDs
        mfma_busy = pmc["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMD * cyc)
        if pmc.get("pmc_pass_kernel_ns"):
            # shader clock the kernel actually ran at (in the counter pass): busy cycles / wall time.  The power cap holds it
            # well below the 2.4 GHz the peaks are quoted at (MI355X_MICROARCH.md "DVFS give-back")
            clock = cyc / pmc["pmc_pass_kernel_ns"]
    rl = {"bound": "mfma", "achieved": ach, "unit": "TFLOP/s", "traffic": traffic, "launches": launches,
          "avg_launch_us": avg_s * 1e6, "flops_per_launch": flop_launch,
          "algorithmic_bytes": 4.0 * (m * k + n * k + m * n),
          "mfma_busy": mfma_busy, "effective_clock_ghz": clock, "hbm_gbps": (traffic / avg_s / 1e9) if traffic else None,
          "pmc": {k_: v for k_, v in pmc.items() if k_.startswith("pmc_") or k_ in ("FETCH_SIZE", "WRITE_SIZE",
                  "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE")}}
    if split:
        # algorithmic flops = 2MNK of the fp32 product; the kernel executes 3 f16 MFMA products per algorithmic
        # product, so its ceiling is the dense f16 peak / 3
        peak = F16_MFMA_PEAK_TFLOPS / 3.0
        what = ("unet.downs.0.1.blocks.1 Conv1d k=5 as a tap-shifted GEMM" if is_unet else "self_attn.in_proj")
        kname = f"{ran} (" + ("persistent, " if ran == "gemm_h3p_kernel" else "") + ("frames only: " if is_unet and ran == "gemm_h3p_kernel" else "")
        rl.update(kernel=f"{kname}{what}, M={m} N={n} K={k}, 3x v_mfma_f32_32x32x16_f16 per fp32-equivalent "
                         "product" + ("" if is_unet else ", split-rows output") + ")",
                  peak=peak, frac=ach / peak, executed_f16_tflops=3.0 * ach, f16_dense_peak=F16_MFMA_PEAK_TFLOPS,
                  vs_fp32_mfma_peak=ach / FP32_MFMA_PEAK_TFLOPS)
    elif eng.precision == "bf16x6":
        # six exact bf16 partial products per fp32 product: ceiling = dense bf16 peak / 6
        peak = F16_MFMA_PEAK_TFLOPS / 6.0
        rl.update(kernel=f"{ran or 'gemm_x6_kernel'} (self_attn.in_proj, M={m} N={n} K={k}, 6x v_mfma_f32_32x32x16_bf16 per fp32 "
                         "product on exact three-plane bf16 operands)",
                  peak=peak, frac=ach / peak, executed_bf16_tflops=6.0 * ach, bf16_dense_peak=F16_MFMA_PEAK_TFLOPS,
                  vs_fp32_mfma_peak=ach / FP32_MFMA_PEAK_TFLOPS)
    else:
        rl.update(kernel=f"{ran or 'gemm_nt_kernel'} (self_attn.in_proj, M={m} N={n} K={k}, fp32 MFMA 32x32x2)",
                  peak=FP32_MFMA_PEAK_TFLOPS, frac=ach / FP32_MFMA_PEAK_TFLOPS)
    return rl


def roofline_attention(eng, run_loop):
    """The next kernel on the list: the self-attention core, timed like the in_proj GEMM — HIP events around every
    launch in an instrumented pass over the same K steps.  Algorithmic work 4 S^2 d per sequence and layer (QK^T and PV)."""
    eng.profile_select(1)
    eng.profile_enable(True)
    run_loop()
    torch.cuda.synchronize()
    ms, launches, (nseq, S, H) = eng.profile_read()
    eng.profile_enable(False)
    eng.profile_select(0)
    if launches == 0:
        return None
    avg_s = ms / launches * 1e-3
    d_head = 128
    flop = 4.0 * nseq * H * S * S * d_head
    ach = flop / avg_s / 1e12
    split = eng.precision == "f16x3"
    peak = F16_MFMA_PEAK_TFLOPS / 3.0 if split else FP32_MFMA_PEAK_TFLOPS
    return {"kernel": ("attention_h3_kernel (softmax(QK^T / sqrt(128)) V on split-f16 products: 3 MFMAs per fp32-equivalent "
                       "product)" if split else "attention_fwd_kernel (fp32 MFMA 16x16x4)") + f", {nseq} sequences x {H} heads x {S} tokens",
            "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "launches": launches,
            "avg_launch_us": avg_s * 1e6, "flops_per_launch": flop,
            "algorithmic_bytes": 4.0 * nseq * S * 4 * H * d_head,       # q, k, v in + o out, 4 bytes per value
            "hbm_frac": 4.0 * nseq * S * 4 * H * d_head / avg_s / 1e9 / HBM_PEAK_GBPS,
            "note": "one block per (sequence, head) and CU: a load burst (Q + first K/V stages), the key loop, a store burst, entered "
                    "by all CUs together; with every MFMA / softmax / LDS read removed the same kernel takes 22.5 us at 64 "
                    "sequences (its memory floor, profiles/r03_attention_pipe_experiment.txt); the arithmetic of the key loop "
                    "comes on top because 8 waves x 256 registers leave no room for a second work item per CU (DESIGN.md 3)"}


def job_layout(cfg: dict, world: int, rank: int, shard_bounds) -> dict:
    """The rank arithmetic of a run, a pure function (tests/test_host_logic.py checks it for N = 1, 2, 4, 8 without a GPU).
    Weak scaling (c2 / c3 / c4): every rank steps its own batch of cfg['B'], the job's batch is N * B.  Strong scaling (c5):
    ONE global batch of cfg['B'] cut into contiguous shards [lo, hi) by utils.dist_util.shard_bounds."""
    strong = bool(cfg.get("strong"))
    global_batch = cfg["B"] if strong else world * cfg["B"]
    lo, hi = shard_bounds(global_batch, rank, world)
    return {"strong": strong, "global_batch": global_batch, "lo": lo, "hi": hi, "batch": hi - lo}


def job_rates(layout: dict, world: int, K: int, elapsed_s: float, n_chain: int) -> dict:
    """Whole-job numbers from the max-over-ranks time of K steps.  Weak scaling: the job advances N batch-steps per step
    time; strong scaling: ONE global batch, so steps/s is 1 / step time and motions/s carries the scaling."""
    steps_per_s = (1 if layout["strong"] else world) * K / elapsed_s
    return {"steps_per_s": steps_per_s, "ms_per_step": elapsed_s / K * 1e3,
            "motions_per_sec": layout["global_batch"] / (n_chain * elapsed_s / K)}


def self_launch_argv(argv, n_gpus, port=None):
    """The torch.distributed.run command line `python bench.py --gpus N ...` turns into when it was started bare (no
    WORLD_SIZE): one rank per GPU on this node, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), str(Path(__file__).resolve())] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 200, or what the config's chain leaves)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps before them (default 20, at most a tenth of the chain)")
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc sub-runs (traffic / mfma_busy = null)")
    ap.add_argument("--no-f32", action="store_true", help="skip the exact-fp32 engine's leg")
    ap.add_argument("--graph", action="store_true", help="replay each denoising step as a hipGraph")
    ap.add_argument("--no-graph-leg", action="store_true", help="skip the eager-vs-hipGraph comparison legs")
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch of the config (exploration)")
    ap.add_argument("--precision", default=None, choices=["f32", "f16x3", "bf16x6"],
                    help="encoder GEMM arithmetic (include/condmdi.h CMDI_PREC_*); default: the library's")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dry-launch", action="store_true", help=argparse.SUPPRESS)    # CPU test of the launch path (gloo, no GPU work)
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.pmc_child:
        # started bare: become the launcher of N ranks; their rank 0 prints the JSON line
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(self_launch_argv(sys.argv[1:], args.gpus), env=env))
    if args.dry_launch:
        # tests/test_host_logic.py: the bare `python bench.py --gpus 2` must arrive HERE as 2 ranks with a working process
        # group (gloo on the CPU-only container) — the rendezvous, rank arithmetic and one collective, no device work
        world = int(os.environ.get("WORLD_SIZE", "1"))
        assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", init_method="env://")
            ones = torch.ones(1)
            dist.all_reduce(ones)
            layout = job_layout(cfg, world, dist.get_rank(), sub("utils.dist_util").shard_bounds)
            spans = [None] * world
            dist.all_gather_object(spans, (layout["lo"], layout["hi"]))
            if dist.get_rank() == 0:
                print(json.dumps({"dry_launch": True, "n_gpus": args.gpus, "world": dist.get_world_size(), "ranks_summed": int(ones.item()),
                                  "shards": spans, "global_batch": layout["global_batch"]}), flush=True)
            dist.destroy_process_group()
        else:
            print(json.dumps({"dry_launch": True, "n_gpus": 1, "world": 1}), flush=True)
        return

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # launched by torch.distributed.run (the driver's form for N > 1; a 1-process launch takes the same path, so the
    # process group, the barrier and the collectives are exercised on a single GPU too)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU path exists)"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if use_dist:
        dist.init_process_group("nccl", init_method="env://", device_id=dev)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    n_ranks_seen = dist.get_world_size() if use_dist else 1

    gd, rs, du = sub("diffusion.gaussian_diffusion"), sub("diffusion.respace"), sub("utils.dist_util")
    N = sub("_native")
    if args.batch > 0:
        cfg["B"] = args.batch
        if not args.pmc_child:
            cfg["desc"] += f" [batch overridden to {args.batch}]"
    layout = job_layout(cfg, world, rank, du.shard_bounds)
    strong, global_batch, lo, hi = layout["strong"], layout["global_batch"], layout["lo"], layout["hi"]
    B = layout["batch"]              # this rank's samples [lo, hi) of the global batch
    K, W = args.steps, args.warmup
    is_unet = bool(cfg.get("unet"))
    model, sd = build_unet(dev) if is_unet else build_model(cfg["cfg"], dev)
    diffusion = rs.SpacedDiffusion(rs.space_timesteps(1000, cfg["respacing"]),
                                   gd.DiffusionConfig(betas=gd.get_named_beta_schedule("cosine", 1000)))
    n_chain = diffusion.num_timesteps
    if args.warmup is None:
        W = min(20, n_chain // 10)
    if args.steps is None:
        K = min(200, n_chain - W)
    if args.pmc_child:
        K, W = 3, 2
    assert K + W <= n_chain, f"steps + warmup must be <= {n_chain}"
    sampler = N.CMDI_SAMPLER_DDIM if cfg["sampler"] == "ddim" else N.CMDI_SAMPLER_DDPM
    seed = 20260925

    # synthetic per-rank inputs keyed by GLOBAL sample index (this rank owns samples [lo, hi))
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    enc = torch.randn(B, 512, generator=g).to(dev)
    scale = torch.full((B,), 2.5, device=dev)
    cond = dict(batch=B, n_frames=T_FRAMES, cfg=cfg["cfg"], enc_text=enc, text_scale=scale)
    if is_unet:   # sparse keyframes (every 5th frame, all features): observed by the U-Net and imputed
        x0 = torch.randn(B, N_FEATS, 1, T_FRAMES, generator=g).to(dev)
        mask = torch.zeros(B, N_FEATS, 1, T_FRAMES, dtype=torch.bool)
        mask[..., ::5] = True
        cond.update(inpaint_mask=mask.to(dev), inpaint_motion=x0, imputate=True, stop_imputation_at=1,
                    obs_x0=x0, obs_mask=mask.to(dev))
    if cfg["edit"]:
        if not is_unet:
            x0 = torch.randn(B, N_FEATS, 1, T_FRAMES, generator=g).to(dev)
            mask = torch.zeros(B, N_FEATS, 1, T_FRAMES, dtype=torch.bool)
            mask[..., ::5] = True  # benchmark_sparse, trans_length=5, all features (pos_rot_vel)
        cond.update(inpaint_mask=mask.to(dev), inpaint_motion=x0, imputate=True, stop_imputation_at=1,
                    recon_guidance=True, stop_recguidance_at=0,
                    recon_w=np.full((n_chain,), 20.0, dtype=np.float32))

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed_run(precision):
        """W untimed warm-up steps from the top of the chain, then exactly K timed steps (the last K of the chain),
        barrier + synchronize on both sides, max over ranks.  Returns (seconds, engine, samples, loop closure)."""
        model.native_precision = precision
        eng = model.engine(dev, max_batch=B, max_frames=T_FRAMES, want_grad=cfg["edit"])
        eng.set_graph(args.graph)
        eng.set_schedule(diffusion.engine_tables(), key=None)
        eng.set_condition(**cond)
        x = eng.randn((B, N_FEATS, 1, T_FRAMES), seed=seed, first_sample=lo)
        if W > 0:
            eng.sample_loop(x, n_chain - 1, n_chain - W, sampler=sampler, seed=seed, first_sample=lo)
        loop = lambda: eng.sample_loop(x, K - 1, 0, sampler=sampler, seed=seed, first_sample=lo)
        barrier()
        t0 = time.perf_counter()
        loop()
        barrier()
        elapsed = time.perf_counter() - t0
        if use_dist:
            tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        assert torch.isfinite(x).all(), "non-finite samples"
        eng.check_range()
        return elapsed, eng, x, loop

    elapsed, eng, x, loop = timed_run(args.precision)
    if args.pmc_child:
        return
    x_default = x.clone()   # (the roofline leg below replays the K steps on x in place)
    split = eng.precision == "f16x3"
    main_prec = eng.precision

    # the path's only collective: reassemble the generated sequences (outside the timed steps)
    t1 = time.perf_counter()
    full = du.all_gather_batch(x, global_batch)
    torch.cuda.synchronize(dev)
    gather_ms = (time.perf_counter() - t1) * 1e3
    assert full.shape[0] == global_batch

    # whole-job throughput.  Weak scaling: every rank steps its own batch, so the job advances N batch-steps per
    # step time; strong scaling (c5): ONE global batch, so steps/s is 1 / step time and motions/s carries the scaling
    rates = job_rates(layout, world, K, elapsed, n_chain)
    steps_per_s = rates["steps_per_s"]
    passes = 2 if cfg["cfg"] else 1
    flop_step = B * passes * flops_per_sample_eval() * (30.68 / 14.706 if cfg["edit"] else 1.0)
    if is_unet:   # the input-VJP repeats every convolution GEMM once (dX only, no weight gradients)
        flop_step = B * passes * unet_flops_per_sample_eval() * (2.0 if cfg["edit"] else 1.0)
    out = {
        "metric": "diffusion denoising steps/sec", "value": steps_per_s, "unit": "steps/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": rates["ms_per_step"],
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": {"f16x3": "f32 (products as 3 split-f16 MFMAs on 22-bit operand pairs, fp32 accumulate)",
                  "bf16x6": "f32 (operands carried exactly as three bf16 planes, 6 bf16 MFMA partial products per fp32 "
                            "product, fp32 accumulate)",
                  "f32": "f32 (v_mfma_f32_32x32x2_f32)"}[eng.precision],
        "precision_mode": eng.precision, "hip_graph": bool(args.graph),
        "data": "synthetic (random-init MDM weights, z-scored N(0,1) motions, fake CLIP embeddings)",
        "config": {"workload": cfg["desc"], "batch_per_gpu": B, "global_batch": global_batch,
                   "n_frames": T_FRAMES, "n_feats": N_FEATS, "chain_steps": n_chain,
                   "parallelism": f"batch-sharded x{world}"},
        "motions_per_sec": rates["motions_per_sec"],
        "step_tflops": flop_step / (elapsed / K) / 1e12,
        # fraction of the pipe the default mode's products run on: dense f16 peak / 3 products per fp32-equivalent product
        # (against the fp32-MFMA peak the figure exceeds 1 by construction: dropped in round 6)
        "step_frac_of_f16x3_peak": (flop_step / (elapsed / K) / 1e12 / (F16_MFMA_PEAK_TFLOPS / 3.0)) if split else None,
        "allgather_ms": gather_ms, "n_ranks_seen": n_ranks_seen,
        "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if use_dist else None,
        "pipeline_parts": eng.pipeline_parts(),
    }

    want_pmc = rank == 0 and world == 1 and not args.no_pmc and not args.no_roofline
    if rank == 0 and not args.no_roofline:
        # (U-Net: the dominant launches are the level-0 k=5 convolution GEMMs — the largest MFMA count of the step — found by
        # the same rule as the transformer's in_proj; counters measured IN this run since round 3)
        timed = timed_launches(eng, loop)
        pmc = pmc_counters(args.config, eng.precision, B, timeout_s=240.0 if is_unet else 150.0, timed=timed) if want_pmc else {}
        out["roofline"] = roofline_from(eng, timed, split, is_unet, pmc)
        if not is_unet:
            out["roofline_attention"] = roofline_attention(eng, loop)
    if use_dist:
        dist.barrier()

    # the other two arithmetic modes on the same K steps, each with its own driver-timed number and roofline:
    #   f32_exact  CMDI_PREC_F32: v_mfma_f32_32x32x2_f32 products
    #   bf16x6     CMDI_PREC_BF16X6: exact three-plane bf16 operands, six MFMA products (what the f16-range guard of the
    #              default mode falls back to)
    #   (MDM_UNET: bf16x6 only — its convolutions on gemm_x6, round 5; the fp32-MFMA engine is not built for that architecture)
    if rank == 0 and world == 1 and not args.no_f32 and args.precision is None:
        for key, prec in ((("bf16x6", "bf16x6"),) if is_unet else (("bf16x6", "bf16x6"), ("f32_exact", "f32"))):
            if prec == eng.precision:
                continue
            e2, eng2, x2, loop2 = timed_run(prec)
            leg = {"value": K / e2, "unit": "steps/s", "ms_per_step": e2 / K * 1e3, "precision_mode": eng2.precision,
                   "step_tflops": flop_step / (e2 / K) / 1e12,
                   "step_frac_of_pipe_peak": flop_step / (e2 / K) / 1e12 / (F16_MFMA_PEAK_TFLOPS / 6.0 if prec == "bf16x6" else FP32_MFMA_PEAK_TFLOPS),
                   "max_abs_diff_vs_default": float((x2 - x_default).abs().max()),
                   "rel_l2_vs_default": float((x2 - x_default).norm() / x_default.norm())}
            if is_unet and cfg["edit"]:
                # the guided U-Net chain on random weights amplifies rounding by ~10 x per 10 steps (one fp32 ulp of x_T moves the
                # result by 1e-3 ... 1e-1: profiles/r06_unet_guided_chain_attribution.md) — the two legs' samples say that, not parity
                leg["vs_default_note"] = "chaotic chain on random weights: not a parity figure (see tests: one-ulp sensitivity)"
            if not args.no_roofline and not is_unet:
                timed2 = timed_launches(eng2, loop2)
                pmc2 = pmc_counters(args.config, prec, B, timed=timed2) if (want_pmc and prec == "f32") else {}
                leg["roofline"] = roofline_from(eng2, timed2, False, False, pmc2)
            out[key] = leg
        model.native_precision = None

    # hipGraph replay vs eager launches of the same K steps (VERDICT r3 task 6; SURVEY 8d asks config 4 for step latency with
    # and without hipGraph, through ddim_sample_loop AND p_sample_loop on the 'ddim100' respacing).  The engine replays the
    # schedule it runs eagerly: pipeline parts on their own streams, one graph per (part, kind).
    # (round 6: the U-Net configs too — its embedding kernel reads the step's timestep from the chain's device table)
    if rank == 0 and world == 1 and not args.no_graph_leg and args.precision is None:
        legs = {}
        kinds = ("ddim", "ddpm") if args.config == "c4" else (cfg["sampler"],)
        model.native_precision = main_prec     # (pinned: with None, engine() would hand back the f32 engine of the leg above)
        eng_g = model.engine(dev, max_batch=B, max_frames=T_FRAMES, want_grad=cfg["edit"])
        assert eng_g.precision == main_prec
        for kind in kinds:
            sid = N.CMDI_SAMPLER_DDIM if kind == "ddim" else N.CMDI_SAMPLER_DDPM
            leg = {}
            for mode in ("eager", "graph"):
                eng_g.set_graph(mode == "graph")
                eng_g.set_schedule(diffusion.engine_tables(), key=None)
                eng_g.set_condition(**cond)
                xg = eng_g.randn((B, N_FEATS, 1, T_FRAMES), seed=seed, first_sample=lo)
                if W > 0:
                    eng_g.sample_loop(xg, n_chain - 1, n_chain - W, sampler=sid, seed=seed, first_sample=lo)
                # (graph capture happens on the first two steps of a kind: two untimed steps in both modes keep it out of the K)
                hi2 = min(K + 1, n_chain - 1)
                eng_g.sample_loop(xg, hi2, hi2 - 1, sampler=sid, seed=seed, first_sample=lo)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                eng_g.sample_loop(xg, K - 1, 0, sampler=sid, seed=seed, first_sample=lo)
                torch.cuda.synchronize(dev)
                leg[f"{mode}_ms_per_step"] = (time.perf_counter() - t0) / K * 1e3
                if mode == "eager":
                    ref_x = xg.clone()
                else:
                    leg["bitwise_equal"] = bool(torch.equal(ref_x, xg))
            leg["pipeline_parts"] = eng_g.pipeline_parts()
            legs[{"ddim": "ddim_sample_loop", "ddpm": "p_sample_loop"}[kind]] = leg
        eng_g.set_graph(args.graph)
        model.native_precision = None
        out["hip_graph_legs"] = legs

    if rank == 0 and world == 1 and not args.no_cpu and not (is_unet and cfg["edit"]):
        Bc = min(B, 32)   # bounded sample: at most the c2 batch (c4/c5: per-sample cost is the same, scaled below)
        base = cpu_baseline_unet(sd, Bc) if is_unet else cpu_baseline(sd, Bc)
        if Bc != B:
            base["value"] *= Bc / B
            base["sample"] += f"; measured at B={Bc} and scaled by {Bc}/{B} to this config's batch (cost is linear in B)"
        out["cpu_baseline"] = base
        out["gpu_over_cpu"] = steps_per_s / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
