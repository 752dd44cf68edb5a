"""Parity tests proper: the HIP engine (through the C-ABI and the public Python API) against the
golden vectors of the REAL reference and against the numpy oracle.  Needs an MI355X: `-m gpu`.

Every model-level test runs in BOTH arithmetic modes of the engine (include/condmdi.h CMDI_PREC_*):
"f32" (exact fp32 MFMA products) and "f16x3" (fp32-equivalent split-f16 products, the default) —
and both are held to the SAME tolerances.

Stated fp32 tolerances (SURVEY.md §8c asks to state and measure them):
    one denoiser evaluation (8 layers, T<=196)  max-abs <= 1e-4, rel-L2 <= 2e-5 vs reference CPU
    input-VJP of the CFG denoiser               rel-L2 <= 5e-5
    sampler arithmetic given the model output   bit-exact vs oracle
    10-step chains on injected noise            rel-L2 <= 1e-4 on x_0 and on every stored pred_xstart
    Philox4x32-10 words                         bit-exact; N(0,1) values abs <= 2e-6
"""
from types import SimpleNamespace

import os

import numpy as np
import pytest
import torch

from conftest import check_fingerprint, load_golden, max_abs, ok, rel_l2, sub
from oracle import diffusion_oracle as do
from oracle import weights
from oracle.mdm_oracle import MDMOracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def tt(a, dev=DEV):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


PRECISIONS = ["f16x3", "f32", "bf16x6"]


def make_model(case, cfg=None, layers=8, precision=None):
    mu = sub("utils.model_util")
    args = SimpleNamespace(dataset="humanml", unconstrained=not case["text"], layers=layers)
    model, _ = mu.create_model_and_diffusion(args, None)
    sd = weights.make_state_dict(case["weight_seed"], text=case["text"], n_layers=layers)
    mu.load_model_wo_clip(model, weights.to_torch(sd))
    model.to(DEV).eval()
    model.native_precision = precision
    if cfg if cfg is not None else case.get("cfg", False):
        model = sub("model.cfg_sampler").ClassifierFreeSampleModel(model)
        model.eval()
    return model, sd


def make_diffusion(respacing):
    gd, rs = sub("diffusion.gaussian_diffusion"), sub("diffusion.respace")
    betas = gd.get_named_beta_schedule("cosine", 1000)
    return rs.SpacedDiffusion(rs.space_timesteps(1000, respacing or [1000]),
                              gd.DiffusionConfig(betas=betas))


# ---- runtime plumbing ----------------------------------------------------------------------------
def test_single_hip_runtime_and_native_lib_loaded(condmdi):
    condmdi._native.load()
    maps = open("/proc/self/maps").read()
    hip = {ln.split()[-1] for ln in maps.splitlines() if "libamdhip64" in ln}
    assert len(hip) == 1, f"two HIP runtimes in one process: {hip}"
    assert "libcondmdi_hip.so" in maps


# ---- GEMM ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 11, 12, 13, 14, 15, 21, 22, 23, 24, 25])
@pytest.mark.parametrize("shape", [(333, 512, 512), (197 * 4, 1536, 512), (1000, 512, 1024)])
def test_gemm_nt(tile, shape):
    eng = sub("engine")
    m, n, k = shape
    g = torch.Generator().manual_seed(m * 7 + n)
    a = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) * (torch.arange(n).float()[:, None] % 7 + 1)  # asymmetric
    b = torch.randn(n, generator=g)
    r = torch.randn(m, n, generator=g)
    ref = (a.double() @ w.double().T + b.double())
    out = eng.gemm_nt(a.to(DEV), w.to(DEV), b.to(DEV), tile=tile).cpu()
    assert ok("gemm_nt.rel_l2.0", rel_l2(out.numpy(), ref.numpy()), 2e-6)
    out = eng.gemm_nt(a.to(DEV), w.to(DEV), b.to(DEV), tile=tile, epi=3, resid=r.to(DEV)).cpu()
    assert ok("gemm_nt.rel_l2.1", rel_l2(out.numpy(), (ref + r.double()).numpy()), 2e-6)
    out = eng.gemm_nt(a.to(DEV), w.to(DEV), b.to(DEV), tile=tile, epi=1).cpu()
    assert ok("gemm_nt.rel_l2.2", rel_l2(out.numpy(), torch.nn.functional.gelu(ref).numpy()), 2e-6)


# ---- split-f16 (fp32-equivalent) GEMM family -------------------------------------------------------
def test_split_f16_roundtrip():
    """x = hi + lo * 2^-11 to 22 significant bits over the whole f16 range (abs floor 2^-35)."""
    eng = sub("engine")
    g = torch.Generator().manual_seed(3)
    x = torch.randn(257, 512, generator=g) * torch.pow(10.0, torch.rand(257, 512, generator=g) * 9.5 - 6)
    x[0, :8] = torch.tensor([0.0, -0.0, 65000.0, -65000.0, 1e-7, 6e-8, 2.0 ** -14, 1.0])
    s = eng.split_f16(x.to(DEV))
    assert s.shape == (257, 1024) and s.dtype == torch.float16
    back = eng.unsplit_f16(s).cpu().double()
    err = (back - x.double()).abs()
    bound = torch.maximum(x.double().abs() * 2.0 ** -21.5, torch.tensor(2.0 ** -35, dtype=torch.float64))
    assert bool((err <= bound).all()), float((err / bound).max())
    hi_plane = s.view(257, 16, 2, 32)[:, :, 0].reshape(257, 512)
    assert torch.equal(hi_plane.cpu(), x.half())  # hi plane = round-to-nearest f16


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 20])
@pytest.mark.parametrize("shape", [(333, 512, 512), (197 * 4, 1536, 512), (1000, 512, 1024), (130, 256, 32),
                                   (12608, 1536, 64)])
def test_gemm_h3(tile, shape):
    """Same inputs, same float64 reference and the same 2e-6 bound as the exact-fp32 kernels; the
    split-f16 error must also stay within 1.5x of the fp32-MFMA kernel's own error."""
    eng = sub("engine")
    m, n, k = shape
    g = torch.Generator().manual_seed(m * 7 + n)
    a = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) * (torch.arange(n).float()[:, None] % 7 + 1)  # asymmetric
    b = torch.randn(n, generator=g)
    r = torch.randn(m, n, generator=g)
    ref = (a.double() @ w.double().T + b.double())
    a_s, w_s = eng.split_f16(a.to(DEV)), eng.split_f16(w.to(DEV))
    out = eng.gemm_h3(a_s, w_s, b.to(DEV), tile=tile).cpu()
    e_h3 = rel_l2(out.numpy(), ref.numpy())
    e_f32 = rel_l2(eng.gemm_nt(a.to(DEV), w.to(DEV), b.to(DEV)).cpu().numpy(), ref.numpy())
    assert ok("gemm_h3.rel_l2.plain", e_h3, 2e-6) and e_h3 <= 1.5 * e_f32 + 1e-8, (e_h3, e_f32)
    out = eng.gemm_h3(a_s, w_s, b.to(DEV), tile=tile, epi=3, resid=r.to(DEV)).cpu()
    assert ok("gemm_h3.rel_l2.0", rel_l2(out.numpy(), (ref + r.double()).numpy()), 2e-6)
    if n % 32 == 0:   # the residual handed over as split rows (what LayerNorm leaves behind for the next GEMM)
        r_s = eng.split_f16(r.to(DEV))
        out = eng.gemm_h3(a_s, w_s, b.to(DEV), tile=tile, epi=4, resid=r_s).cpu()
        assert ok("gemm_h3.rel_l2.1", rel_l2(out.numpy(), (ref + eng.unsplit_f16(r_s).cpu().double()).numpy()), 2e-6)
        assert ok("gemm_h3.rel_l2.2", rel_l2(out.numpy(), (ref + r.double()).numpy()), 2e-6)
    out = eng.unsplit_f16(eng.gemm_h3(a_s, w_s, b.to(DEV), tile=tile, epi=1)).cpu()
    assert ok("gemm_h3.rel_l2.3", rel_l2(out.numpy(), torch.nn.functional.gelu(ref).numpy()), 2e-6)
    out = eng.unsplit_f16(eng.gemm_h3(a_s, w_s, b.to(DEV), tile=tile, split_out=True)).cpu()
    assert ok("gemm_h3.rel_l2.4", rel_l2(out.numpy(), ref.numpy()), 2e-6)


@pytest.mark.parametrize("shape", [(333, 512, 512), (128, 256, 64), (129, 256, 96), (197 * 4, 1536, 512), (1000, 512, 1024),
                                   (12608, 1536, 64), (12608, 1536, 512), (6304, 1024, 512), (40000, 256, 512), (33000, 768, 128)])
def test_gemm_h3_persistent_is_bitwise_the_tiled_kernel(shape):
    """gemm_h3p.hpp (tile id 50: persistent grid of 128 x 256 tiles + 128 x 128 half tiles in the last partial round, two wave
    groups alternating between fragment reads and products, LDS-DMA ring of three stages running ahead across tile
    boundaries, epilogue with 8 columns per lane) must produce the SAME BITS as the one-tile-per-block kernel (tile 8) for
    every epilogue: which of the two runs is a speed decision taken from M (gemm_h3_persistent_for), and a sample must not
    depend on the size of the batch it is sampled in.  Shapes: ragged last row tile, K of 2 / 3 / 16 / 32 steps, fewer tiles
    than CUs, rounds with and without half tiles."""
    eng = sub("engine")
    m, n, k = shape
    g = torch.Generator().manual_seed(m * 7 + n + k)
    a = torch.randn(m, k, generator=g).to(DEV)
    w = (torch.randn(n, k, generator=g) * (torch.arange(n).float()[:, None] % 7 + 1) * 0.05).to(DEV)
    b = torch.randn(n, generator=g).to(DEV)
    r = torch.randn(m, n, generator=g).to(DEV)
    a_s, w_s, r_s = eng.split_f16(a), eng.split_f16(w), eng.split_f16(r)
    ref64 = a.double() @ w.double().T + b.double()
    for epi, kw in ((0, dict(split_out=True)), (0, {}), (1, {}), (3, dict(resid=r)), (4, dict(resid=r_s))):
        ref = eng.gemm_h3(a_s, w_s, b, tile=8, epi=epi, **kw)
        out = eng.gemm_h3(a_s, w_s, b, tile=50, epi=epi, **kw)
        assert torch.equal(out, ref), (shape, epi, kw.keys())
    assert ok("gemm_h3_persistent_is_bitwise_the_tiled_kernel.rel_l2.0", rel_l2(eng.gemm_h3(a_s, w_s, b, tile=50).cpu().numpy(), ref64.cpu().numpy()), 2e-6)


def test_persistent_gemm_keep_path_is_bitwise_the_tiled_schedule(cases, tmp_path):
    """at M >= 32,768 rows (B = 256) the stashing forward pass of a reconstruction-guidance step runs its folded-
    LayerNorm GEMMs (ln_part, ln_c1, ln_rg, out_part, ln_stats, aux under a folded A operand, C + Cs together) on the
    PERSISTENT kernel, which no golden reaches at that size.  Here the same schedule is forced on the small goldens:
    CMDI_H3_PERSIST=1 (read once per process, hence the subprocesses) sends every supported GEMM to gemm_h3p.  Forward with
    stash, input-VJP and a reconstruction-guidance chain must equal the tiled schedule BIT FOR BIT, and meet the
    reference's golden values."""
    import os
    import subprocess
    import sys
    helper = str(__import__("pathlib").Path(__file__).resolve().parent / "helpers" / "persist_probe.py")
    outs = {}
    for mode in ("0", "1"):
        path = tmp_path / f"persist{mode}.npz"
        env = dict(os.environ, CMDI_H3_PERSIST=mode)
        r = subprocess.run([sys.executable, helper, str(path)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[mode] = np.load(path)
    for key in ("vjp_out", "vjp_gx", "chain_final"):
        assert np.array_equal(outs["0"][key], outs["1"][key]), key
    g = load_golden("vjp_text_cfg")
    assert ok("persistent_keep_path.fwd", rel_l2(outs["1"]["vjp_out"], g["out"]), 2e-5)
    assert ok("persistent_keep_path.vjp", rel_l2(outs["1"]["vjp_gx"], g["gx"]), 5e-5)
    assert ok("persistent_keep_path.chain", rel_l2(outs["1"]["chain_final"], load_golden("chain_edit_recon")["final"]), 1e-4)


@pytest.mark.parametrize("m", [1, 31, 32, 33, 333, 788, 3940, 6304, 12608, 40000])
@pytest.mark.parametrize("n", [128, 512, 1024, 1536])
def test_gemm_h3w_is_bitwise_the_tiled_kernel(m, n):
    """gemm_h3w.hpp (tile id 60; round 6): the weight-stationary form for K = 512 — a wave owns 32 output columns for the whole K
    (its W fragments fill the 256 accumulation registers, read from a fragment-ordered copy of W), A streams through LDS in
    32-row tiles, the MFMA stream is hand-issued, the epilogue of an interior tile rides in the MFMA gaps of the next tile's
    stream — must produce the SAME BITS as the one-tile-per-block kernel (tile 8) for every epilogue: products and their
    order per output and the epilogue arithmetic are the same, and a sample must not depend on which kernel its batch size
    selects.  Row counts: one ragged tile, fewer tiles than XCDs, sub-ranges of 0 / 1 / several tiles, 1 - 3 passes."""
    eng = sub("engine")
    k = 512
    g = torch.Generator().manual_seed(m * 7 + n)
    a = torch.randn(m, k, generator=g).to(DEV)
    w = (torch.randn(n, k, generator=g) * (torch.arange(n).float()[:, None] % 7 + 1) * 0.05).to(DEV)
    b = torch.randn(n, generator=g).to(DEV)
    r = torch.randn(m, n, generator=g).to(DEV)
    a_s, w_s, r_s = eng.split_f16(a), eng.split_f16(w), eng.split_f16(r)
    for epi, kw in ((0, dict(split_out=True)), (0, {}), (1, {}), (3, dict(resid=r)), (4, dict(resid=r_s))):
        ref = eng.gemm_h3(a_s, w_s, b, tile=8, epi=epi, **kw)
        out = eng.gemm_h3(a_s, w_s, b, tile=60, epi=epi, **kw)
        assert torch.equal(out, ref), (m, n, epi, kw.keys())
    ref64 = a.double() @ w.double().T + b.double()
    assert ok("gemm_h3w_is_bitwise_the_tiled_kernel.rel_l2.0", rel_l2(eng.gemm_h3(a_s, w_s, b, tile=60).cpu().numpy(), ref64.cpu().numpy()), 2e-6)


def test_weight_stationary_schedule_is_bitwise_the_tiled_schedule(cases, tmp_path):
    """CMDI_H3W=1 routes every K = 512 GEMM of the engine whose launch has at least CMDI_H3W_MIN_M rows to gemm_h3w — with the
    fragment-ordered weight copies packed at cmdi_finalize_weights, the folded LayerNorms (ln_part through LDS-DMA, ln_c1, ln_rg,
    out_part, ln_stats), the stash (aux) and the backward GEMMs (GELU-gradient and residual epilogues on the packed transposes).
    Forced on the small goldens (MIN_M = 1; subprocesses: the switches are read when an engine is created): forward with stash,
    input-VJP and a reconstruction-guidance chain must equal the tiled schedule BIT FOR BIT."""
    import os
    import subprocess
    import sys
    helper = str(__import__("pathlib").Path(__file__).resolve().parent / "helpers" / "persist_probe.py")
    outs = {}
    for mode in ("0", "1"):
        path = tmp_path / f"h3w{mode}.npz"
        env = dict(os.environ, CMDI_H3W=mode, CMDI_H3W_MIN_M="1")
        r = subprocess.run([sys.executable, helper, str(path)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[mode] = np.load(path)
    for key in ("vjp_out", "vjp_gx", "chain_final"):
        assert np.array_equal(outs["0"][key], outs["1"][key]), key


def test_attention_split_schedule_is_bitwise_identical(tmp_path):
    """Round 4: below half a chip of (sequence, head) pairs the attention core runs as two 4-wave blocks per pair
    (attention_h3.hip launch_attention_h3).  A sample must not depend on the schedule its batch size selects: forced off,
    forced on and the default give the same bits, at a small batch, at S = 129 (second block holds one query) and at a
    batch that fills the chip."""
    import os
    import subprocess
    import sys
    helper = str(__import__("pathlib").Path(__file__).resolve().parent / "helpers" / "attn_split_probe.py")
    outs = {}
    for mode in ("0", "1", None):
        path = tmp_path / f"split{mode}.npz"
        env = {k: v for k, v in os.environ.items() if k != "CMDI_ATTN_SPLIT"}
        if mode is not None:
            env["CMDI_ATTN_SPLIT"] = mode
        r = subprocess.run([sys.executable, helper, str(path)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[mode] = np.load(path)
    for key in ("small", "edge", "big"):
        assert np.isfinite(outs["0"][key]).all()
        assert np.array_equal(outs["0"][key], outs["1"][key]), key
        assert np.array_equal(outs["0"][key], outs[None][key]), key


def test_attention_persistent_schedule_is_bitwise_identical(tmp_path):
    """Round 6: with at least two (sequence, head) pairs per CU the attention core runs as ONE persistent block per CU that walks
    its share of the pairs (attention_h3.hip PERSIST: the next pair's first requests travel under the previous pair's output
    stores) instead of one block per pair.  A wave runs the same instruction sequence per pair in both: forced off and on
    must give the same bits — pairs split unevenly over the blocks (532 on 256), S = 197 and S = 150; the batches below the
    threshold take the one-block-per-pair kernel either way."""
    import os
    import subprocess
    import sys
    helper = str(__import__("pathlib").Path(__file__).resolve().parent / "helpers" / "attn_split_probe.py")
    outs = {}
    for mode in ("0", "1"):
        path = tmp_path / f"persist{mode}.npz"
        env = dict(os.environ, CMDI_ATTN_PERSIST=mode)
        r = subprocess.run([sys.executable, helper, str(path)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[mode] = np.load(path)
    for key in ("small", "big", "many", "many_short"):
        assert np.isfinite(outs["1"][key]).all()
        assert np.array_equal(outs["0"][key], outs["1"][key]), key


# ---- bf16x6: exact three-plane bf16 operands, six MFMA products (fp32-class, no operand truncation) -----------------
def test_pack_x6_is_exact():
    """W = p0 + p1 + p2 EXACTLY for every finite binary32 (24 significant bits = three bf16 mantissas), over the whole
    fp32 exponent range — no range limit, unlike the split-f16 format."""
    eng = sub("engine")
    g = torch.Generator().manual_seed(4)
    x = torch.randn(129, 512, generator=g) * torch.pow(10.0, torch.rand(129, 512, generator=g) * 60 - 30)
    x[0, :8] = torch.tensor([0.0, -0.0, 65504.0, -7e4, 3.38e38, 1e-30, 2.0 ** -14, 1.0])   # bf16 max = 3.3895e38
    planes = eng.pack_x6(x.to(DEV)).cpu()
    assert planes.shape == (129, 16, 3, 32) and planes.dtype == torch.bfloat16
    back = planes.double().sum(dim=2).reshape(129, 512)
    assert torch.equal(back, x.double())
    assert torch.equal(planes[:, :, 0].reshape(129, 512), x.bfloat16())   # leading plane = round-to-nearest bf16


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("shape", [(333, 512, 512), (197 * 4, 1536, 512), (1000, 512, 1024), (130, 256, 32),
                                   (12608, 1536, 64), (12608, 1024, 512), (257 * 128 + 5, 512, 512)])
def test_gemm_x6(variant, shape):
    """Same inputs, float64 reference and 2e-6 bound as the fp32-MFMA kernels; the bf16x6 error must not exceed the
    fp32-MFMA kernel's by more than 10 % (it carries all 24 bits of both operands)."""
    eng = sub("engine")
    m, n, k = shape
    g = torch.Generator().manual_seed(m * 7 + n)
    a = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) * (torch.arange(n).float()[:, None] % 7 + 1)  # asymmetric
    b = torch.randn(n, generator=g)
    r = torch.randn(m, n, generator=g)
    ref = (a.double() @ w.double().T + b.double())
    wx = eng.pack_x6(w.to(DEV))
    out = eng.gemm_x6(a.to(DEV), wx, b.to(DEV), variant=variant).cpu()
    e_x6 = rel_l2(out.numpy(), ref.numpy())
    e_f32 = rel_l2(eng.gemm_nt(a.to(DEV), w.to(DEV), b.to(DEV)).cpu().numpy(), ref.numpy())
    assert ok("gemm_x6.rel_l2.plain", e_x6, 2e-6) and e_x6 <= 1.1 * e_f32 + 1e-8, (e_x6, e_f32)
    out = eng.gemm_x6(a.to(DEV), wx, b.to(DEV), epi=3, resid=r.to(DEV), variant=variant).cpu()
    assert ok("gemm_x6.rel_l2.0", rel_l2(out.numpy(), (ref + r.double()).numpy()), 2e-6)
    out = eng.gemm_x6(a.to(DEV), wx, b.to(DEV), epi=1, variant=variant).cpu()
    assert ok("gemm_x6.rel_l2.1", rel_l2(out.numpy(), torch.nn.functional.gelu(ref).numpy()), 2e-6)
    # no range limit: operands far outside the f16 range
    big = eng.gemm_x6((a * 1e6).to(DEV), eng.pack_x6((w * 1e5).to(DEV)), None, variant=variant).cpu()
    assert ok("gemm_x6.rel_l2.2", rel_l2(big.numpy(), (ref - b.double()).numpy() * 1e11), 2e-6)


@pytest.mark.parametrize("shape", [(333, 512), (12608, 1024), (64, 32)])
def test_fused_layernorm_gemm(shape):
    """out = LayerNorm(A·Wᵀ + b + R) with the normalisation inside the GEMM epilogue, vs float64."""
    eng = sub("engine")
    m, k = shape
    g = torch.Generator().manual_seed(m + k)
    a = torch.randn(m, k, generator=g)
    w = torch.randn(512, k, generator=g) * 0.05 * (torch.arange(512).float()[:, None] % 5 + 1)
    b, r = torch.randn(512, generator=g), torch.randn(m, 512, generator=g)
    gamma, beta = torch.rand(512, generator=g) + 0.5, torch.randn(512, generator=g)
    x = a.double() @ w.double().T + b.double() + r.double()
    ref = torch.nn.functional.layer_norm(x, (512,), gamma.double(), beta.double(), 1e-5)
    y, ys = eng.gemm_h3_ln(eng.split_f16(a.to(DEV)), eng.split_f16(w.to(DEV)), b.to(DEV), r.to(DEV),
                           gamma.to(DEV), beta.to(DEV), want_split=True)
    assert ok("fused_layernorm_gemm.rel_l2.0", rel_l2(y.cpu().numpy(), ref.numpy()), 2e-6), rel_l2(y.cpu().numpy(), ref.numpy())
    assert ok("fused_layernorm_gemm.rel_l2.1", rel_l2(eng.unsplit_f16(ys).cpu().numpy(), ref.numpy()), 2e-6)


def test_f16x3_range_guard():
    """|x| >= 65504 cannot be split: an engine PINNED to f16x3 refuses such weights at finalize; the default precision
    falls back to bf16x6 (exact three-plane operands, fp32's exponent range) and matches the oracle."""
    N = sub("_native")
    case = dict(text=False, weight_seed=5)
    model, sd = make_model(case, layers=1, precision="f16x3")
    with torch.no_grad():
        model.seqTransEncoder.layers[0].linear1.weight[3, 7] = 7e4
    with pytest.raises(N.RangeError, match="f16 range"):
        model.engine(torch.device(DEV), max_batch=2, max_frames=20)
    # seeded input: a 7e4 weight amplifies the fp32 rounding of one token feature by 7e4 in front of a GELU, so an unlucky
    # draw (that pre-activation near 0 for some token) moves single outputs by 1e-3 relative in ANY fp32 implementation
    # (3.6e-5 rel-L2 seen once in round 3 with an unseeded draw; seeds 0-7: 5.8e-7 ... 6.4e-7 in all three modes)
    x = torch.randn(2, 263, 1, 20, generator=torch.Generator().manual_seed(0)).to(DEV)
    t = torch.tensor([5, 9], device=DEV)
    sd2 = dict(sd)
    sd2["seqTransEncoder.layers.0.linear1.weight"] = sd["seqTransEncoder.layers.0.linear1.weight"].copy()
    sd2["seqTransEncoder.layers.0.linear1.weight"][3, 7] = 7e4
    want = MDMOracle(sd2).forward(x.cpu().numpy(), t.cpu().numpy())
    for precision in (None, "bf16x6", "f32"):     # None = library default: f16x3, falling back to bf16x6 here
        model.native_precision = precision
        model.invalidate_engine()
        out = model(x, t, y={})
        assert model._engine.precision == (precision or "bf16x6")
        assert ok("f16x3_range_guard.rel_l2.0", rel_l2(out.cpu().numpy(), want), 2e-5, precision=model._engine.precision), \
            (precision, rel_l2(out.cpu().numpy(), want))


@pytest.mark.parametrize("scale,expect", [(5.0e3, "f16x3"), (4.0e4, "bf16x6")])
def test_real_scale_activations_and_range_fallback(scale, expect):
    """'real-scale weights: a range test with LayerNorm-free FFN pre-activations near 1e4'.  linear1 is scaled
    so that the FFN pre-activations (and the GELU outputs the f16x3 path has to split) reach ~1e4: still inside the f16
    range -> the default engine stays f16x3 and matches the oracle; scaled 20x further they pass 65504, the device flag
    fires and p_sample_loop re-runs the SAME chain (same seed) on a bf16x6 engine — result again within tolerance."""
    case = dict(text=False, weight_seed=6)
    model, sd = make_model(case, layers=2)
    key = "seqTransEncoder.layers.0.linear1.weight"
    sd = dict(sd)
    sd[key] = sd[key] * np.float32(scale)
    sub("utils.model_util").load_model_wo_clip(model, weights.to_torch(sd))
    model.to(DEV).eval()
    diffusion = make_diffusion([3])
    B, T = 2, 40
    rng = np.random.default_rng(8)
    shape = (B, 263, 1, T)
    x_T = rng.standard_normal(shape).astype(np.float32)
    noise = rng.standard_normal((3,) + shape).astype(np.float32)
    y = {"mask": torch.ones(B, 1, 1, T, dtype=torch.bool, device=DEV), "lengths": torch.full((B,), T)}
    diffusion.injected_noise = tt(noise)
    out = diffusion.p_sample_loop(model, shape, noise=tt(x_T), clip_denoised=False, model_kwargs={"y": y}).cpu().numpy()
    assert model._engine.precision == expect
    oracle = MDMOracle(sd)
    keep = []
    oracle.forward(x_T, np.full(B, 999), keep=keep)
    pre = np.abs(keep[0]["u"]).max()          # linear1 output of layer 0 = the FFN pre-activation
    assert (pre < 65504.0) == (expect == "f16x3") and pre > 5e3, pre
    sch = do.Schedule(do.named_betas("cosine", 1000), do.space_timesteps(1000, [3]))
    want = do.sample_loop(sch, oracle, x_T, noise)
    assert ok("real_scale_activations_and_range_fallback.rel_l2.0", rel_l2(out, want), 1e-4), rel_l2(out, want)


# ---- attention -----------------------------------------------------------------------------------
@pytest.mark.parametrize("kernel", ["attention_fwd", "attention_fwd_h3"])
@pytest.mark.parametrize("S", [197, 61, 16, 17, 224, 33, 193, 1])
def test_attention_core_vs_torch(S, kernel):
    eng = sub("engine")
    n_seq, H = 3, 4
    g = torch.Generator().manual_seed(S)
    qkv = torch.randn(n_seq * S, 3 * H * 128, generator=g)
    qkv[:, :512] *= 3.0  # sharper softmax; a spiked key forces the online-max rescale late in the row
    qkv[S - 1, 512:1024] *= 4.0
    q, k, v = (qkv[:, i * 512:(i + 1) * 512].double().view(n_seq, S, H, 128).transpose(1, 2) for i in range(3))
    p = torch.softmax(q @ k.transpose(-1, -2) / 128 ** 0.5, dim=-1)
    ref = (p @ v).transpose(1, 2).reshape(n_seq * S, 512)
    out = getattr(eng, kernel)(qkv.to(DEV), n_seq, S, H).cpu()
    assert ok("attention_core_vs_torch.rel_l2.0", rel_l2(out.numpy(), ref.numpy()), 2e-6), rel_l2(out.numpy(), ref.numpy())


# ---- RNG -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("S", [197, 61, 33])
def test_attention_h3_ignores_memory_behind_a_sequence(S):
    """the K / V stages of the split-f16 kernel read whole 32-key blocks; keys past the sequence are fetched
    through a buffer descriptor that ends with the sequence, with the stage position in the range-checked vector offset.
    NaN patterns directly behind the tensor (last sequence) and Inf rows in the NEXT sequence (first sequence) must not
    reach the output: 0 * NaN in the P·V product would.  Output bit-identical to the run with zeros there."""
    eng = sub("engine")
    n_seq, H = 2, 4
    M = n_seq * S
    g = torch.Generator().manual_seed(100 + S)
    qs = eng.split_f16(torch.randn(M, 3 * H * 128, generator=g).to(DEV))
    buf = torch.zeros(M + 64, qs.shape[1], dtype=torch.float16, device=DEV)
    buf[:M] = qs
    clean = eng.attention_fwd_h3_split(buf, n_seq, S, H)
    buf[M:] = float("nan")
    dirty = eng.attention_fwd_h3_split(buf, n_seq, S, H)
    assert torch.isfinite(dirty).all() and torch.equal(clean, dirty)
    # sequence 0 must not see sequence 1's rows either: make them Inf / NaN and compare sequence 0's output
    buf[S:M] = float("inf")
    buf[S:M, ::3] = float("nan")
    first = eng.attention_fwd_h3_split(buf, n_seq, S, H)[:S]
    assert torch.isfinite(first).all() and torch.equal(first, clean[:S])


@pytest.mark.parametrize("S", [197, 61, 16, 17, 224, 33, 193, 1, 129, 160])
def test_attention_vjp_h3_vs_torch_autograd(S):
    """The input-VJP of the attention core alone (dQ / dK / dV kernels of attention_h3.hip through cmdi_attention_vjp_h3)
    against float64 torch.autograd of softmax(QK^T / sqrt(128)) V — every tile count 1..7 on both axes, tails of 1, 5 and
    16 rows, a sharpened softmax and a spiked key."""
    eng = sub("engine")
    n_seq, H = 3, 4
    g = torch.Generator().manual_seed(1000 + S)
    qkv = torch.randn(n_seq * S, 3 * H * 128, generator=g)
    qkv[:, :512] *= 3.0
    qkv[S - 1, 512:1024] *= 4.0
    dout = torch.randn(n_seq * S, H * 128, generator=g)
    x = qkv.double().requires_grad_(True)
    q, k, v = (x[:, i * 512:(i + 1) * 512].view(n_seq, S, H, 128).transpose(1, 2) for i in range(3))
    p = torch.softmax(q @ k.transpose(-1, -2) / 128 ** 0.5, dim=-1)
    out = (p @ v).transpose(1, 2).reshape(n_seq * S, 512)
    want, = torch.autograd.grad((out * dout.double()).sum(), x)
    got = eng.attention_vjp_h3(qkv.to(DEV), dout.to(DEV), n_seq, S, H).cpu()
    whole = float(np.linalg.norm(want.numpy()))
    for i, part in enumerate("qkv"):
        a, b = got[:, i * 512:(i + 1) * 512].numpy(), want[:, i * 512:(i + 1) * 512].numpy()
        # (a part that is small next to the whole gradient is measured against a tenth of the whole; S = 1: P = 1, so dQ = dK = 0
        # exactly in the reference and the rounding residue of dP - D is measured against the whole gradient)
        err = float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), (1.0 if S == 1 else 0.1) * whole))
        assert ok(f"attention_vjp_h3.d{part}", err, 5e-6), (part, err)


def test_attention_vjp_h3_has_no_row_outliers_under_sparse_large_gradients():
    """Regression test of round 5's finding (profiles/r05_guided_error_attribution.md).  Reconstruction guidance drives the
    attention backward with d out rows that are large on a FEW queries (the keyframes) and zero elsewhere; a single softmax
    probability that is off by one f16 ulp — rounds 1-4: hi and lo of the split P operand of dV taken from different roundings of
    a contracted product, one entry in ~4,000 — then stands out in ONE row of dV by two orders of magnitude while every
    whole-tensor norm stays fine (the test above never saw it).  Here: 4 sequences x 197 tokens, diffuse attention, d out on every
    fifth query only, and the error is measured PER ROW of dq / dk / dv against float64 autograd.  On the round-4 kernel (library
    variant built from the parent commit of the fix, same box) this test FAILS with a worst dV row at 6.3e-5 of its norm, median
    row 3.6e-6; at HEAD the worst row of dq / dk / dv is 2.2e-7."""
    eng = sub("engine")
    n_seq, S, H = 4, 197, 4
    g = torch.Generator().manual_seed(20260927)
    qkv = torch.randn(n_seq * S, 3 * H * 128, generator=g)
    qkv[:, :1024] *= 0.3                                   # diffuse softmax rows (P ~ 1 / S), as under the random-init denoiser
    dout = torch.zeros(n_seq * S, H * 128)
    key_rows = torch.arange(0, n_seq * S, 5)
    dout[key_rows] = torch.randn(len(key_rows), H * 128, generator=g) * 8.0
    x = qkv.double().requires_grad_(True)
    q, k, v = (x[:, i * 512:(i + 1) * 512].view(n_seq, S, H, 128).transpose(1, 2) for i in range(3))
    p = torch.softmax(q @ k.transpose(-1, -2) / 128 ** 0.5, dim=-1)
    out = (p @ v).transpose(1, 2).reshape(n_seq * S, 512)
    want, = torch.autograd.grad((out * dout.double()).sum(), x)
    got = eng.attention_vjp_h3(qkv.to(DEV), dout.to(DEV), n_seq, S, H).cpu().double()
    for i, part in enumerate("qkv"):
        a, b = got[:, i * 512:(i + 1) * 512], want[:, i * 512:(i + 1) * 512]
        row_norm = b.norm(dim=1)
        floor = 0.05 * float(row_norm.median()) if float(row_norm.median()) > 0 else 1e-30   # (dq rows of queries without d out are 0)
        rel = ((a - b).norm(dim=1) / torch.clamp(row_norm, min=max(floor, 1e-30)))
        rel = rel[row_norm > 0] if part != "q" else rel[key_rows]
        worst, med = float(rel.max()), float(rel.median())
        assert ok(f"attention_vjp_rows.d{part}.worst_row", worst, 1e-5), (part, worst, med)
        assert worst <= 20 * med + 1e-12, (part, worst, med)       # no heavy tail: the worst row is an ordinary row (it was 100-700 x)


def test_engine_rng_matches_oracle():
    Engine = sub("engine").Engine
    e = Engine(n_layers=0, d_model=0, d_ff=0, n_heads=0, n_feats=263, max_frames=196, max_batch=4,
               device=DEV)
    z = e.randn((4, 263, 1, 196), seed=0x1234ABCD5678, first_sample=3, step=7).cpu().numpy()
    want = do.engine_randn(4, 263 * 196, seed=0x1234ABCD5678, first_sample=3, step=7)
    assert ok("engine_rng_matches_oracle.max_abs.0", max_abs(z.reshape(4, -1), want), 2e-6)
    odd = e.randn((3, 263, 1, 59), seed=5, step=-1).cpu().numpy()  # per-sample size not % 4
    assert ok("engine_rng_matches_oracle.max_abs.1", max_abs(odd.reshape(3, -1), do.engine_randn(3, 263 * 59, seed=5)), 2e-6)


# ---- denoiser --------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", PRECISIONS)
def test_forward_uncond_vs_reference(cases, precision):
    case = cases.CASES["fwd_uncond"]
    inp = cases.make_inputs(case)
    check_fingerprint(cases, "fwd_uncond", inp)
    model, sd = make_model(case, precision=precision)
    assert model.engine(torch.device(DEV), max_batch=inp["x"].shape[0],
                        max_frames=inp["x"].shape[-1]).precision == precision
    out = model(tt(inp["x"]), tt(inp["t"]), y={}).cpu().numpy()
    ref = load_golden("fwd_uncond")["out"]
    assert ok("forward_uncond_vs_reference.max_abs.0", max_abs(out, ref), 1e-4) and ok("forward_uncond_vs_reference.rel_l2.0", rel_l2(out, ref), 2e-5), (max_abs(out, ref), rel_l2(out, ref))
    assert ok("forward_uncond_vs_reference.rel_l2.1", rel_l2(out, MDMOracle(sd).forward(inp["x"], inp["t"])), 2e-5)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_forward_text_cfg_vs_reference(cases, precision):
    case = cases.CASES["fwd_text"]
    inp = cases.make_inputs(case)
    check_fingerprint(cases, "fwd_text", inp)
    model, _ = make_model(case, cfg=False, precision=precision)
    g = load_golden("fwd_text")
    x, t = tt(inp["x"]), tt(inp["t"])
    y = {"text_embed": tt(inp["enc_text"])}
    oc = model(x, t, y=y).cpu().numpy()
    ou = model(x, t, y=dict(y, uncond=True)).cpu().numpy()
    wrapped = sub("model.cfg_sampler").ClassifierFreeSampleModel(model)
    cfg = wrapped(x, t, y=dict(y, text_scale=tt(inp["text_scale"]))).cpu().numpy()
    for mine, key in ((oc, "out_cond"), (ou, "out_uncond"), (cfg, "out_cfg")):
        assert ok("forward_text_cfg_vs_reference.max_abs.0", max_abs(mine, g[key]), 2e-4) and ok("forward_text_cfg_vs_reference.rel_l2.0", rel_l2(mine, g[key]), 2e-5), \
            (key, max_abs(mine, g[key]), rel_l2(mine, g[key]))


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("B,T", [(1, 1), (1, 2), (2, 31), (3, 64), (5, 100), (1, 196), (2, 223), (9, 33)])
def test_forward_other_shapes_vs_oracle(precision, B, T):
    """Frame counts and batch sizes off the golden shapes — one frame, the largest sequence the attention kernels take
    (T = 223: S = 224), row counts that leave ragged last tiles — CFG forward vs the numpy oracle, plus the input-VJP at
    the same shapes (the stashing forward takes the un-folded LayerNorm schedule)."""
    case = dict(text=True, weight_seed=17, cfg=True)
    model, sd = make_model(case, layers=2, precision=precision)
    oracle = MDMOracle(sd)
    rng = np.random.default_rng(3000 + 17 * B + T)
    shape = (B, 263, 1, T)
    x = rng.standard_normal(shape).astype(np.float32)
    t = rng.integers(0, 1000, B)
    enc = rng.standard_normal((B, 512)).astype(np.float32)
    scale = rng.uniform(0.0, 3.0, B).astype(np.float32)
    y = {"text_embed": tt(enc), "text_scale": tt(scale)}
    want, _, _ = oracle.forward_cfg(x, t, enc, scale)
    got = model(tt(x), tt(t), y=y).cpu().numpy()
    assert np.isfinite(got).all()
    assert ok("forward_other_shapes_vs_oracle.max_abs.0", max_abs(got, want), 2e-4) and ok("forward_other_shapes_vs_oracle.rel_l2.0", rel_l2(got, want), 2e-5), (max_abs(got, want), rel_l2(got, want))
    gout = rng.standard_normal(shape).astype(np.float32)
    z = tt(x).requires_grad_(True)
    with torch.enable_grad():
        out = model(z, tt(t), y=y)
        gx, = torch.autograd.grad((out * tt(gout)).sum(), z)
    assert ok("forward_other_shapes_vs_oracle.rel_l2.1", rel_l2(out.detach().cpu().numpy(), want), 2e-5)
    want_gx = oracle.vjp_cfg(x, t, gout, enc, scale)
    want_gx = want_gx[0] if isinstance(want_gx, tuple) else want_gx
    assert ok("forward_other_shapes_vs_oracle.rel_l2.2", rel_l2(gx.cpu().numpy(), want_gx), 5e-5), rel_l2(gx.cpu().numpy(), want_gx)


@pytest.mark.parametrize("fold", ["0", "1"])
def test_forward_layernorm_schedules(cases, monkeypatch, fold):
    """f16x3 forward with the LayerNorms folded into their consuming GEMMs (default: no LayerNorm pass, the residual stream
    travels pre-LayerNorm with per-row partial statistics) and with the separate LayerNorm kernels (CMDI_LN_FOLD=0): both
    meet the reference's golden outputs at the same tolerance, and agree with each other far below it."""
    monkeypatch.setenv("CMDI_LN_FOLD", fold)
    case = cases.CASES["fwd_text"]
    inp = cases.make_inputs(case)
    model, _ = make_model(case, cfg=True, precision="f16x3")
    g = load_golden("fwd_text")
    out = model(tt(inp["x"]), tt(inp["t"]), y={"text_embed": tt(inp["enc_text"]), "text_scale": tt(inp["text_scale"])})
    out = out.cpu().numpy()
    assert ok("forward_layernorm_schedules.max_abs.0", max_abs(out, g["out_cfg"]), 2e-4) and ok("forward_layernorm_schedules.rel_l2.0", rel_l2(out, g["out_cfg"]), 2e-5), (max_abs(out, g["out_cfg"]), rel_l2(out, g["out_cfg"]))


@pytest.mark.parametrize("fold_keep", ["0", "1"])
def test_vjp_stash_formats(cases, monkeypatch, fold_keep):
    """The input-VJP over both stash formats of the f16x3 forward pass (round 4): split rows written once by the folded
    schedule (default, the stash is the stream) and the fp32 rows of the schedule with separate LayerNorm kernels
    (CMDI_LN_FOLD_KEEP=0).  Same golden, same bound."""
    monkeypatch.setenv("CMDI_LN_FOLD_KEEP", fold_keep)
    case = cases.CASES["vjp_text_cfg"]
    inp = cases.make_inputs(case)
    model, _ = make_model(case, precision="f16x3")
    B, _, _, T = inp["x"].shape
    eng = model.model.engine(torch.device(DEV), max_batch=B, max_frames=T, want_grad=True)
    eng.set_condition(batch=B, n_frames=T, cfg=True, enc_text=tt(inp["enc_text"]), text_scale=tt(inp["text_scale"]))
    g = load_golden("vjp_text_cfg")
    out = eng.mdm_forward(tt(inp["x"]), tt(inp["t"])).cpu().numpy()
    assert ok("vjp_stash_formats.fwd", rel_l2(out, g["out"]), 2e-5)
    gx = eng.mdm_vjp(tt(inp["gout"])).cpu().numpy()
    assert ok("vjp_stash_formats.vjp", rel_l2(gx, g["gx"]), 5e-5), rel_l2(gx, g["gx"])


@pytest.mark.parametrize("precision", PRECISIONS)
def test_vjp_vs_reference_autograd(cases, precision):
    case = cases.CASES["vjp_text_cfg"]
    inp = cases.make_inputs(case)
    check_fingerprint(cases, "vjp_text_cfg", inp)
    model, _ = make_model(case, precision=precision)
    mdm = model.model
    B, _, _, T = inp["x"].shape
    eng = mdm.engine(torch.device(DEV), max_batch=B, max_frames=T, want_grad=True)
    eng.set_condition(batch=B, n_frames=T, cfg=True, enc_text=tt(inp["enc_text"]),
                      text_scale=tt(inp["text_scale"]))
    g = load_golden("vjp_text_cfg")
    out = eng.mdm_forward(tt(inp["x"]), tt(inp["t"])).cpu().numpy()
    assert ok("vjp_vs_reference_autograd.rel_l2.0", rel_l2(out, g["out"]), 2e-5)
    gx = eng.mdm_vjp(tt(inp["gout"])).cpu().numpy()
    assert ok("vjp_vs_reference_autograd.rel_l2.1", rel_l2(gx, g["gx"]), 5e-5), rel_l2(gx, g["gx"])
    # the VJP is linear in gout: tiny and huge output gradients must come back to the same
    # tolerance (f16x3: the power-of-two gradient scale keeps them inside the f16 range)
    for k in (1e-12, 1e9):
        gk = eng.mdm_vjp(tt(inp["gout"] * np.float32(k))).cpu().numpy().astype(np.float64) / k
        assert ok("vjp_vs_reference_autograd.rel_l2.2", rel_l2(gk, g["gx"]), 5e-5), (k, rel_l2(gk, g["gx"]))


@pytest.mark.parametrize("bits", ["1", "6", "7"])
def test_vjp_with_fp32_copies_beside_the_split_stash(cases, bits, monkeypatch):
    """CMDI_STASH_F32 (round 5; the attribution switches of profiles/r05_guided_error_attribution.md, kept as a selectable
    schedule): the folded stashing forward also writes fp32 copies of the attention output (bit 1) / pre1 (2) / pre2 (4) and the
    backward reads those instead of the split rows.  Same forward bits, an input-VJP within the usual bound of the reference's
    autograd, and within rounding of the default data flow (the 22-bit stash is NOT where the guided path lost accuracy)."""
    case = cases.CASES["vjp_text_cfg"]
    inp = cases.make_inputs(case)
    g = load_golden("vjp_text_cfg")
    B, _, _, T = inp["x"].shape
    outs = {}
    for mode in ("0", bits):
        monkeypatch.setenv("CMDI_STASH_F32", mode)
        model, _ = make_model(case, precision="f16x3")
        eng = model.model.engine(torch.device(DEV), max_batch=B, max_frames=T, want_grad=True)
        eng.set_condition(batch=B, n_frames=T, cfg=True, enc_text=tt(inp["enc_text"]), text_scale=tt(inp["text_scale"]))
        out = eng.mdm_forward(tt(inp["x"]), tt(inp["t"])).cpu().numpy()
        gx = eng.mdm_vjp(tt(inp["gout"])).cpu().numpy()
        outs[mode] = (out, gx)
        assert ok("vjp_stash_f32.vjp", rel_l2(gx, g["gx"]), 5e-5), (mode, rel_l2(gx, g["gx"]))
    assert np.array_equal(outs["0"][0], outs[bits][0])                       # the copies do not touch the forward values
    assert ok("vjp_stash_f32.vs_default", rel_l2(outs[bits][1], outs["0"][1]), 2e-6), rel_l2(outs[bits][1], outs["0"][1])


def test_torch_autograd_through_the_native_denoiser(cases):
    """torch.autograd.grad(loss(model(z, t, **kw)), z) — the reference's reconstruction-guidance / cond_fn pattern
    (gaussian_diffusion.py:411-416) — runs the native forward + input-VJP behind an autograd.Function."""
    case = cases.CASES["vjp_text_cfg"]
    inp = cases.make_inputs(case)
    model, _ = make_model(case)
    g = load_golden("vjp_text_cfg")
    z = tt(inp["x"]).requires_grad_(True)
    y = {"text_embed": tt(inp["enc_text"]), "text_scale": tt(inp["text_scale"])}
    with torch.enable_grad():
        out = model(z, tt(inp["t"]), y=y)
        assert out.requires_grad and ok("torch_autograd_through_the_native_denoiser.rel_l2.0", rel_l2(out.detach().cpu().numpy(), g["out"]), 2e-5)
        loss = (out * tt(inp["gout"])).sum()           # d loss / d out = gout
        gx, = torch.autograd.grad(loss, z)
    assert ok("torch_autograd_through_the_native_denoiser.rel_l2.1", rel_l2(gx.cpu().numpy(), g["gx"]), 5e-5), rel_l2(gx.cpu().numpy(), g["gx"])
    # a second forward invalidates the first graph's stash: the stale backward must fail loudly
    with torch.enable_grad():
        o1 = model(z, tt(inp["t"]), y=y)
        o2 = model(z, tt(inp["t"]), y=y)
        with pytest.raises(RuntimeError):
            torch.autograd.grad(o1.sum(), z)
        torch.autograd.grad(o2.sum(), z)
    with torch.no_grad():                                # no graph requested: plain forward
        assert not model(z, tt(inp["t"]), y=y).requires_grad


# ---- sampler arithmetic ----------------------------------------------------------------------------
@pytest.mark.parametrize("sampler,eta", [("ddpm", 0.0), ("ddim", 0.0), ("ddim", 0.7)])
@pytest.mark.parametrize("mode", ["plain", "impute", "recon", "recon_impute"])
def test_sampler_update_bit_exact(sampler, eta, mode):
    Engine, N = sub("engine").Engine, sub("_native")
    B, C, T = 3, 263, 52
    rng = np.random.default_rng(7)
    sch = do.Schedule(do.named_betas("cosine", 1000), do.space_timesteps(1000, "ddim100"))
    diff = make_diffusion("ddim100")
    e = Engine(n_layers=0, d_model=0, d_ff=0, n_heads=0, n_feats=C, max_frames=T, max_batch=B, device=DEV)
    e.set_schedule(diff.engine_tables())
    shape = (B, C, 1, T)
    x, hat, nz, inp, grad = (rng.standard_normal(shape).astype(np.float32) for _ in range(5))
    mask = rng.random(shape) < 0.3
    rw = (np.linspace(0.5, 1.5, 100).astype(np.float32) * np.float32(20.0))
    impute, recon = mode in ("impute", "recon_impute"), mode in ("recon", "recon_impute")
    kw = {}
    if impute or recon:
        kw = dict(inpaint_mask=tt(mask), inpaint_motion=tt(inp), imputate=impute,
                  stop_imputation_at=0, recon_guidance=recon, stop_recguidance_at=0,
                  recon_w=rw if recon else None)
    e.set_condition(batch=B, n_frames=T, **kw)
    sid = N.CMDI_SAMPLER_DDIM if sampler == "ddim" else N.CMDI_SAMPLER_DDPM
    for step in (99, 50, 1, 0):
        want, want_x0 = do.step_update(sch, step, x, hat, nz, sampler=sampler, eta=eta, mask=mask,
                                       inpaint=inp, impute=impute, recon=recon, grad=grad,
                                       recon_w=rw[step])
        xd, pred = tt(x).clone(), torch.empty(shape, device=DEV)
        e.sampler_update(xd, tt(hat), step, sampler=sid, eta=eta, noise=tt(nz), pred_xstart=pred,
                         recon_grad=tt(grad) if recon else None)
        assert np.array_equal(pred.cpu().numpy(), want_x0), (mode, sampler, step)
        assert np.array_equal(xd.cpu().numpy(), want), (mode, sampler, step)


# ---- chains through the public API -----------------------------------------------------------------
def run_chain(cases, name, precision):
    case = cases.CASES[name]
    inp = cases.make_inputs(case)
    check_fingerprint(cases, name, inp)
    model, _ = make_model(case, precision=precision)
    diffusion = make_diffusion(case["respacing"])
    B = inp["x_T"].shape[0]
    y = {"mask": tt(inp["len_mask"]), "lengths": tt(inp["lengths"])}
    if case["text"]:
        y.update(text_embed=tt(inp["enc_text"]), text_scale=tt(inp["text_scale"]))
    if case.get("edit"):
        y.update(inpainting_mask=tt(inp["inpaint_mask"]), inpainted_motion=tt(inp["x0"]),
                 imputate=case["imputate"], stop_imputation_at=case["stop_imputation_at"],
                 replacement_distribution=case.get("replacement", "conditional"), reconstruction_guidance=case["recon"],
                 reconstruction_weight=case["recon_weight"], gradient_schedule=case["grad_schedule"],
                 diffusion_steps=1000, stop_recguidance_at=case["stop_recguidance_at"])
    loop = diffusion.ddim_sample_loop if case["sampler"] == "ddim" else diffusion.p_sample_loop
    kw = dict(noise=tt(inp["x_T"]), clip_denoised=False, model_kwargs={"y": y},
              skip_timesteps=case.get("skip", 0),
              init_image=tt(inp["init_image"]) if "init_image" in inp else None)
    if case["sampler"] == "ddim":
        kw["eta"] = case["eta"]
    diffusion.injected_noise = tt(inp["noise"])
    final = loop(model, inp["x_T"].shape, **kw).cpu().numpy()
    dumps = loop(model, inp["x_T"].shape, dump_steps=list(cases.DUMP_STEPS), **kw)
    return final, [d.cpu().numpy() for d in dumps], load_golden(name)


@pytest.mark.parametrize("name", ["chain_uncond_ddpm", "chain_impute_only", "chain_edit_recon",
                                  "chain_ddim_eta0", "chain_ddim_eta05", "chain_skip_init", "chain_marginal_recon"])
@pytest.mark.parametrize("precision", PRECISIONS)
def test_chain_vs_reference(cases, name, precision):
    final, dumps, g = run_chain(cases, name, precision)
    assert ok("chain_vs_reference.rel_l2.0", rel_l2(final, g["final"]), 1e-4), rel_l2(final, g["final"])
    assert len(dumps) == g["pred_xstart"].shape[0]
    for k, d in enumerate(dumps):
        assert ok("chain_vs_reference.rel_l2.1", rel_l2(d, g["pred_xstart"][k]), 1e-4), (k, rel_l2(d, g["pred_xstart"][k]))


# ---- BASELINE shapes and chain lengths vs the reference (tests/golden/make_golden_big.py) ------------------
_BIG_NOISE = {}     # the injected noise stream of the most recent BIG case, on the device


def big_setup(cases, name, precision):
    case = cases.BIG_CASES[name]
    inp = cases.make_big_inputs(case)
    g = load_golden(name)
    assert np.array_equal(g["fingerprint"], cases.fingerprint(inp)), f"inputs of {name} drifted"
    model, _ = make_model(case, precision=precision)
    gd, rs = sub("diffusion.gaussian_diffusion"), sub("diffusion.respace")
    conf = gd.DiffusionConfig(betas=gd.get_named_beta_schedule("cosine", 1000))
    if case.get("mean_type") == "eps":
        conf.model_mean_type = gd.ModelMeanType.EPSILON
    diffusion = rs.SpacedDiffusion(rs.space_timesteps(1000, case.get("respacing") or [1000]), conf)
    B = case["B"]
    y = {"mask": tt(inp["len_mask"]), "lengths": tt(inp["lengths"])}
    if case["text"]:
        y.update(text_embed=tt(inp["enc_text"]), text_scale=tt(inp["text_scale"]))
    if case.get("edit"):
        y.update(inpainting_mask=tt(inp["inpaint_mask"]), inpainted_motion=tt(inp["x0"]), imputate=case["imputate"],
                 stop_imputation_at=case["stop_imputation_at"], replacement_distribution='conditional',
                 reconstruction_guidance=case["recon"], reconstruction_weight=case["recon_weight"],
                 gradient_schedule=case["grad_schedule"], diffusion_steps=1000,
                 stop_recguidance_at=case["stop_recguidance_at"])
    n = diffusion.num_timesteps - case.get("skip", 0)
    if name not in _BIG_NOISE:     # (big_c2_long: 1000 draws x 6.6 MB, built once for the three precision modes)
        noise = torch.empty((n,) + inp["draw0"].shape, dtype=torch.float32, device=DEV)
        for k in range(n):
            noise[k].copy_(torch.from_numpy(cases.big_draw(case, 1 + k)))
        _BIG_NOISE.clear()
        _BIG_NOISE[name] = noise
    diffusion.injected_noise = _BIG_NOISE[name]
    kw = dict(noise=tt(inp["draw0"]), clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=case.get("skip", 0),
              init_image=tt(inp["init_image"]) if "init_image" in inp else None)
    if case["sampler"] == "ddim":
        kw["eta"] = case["eta"]
    return case, inp, g, model, diffusion, kw


def stats_err(got, want):
    """Per-sample (sum, sum^2) of every sample vs the reference's float64 values, as ONE relative figure: the largest of
    |d sum| / sqrt(N * sum^2) (the scale of a sum of N terms) and |d sum^2| / sum^2 over the samples."""
    n = 263 * 196
    return float(max(np.max(np.abs(got[:, 0] - want[:, 0]) / np.sqrt(n * want[:, 1])),
                     np.max(np.abs(got[:, 1] - want[:, 1]) / want[:, 1])))


@pytest.mark.parametrize("name", ["big_c2", "big_c3"])
@pytest.mark.parametrize("precision", PRECISIONS)
def test_baseline_shape_chain_vs_reference(cases, name, precision):
    """BASELINE configs 2 and 3 at their own shape (B=32, T=196, CFG; c3 = imputation + reconstruction guidance),
    20 steps of the 1000-step chain, ragged lengths: p_sample_loop takes the engine's one-call path, which at this
    size cuts the batch into TWO independent pipelines (api_sampler.hip n_parts / part_forward / part_backward) — compared
    here with values of the real reference: six stored samples (three per pipeline) and (sum, sum^2) of all 32."""
    case, inp, g, model, diffusion, kw = big_setup(cases, name, precision)
    loop = diffusion.p_sample_loop
    final = loop(model, inp["draw0"].shape, **kw)
    eng = model.model._engine
    assert eng.pipeline_parts() == 2, "the two-pipeline schedule was not taken at the BASELINE shape"
    final = final.cpu().numpy()
    keep = list(case["keep"])
    err = rel_l2(final[keep], g["final"])
    per = [rel_l2(final[k], g["final"][i]) for i, k in enumerate(keep)]
    assert ok("baseline_shape_chain.rel_l2", err, 1e-4) and ok("baseline_shape_chain.per_sample", max(per), 2e-4), (err, per)
    assert ok("baseline_shape_chain.stats", stats_err(cases.sample_stats(final), g["stats"]), 2e-4)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_baseline_config2_full_1000_step_chain_vs_reference(cases, precision):
    """BASELINE config 2 END TO END (VERDICT r3 task 5b): B=32 x 196 frames, text CFG, ragged lengths, all 1000 ancestral steps
    on injected noise through the one-call path — i.e. the two-pipeline, two-stream schedule that bench.py times — against the
    real reference's CPU chain (tests/golden/make_golden_big.py big_c2_long, ~13 min of reference time): six stored samples
    (three per pipeline) and float64 (sum, sum^2) of all 32.  The per-step generator (cmdi_step per yield) gives the same
    chain bit for bit, and sample 0's x_t is compared with the reference every 100 steps on the way."""
    name = "big_c2_long"
    case, inp, g, model, diffusion, kw = big_setup(cases, name, precision)
    final = diffusion.p_sample_loop(model, inp["draw0"].shape, **kw)
    eng = model.model._engine
    assert eng.pipeline_parts() == 2, "the two-pipeline schedule was not taken at the BASELINE shape"
    final = final.cpu().numpy()
    keep = list(case["keep"])
    err = rel_l2(final[keep], g["final"])
    per = [rel_l2(final[k], g["final"][i]) for i, k in enumerate(keep)]
    assert np.isfinite(final).all()
    assert ok("baseline_config2_full_chain.rel_l2", err, 1e-4) and ok("baseline_config2_full_chain.per_sample", max(per), 2e-4), (err, per)
    assert ok("baseline_config2_full_chain.stats", stats_err(cases.sample_stats(final), g["stats"]), 2e-4)
    if precision == PRECISIONS[0]:
        at = {int(i): k for k, i in enumerate(g["dump_at"])}
        last = None
        for i, out in enumerate(diffusion.p_sample_loop_progressive(model, inp["draw0"].shape, **kw)):
            last = out["sample"]
            if i in at:
                assert ok("baseline_config2_full_chain.on_the_way", rel_l2(last[:1].cpu().numpy(), g["dumps"][at[i]]), 1e-4), i
        assert np.array_equal(last.cpu().numpy(), final)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_baseline_config5_rank_share_full_1000_step_chain_vs_reference(cases, precision):
    """BASELINE config 5, one rank's share END TO END (round 6): B=128 x 196 frames (1024 / 8 GPUs), text CFG, ragged lengths, all
    1000 ancestral steps on injected noise through the one-call two-pipeline path (M = 2 x 25,216 rows per evaluation) against
    the REAL reference's CPU chain (make_golden_big.py c5_rank_long, ~70 min of reference time): six stored samples, float64
    (sum, sum^2) of all 128, sample 0's x_t every 100 steps on the way.  With configs 2, 3 and 4 every single-GPU BASELINE
    configuration is pinned whole; the other ranks of config 5 run the same kernels on other samples (shard invariance is
    bitwise: test_full_size_batch_independence_and_sharding, test_sharded_p_sample_loop_equals_the_full_batch)."""
    from conftest import GOLDEN
    name = "c5_rank_long"
    if not (GOLDEN / f"{name}.npz").exists():
        pytest.skip(f"{name}.npz not generated")
    case, inp, g, model, diffusion, kw = big_setup(cases, name, precision)
    final = diffusion.p_sample_loop(model, inp["draw0"].shape, **kw)
    eng = model.model._engine
    assert eng.pipeline_parts() == C4C5_PARTS[case["B"]], eng.pipeline_parts()
    final = final.cpu().numpy()
    keep = list(case["keep"])
    err = rel_l2(final[keep], g["final"])
    per = [rel_l2(final[k], g["final"][i]) for i, k in enumerate(keep)]
    print(json_line({"case": name, "precision": precision, "rel_l2": err, "per_sample": per}))
    assert np.isfinite(final).all()
    assert ok("baseline_config5_rank_full_chain.rel_l2", err, 1e-4) and ok("baseline_config5_rank_full_chain.per_sample", max(per), 2e-4), (err, per)
    assert ok("baseline_config5_rank_full_chain.stats", stats_err(cases.sample_stats(final), g["stats"]), 2e-4)
    if precision == PRECISIONS[0]:
        at = {int(i): k for k, i in enumerate(g["dump_at"])}
        last = None
        for i, out in enumerate(diffusion.p_sample_loop_progressive(model, inp["draw0"].shape, **kw)):
            last = out["sample"]
            if i in at:
                assert ok("baseline_config5_rank_full_chain.on_the_way", rel_l2(last[:1].cpu().numpy(), g["dumps"][at[i]]), 1e-4), i
        assert np.array_equal(last.cpu().numpy(), final)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_baseline_config3_full_1000_step_guided_chain_vs_reference(cases, precision):
    """BASELINE config 3 END TO END (VERDICT r4 task 1a): B=32 x 196 frames, text CFG, ragged lengths, 'benchmark_sparse'
    keyframes, imputation + reconstruction guidance (weight 20) on ALL 1000 ancestral steps — the hand-written input-VJP, the
    power-of-two gradient scale and the split-row stash on every step — through the one-call two-pipeline path
    (part_forward / part_backward) against the REAL reference's CPU chain (make_golden_big.py big_c3_long; reference
    gaussian_diffusion.py:405-435 on every step, README.md:161): six stored samples (three per pipeline), float64
    (sum, sum^2) of all 32, and sample 0's x_t every 100 steps on the way (per-step generator, default precision)."""
    name = "big_c3_long"
    case, inp, g, model, diffusion, kw = big_setup(cases, name, precision)
    final = diffusion.p_sample_loop(model, inp["draw0"].shape, **kw)
    eng = model.model._engine
    assert eng.pipeline_parts() == 2, "the two-pipeline schedule was not taken at the BASELINE shape"
    final = final.cpu().numpy()
    keep = list(case["keep"])
    err = rel_l2(final[keep], g["final"])
    per = [rel_l2(final[k], g["final"][i]) for i, k in enumerate(keep)]
    print(json_line({"case": name, "precision": precision, "rel_l2": err, "per_sample": per}))
    assert np.isfinite(final).all()
    assert ok("baseline_config3_full_chain.rel_l2", err, 1e-4) and ok("baseline_config3_full_chain.per_sample", max(per), 2e-4), (err, per)
    assert ok("baseline_config3_full_chain.stats", stats_err(cases.sample_stats(final), g["stats"]), 2e-4)
    if precision == PRECISIONS[0]:
        at = {int(i): k for k, i in enumerate(g["dump_at"])}
        last = None
        for i, out in enumerate(diffusion.p_sample_loop_progressive(model, inp["draw0"].shape, **kw)):
            last = out["sample"]
            if i in at:
                assert ok("baseline_config3_full_chain.on_the_way", rel_l2(last[:1].cpu().numpy(), g["dumps"][at[i]]), 1e-4), i
        # the per-step generator runs the same kernels on the same two parts: the same chain (bitwise where the parts' power-of-two
        # gradient scales agree, which they do — one scale per part in both paths)
        assert ok("baseline_config3_full_chain.per_step_vs_one_call", rel_l2(last.cpu().numpy(), final), 1e-6), \
            rel_l2(last.cpu().numpy(), final)


def json_line(obj):
    import json
    return json.dumps(obj)


@pytest.mark.parametrize("name", ["c4_ddim", "c4_ddpm", "c5_rank"])
@pytest.mark.parametrize("precision", PRECISIONS)
def test_c4_c5_shape_chains_vs_reference(cases, name, precision):
    """CHAINS (not single evaluations) at BASELINE config 4's and 5's shapes vs the real reference: B=256 on the 'ddim100'
    respacing — its last 10 steps from a noised init_image, through ddim_sample_loop (eta 0) and through p_sample_loop (what
    the sample scripts call on that respacing) — and one rank's share of config 5 (B=128, 10 respaced steps).  CFG, ragged
    lengths.  Tile counts, split decisions, workspace offsets and the number of pipelines differ from B=32; the schedule
    that was actually taken is asserted as observed."""
    case, inp, g, model, diffusion, kw = big_setup(cases, name, precision)
    loop = diffusion.ddim_sample_loop if case["sampler"] == "ddim" else diffusion.p_sample_loop
    final = loop(model, inp["draw0"].shape, **kw)
    eng = model.model._engine
    assert eng.pipeline_parts() == C4C5_PARTS[case["B"]], eng.pipeline_parts()
    final = final.cpu().numpy()
    keep = list(case["keep"])
    err = rel_l2(final[keep], g["final"])
    per = [rel_l2(final[k], g["final"][i]) for i, k in enumerate(keep)]
    assert ok("c4_c5_shape_chains.rel_l2", err, 1e-4) and ok("c4_c5_shape_chains.per_sample", max(per), 2e-4), (err, per)
    assert ok("c4_c5_shape_chains.stats", stats_err(cases.sample_stats(final), g["stats"]), 2e-4)


C4C5_PARTS = {256: 2, 128: 2}     # engine pipelines at these batch sizes (api_sampler.hip n_parts), as observed


@pytest.mark.parametrize("name", ["c4_ddim_long", "c4_ddpm_long"])
@pytest.mark.parametrize("precision", PRECISIONS)
def test_baseline_config4_full_100_step_chain_vs_reference(cases, name, precision):
    """BASELINE config 4 END TO END (VERDICT r5 task 1a): B=256 x 196 frames on the 'ddim100' respacing, text CFG, ragged lengths,
    ALL 100 steps from pure noise — once through ddim_sample_loop (eta 0; reference gaussian_diffusion.py:1454-1587) and once
    through p_sample_loop on the respaced chain (what the sample scripts call) — through the one-call path at M = 2 x 50,432 rows
    per evaluation (two pipelines; the persistent / weight-stationary GEMM routes of that height), against the REAL reference's
    CPU chains (make_golden_big.py c4_ddim_long / c4_ddpm_long, ~25 min each): six stored samples, float64 (sum, sum^2) of all
    256, and sample 0's x_t every 10 steps on the way (per-step generator, default precision)."""
    case, inp, g, model, diffusion, kw = big_setup(cases, name, precision)
    loop = diffusion.ddim_sample_loop if case["sampler"] == "ddim" else diffusion.p_sample_loop
    final = loop(model, inp["draw0"].shape, **kw)
    eng = model.model._engine
    assert eng.pipeline_parts() == C4C5_PARTS[case["B"]], eng.pipeline_parts()
    final = final.cpu().numpy()
    keep = list(case["keep"])
    err = rel_l2(final[keep], g["final"])
    per = [rel_l2(final[k], g["final"][i]) for i, k in enumerate(keep)]
    print(json_line({"case": name, "precision": precision, "rel_l2": err, "per_sample": per}))
    assert np.isfinite(final).all()
    assert ok("baseline_config4_full_chain.rel_l2", err, 1e-4) and ok("baseline_config4_full_chain.per_sample", max(per), 2e-4), (err, per)
    assert ok("baseline_config4_full_chain.stats", stats_err(cases.sample_stats(final), g["stats"]), 2e-4)
    if precision == PRECISIONS[0]:
        prog = diffusion.ddim_sample_loop_progressive if case["sampler"] == "ddim" else diffusion.p_sample_loop_progressive
        at = {int(i): k for k, i in enumerate(g["dump_at"])}
        last = None
        for i, out in enumerate(prog(model, inp["draw0"].shape, **kw)):
            last = out["sample"]
            if i in at:
                assert ok("baseline_config4_full_chain.on_the_way", rel_l2(last[:1].cpu().numpy(), g["dumps"][at[i]]), 1e-4), i
        assert np.array_equal(last.cpu().numpy(), final)


@pytest.mark.parametrize("name", ["synthesize", "edit", "conditional_synthesis"])
@pytest.mark.parametrize("progress", [True, False])
def test_reference_callers_replayed_on_the_gpu(cases, name, progress):
    """tests/golden/caller_<name>.npz holds the EXACT shape / model_kwargs / keyword arguments the reference's own
    sample/<name>.py main() passes to diffusion.p_sample_loop (recorded while the script ran unchanged on the reference's
    modules; tests/test_reference_callers.py asserts that the script builds the same arguments on THIS package) and the real
    reference sampler's output for that call on the [10] respacing with injected noise.  Here the recorded call goes through
    the native p_sample_loop — nothing mocked.  Deviations, both forced by the offline box: CLIP is absent, so y['text']
    comes with its stand-in embedding as y['text_embed']; the noise stream is injected (torch CPU randn != device Philox).
    progress=True is what the scripts pass (per-step cmdi_step path); False takes the one-call cmdi_sample_loop path."""
    import json
    import sys
    helpers = str(__import__("pathlib").Path(__file__).resolve().parent / "helpers")
    if helpers not in sys.path:
        sys.path.insert(0, helpers)
    import caller_setup
    from run_reference_caller import replay_draw
    case = cases.CALLER_CASES[name]
    g = load_golden(f"caller_{name}")
    meta = json.loads(str(g["meta"]))
    mu, rs, gd = sub("utils.model_util"), sub("diffusion.respace"), sub("diffusion.gaussian_diffusion")
    # what the script does: create_model_and_diffusion(args from the checkpoint's args.json) + load_saved_model
    args = SimpleNamespace(**dict(case["model_args"], abs_3d=True, latent_dim=512))
    model, diffusion = mu.create_model_and_diffusion(args, None)
    mu.load_model_wo_clip(model, weights.to_torch(caller_setup.caller_state_dict(case["model_args"], case["weight_seed"])))
    model = sub("model.cfg_sampler").ClassifierFreeSampleModel(model)       # guidance_param != 1 in all three scripts
    model.to(DEV).eval()
    assert diffusion.num_timesteps == 1000
    conf = diffusion.conf
    conf.betas = gd.get_named_beta_schedule("cosine", 1000)
    short = rs.SpacedDiffusion(rs.space_timesteps(1000, [10]), conf)
    shape = tuple(meta["shape"])
    short.injected_noise = torch.from_numpy(np.stack([replay_draw(shape, k) for k in range(11)])).to(DEV)
    y = dict(meta["y"])
    y.update({k[2:]: tt(g[k]) for k in g.files if k.startswith("y.")})
    y["text_embed"] = tt(g["text_embed"])
    model_kwargs = {"y": y}
    model_kwargs.update(meta["mk"])
    model_kwargs.update({k[3:]: tt(g[k]) for k in g.files if k.startswith("mk.")})
    kw = dict(meta["kw"])
    kw.update({k[3:]: tt(g[k]) for k in g.files if k.startswith("kw.")})
    assert kw["progress"] is True and kw["noise"] is None and kw["clip_denoised"] is False
    kw["progress"] = progress
    out = short.p_sample_loop(model, shape, model_kwargs=model_kwargs, **kw).cpu().numpy()
    err = rel_l2(out, g["ref_sample"])
    assert np.isfinite(out).all() and ok(f"reference_callers_replayed.{name}", err, 2e-4 if name == "conditional_synthesis" else 1e-4), err


@pytest.mark.parametrize("precision", PRECISIONS)
def test_forced_two_pipelines_on_small_golden(cases, precision, monkeypatch):
    """CMDI_GROUPS=2 forces the part_forward / part_backward schedule on a small reference chain (B=2: one sample per
    pipeline) with imputation + reconstruction guidance and ragged lengths."""
    monkeypatch.setenv("CMDI_GROUPS", "2")
    name = "chain_edit_recon"
    case = cases.CASES[name]
    inp = cases.make_inputs(case)
    model, _ = make_model(case, precision=precision)
    diffusion = make_diffusion(case["respacing"])
    y = {"mask": tt(inp["len_mask"]), "lengths": tt(inp["lengths"]), "text_embed": tt(inp["enc_text"]),
         "text_scale": tt(inp["text_scale"]), "inpainting_mask": tt(inp["inpaint_mask"]), "inpainted_motion": tt(inp["x0"]),
         "imputate": True, "stop_imputation_at": case["stop_imputation_at"], "replacement_distribution": "conditional",
         "reconstruction_guidance": True, "reconstruction_weight": case["recon_weight"], "gradient_schedule": None,
         "diffusion_steps": 1000, "stop_recguidance_at": case["stop_recguidance_at"]}
    diffusion.injected_noise = tt(inp["noise"])
    final = diffusion.p_sample_loop(model, inp["x_T"].shape, noise=tt(inp["x_T"]), clip_denoised=False,
                                    model_kwargs={"y": y})
    assert model.model._engine.pipeline_parts() == 2
    g = load_golden(name)
    assert ok("forced_two_pipelines_on_small_golden.rel_l2.0", rel_l2(final.cpu().numpy(), g["final"]), 1e-4), rel_l2(final.cpu().numpy(), g["final"])


# measured drift (gpurun_out/drift_*.json): bound = 4x the larger of the two engines' measured value
LONG_TOL = {"long_ddpm": 1e-4, "long_ddim100": 1e-4, "long_c3": 1e-4}


@pytest.mark.parametrize("name", ["long_ddpm", "long_ddim100", "long_c3"])
@pytest.mark.parametrize("precision", PRECISIONS)
def test_long_chain_drift_vs_reference(cases, name, precision):
    """The FULL 1000-step ancestral chain (and ddim_sample_loop on 'ddim100') on shared noise: x_t every 100 (10) steps
    against the reference's fp32 chain and against the same chain run by the reference in float64 (ground truth).
    SURVEY 8c: 'measure, do not assume' — the per-checkpoint errors are written to gpurun_out/ for DESIGN.md."""
    import json
    import os
    case, inp, g, model, diffusion, kw = big_setup(cases, name, precision)
    prog = diffusion.ddim_sample_loop_progressive if case["sampler"] == "ddim" else diffusion.p_sample_loop_progressive
    at = {int(i): k for k, i in enumerate(g["dump_at"])}
    rows, final = [], None
    for i, out in enumerate(prog(model, inp["draw0"].shape, **kw)):
        final = out["sample"]
        if i in at:
            x = out["sample"][:1].cpu().numpy()
            k = at[i]
            rows.append({"step": i + 1, "vs_ref_f32": rel_l2(x, g["dumps"][k]), "vs_ref_f64": rel_l2(x, g["dumps_f64"][k]),
                         "ref_f32_vs_f64": rel_l2(g["dumps"][k], g["dumps_f64"][k]),
                         "max_abs_vs_f64": max_abs(x, g["dumps_f64"][k])})
    final = final.cpu().numpy()
    summary = {"case": name, "precision": precision, "final_vs_ref_f32": rel_l2(final, g["final"]),
               "final_vs_ref_f64": rel_l2(final, g["final_f64"]),
               "ref_f32_vs_f64": rel_l2(g["final"], g["final_f64"]), "rows": rows}
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/drift_{name}_{precision}.json", "w") as fh:
        json.dump(summary, fh, indent=1)
    print(json.dumps(summary))
    assert np.isfinite(final).all()
    assert ok(f"long_chain_drift.{name}.vs_f64", summary["final_vs_ref_f64"], LONG_TOL[name]), summary
    assert ok(f"long_chain_drift.{name}.vs_f32", summary["final_vs_ref_f32"], LONG_TOL[name]), summary
    # the one-call loop (what p_sample_loop runs) gives the same chain as the per-step generator
    whole = (diffusion.ddim_sample_loop if case["sampler"] == "ddim" else diffusion.p_sample_loop)(
        model, inp["draw0"].shape, **kw).cpu().numpy()
    assert np.array_equal(whole, final)


@pytest.mark.parametrize("name", ["eps_ddpm", "eps_ddim"])
@pytest.mark.parametrize("precision", PRECISIONS)
def test_epsilon_model_chain_vs_reference(cases, name, precision):
    """ModelMeanType.EPSILON through the native loop (sampler.hip: x0 = sra x - srm1a eps) vs the real reference."""
    case, inp, g, model, diffusion, kw = big_setup(cases, name, precision)
    prog = diffusion.ddim_sample_loop_progressive if case["sampler"] == "ddim" else diffusion.p_sample_loop_progressive
    at = {int(i): k for k, i in enumerate(g["dump_at"])}
    final = None
    for i, out in enumerate(prog(model, inp["draw0"].shape, **kw)):
        final = out["sample"]
        if i in at:
            assert ok("epsilon_model_chain_vs_reference.rel_l2.0", rel_l2(out["sample"][:1].cpu().numpy(), g["dumps"][at[i]]), 1e-4)
    assert ok("epsilon_model_chain_vs_reference.rel_l2.1", rel_l2(final.cpu().numpy(), g["final"]), 1e-4), rel_l2(final.cpu().numpy(), g["final"])
    # imputation / reconstruction guidance on an eps-model: refused like the reference (:407,430)
    y = dict(kw["model_kwargs"]["y"], imputate=True, stop_imputation_at=0, inpainting_mask=torch.zeros(inp["draw0"].shape, dtype=torch.bool, device=DEV),
             inpainted_motion=tt(inp["draw0"]), replacement_distribution='conditional')
    kw.pop("eta", None)
    with pytest.raises(AssertionError, match="X_start"):
        diffusion.p_sample_loop(model, inp["draw0"].shape, **dict(kw, model_kwargs={"y": y}))
    # clip_denoised on an eps-model is NotImplemented outside abs_3d trajectory models (:500-505)
    with pytest.raises(NotImplementedError):
        diffusion.p_sample_loop(model, inp["draw0"].shape, **dict(kw, clip_denoised=True))


def test_epsilon_clip_bit_exact():
    """process_xstart's clamp for abs_3d trajectory eps-models (clip_x0 of cmdi_schedule) vs the oracle, bit for bit."""
    Engine, N = sub("engine").Engine, sub("_native")
    gd, rs = sub("diffusion.gaussian_diffusion"), sub("diffusion.respace")
    B, C, T = 2, 263, 40
    rng = np.random.default_rng(11)
    conf = gd.DiffusionConfig(betas=gd.get_named_beta_schedule("cosine", 1000), model_mean_type=gd.ModelMeanType.EPSILON,
                              abs_3d=True, traj_only=True, clip_range=6.0)
    diff = rs.SpacedDiffusion(rs.space_timesteps(1000, "ddim100"), conf)
    sch = do.Schedule(do.named_betas("cosine", 1000), do.space_timesteps(1000, "ddim100"))
    e = Engine(n_layers=0, d_model=0, d_ff=0, n_heads=0, n_feats=C, max_frames=T, max_batch=B, device=DEV)
    e.set_schedule(diff.engine_tables(diff._clip_x0(True)))
    e.set_condition(batch=B, n_frames=T)
    shape = (B, C, 1, T)
    x, eps, nz = (rng.standard_normal(shape).astype(np.float32) for _ in range(3))
    for step in (90, 40, 0):
        want, want_x0 = do.step_update(sch, step, x, eps, nz, mean_eps=True, clip=6.0)
        xd, pred = tt(x).clone(), torch.empty(shape, device=DEV)
        e.sampler_update(xd, tt(eps), step, noise=tt(nz), pred_xstart=pred)
        assert np.array_equal(pred.cpu().numpy(), want_x0) and np.array_equal(xd.cpu().numpy(), want), step
    assert float(np.abs(want_x0).max()) <= 6.0


@pytest.mark.parametrize("precision", PRECISIONS)
def test_forward_b256_vs_reference(cases, precision):
    """One CFG evaluation at BASELINE config 4's batch (B=256: GEMM height M = 2 * 256 * 197 = 100,864 rows)."""
    case = cases.BIG_CASES["fwd_b256"]
    inp = cases.make_big_inputs(case)
    g = load_golden("fwd_b256")
    assert np.array_equal(g["fingerprint"], cases.fingerprint(inp))
    model, _ = make_model(case, cfg=True, precision=precision)
    out = model(tt(inp["x"]), tt(inp["t"]), y={"text_embed": tt(inp["enc_text"]), "text_scale": tt(inp["text_scale"])})
    out = out.cpu().numpy()
    keep = list(case["keep"])
    assert ok("forward_b256_vs_reference.max_abs.0", max_abs(out[keep], g["out_cfg"]), 2e-4) and ok("forward_b256_vs_reference.rel_l2.0", rel_l2(out[keep], g["out_cfg"]), 2e-5), \
        (max_abs(out[keep], g["out_cfg"]), rel_l2(out[keep], g["out_cfg"]))
    assert ok("forward_b256_vs_reference.stats", stats_err(cases.sample_stats(out), g["stats"]), 5e-5)


def test_sharded_p_sample_loop_equals_the_full_batch():
    """the documented multi-GPU flow (same torch seed on every rank + dist_util.shard_call) must reproduce
    the single-device batch — y['first_sample'] keys x_T and every step's noise by the GLOBAL sample index."""
    du = sub("utils.dist_util")
    case = dict(text=True, weight_seed=23, cfg=True)
    model, _ = make_model(case, layers=2)
    diffusion = make_diffusion([5])
    B, T = 6, 40
    rng = np.random.default_rng(3)
    y = {"mask": torch.ones(B, 1, 1, T, dtype=torch.bool, device=DEV), "lengths": torch.full((B,), T),
         "text_embed": tt(rng.standard_normal((B, 512)).astype(np.float32)), "text_scale": torch.full((B,), 2.5, device=DEV)}
    shape = (B, 263, 1, T)

    def run(lo, hi):
        torch.manual_seed(1234)                       # utils.fixseed(seed) on every rank
        kw = du.shard_batch({"y": y}, 0, 1, B)
        kw = {"y": {k: (v[lo:hi] if torch.is_tensor(v) and v.shape[0] == B else v) for k, v in kw["y"].items()}}
        kw["y"]["first_sample"] = lo
        return diffusion.p_sample_loop(model, (hi - lo,) + shape[1:], clip_denoised=False, model_kwargs=kw)

    full = run(0, B)
    parts = torch.cat([run(0, 2), run(2, 6)])
    assert torch.isfinite(full).all() and torch.equal(full, parts)
    assert not torch.equal(full[:2], full[2:4])       # different samples do get different noise
    # the progressive (per-step) path keys the noise the same way
    torch.manual_seed(1234)
    *_, last = diffusion.p_sample_loop_progressive(model, (4,) + shape[1:], clip_denoised=False,
                                                   model_kwargs={"y": dict({k: (v[2:6] if torch.is_tensor(v) and v.shape[0] == B else v)
                                                                            for k, v in y.items()}, first_sample=2)})
    assert torch.equal(last["sample"], full[2:6])


def test_out_of_range_timestep_is_reported():
    """The reference raises IndexError at pe[timesteps]; the engine clamps the row and raises the status bit."""
    case = dict(text=False, weight_seed=5)
    model, _ = make_model(case, layers=1)
    x = torch.randn(2, 263, 1, 20, device=DEV)
    model(x, torch.tensor([5, 4999], device=DEV), y={})
    model._engine.check_range()
    model(x, torch.tensor([5, 5000], device=DEV), y={})
    with pytest.raises(IndexError):
        model._engine.check_range()
    with pytest.raises(IndexError):
        model(x, torch.tensor([5, 5000]), y={})      # host tensor: checked before the launch
    model(x, torch.tensor([5, 5000], device=DEV), y={})
    with pytest.raises(IndexError):
        model.check_range()                          # the public form of the same check


def test_status_flag_is_per_call(cases):
    """(1) a stale flag left by an unchecked forward call must not fire at the end of the next, unrelated chain —
    the loops clear it before their first step; (2) the public single-step samplers check it themselves (a 10-step schedule
    keeps timesteps in range, so the stale flag is the only event: p_sample / ddim_sample must NOT raise)."""
    case = cases.CASES["chain_uncond_ddpm"]
    inp = cases.make_inputs(case)
    model, _ = make_model(case, layers=1)
    diffusion = make_diffusion(case["respacing"])
    x = tt(inp["x_T"])
    y = {"mask": tt(inp["len_mask"]), "lengths": tt(inp["lengths"])}
    model(x, torch.tensor([5, 5000], device=DEV), y={})          # raises the timestep bit, nobody checks
    out = diffusion.p_sample_loop(model, x.shape, noise=x, clip_denoised=False, model_kwargs={"y": y})
    assert torch.isfinite(out).all()
    model(x, torch.tensor([5, 5000], device=DEV), y={})
    step = diffusion.p_sample(model, x, torch.full((2,), 3, device=DEV), clip_denoised=False, model_kwargs={"y": y})
    assert torch.isfinite(step["sample"]).all()
    model(x, torch.tensor([5, 5000], device=DEV), y={})
    step = diffusion.ddim_sample(model, x, torch.full((2,), 3, device=DEV), clip_denoised=False, model_kwargs={"y": y})
    assert torch.isfinite(step["sample"]).all()
    model._engine.check_range()                                   # nothing pending after the checked calls


def test_range_probe_switches_before_a_long_chain(cases, monkeypatch):
    """VERDICT r2 task 7: weights whose activations leave the f16 range are detected by the two-evaluation probe at the START
    of a long chain — the chain itself then runs once, on bf16x6 — instead of at the end of 100 wasted f16x3 steps."""
    gd = sub("diffusion.gaussian_diffusion")
    case = dict(text=False, weight_seed=6)
    model, sd = make_model(case, layers=2)
    key = "seqTransEncoder.layers.0.linear1.weight"      # FFN pre-activations of ~1e5: beyond f16 (the scale of
    sd = dict(sd)                                         # test_real_scale_activations_and_range_fallback's second case)
    sd[key] = sd[key] * np.float32(4.0e4)
    sub("utils.model_util").load_model_wo_clip(model, weights.to_torch(sd))
    model.to(DEV).eval()
    diffusion = make_diffusion("ddim100")
    x = torch.randn(2, 263, 1, 40, device=DEV)
    y = {"mask": torch.ones(2, 1, 1, 40, dtype=torch.bool, device=DEV), "lengths": torch.full((2,), 40, device=DEV)}
    calls = []
    real = sub("engine").Engine.sample_loop
    monkeypatch.setattr(sub("engine").Engine, "sample_loop",
                        lambda self, *a, **k: (calls.append(self.precision), real(self, *a, **k))[1])
    out = diffusion.p_sample_loop(model, x.shape, noise=x, clip_denoised=False, model_kwargs={"y": y})
    assert calls == ["bf16x6"], calls                  # the f16x3 engine never ran the chain
    assert model._engine.precision == "bf16x6" and torch.isfinite(out).all()


def test_single_step_range_fallback(cases):
    """p_sample on a model whose activations leave the f16 range falls back to bf16x6 like the loops do (instead of
    returning overflowed values); with a pinned f16x3 precision it raises."""
    N_ = sub("_native")
    case = cases.CASES["chain_uncond_ddpm"]
    inp = cases.make_inputs(case)
    diffusion = make_diffusion(case["respacing"])
    y = {"mask": tt(inp["len_mask"]), "lengths": tt(inp["lengths"])}
    x = tt(inp["x_T"]) * 3.0e4            # token values of 1e5 .. 1e6 after the input projection: beyond f16
    t3 = torch.full((2,), 3, device=DEV)
    model, _ = make_model(case, layers=1, precision="f16x3")
    with pytest.raises(N_.RangeError):
        diffusion.p_sample(model, x, t3, clip_denoised=False, model_kwargs={"y": y})
    model, _ = make_model(case, layers=1)            # default precision: automatic fallback
    out = diffusion.p_sample(model, x, t3, clip_denoised=False, model_kwargs={"y": y})
    assert model._engine.precision == "bf16x6" and torch.isfinite(out["sample"]).all()


# ---- cond_fn guidance (p_sample_with_grad / ddim_sample_with_grad) -------------------------------
@pytest.mark.parametrize("name", ["chain_condfn_ddpm", "chain_condfn_ddim"])
@pytest.mark.parametrize("precision", PRECISIONS)
def test_cond_fn_chain_vs_reference(cases, name, precision):
    """p_sample_loop(cond_fn=..., cond_fn_with_grad=True) and ddim_sample_loop(cond_fn=...) with a quadratic key-location
    style cond_fn whose gradient is taken by torch.autograd THROUGH the native CFG denoiser (cmdi_mdm_vjp), vs the real
    reference's chain on the same noise."""
    case = cases.CASES[name]
    inp = cases.make_inputs(case)
    check_fingerprint(cases, name, inp)
    model, _ = make_model(case, precision=precision)
    diffusion = make_diffusion(case["respacing"])
    y = {"mask": tt(inp["len_mask"]), "lengths": tt(inp["lengths"]), "text_embed": tt(inp["enc_text"]),
         "text_scale": tt(inp["text_scale"])}
    cond_fn = cases.make_cond_fn(tt(inp["x0"]), tt(inp["inpaint_mask"] & inp["len_mask"]).float(), case["cond_weight"])
    loop = diffusion.ddim_sample_loop if case["sampler"] == "ddim" else diffusion.p_sample_loop
    kw = dict(noise=tt(inp["x_T"]), clip_denoised=False, model_kwargs={"y": y}, cond_fn=cond_fn, cond_fn_with_grad=True)
    if case["sampler"] == "ddim":
        kw["eta"] = case["eta"]
    diffusion.injected_noise = tt(inp["noise"])
    final = loop(model, inp["x_T"].shape, **kw).cpu().numpy()
    dumps = loop(model, inp["x_T"].shape, dump_steps=list(cases.DUMP_STEPS), **kw)
    g = load_golden(name)
    assert ok("cond_fn_chain_vs_reference.rel_l2.0", rel_l2(final, g["final"]), 1e-4), rel_l2(final, g["final"])
    for k, d in enumerate(dumps):
        assert ok("cond_fn_chain_vs_reference.rel_l2.1", rel_l2(d.cpu().numpy(), g["pred_xstart"][k]), 1e-4), (k, rel_l2(d.cpu().numpy(), g["pred_xstart"][k]))
    # is the chain's distance from the reference (1.3e-5 ... 2.0e-5 at step 1) rounding or a defect?  The
    # chain's own sensitivity answers it: two runs of THIS engine whose x_T differ by one fp32 ulp (6e-8 relative) end this far apart
    moved = loop(model, inp["x_T"].shape, **dict(kw, noise=torch.nextafter(kw["noise"], torch.full_like(kw["noise"], float("inf"))))).cpu().numpy()
    sens = rel_l2(moved, final)
    print(json_line({"case": name, "precision": precision, "vs_reference": rel_l2(final, g["final"]), "one_ulp_of_x_T": sens}))
    assert rel_l2(final, g["final"]) <= 25.0 * sens + 2e-6, (rel_l2(final, g["final"]), sens)
    # the guidance is not a no-op: the unguided chain on the same noise ends elsewhere
    plain = loop(model, inp["x_T"].shape, **{k: v for k, v in kw.items() if k not in ("cond_fn", "cond_fn_with_grad")})
    assert rel_l2(plain.cpu().numpy(), g["final"]) > 1e-2
    if case["sampler"] == "ddpm":     # p_sample asserts cond_fn is None (:685): without cond_fn_with_grad the loop refuses
        with pytest.raises(AssertionError):
            loop(model, inp["x_T"].shape, **dict(kw, cond_fn_with_grad=False))


# ---- full-size properties (BASELINE config 2 shape: B=32, T=196, CFG) --------------------------------
@pytest.mark.parametrize("precision", PRECISIONS)
def test_full_size_batch_independence_and_sharding(precision):
    case = dict(text=True, weight_seed=21, cfg=True)
    model, _ = make_model(case, precision=precision)
    diffusion = make_diffusion([4])  # 4 steps of the 1000-step chain: t = 0, 333, 666, 999
    B, T = 32, 196
    rng = np.random.default_rng(5)
    emb = tt(rng.standard_normal((B, 512)).astype(np.float32))
    scale = torch.full((B,), 2.5, device=DEV)
    shape = (B, 263, 1, T)

    def sample(lo, hi, seed):
        torch.manual_seed(seed)
        y = {"mask": torch.ones(hi - lo, 1, 1, T, dtype=torch.bool, device=DEV),
             "lengths": torch.full((hi - lo,), T), "text_embed": emb[lo:hi], "text_scale": scale[lo:hi]}
        eng = model.model.engine(torch.device(DEV), max_batch=B, max_frames=T)
        eng.set_schedule(diffusion.engine_tables(), key="t")
        eng.set_condition(batch=hi - lo, n_frames=T, cfg=True, enc_text=y["text_embed"],
                          text_scale=y["text_scale"])
        x = eng.randn((hi - lo, 263, 1, T), seed=99, first_sample=lo)
        eng.sample_loop(x, 3, 0, seed=99, first_sample=lo)
        return x

    full = sample(0, B, 0)
    assert torch.isfinite(full).all()
    again = sample(0, B, 1)
    assert torch.equal(full, again), "sampling is not deterministic for a fixed engine seed"
    halves = torch.cat([sample(0, 16, 2), sample(16, 32, 3)])
    assert torch.equal(full, halves), "sharded batch differs from the single-device batch"
    single = sample(5, 6, 4)
    assert torch.equal(full[5:6], single), "a sample depends on its batch neighbours"


# ---- bench.py ------------------------------------------------------------------------------------------
def test_gelu_epilogue_accuracy():
    """The GELU of the linear1 epilogue (common.hpp: branch-free erf) against float64 over [-8, 8]: fp32-class — within
    2.5e-7 of the exact value (times |x| beyond 1), i.e. what a correctly rounded erf followed by two fp32 roundings gives
    to within a factor of ~2; identity weights make the fp32-MFMA product exact, so the epilogue is what is measured."""
    from scipy import special
    E = sub("engine")
    x = np.concatenate([np.linspace(-8.0, 8.0, 32 * 8192 - 32 * 1024), np.random.default_rng(5).standard_normal(32 * 1024) * 2])
    x = x.astype(np.float32).reshape(-1, 32)
    got = E.gemm_nt(tt(x), torch.eye(32, device=DEV), torch.zeros(32, device=DEV), epi=1).cpu().numpy().astype(np.float64)
    x64 = x.astype(np.float64)
    want = 0.5 * x64 * (1.0 + special.erf(x64 / np.sqrt(2.0)))
    err = np.abs(got - want) / np.maximum(1.0, np.abs(x64))
    assert ok("gelu_epilogue_accuracy.max_err", err.max(), 2.5e-7), (err.max(), x64.flat[err.argmax()])


def test_bench_c5_runs_on_one_gpu():
    """BASELINE configs[4] (batch 1024 sharded over the node, strong scaling): the driver's 8-GPU run must not be the
    first execution of this path — one GPU takes the whole global batch (N = 1 point of the strong-scaling curve)."""
    import json
    import subprocess
    import sys
    from conftest import REPO
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "1", "--config", "c5", "--steps", "4",
                        "--warmup", "1", "--no-cpu", "--no-pmc", "--no-f32"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["scaling"] == "strong" and line["n_gpus"] == 1 and line["steps"] == 4
    assert line["config"]["global_batch"] == 1024 and line["config"]["batch_per_gpu"] == 1024
    assert line["value"] > 0 and abs(line["value"] * line["ms_per_step"] - 1000.0) < 1e-6 * 1000
    assert line["roofline"]["frac"] > 0.05 and line["pipeline_parts"] == 2


def test_bench_under_torchrun_one_process_uses_rccl():
    """The driver's multi-GPU launch form (python -m torch.distributed.run ... bench.py --gpus N) with ONE process: the
    process group (backend nccl = RCCL), the barriers, the max-over-ranks all-reduce and the all-gather of the generated
    batch all execute on the single GPU, so the 8-GPU run is not the first time that code runs."""
    import json
    import socket
    import subprocess
    import sys
    from conftest import REPO
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(REPO / "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--no-cpu",
           "--no-pmc", "--no-f32"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(REPO))
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    line = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["n_ranks_seen"] == 1 and line["steps"] == 4
    assert line["rccl_version"], "the process group was not initialised"
    assert line["allgather_ms"] >= 0 and line["value"] > 0


# ---- hipGraph replay -----------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("sampler", ["ddpm", "ddim"])
def test_graph_replay_is_bitwise_identical(precision, sampler):
    """cmdi_sample_loop with hipGraph replay (device step tables + cursor) == the eager loop, bit for bit;
    the chain crosses both graph kinds (reconstruction guidance stops at step 4) and the imputation gate."""
    N = sub("_native")
    case = dict(text=True, weight_seed=11, cfg=True)
    model, _ = make_model(case, layers=2, precision=precision)
    diffusion = make_diffusion([12])
    B, T = 3, 52
    rng = np.random.default_rng(9)
    shape = (B, 263, 1, T)
    emb = tt(rng.standard_normal((B, 512)).astype(np.float32))
    x0 = tt(rng.standard_normal(shape).astype(np.float32))
    mask = tt(rng.random(shape) < 0.3)
    eng = model.model.engine(torch.device(DEV), max_batch=B, max_frames=T, want_grad=True)
    eng.set_schedule(diffusion.engine_tables(), key="g")
    sid = N.CMDI_SAMPLER_DDIM if sampler == "ddim" else N.CMDI_SAMPLER_DDPM

    def run(graph):
        eng.set_graph(graph)
        eng.set_condition(batch=B, n_frames=T, cfg=True, enc_text=emb, text_scale=torch.full((B,), 2.5, device=DEV),
                          inpaint_mask=mask, inpaint_motion=x0, imputate=True, stop_imputation_at=1,
                          recon_guidance=True, stop_recguidance_at=4,
                          recon_w=np.full((12,), 20.0, dtype=np.float32))
        x = eng.randn(shape, seed=5)
        eng.sample_loop(x, 11, 0, sampler=sid, eta=0.3, seed=77, first_sample=2)
        eng.check_range()
        return x.clone()

    eager = run(False)
    replay = run(True)
    again = run(True)
    eng.set_graph(False)
    assert torch.isfinite(eager).all()
    assert torch.equal(eager, replay) and torch.equal(eager, again)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("groups", ["2", "3"])
def test_graph_replay_of_the_pipelined_schedule_is_bitwise_identical(precision, groups, monkeypatch):
    """hipGraph replay of the PIPELINED schedule (VERDICT r3 task 6): the batch is cut into independent parts, each part's
    denoising step is captured on its own stream (per-part device cursor) and replayed into it; the chain crosses both
    graph kinds (reconstruction guidance stops at step 4) and the imputation gate, DDPM and DDIM.  Bitwise the eager
    pipelined chain, which is bitwise the single-stream chain."""
    N = sub("_native")
    monkeypatch.setenv("CMDI_GROUPS", groups)
    case = dict(text=True, weight_seed=11, cfg=True)
    model, _ = make_model(case, layers=2, precision=precision)
    diffusion = make_diffusion([12])
    B, T = 5, 52
    rng = np.random.default_rng(10)
    shape = (B, 263, 1, T)
    emb = tt(rng.standard_normal((B, 512)).astype(np.float32))
    x0 = tt(rng.standard_normal(shape).astype(np.float32))
    mask = tt(rng.random(shape) < 0.3)
    eng = model.model.engine(torch.device(DEV), max_batch=B, max_frames=T, want_grad=True)
    eng.set_schedule(diffusion.engine_tables(), key="gp")

    def run(graph, sid):
        eng.set_graph(graph)
        eng.set_condition(batch=B, n_frames=T, cfg=True, enc_text=emb, text_scale=torch.full((B,), 2.5, device=DEV),
                          inpaint_mask=mask, inpaint_motion=x0, imputate=True, stop_imputation_at=1,
                          recon_guidance=True, stop_recguidance_at=4,
                          recon_w=np.full((12,), 20.0, dtype=np.float32))
        x = eng.randn(shape, seed=5)
        eng.sample_loop(x, 11, 0, sampler=sid, eta=0.3, seed=78, first_sample=3)
        eng.check_range()
        assert eng.pipeline_parts() == int(groups)
        return x.clone()

    for sid in (N.CMDI_SAMPLER_DDPM, N.CMDI_SAMPLER_DDIM):
        eager = run(False, sid)
        replay = run(True, sid)
        again = run(True, sid)
        assert torch.isfinite(eager).all()
        assert torch.equal(eager, replay) and torch.equal(eager, again)
    eng.set_graph(False)


@pytest.mark.parametrize("groups", ["1", "2"])
def test_graph_cache_hit_replays_the_graphs_of_an_earlier_call(groups, monkeypatch):
    """the part-graph cache (api_sampler.hip: key = stream, seed, first_sample, x pointer, parts) was never HIT in
    a test — every run above calls set_condition first, which drops the graphs.  Here a chain is cut into two
    cmdi_sample_loop calls on the SAME x tensor with no set_condition in between: the second call replays the graphs the
    first one captured (cursors and per-step tables re-uploaded for its step range) and must continue the chain bit for bit
    as the eager single call does — across the guidance gate (stop_recguidance_at = 4) and the imputation gate."""
    N = sub("_native")
    monkeypatch.setenv("CMDI_GROUPS", groups)
    case = dict(text=True, weight_seed=11, cfg=True)
    model, _ = make_model(case, layers=2, precision="f16x3")
    diffusion = make_diffusion([12])
    B, T = 5, 52
    rng = np.random.default_rng(11)
    shape = (B, 263, 1, T)
    emb = tt(rng.standard_normal((B, 512)).astype(np.float32))
    x0 = tt(rng.standard_normal(shape).astype(np.float32))
    mask = tt(rng.random(shape) < 0.3)
    eng = model.model.engine(torch.device(DEV), max_batch=B, max_frames=T, want_grad=True)
    eng.set_schedule(diffusion.engine_tables(), key="gc")
    cond = lambda: eng.set_condition(batch=B, n_frames=T, cfg=True, enc_text=emb, text_scale=torch.full((B,), 2.5, device=DEV),
                                     inpaint_mask=mask, inpaint_motion=x0, imputate=True, stop_imputation_at=1, recon_guidance=True,
                                     stop_recguidance_at=4, recon_w=np.full((12,), 20.0, dtype=np.float32))
    for sid in (N.CMDI_SAMPLER_DDPM, N.CMDI_SAMPLER_DDIM):
        eng.set_graph(False)
        cond()
        want = eng.randn(shape, seed=6)
        eng.sample_loop(want, 11, 0, sampler=sid, eta=0.3, seed=79, first_sample=2)
        eng.set_graph(True)
        cond()                                   # drops every graph: the first call below captures afresh
        x = eng.randn(shape, seed=6)
        eng.sample_loop(x, 11, 7, sampler=sid, eta=0.3, seed=79, first_sample=2)     # captures (kind: with guidance)
        eng.sample_loop(x, 6, 3, sampler=sid, eta=0.3, seed=79, first_sample=2)      # cache hit + the second kind from step 3
        eng.sample_loop(x, 2, 0, sampler=sid, eta=0.3, seed=79, first_sample=2)      # cache hit, both kinds known
        eng.check_range()
        assert eng.pipeline_parts() == int(groups)
        assert torch.isfinite(want).all() and torch.equal(x, want), float((x - want).abs().max())
    eng.set_graph(False)


# ---- post-sampling step -----------------------------------------------------------
@pytest.mark.parametrize("abs_3d", [False, True])
def test_recover_xyz_vs_reference(cases, abs_3d):
    """inv_transform + recover_from_ric on the device vs the real reference's output and the numpy oracle."""
    from oracle.post_oracle import recover_xyz
    mp = sub("data_loaders.humanml.scripts.motion_process")
    inp = cases.make_post_inputs()
    g = load_golden("post_ric")
    assert np.array_equal(g["fingerprint"], cases.fingerprint(inp))
    ref = g[f"xyz_abs{int(abs_3d)}"]
    out = mp.sample_to_xyz(tt(inp["sample"]), inp["mean"], inp["std"], 22, abs_3d).cpu().numpy()
    scale = max(1.0, float(np.abs(ref).max()))     # relative to the largest coordinate
    assert out.shape == ref.shape and ok("recover_xyz.vs_reference", max_abs(out, ref) / scale, 2e-5), max_abs(out, ref)
    assert ok("recover_xyz.vs_oracle", max_abs(out, recover_xyz(inp["sample"], inp["mean"], inp["std"], 22, abs_3d)) / scale, 2e-5)
    # the reference-signature wrapper: un-normalised [B, 1, T, 263] in, [B, 1, T, 22, 3] out
    data = tt(inp["sample"]).permute(0, 2, 3, 1) * tt(inp["std"]) + tt(inp["mean"])
    xyz = mp.recover_from_ric(data, 22, abs_3d)
    assert xyz.shape == (3, 1, 196, 22, 3)
    assert ok("recover_xyz.wrapper", max_abs(xyz[:, 0].permute(0, 2, 3, 1).cpu().numpy(), ref) / scale, 2e-5)


# ---- convolution over token rows (UNET building block) ------------------------------------------------------
@pytest.mark.parametrize("kind", ["k5", "k1", "down", "up"])
def test_conv_rows_h3_vs_torch(kind):
    """conv1d (k=5 pad 2; k=1), stride-2 conv (k=3, pad 1) and ConvTranspose1d (k=4, s=2, p=1) as split-f16 GEMMs
    over tap-shifted token rows with zero halo frames, vs torch (float64)."""
    eng, N = sub("engine"), sub("_native")
    lib = N.load()
    g = torch.Generator().manual_seed(hash(kind) % 1000)
    B, cin, cout = 3, 64, 96
    T_in, h_in = (56, 4)
    x = torch.randn(B, cin, T_in, generator=g)
    if kind in ("k5", "k1"):
        k, T_out, h_out = (5 if kind == "k5" else 1), T_in, h_in
        w = torch.randn(cout, cin, k, generator=g) * 0.1
        ref = torch.nn.functional.conv1d(x.double(), w.double(), padding=k // 2)
        wg = [w.permute(0, 2, 1).reshape(cout, k * cin)]                    # [n, tap * cin + c]
        launches = [dict(taps=k, pad=k // 2, a_mul=1, c_mul=0, c_add=0)]
    elif kind == "down":
        T_out, h_out = T_in // 2, h_in // 2
        w = torch.randn(cout, cin, 3, generator=g) * 0.1
        ref = torch.nn.functional.conv1d(x.double(), w.double(), stride=2, padding=1)
        wg = [w.permute(0, 2, 1).reshape(cout, 3 * cin)]
        launches = [dict(taps=3, pad=1, a_mul=2, c_mul=0, c_add=0)]
    else:
        T_out, h_out = T_in * 2, h_in * 2
        w = torch.randn(cin, cout, 4, generator=g) * 0.1                     # ConvTranspose1d weight [in, out, k]
        ref = torch.nn.functional.conv_transpose1d(x.double(), w.double(), stride=2, padding=1)
        # out[2j] = x[j-1] W3 + x[j] W1 ; out[2j+1] = x[j] W2 + x[j+1] W0
        wg = [torch.cat([w[:, :, 3].T, w[:, :, 1].T], dim=1), torch.cat([w[:, :, 2].T, w[:, :, 0].T], dim=1)]
        launches = [dict(taps=2, pad=1, a_mul=1, c_mul=2, c_add=0), dict(taps=2, pad=0, a_mul=1, c_mul=2, c_add=1)]
    bias = torch.randn(cout, generator=g)
    tp_in, tp_out = T_in + 2 * h_in, T_out + 2 * h_out
    guard = 8
    rows = torch.zeros(guard + B * tp_in + guard, cin)
    for b in range(B):
        rows[guard + b * tp_in + h_in: guard + b * tp_in + h_in + T_in] = x[b].T
    a_s = eng.split_f16(rows.to(DEV))                                       # [rows, 2 cin]
    out = torch.zeros(B * tp_out, cout, device=DEV)
    m_gemm = B * tp_in if kind == "up" else B * tp_out                       # GEMM rows = output (or input, up) rows
    for wmat, L in zip(wg, launches):
        w_s = eng.split_f16(eng.conv_weight_k_order(wmat.contiguous(), L["taps"]).to(DEV))   # tap-major -> the GEMM's K order
        a_ptr = a_s.data_ptr() + guard * (2 * cin) * 2
        with torch.cuda.device(DEV):
            N.check(lib.cmdi_conv_rows_h3(a_ptr, 2 * cin, N.ptr(w_s), N.ptr(bias.to(DEV)), 0, N.ptr(out), 0,
                                          m_gemm, cout, cin, L["taps"], L["pad"], L["a_mul"], L["c_mul"],
                                          L["c_add"], tp_out, h_out, h_out + T_out, 0,
                                          N.current_stream(torch.device(DEV))))
    got = out.cpu().view(B, tp_out, cout)
    assert float(got[:, :h_out].abs().max()) == 0.0 and float(got[:, h_out + T_out:].abs().max()) == 0.0  # halo
    got = got[:, h_out:h_out + T_out].permute(0, 2, 1)
    want = ref + bias.double()[None, :, None]
    assert ok("conv_rows_h3_vs_torch.rel_l2.0", rel_l2(got.numpy(), want.numpy()), 2e-6), rel_l2(got.numpy(), want.numpy())


@pytest.mark.parametrize("variant", [2, 0])
@pytest.mark.parametrize("kind", ["k5", "k1", "down", "up"])
def test_conv_rows_x6_vs_torch(kind, variant):
    """The U-Net's bf16x6 mode (round 5): conv1d (k=5 pad 2; k=1), the stride-2 convolution and ConvTranspose1d as gemm_x6
    GEMMs over tap-shifted fp32 rows (cmdi_conv_rows_x6: exact three-plane operands) vs torch in float64 — inputs scaled to
    3e5, beyond the f16 range the split-f16 form of the same convolutions is limited to; residual + second output covered."""
    eng, N = sub("engine"), sub("_native")
    lib = N.load()
    g = torch.Generator().manual_seed(len(kind) * 7 + variant)
    B, cin, cout = 3, 64, 96
    T_in, h_in = (56, 4)
    x = torch.randn(B, cin, T_in, generator=g) * 3e5
    if kind in ("k5", "k1"):
        k, T_out, h_out = (5 if kind == "k5" else 1), T_in, h_in
        w = torch.randn(cout, cin, k, generator=g) * 0.1
        ref = torch.nn.functional.conv1d(x.double(), w.double(), padding=k // 2)
        wg = [w.permute(0, 2, 1).reshape(cout, k * cin)]
        launches = [dict(taps=k, pad=k // 2, a_mul=1, c_mul=0, c_add=0)]
    elif kind == "down":
        T_out, h_out = T_in // 2, h_in // 2
        w = torch.randn(cout, cin, 3, generator=g) * 0.1
        ref = torch.nn.functional.conv1d(x.double(), w.double(), stride=2, padding=1)
        wg = [w.permute(0, 2, 1).reshape(cout, 3 * cin)]
        launches = [dict(taps=3, pad=1, a_mul=2, c_mul=0, c_add=0)]
    else:
        T_out, h_out = T_in * 2, h_in * 2
        w = torch.randn(cin, cout, 4, generator=g) * 0.1
        ref = torch.nn.functional.conv_transpose1d(x.double(), w.double(), stride=2, padding=1)
        wg = [torch.cat([w[:, :, 3].T, w[:, :, 1].T], dim=1), torch.cat([w[:, :, 2].T, w[:, :, 0].T], dim=1)]
        launches = [dict(taps=2, pad=1, a_mul=1, c_mul=2, c_add=0), dict(taps=2, pad=0, a_mul=1, c_mul=2, c_add=1)]
    bias = torch.randn(cout, generator=g) * 1e4
    tp_in, tp_out = T_in + 2 * h_in, T_out + 2 * h_out
    guard = 8
    rows = torch.zeros(guard + B * tp_in + guard, cin)
    for b in range(B):
        rows[guard + b * tp_in + h_in: guard + b * tp_in + h_in + T_in] = x[b].T
    a = rows.to(DEV)
    resid = (torch.randn(B * tp_out, cout, generator=g) * 1e5).to(DEV)
    out = torch.zeros(B * tp_out, cout, device=DEV)
    out2 = torch.zeros(B * tp_out, 2 * cout, device=DEV)            # second output with its own row stride (the skip's concat slot)
    m_gemm = B * tp_in if kind == "up" else B * tp_out
    for wmat, L in zip(wg, launches):
        w_p = eng.pack_x6(eng.conv_weight_k_order(wmat.contiguous(), L["taps"]).to(DEV))
        with torch.cuda.device(DEV):
            N.check(lib.cmdi_conv_rows_x6(a.data_ptr() + guard * cin * 4, cin, N.ptr(w_p), N.ptr(bias.to(DEV)), N.ptr(resid), N.ptr(out),
                                          out2.data_ptr() + cout * 4, 2 * cout, m_gemm, cout, cin, L["taps"], L["pad"], L["a_mul"],
                                          L["c_mul"], L["c_add"], tp_out, h_out, h_out + T_out, variant,
                                          N.current_stream(torch.device(DEV))))
    got = out.cpu().view(B, tp_out, cout)
    assert float(got[:, :h_out].abs().max()) == 0.0 and float(got[:, h_out + T_out:].abs().max()) == 0.0  # halo rows untouched
    assert torch.equal(out2[:, cout:].cpu().view(B, tp_out, cout), got) and float(out2[:, :cout].abs().max()) == 0.0
    got = got[:, h_out:h_out + T_out].permute(0, 2, 1)
    want = ref + bias.double()[None, :, None] + resid.cpu().double().view(B, tp_out, cout)[:, h_out:h_out + T_out].permute(0, 2, 1)
    assert ok("conv_rows_x6_vs_torch.rel_l2", rel_l2(got.numpy(), want.numpy()), 2e-6), rel_l2(got.numpy(), want.numpy())


@pytest.mark.parametrize("kind", ["k5", "down", "up", "k1"])
def test_conv_rows_persistent_is_bitwise_the_tiled_kernel(kind):
    """Round 4: the U-Net's long-K convolutions run on the persistent GEMM (gemm_h3p.hpp with tap-shifted A requests and the
    convolution's output-row rule).  Same products in the same order: its fp32 output must equal the tiled kernel's bit for
    bit — every addressing mode (k=5, stride 2, transposed, 1x1), a row count that leaves edge tiles and a half-tile
    remainder, halo rows untouched (the output buffer is pre-filled with a sentinel)."""
    eng, N = sub("engine"), sub("_native")
    lib = N.load()
    g = torch.Generator().manual_seed(200 + len(kind))
    B, cin, cout = 37, 64, 512
    T_in, h_in = 56, 4
    taps, pad, a_mul, c_mul, c_add = dict(k5=(5, 2, 1, 0, 0), k1=(1, 0, 1, 0, 0), down=(3, 1, 2, 0, 0), up=(2, 1, 1, 2, 0))[kind]
    T_out, h_out = (T_in // 2, h_in // 2) if kind == "down" else ((T_in * 2, h_in * 2) if kind == "up" else (T_in, h_in))
    tp_in, tp_out = T_in + 2 * h_in, T_out + 2 * h_out
    guard = 8
    rows = torch.zeros(guard + B * tp_in + guard, cin)
    rows[guard:guard + B * tp_in] = torch.randn(B * tp_in, cin, generator=g)
    a_s = eng.split_f16(rows.to(DEV))
    w_s = eng.split_f16((torch.randn(cout, taps * cin, generator=g) * 0.1).to(DEV))
    bias = torch.randn(cout, generator=g).to(DEV)
    m_gemm = B * tp_in if kind == "up" else B * tp_out
    outs = []
    for tile in (0, 50) + ((51,) if kind in ("k5", "k1") else ()):    # 51: the persistent kernel over frames only (rc_tv)
        out = torch.full((B * tp_out, cout), -7.25, device=DEV)
        with torch.cuda.device(DEV):
            N.check(lib.cmdi_conv_rows_h3(a_s.data_ptr() + guard * (2 * cin) * 2, 2 * cin, N.ptr(w_s), N.ptr(bias), 0, N.ptr(out), 0,
                                          m_gemm, cout, cin, taps, pad, a_mul, c_mul, c_add, tp_out, h_out, h_out + T_out, tile,
                                          N.current_stream(torch.device(DEV))))
        outs.append(out.cpu())
    assert torch.isfinite(outs[0]).all() and all(torch.equal(outs[0], o) for o in outs[1:])
    v = outs[1].view(B, tp_out, cout)
    assert float((v[:, :h_out] + 7.25).abs().max()) == 0.0 and float((v[:, h_out + T_out:] + 7.25).abs().max()) == 0.0
    assert float((v[:, h_out:h_out + T_out:2 if kind == "up" else 1] + 7.25).abs().min()) > 0.0   # frames were written


def test_conv_rows_persistent_refuses_what_it_does_not_compute():
    """tiles 50 / 51 of cmdi_conv_rows_h3 reach the persistent kernel, which computes plain fp32
    convolution outputs with n % 256 == 0 only (tile 51: stride-1 rows).  Every other combination must come back as an
    error with the output untouched — never CMDI_OK with columns unwritten or rows addressed as a plain GEMM."""
    eng, N = sub("engine"), sub("_native")
    lib = N.load()
    g = torch.Generator().manual_seed(77)
    B, cin, T, h = 5, 64, 56, 4
    tp = T + 2 * h
    a_s = eng.split_f16(torch.randn(8 + B * tp + 8, cin, generator=g).to(DEV))
    a_ptr = a_s.data_ptr() + 8 * (2 * cin) * 2
    stream = N.current_stream(torch.device(DEV))

    def call(cout, tile, split=False, resid=False, taps=5, pad=2, a_mul=1, c_mul=0):
        w_s = eng.split_f16((torch.randn(cout, taps * cin, generator=g) * 0.1).to(DEV))
        bias = torch.zeros(cout, device=DEV)
        out = torch.full((2 * B * tp, cout), -7.25, device=DEV)
        out_s = torch.zeros((2 * B * tp, 2 * cout), dtype=torch.float16, device=DEV) if split else None
        res = torch.zeros((2 * B * tp, cout), device=DEV) if resid else None
        with torch.cuda.device(DEV):
            rc = lib.cmdi_conv_rows_h3(a_ptr, 2 * cin, N.ptr(w_s), N.ptr(bias), N.ptr(res), 0 if split else N.ptr(out), N.ptr(out_s),
                                       B * tp, cout, cin, taps, pad, a_mul, c_mul, 0, tp, h, h + T, tile, stream)
        torch.cuda.synchronize()
        untouched = bool((out == -7.25).all()) and (out_s is None or bool((out_s == 0).all()))
        return rc, untouched

    assert call(256, 50)[0] == 0 and call(256, 51)[0] == 0                     # the supported form still runs
    for kw in (dict(cout=320, tile=50), dict(cout=128, tile=50), dict(cout=320, tile=51),      # n % 256 != 0, n < 256
               dict(cout=256, tile=50, split=True), dict(cout=256, tile=51, split=True),       # split output
               dict(cout=256, tile=50, resid=True), dict(cout=256, tile=51, resid=True),       # residual epilogue
               dict(cout=256, tile=51, taps=3, pad=1, a_mul=2),                                # stride-2 rows as logical rows
               dict(cout=256, tile=51, taps=2, pad=1, c_mul=2)):                               # transposed-convolution rows
        rc, untouched = call(**kw)
        assert rc != 0 and untouched, (kw, rc, untouched)
        assert b"tile" in lib.cmdi_last_error() or b"launch_gemm_h3" in lib.cmdi_last_error()


# ---- MDM_UNET denoiser -------------------------------------------------------------
UNET_PRECISIONS = ["f16x3", "bf16x6"]     # the native U-Net's arithmetic modes (bf16x6: round 5, exact operands, no f16 range limit)


def make_unet(cases, precision=None):
    mu = sub("utils.model_util")
    case = cases.UNET_CASE
    args = SimpleNamespace(dataset="humanml", arch="unet", keyframe_conditioned=True, dim_mults=case["dim_mults"],
                           cond_mask_prob=0.1)
    model, _ = mu.create_model_and_diffusion(args, None)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    g = load_golden("unet_fwd")
    assert sorted(shapes) == list(g["names"]), "state-dict names differ from the reference's MDM_UNET"
    mu.load_model_wo_clip(model, weights.to_torch(weights.fill_like(shapes, case["weight_seed"])) |
                          {k: v for k, v in model.state_dict().items() if k.endswith(".pe")})
    model.native_precision = precision
    return model.to(DEV).eval(), g


@pytest.mark.parametrize("precision", UNET_PRECISIONS)
def test_unet_forward_vs_reference(cases, precision):
    """MDM_UNET (keyframe-conditioned, text, CFG) on the device vs the real reference's CPU outputs."""
    inp = cases.make_unet_inputs()
    model, g = make_unet(cases, precision)
    assert np.array_equal(g["fingerprint"], cases.fingerprint(inp))
    x, t = tt(inp["x"]), tt(inp["t"])
    kw = dict(obs_x0=tt(inp["obs_x0"]), obs_mask=tt(inp["obs_mask"]))
    y = {"text_embed": tt(inp["enc_text"])}
    oc = model(x, t, y=y, **kw).cpu().numpy()
    assert model._engine.precision == precision
    assert np.array_equal(oc, model(x, t, y=y, **kw).cpu().numpy())   # split-K accumulation is order-independent
    ou = model(x, t, y=dict(y, uncond=True), **kw).cpu().numpy()
    wrapped = sub("model.cfg_sampler").ClassifierFreeSampleModel(model)
    cfg = wrapped(x, t, y=dict(y, text_scale=tt(inp["text_scale"])), **kw).cpu().numpy()
    for mine, key in ((oc, "out_cond"), (ou, "out_uncond"), (cfg, "out_cfg")):
        assert np.isfinite(mine).all()
        assert ok("unet_forward_vs_reference.max_abs.0", max_abs(mine, g[key]), 2e-4) and ok("unet_forward_vs_reference.rel_l2.0", rel_l2(mine, g[key]), 2e-5), \
            (key, max_abs(mine, g[key]), rel_l2(mine, g[key]))


@pytest.mark.parametrize("precision", UNET_PRECISIONS)
@pytest.mark.parametrize("B,T", [(3, 100), (1, 224), (2, 17)])
def test_unet_vs_oracle_other_shapes(cases, B, T, precision):
    """Frame counts other than the golden 196 (right-padded to 224 inside the model) vs the numpy oracle."""
    from oracle.unet_oracle import UnetOracle
    model, _ = make_unet(cases, precision)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    rng = np.random.default_rng(1000 + 7 * B + T)
    shape = (B, 263, 1, T)
    x = rng.standard_normal(shape).astype(np.float32)
    obs = rng.standard_normal(shape).astype(np.float32)
    m = rng.random(shape) < 0.2
    t = rng.integers(0, 1000, B)
    enc = rng.standard_normal((B, 512)).astype(np.float32)
    scale = np.full(B, 2.5, np.float32)
    want, _, _ = UnetOracle(sd).forward_cfg(x, t, enc, scale, obs, m)
    wrapped = sub("model.cfg_sampler").ClassifierFreeSampleModel(model)
    got = wrapped(tt(x), tt(t), y={"text_embed": tt(enc), "text_scale": tt(scale)}, obs_x0=tt(obs), obs_mask=tt(m)).cpu().numpy()
    assert ok("unet_vs_oracle_other_shapes.max_abs.0", max_abs(got, want), 2e-4) and ok("unet_vs_oracle_other_shapes.rel_l2.0", rel_l2(got, want), 2e-5), (max_abs(got, want), rel_l2(got, want))


@pytest.mark.parametrize("precision", UNET_PRECISIONS)
def test_unet_chain_vs_reference(cases, precision):
    """The conditional_synthesis.py call (p_sample_loop, CFG wrapper, obs_x0 / obs_mask, imputation) with the
    native MDM_UNET vs the real reference's chain on the same injected noise."""
    cc = cases.UNET_CHAIN
    ci = cases.make_unet_chain_inputs()
    g = load_golden("unet_chain")
    assert np.array_equal(g["fingerprint"], cases.fingerprint(ci))
    model, _ = make_unet(cases, precision)
    wrapped = sub("model.cfg_sampler").ClassifierFreeSampleModel(model)
    diffusion = make_diffusion(cc["respacing"])
    obs_mask = tt(ci["obs_mask"])
    y = {"mask": tt(ci["len_mask"]), "lengths": tt(ci["lengths"]), "text_embed": tt(ci["enc_text"]),
         "text_scale": tt(ci["text_scale"]), "inpainting_mask": obs_mask, "inpainted_motion": tt(ci["x0"]),
         "imputate": True, "stop_imputation_at": cc["stop_imputation_at"], "replacement_distribution": "conditional",
         "reconstruction_guidance": False, "diffusion_steps": 1000}
    diffusion.injected_noise = tt(ci["noise"])
    final = diffusion.p_sample_loop(wrapped, ci["x_T"].shape, noise=tt(ci["x_T"]), clip_denoised=False,
                                    model_kwargs={"y": y, "obs_x0": tt(ci["x0"]), "obs_mask": obs_mask}).cpu().numpy()
    assert ok("unet_chain_vs_reference.rel_l2.0", rel_l2(final, g["final"]), 1e-4), rel_l2(final, g["final"])


@pytest.mark.parametrize("precision", UNET_PRECISIONS)
def test_unet_vjp_vs_reference_autograd(cases, precision):
    """Input-VJP of ClassifierFreeSampleModel(MDM_UNET) (unet.hip::unet_backward) vs torch autograd through the real
    reference on CPU; also reached through torch.autograd.grad on the native module, and linear in gout over 21 decades."""
    vi = cases.make_unet_vjp_inputs()
    g = load_golden("unet_vjp")
    assert np.array_equal(g["fingerprint"], cases.fingerprint(vi))
    model, _ = make_unet(cases, precision)
    wrapped = sub("model.cfg_sampler").ClassifierFreeSampleModel(model)
    y = {"text_embed": tt(vi["enc_text"]), "text_scale": tt(vi["text_scale"])}
    kw = dict(obs_x0=tt(vi["obs_x0"]), obs_mask=tt(vi["obs_mask"]))
    z = tt(vi["x"]).requires_grad_(True)
    with torch.enable_grad():
        out = wrapped(z, tt(vi["t"]), y=y, **kw)
        assert ok("unet_vjp_vs_reference_autograd.rel_l2.0", rel_l2(out.detach().cpu().numpy(), g["out"]), 2e-5)
        gx, = torch.autograd.grad((out * tt(vi["gout"])).sum(), z)
    gx = gx.cpu().numpy()
    assert float(np.abs(gx[vi["obs_mask"]]).max()) == 0.0          # observed entries are replaced by obs_x0
    assert ok("unet_vjp_vs_reference_autograd.rel_l2.1", rel_l2(gx, g["gx"]), 5e-5), rel_l2(gx, g["gx"])
    eng = model._engine            # the engine (and activation stash) of the forward pass above
    assert eng is not None and eng.want_grad
    for k in (1e-12, 1e9):
        gk = eng.mdm_vjp(tt(vi["gout"] * np.float32(k))).cpu().numpy().astype(np.float64) / k
        assert ok("unet_vjp_vs_reference_autograd.rel_l2.2", rel_l2(gk, g["gx"]), 5e-5), (k, rel_l2(gk, g["gx"]))


@pytest.mark.parametrize("precision", UNET_PRECISIONS)
def test_unet_graph_replay_is_bitwise_identical(cases, precision):
    """Round 6 (VERDICT r5 task 6): cmdi_sample_loop replays MDM_UNET steps as hipGraphs too — the embedding kernel reads the
    step's timestep from the chain's device table through the cursor, like token0_kernel.  A 12-step chain with keyframe
    conditioning, imputation and reconstruction guidance that stops at step 4 (both graph kinds: forward + input-VJP, forward
    only) == the eager loop, bit for bit, also when the graphs are replayed by a second call."""
    N = sub("_native")
    model, _ = make_unet(cases, precision)
    diffusion = make_diffusion([12])
    B, T = 2, 100
    rng = np.random.default_rng(19)
    shape = (B, 263, 1, T)
    emb = tt(rng.standard_normal((B, 512)).astype(np.float32))
    x0 = tt(rng.standard_normal(shape).astype(np.float32))
    mask = tt(rng.random(shape) < 0.3)
    eng = model.engine(torch.device(DEV), max_batch=B, max_frames=T, want_grad=True)
    eng.set_schedule(diffusion.engine_tables(), key="ug")

    def run(graph):
        eng.set_graph(graph)
        eng.set_condition(batch=B, n_frames=T, cfg=True, enc_text=emb, text_scale=torch.full((B,), 2.5, device=DEV),
                          inpaint_mask=mask, inpaint_motion=x0, imputate=True, stop_imputation_at=1,
                          recon_guidance=True, stop_recguidance_at=4, recon_w=np.full((12,), 20.0, dtype=np.float32),
                          obs_x0=x0, obs_mask=mask)
        x = eng.randn(shape, seed=5)
        eng.sample_loop(x, 11, 0, sampler=N.CMDI_SAMPLER_DDPM, seed=77, first_sample=2)
        eng.check_range()
        return x.clone()

    eager = run(False)
    replay = run(True)
    again = run(True)
    eng.set_graph(False)
    assert torch.isfinite(eager).all() and float(eager.abs().max()) > 0
    assert torch.equal(eager, replay) and torch.equal(eager, again)


def make_unet_attention(cases):
    mm = sub("model.mdm_unet")
    mu = sub("utils.model_util")
    case = cases.UNET_ATTN_CASE
    model = mm.MDM_UNET(njoints=263, nfeats=1, latent_dim=512, dim_mults=case["dim_mults"], attention=True,
                        keyframe_conditioned=True, cond_mode="text", cond_mask_prob=0.1)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.startswith("clip_model.")}
    g = load_golden("unet_attn")
    assert sorted(shapes) == list(g["names"]), "state-dict names differ from the reference's MDM_UNET(attention=True)"
    mu.load_model_wo_clip(model, weights.to_torch(weights.fill_like(shapes, case["weight_seed"])) |
                          {k: v for k, v in model.state_dict().items() if k.endswith(".pe")})
    return model.to(DEV).eval(), g


def test_unet_attention_forward_and_vjp_vs_reference(cases):
    """MDM_UNET(attention=True): the eight Residual(PreNorm(LinearAttention)) sites (reference model/mdm_unet.py:102-156)
    forward (cond / uncond / CFG) and input-VJP vs the real reference's CPU outputs / torch autograd."""
    inp = cases.make_unet_vjp_inputs(cases.UNET_ATTN_CASE)
    model, g = make_unet_attention(cases)
    assert np.array_equal(g["fingerprint"], cases.fingerprint(inp))
    x, t = tt(inp["x"]), tt(inp["t"])
    kw = dict(obs_x0=tt(inp["obs_x0"]), obs_mask=tt(inp["obs_mask"]))
    y = {"text_embed": tt(inp["enc_text"])}
    oc = model(x, t, y=y, **kw).cpu().numpy()
    ou = model(x, t, y=dict(y, uncond=True), **kw).cpu().numpy()
    wrapped = sub("model.cfg_sampler").ClassifierFreeSampleModel(model)
    cfg = wrapped(x, t, y=dict(y, text_scale=tt(inp["text_scale"])), **kw).cpu().numpy()
    for mine, key in ((oc, "out_cond"), (ou, "out_uncond"), (cfg, "out_cfg")):
        assert np.isfinite(mine).all()
        assert ok("unet_attention_forward_and_vjp_vs_reference.max_abs.0", max_abs(mine, g[key]), 2e-4) and ok("unet_attention_forward_and_vjp_vs_reference.rel_l2.0", rel_l2(mine, g[key]), 2e-5), \
            (key, max_abs(mine, g[key]), rel_l2(mine, g[key]))
    z = tt(inp["x"]).requires_grad_(True)
    with torch.enable_grad():
        out = wrapped(z, t, y=dict(y, text_scale=tt(inp["text_scale"])), **kw)
        gx, = torch.autograd.grad((out * tt(inp["gout"])).sum(), z)
    gx = gx.cpu().numpy()
    assert float(np.abs(gx[inp["obs_mask"]]).max()) == 0.0
    assert ok("unet_attention_forward_and_vjp_vs_reference.rel_l2.1", rel_l2(gx, g["gx"]), 5e-5), rel_l2(gx, g["gx"])
    eng = model._engine
    for k in (1e-12, 1e9):   # linear in gout: the power-of-two gradient scale passes through the attention sites
        gk = eng.mdm_vjp(tt(inp["gout"] * np.float32(k))).cpu().numpy().astype(np.float64) / k
        assert ok("unet_attention_forward_and_vjp_vs_reference.rel_l2.2", rel_l2(gk, g["gx"]), 5e-5), (k, rel_l2(gk, g["gx"]))


def test_unet_attention_out_of_range_raises_and_the_module_keeps_working(cases):
    """MDM_UNET(attention=True) exists in f16x3 only (the LinearAttention sites have no bf16x6 form), so an
    evaluation that leaves the f16 range must raise RangeError — through check_range() and through a sampling loop's own probe —
    WITHOUT taking the bf16x6 fallback (range_fallback() False, no engine swap), and the module must keep giving the golden
    result on in-range inputs afterwards."""
    N = sub("_native")
    inp = cases.make_unet_vjp_inputs(cases.UNET_ATTN_CASE)
    model, g = make_unet_attention(cases)
    x, t = tt(inp["x"]), tt(inp["t"])
    kw = dict(obs_x0=tt(inp["obs_x0"]), obs_mask=tt(inp["obs_mask"]))
    y = {"text_embed": tt(inp["enc_text"])}
    first = model(x, t, y=y, **kw).cpu().numpy()
    model.check_range()
    eng = model._engine
    model(x * 1e6, t, y=y, **kw)                       # 1e6 x the data scale: the frame rows leave the f16 range
    with pytest.raises(N.RangeError):
        model.check_range()
    assert model.range_fallback() is False and not getattr(model, "_range_fallback", False)
    assert model._engine is eng and eng.precision == "f16x3"
    again = model(x, t, y=y, **kw).cpu().numpy()
    model.check_range()                                # the flag was cleared by the read-back that raised
    assert np.array_equal(again, first)
    assert rel_l2(again, g["out_cond"]) <= 2e-5
    # the sampling loop's own probe: the error reaches the caller, no silent retry on an engine that does not exist
    diffusion = make_diffusion([4])
    B, _, _, T = inp["x"].shape
    yy = {"mask": torch.ones(B, 1, 1, T, dtype=torch.bool, device=DEV), "lengths": torch.full((B,), T), "text_embed": tt(inp["enc_text"]),
          "text_scale": tt(inp["text_scale"])}
    wrapped = sub("model.cfg_sampler").ClassifierFreeSampleModel(model)
    with pytest.raises(N.RangeError):
        diffusion.p_sample_loop(wrapped, inp["x"].shape, noise=x * 1e6, clip_denoised=False,
                                model_kwargs={"y": yy, "obs_x0": kw["obs_x0"] * 1e6, "obs_mask": kw["obs_mask"]})
    assert model._engine is not None and model._engine.precision == "f16x3"
    assert np.array_equal(model(x, t, y=y, **kw).cpu().numpy(), first)


@pytest.mark.parametrize("B,T", [(3, 100), (1, 224)])
def test_unet_attention_vs_oracle_other_shapes(cases, B, T):
    from oracle.unet_oracle import UnetOracle
    model, _ = make_unet_attention(cases)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    rng = np.random.default_rng(2000 + 7 * B + T)
    shape = (B, 263, 1, T)
    x = rng.standard_normal(shape).astype(np.float32)
    obs = rng.standard_normal(shape).astype(np.float32)
    m = rng.random(shape) < 0.2
    t = rng.integers(0, 1000, B)
    enc = rng.standard_normal((B, 512)).astype(np.float32)
    scale = np.full(B, 2.5, np.float32)
    want, _, _ = UnetOracle(sd).forward_cfg(x, t, enc, scale, obs, m)
    wrapped = sub("model.cfg_sampler").ClassifierFreeSampleModel(model)
    got = wrapped(tt(x), tt(t), y={"text_embed": tt(enc), "text_scale": tt(scale)}, obs_x0=tt(obs), obs_mask=tt(m)).cpu().numpy()
    assert ok("unet_attention_vs_oracle_other_shapes.max_abs.0", max_abs(got, want), 2e-4) and ok("unet_attention_vs_oracle_other_shapes.rel_l2.0", rel_l2(got, want), 2e-5), (max_abs(got, want), rel_l2(got, want))


@pytest.mark.parametrize("fuse", ["0", "2", "3"])
def test_unet_forward_groupnorm_fusion_modes(cases, monkeypatch, fuse):
    """CMDI_UNET_FUSE_GN = 0 (no convolution + GroupNorm fusion: the default since round 4), 2 (fused epilogue at level 1: the
    default of rounds 2-3) and 3 (levels 0 AND 1) give the reference's output to the same tolerance."""
    monkeypatch.setenv("CMDI_UNET_FUSE_GN", fuse)
    inp = cases.make_unet_inputs()
    model, g = make_unet(cases)
    wrapped = sub("model.cfg_sampler").ClassifierFreeSampleModel(model)
    cfg = wrapped(tt(inp["x"]), tt(inp["t"]), y={"text_embed": tt(inp["enc_text"]), "text_scale": tt(inp["text_scale"])},
                  obs_x0=tt(inp["obs_x0"]), obs_mask=tt(inp["obs_mask"])).cpu().numpy()
    assert ok("unet_forward_groupnorm_fusion_modes.max_abs.0", max_abs(cfg, g["out_cfg"]), 2e-4) and ok("unet_forward_groupnorm_fusion_modes.rel_l2.0", rel_l2(cfg, g["out_cfg"]), 2e-5), rel_l2(cfg, g["out_cfg"])


def test_unet_groupnorm_one_pass_is_the_two_kernel_groupnorm(cases, monkeypatch):
    """Round 4: GroupNorm as ONE register-resident pass (unet.hip gn_fused_kernel) against the statistics + apply kernels it
    replaces (CMDI_UNET_GN1=0).  With 256 threads per (sequence, group) (=1): same element map, same summation order, same
    formulas -> the same bits.  With 1,024 threads (=2, the default) the sums run in another order: equal to 1e-6 of the
    output scale, and to the reference within the usual bound."""
    inp = cases.make_unet_inputs()
    outs = {}
    for mode in ("0", "1", "2"):
        monkeypatch.setenv("CMDI_UNET_GN1", mode)
        monkeypatch.setenv("CMDI_UNET_FUSE_GN", "0")     # every GroupNorm through the kernels under test
        model, g = make_unet(cases)
        wrapped = sub("model.cfg_sampler").ClassifierFreeSampleModel(model)
        outs[mode] = wrapped(tt(inp["x"]), tt(inp["t"]), y={"text_embed": tt(inp["enc_text"]), "text_scale": tt(inp["text_scale"])},
                             obs_x0=tt(inp["obs_x0"]), obs_mask=tt(inp["obs_mask"])).cpu().numpy()
    assert np.isfinite(outs["2"]).all()
    assert np.array_equal(outs["0"], outs["1"]), float(np.abs(outs["0"] - outs["1"]).max())
    assert ok("unet_groupnorm_one_pass.order", rel_l2(outs["2"], outs["0"]), 2e-6)
    assert ok("unet_groupnorm_one_pass.rel_l2", rel_l2(outs["2"], g["out_cfg"]), 2e-5)


def test_unet_gradient_gemms_on_the_persistent_kernel_are_bitwise_the_tiled_route(cases, monkeypatch):
    """Round 6: at the bench's batch the long-K gradient GEMMs of the U-Net's input-VJP (the k=5 convolutions' input gradients
    at levels 0 and 1: M = 64 x Tp rows, K = 5,120) take the forward's route — persistent kernel over frames only.  Same
    products in the same order: the input gradient of the released geometry at B=32 must equal the tiled route's
    (CMDI_UNET_PERSIST_BWD=0) bit for bit, and the forward output with it."""
    mu = sub("utils.model_util")
    case = cases.UNET_XL_CASE
    B, T = 32, 196
    rng = np.random.default_rng(4242)
    shape = (B, 263, 1, T)
    x, obs, gout = (rng.standard_normal(shape).astype(np.float32) for _ in range(3))
    m = rng.random(shape) < 0.2
    enc = rng.standard_normal((B, 512)).astype(np.float32)
    t = rng.integers(0, 1000, B)
    sc = np.full(B, 2.5, np.float32)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("CMDI_UNET_PERSIST_BWD", mode)
        args = SimpleNamespace(dataset="humanml", arch="unet", keyframe_conditioned=True, dim_mults=case["dim_mults"], cond_mask_prob=0.1)
        model, _ = mu.create_model_and_diffusion(args, None)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        mu.load_model_wo_clip(model, weights.to_torch(weights.fill_like(shapes, case["weight_seed"])) |
                              {k: v for k, v in model.state_dict().items() if k.endswith(".pe")})
        model = model.to(DEV).eval()
        wrapped = sub("model.cfg_sampler").ClassifierFreeSampleModel(model)
        z = tt(x).requires_grad_(True)
        with torch.enable_grad():
            out = wrapped(z, tt(t), y={"text_embed": tt(enc), "text_scale": tt(sc)}, obs_x0=tt(obs), obs_mask=tt(m))
            gx, = torch.autograd.grad((out * tt(gout)).sum(), z)
        res[mode] = (out.detach().cpu().numpy(), gx.cpu().numpy())
        model.invalidate_engine()
        del model, wrapped
        torch.cuda.empty_cache()
    assert np.isfinite(res["1"][1]).all() and float(np.abs(res["1"][1]).max()) > 0.0
    assert np.array_equal(res["0"][0], res["1"][0])
    assert np.array_equal(res["0"][1], res["1"][1]), float(np.abs(res["0"][1] - res["1"][1]).max())


@pytest.mark.parametrize("B", [1, 2, 10])
def test_unet_split_k_schedules_agree(cases, monkeypatch, B):
    """Round 6: at small batches the long-K GEMMs of MDM_UNET run as split-K — more slices where the tiles do not fill the chip,
    and (split_k_generic) also the GEMMs whose epilogue adds a residual, writes split rows or maps rows (down / up-sampling
    convolutions, the residual blocks' gradient GEMMs) through scratch slices + sum_slices_kernel.  The three schedules
    (CMDI_UNET_SPLITK = 2: the round-2 rule, 3: without the generic path, 1: default) differ in summation order only: forward
    output and input gradient of the released geometry agree to the rounding level of one evaluation (1e-5 / 2e-5 stated; measured
    2.4e-6 / 4.8e-6), and the default stays within the
    usual bound of the float64-checked oracle."""
    from oracle.unet_oracle import UnetOracle
    mu = sub("utils.model_util")
    case = cases.UNET_XL_CASE
    T = 196
    rng = np.random.default_rng(777 + B)
    shape = (B, 263, 1, T)
    x, obs, gout = (rng.standard_normal(shape).astype(np.float32) for _ in range(3))
    m = rng.random(shape) < 0.2
    enc = rng.standard_normal((B, 512)).astype(np.float32)
    t = rng.integers(0, 1000, B)
    sc = np.full(B, 2.5, np.float32)
    res, sd_np = {}, None
    for mode in ("2", "3", "1"):
        monkeypatch.setenv("CMDI_UNET_SPLITK", mode)
        args = SimpleNamespace(dataset="humanml", arch="unet", keyframe_conditioned=True, dim_mults=case["dim_mults"], cond_mask_prob=0.1)
        model, _ = mu.create_model_and_diffusion(args, None)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        mu.load_model_wo_clip(model, weights.to_torch(weights.fill_like(shapes, case["weight_seed"])) |
                              {k: v for k, v in model.state_dict().items() if k.endswith(".pe")})
        model = model.to(DEV).eval()
        sd_np = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
        wrapped = sub("model.cfg_sampler").ClassifierFreeSampleModel(model)
        z = tt(x).requires_grad_(True)
        with torch.enable_grad():
            out = wrapped(z, tt(t), y={"text_embed": tt(enc), "text_scale": tt(sc)}, obs_x0=tt(obs), obs_mask=tt(m))
            gx, = torch.autograd.grad((out * tt(gout)).sum(), z)
        res[mode] = (out.detach().cpu().numpy(), gx.cpu().numpy())
        model.invalidate_engine()
        del model, wrapped
        torch.cuda.empty_cache()
    for mode in ("2", "3"):
        assert ok("unet_split_k_schedules.forward", rel_l2(res["1"][0], res[mode][0]), 1e-5), (mode, rel_l2(res["1"][0], res[mode][0]))
        assert ok("unet_split_k_schedules.vjp", rel_l2(res["1"][1], res[mode][1]), 2e-5), (mode, rel_l2(res["1"][1], res[mode][1]))
    if B <= 2:
        want, _, _ = UnetOracle(sd_np).forward_cfg(x, t, enc, sc, obs, m)
        assert ok("unet_split_k_schedules.vs_oracle", rel_l2(res["1"][0], want), 2e-5), rel_l2(res["1"][0], want)


@pytest.mark.parametrize("scale,expect", [(4000.0, "f16x3"), (60000.0, "bf16x6"), (60000.0, "pinned")])
def test_unet_real_scale_activations_and_range_probe(cases, scale, expect, monkeypatch):
    """MDM_UNET's range story.  The reference's U-Net is plain fp32 at any activation scale (model/mdm_unet.py:561-849); the
    native default (f16x3) carries operands as split f16, |x| < 65504.  (1) Activations far above the synthetic-weight scale
    stay exact to the usual tolerance on f16x3 — the first block's residual 1x1 convolution is scaled so that the block's
    output (an operand of the next convolution) reaches ~1e4.  (2) Round 5 (VERDICT r4 task 4): activations that DO leave the
    f16 range (the same weights x 15: ~1e5) no longer strand the checkpoint — the pre-chain probe sends the chain to the
    bf16x6 engine (every convolution on gemm_x6: exact three-plane operands, fp32's range) BEFORE its first step, and both the
    forward value and the chain's sample are within the tolerance of the float64-checked oracle.  (3) An engine PINNED to
    f16x3 still refuses — a RangeError that names the cause before any step is spent, never a silent wrong sample."""
    from oracle.unet_oracle import UnetOracle
    from oracle import diffusion_oracle as do
    N = sub("_native")
    mu = sub("utils.model_util")
    case = cases.UNET_CASE
    args = SimpleNamespace(dataset="humanml", arch="unet", keyframe_conditioned=True, dim_mults=case["dim_mults"],
                           cond_mask_prob=0.1)
    model, _ = mu.create_model_and_diffusion(args, None)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = weights.fill_like(shapes, 91)
    first = "unet.downs.0.0.residual_conv.weight"
    assert first in sd, sorted(sd)[:8]
    sd[first] = sd[first] * np.float32(scale)
    mu.load_model_wo_clip(model, weights.to_torch(sd) | {k: v for k, v in model.state_dict().items() if k.endswith(".pe")})
    model = model.to(DEV).eval()
    if expect == "pinned":
        model.native_precision = "f16x3"
    wrapped = sub("model.cfg_sampler").ClassifierFreeSampleModel(model)
    B, T = 2, 64
    rng = np.random.default_rng(92)
    shape = (B, 263, 1, T)
    x = rng.standard_normal(shape).astype(np.float32)
    obs = rng.standard_normal(shape).astype(np.float32)
    m = rng.random(shape) < 0.2
    enc = rng.standard_normal((B, 512)).astype(np.float32)
    sc = np.full(B, 2.5, np.float32)
    t = rng.integers(0, 1000, B)
    full = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    oracle = UnetOracle(full)
    want, _, _ = oracle.forward_cfg(x, t, enc, sc, obs, m)
    call = lambda: wrapped(tt(x), tt(t), y={"text_embed": tt(enc), "text_scale": tt(sc)}, obs_x0=tt(obs), obs_mask=tt(m)).cpu().numpy()
    if expect == "f16x3":
        got = call()
        model.check_range()
        assert model._engine.precision == "f16x3"
        assert ok("unet_real_scale.rel_l2", rel_l2(got, want), 2e-5, precision="f16x3"), rel_l2(got, want)
        return
    # ---- activations beyond the f16 range -------------------------------------------------------------------------------
    n_steps = 12
    monkeypatch.setattr(sub("diffusion.gaussian_diffusion").GaussianDiffusion, "RANGE_PROBE_MIN_STEPS", n_steps)   # the probe runs
    diffusion = make_diffusion([n_steps])
    calls = []
    eng_cls = sub("engine").Engine
    real = eng_cls.sample_loop
    monkeypatch.setattr(eng_cls, "sample_loop", lambda self, *a, **k: (calls.append(self.precision), real(self, *a, **k))[1])
    y = {"mask": torch.ones(B, 1, 1, T, dtype=torch.bool, device=DEV), "lengths": torch.full((B,), T),
         "text_embed": tt(enc), "text_scale": tt(sc)}
    noise = rng.standard_normal((n_steps + 1,) + shape).astype(np.float32)
    diffusion.injected_noise = tt(noise[1:])
    run = lambda: diffusion.p_sample_loop(wrapped, shape, noise=tt(noise[0]), clip_denoised=False,
                                          model_kwargs={"y": y, "obs_x0": tt(obs), "obs_mask": tt(m)})
    if expect == "pinned":
        with pytest.raises(N.RangeError, match="left the f16 range"):
            run()
        assert not calls, "the chain was started although the probe evaluation left the f16 range"
        return
    final = run().cpu().numpy()
    assert calls == ["bf16x6"], calls            # the probe switched BEFORE the chain; no f16x3 step was spent
    assert model._engine.precision == "bf16x6" and np.isfinite(final).all()
    got = call()                                 # the module stays on the unrestricted mode
    assert ok("unet_real_scale.rel_l2", rel_l2(got, want), 2e-5, precision="bf16x6"), rel_l2(got, want)
    sch = do.Schedule(do.named_betas("cosine", 1000), do.space_timesteps(1000, [n_steps]))
    xo = noise[0]
    for k, i in enumerate(range(n_steps - 1, -1, -1)):
        hat, _, _ = oracle.forward_cfg(xo, np.full(B, sch.timestep_map[i]), enc, sc, obs, m)
        xo, _ = do.step_update(sch, i, xo, hat, noise[1 + k])
    assert ok("unet_real_scale.chain", rel_l2(final, xo), 1e-4, precision="bf16x6"), rel_l2(final, xo)


@pytest.mark.parametrize("precision", UNET_PRECISIONS)
def test_unet_xl_geometry_vs_reference(cases, precision):
    """The released geometry (configs/model.py motion_unet_adagn_xl: dim 512 x mults (2,2,2,2) = 1024 channels, 128 per
    GroupNorm group) — forward (cond / uncond / CFG) and input-VJP vs the REAL reference's CPU outputs
    (tests/golden/make_golden_unet_xl.py; rounds 1-3 compared this geometry with the torch port only)."""
    mu = sub("utils.model_util")
    case = cases.UNET_XL_CASE
    inp = cases.make_unet_vjp_inputs(case)
    g = load_golden("unet_xl")
    assert np.array_equal(g["fingerprint"], cases.fingerprint(inp))
    args = SimpleNamespace(dataset="humanml", arch="unet", keyframe_conditioned=True, dim_mults=case["dim_mults"],
                           cond_mask_prob=0.1)
    model, _ = mu.create_model_and_diffusion(args, None)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert sorted(shapes) == list(g["names"]), "state-dict names differ from the reference's MDM_UNET"
    mu.load_model_wo_clip(model, weights.to_torch(weights.fill_like(shapes, case["weight_seed"])) |
                          {k: v for k, v in model.state_dict().items() if k.endswith(".pe")})
    model = model.to(DEV).eval()
    model.native_precision = precision
    net = sub("model.cfg_sampler").ClassifierFreeSampleModel(model)
    x, t = tt(inp["x"]), tt(inp["t"])
    kw = dict(obs_x0=tt(inp["obs_x0"]), obs_mask=tt(inp["obs_mask"]))
    y = {"text_embed": tt(inp["enc_text"])}
    with torch.no_grad():   # plain forward (split-K at the coarse levels, no stash)
        oc = model(x, t, y=y, **kw).cpu().numpy()
        ou = model(x, t, y=dict(y, uncond=True), **kw).cpu().numpy()
        cfg = net(x, t, y=dict(y, text_scale=tt(inp["text_scale"])), **kw).cpu().numpy()
    for mine, key in ((oc, "out_cond"), (ou, "out_uncond"), (cfg, "out_cfg")):
        assert np.isfinite(mine).all()
        assert ok("unet_xl.max_abs", max_abs(mine, g[key]), 2e-4) and ok("unet_xl.rel_l2", rel_l2(mine, g[key]), 2e-5), \
            (key, max_abs(mine, g[key]), rel_l2(mine, g[key]))
    z = tt(inp["x"]).requires_grad_(True)
    with torch.enable_grad():
        out = net(z, t, y=dict(y, text_scale=tt(inp["text_scale"])), **kw)
        got, = torch.autograd.grad((out * tt(inp["gout"])).sum(), z)
    assert ok("unet_xl.stash_fwd", rel_l2(out.detach().cpu().numpy(), g["out_cfg"]), 2e-5)
    got = got.cpu().numpy()
    assert float(np.abs(got[inp["obs_mask"]]).max()) == 0.0          # observed entries are replaced by obs_x0
    assert ok("unet_xl.vjp", rel_l2(got, g["gx"]), 5e-5), rel_l2(got, g["gx"])
    # VERDICT r5 task 1c: against the FLOAT64 evaluation of the reference (make_golden_unet_xl.py) every precision mode must sit
    # within 2 x the distance the reference's own fp32 arithmetic has from it — forward and input-VJP, whole tensor AND per frame
    # row of the gradient (round 5's method: a defect confined to a few rows hides behind a whole-tensor norm)
    fwd = out.detach().cpu().numpy()
    d_fwd, d_vjp = rel_l2(fwd, g["out_cfg_f64"]), rel_l2(got, g["gx_f64"])
    r_fwd, r_vjp = rel_l2(g["out_cfg"], g["out_cfg_f64"]), rel_l2(g["gx"], g["gx_f64"])
    rows = lambda a, b: np.sqrt(((a.astype(np.float64) - b) ** 2).sum(axis=(1, 2)) / (b.astype(np.float64) ** 2).sum(axis=(1, 2)).mean())
    worst, worst_ref = float(rows(got, g["gx_f64"]).max()), float(rows(g["gx"], g["gx_f64"]).max())
    print(json_line({"case": "unet_xl", "precision": precision, "fwd_vs_f64": d_fwd, "reference_fwd_vs_f64": r_fwd, "vjp_vs_f64": d_vjp,
                     "reference_vjp_vs_f64": r_vjp, "vjp_worst_frame_row": worst, "reference_worst_frame_row": worst_ref}))
    assert ok("unet_xl.fwd_vs_f64", d_fwd, 2e-5) and d_fwd <= 2.0 * r_fwd + 2e-7, (d_fwd, r_fwd)
    assert ok("unet_xl.vjp_vs_f64", d_vjp, 5e-5) and d_vjp <= 2.0 * r_vjp + 2e-7, (d_vjp, r_vjp)
    assert worst <= 2.5 * worst_ref + 1e-6, (worst, worst_ref)


@pytest.mark.parametrize("B,T,keyframe,cfg", [(3, 100, True, False), (1, 224, False, True), (2, 33, True, True)])
def test_unet_vjp_other_configs_vs_torch_port(cases, B, T, keyframe, cfg):
    """Other geometries of the U-Net input-VJP (frame counts, no keyframe channels -> 263-channel input, with / without
    CFG) vs torch autograd through the CPU port (oracle/torch_cpu_port.py, itself pinned to the reference's outputs)."""
    from oracle.torch_cpu_port import TorchCpuUNET
    mu = sub("utils.model_util")
    args = SimpleNamespace(dataset="humanml", arch="unet", keyframe_conditioned=keyframe, dim_mults=(1, 1, 1, 1),
                           cond_mask_prob=0.1)
    model, _ = mu.create_model_and_diffusion(args, None)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    mu.load_model_wo_clip(model, weights.to_torch(weights.fill_like(shapes, 77)) |
                          {k: v for k, v in model.state_dict().items() if k.endswith(".pe")})
    model = model.to(DEV).eval()
    port = TorchCpuUNET({k: v.detach().cpu() for k, v in model.state_dict().items()})
    rng = np.random.default_rng(5000 + 13 * B + T)
    shape = (B, 263, 1, T)
    x, gout = (rng.standard_normal(shape).astype(np.float32) for _ in range(2))
    obs = rng.standard_normal(shape).astype(np.float32)
    m = rng.random(shape) < 0.2
    t = rng.integers(0, 1000, B)
    enc = rng.standard_normal((B, 512)).astype(np.float32)
    scale = np.linspace(0.5, 2.5, B).astype(np.float32)
    tc = torch.from_numpy
    okw = dict(obs_x0=tc(obs), obs_mask=tc(m)) if keyframe else {}
    zc = tc(x).clone().requires_grad_(True)
    with torch.enable_grad():
        oc = port.forward_impl(zc, tc(t), tc(enc), False, **okw)
        if cfg:
            ou = port.forward_impl(zc, tc(t), tc(enc), True, **okw)
            oc = ou + tc(scale).view(-1, 1, 1, 1) * (oc - ou)
        want, = torch.autograd.grad((oc * tc(gout)).sum(), zc)
    net = sub("model.cfg_sampler").ClassifierFreeSampleModel(model) if cfg else model
    y = {"text_embed": tt(enc)}
    if cfg:
        y["text_scale"] = tt(scale)
    gkw = dict(obs_x0=tt(obs), obs_mask=tt(m)) if keyframe else {}
    z = tt(x).requires_grad_(True)
    with torch.enable_grad():
        out = net(z, tt(t), y=y, **gkw)
        got, = torch.autograd.grad((out * tt(gout)).sum(), z)
    assert ok("unet_vjp_other_configs_vs_torch_port.rel_l2.0", rel_l2(out.detach().cpu().numpy(), oc.detach().numpy()), 2e-5)
    assert ok("unet_vjp_other_configs_vs_torch_port.rel_l2.1", rel_l2(got.cpu().numpy(), want.numpy()), 5e-5), rel_l2(got.cpu().numpy(), want.numpy())


@pytest.mark.parametrize("precision", UNET_PRECISIONS)
def test_unet_recon_guidance_chain_vs_reference(cases, precision):
    """p_sample_loop with imputation AND reconstruction guidance through the native MDM_UNET vs the real reference."""
    cc = cases.UNET_RECON_CHAIN
    ci = cases.make_unet_chain_inputs(cc)
    g = load_golden("unet_recon_chain")
    assert np.array_equal(g["fingerprint"], cases.fingerprint(ci))
    model, _ = make_unet(cases, precision)
    wrapped = sub("model.cfg_sampler").ClassifierFreeSampleModel(model)
    diffusion = make_diffusion(cc["respacing"])
    obs_mask = tt(ci["obs_mask"])
    y = {"mask": tt(ci["len_mask"]), "lengths": tt(ci["lengths"]), "text_embed": tt(ci["enc_text"]),
         "text_scale": tt(ci["text_scale"]), "inpainting_mask": obs_mask, "inpainted_motion": tt(ci["x0"]),
         "imputate": True, "stop_imputation_at": cc["stop_imputation_at"], "replacement_distribution": "conditional",
         "reconstruction_guidance": True, "reconstruction_weight": cc["recon_weight"], "gradient_schedule": None,
         "stop_recguidance_at": cc["stop_recguidance_at"], "diffusion_steps": 1000}
    diffusion.injected_noise = tt(ci["noise"])
    final = diffusion.p_sample_loop(wrapped, ci["x_T"].shape, noise=tt(ci["x_T"]), clip_denoised=False,
                                    model_kwargs={"y": y, "obs_x0": tt(ci["x0"]), "obs_mask": obs_mask}).cpu().numpy()
    assert np.isfinite(final).all()
    assert ok("unet_recon_guidance_chain_vs_reference.rel_l2.0", rel_l2(final, g["final"]), 2e-4), rel_l2(final, g["final"])
    # VERDICT r5 weak #1 (2.6e-5 / 3.0e-5 after six steps, 10 x the transformer's guided chain, in BOTH precisions): the chain's own
    # sensitivity — x_T moved by one fp32 ulp — says how much of that is the problem's conditioning (GroupNorm + Mish on an
    # untrained net under guidance weight 20; profiles/r06_unet_guided_chain_attribution.md)
    moved = diffusion.p_sample_loop(wrapped, ci["x_T"].shape, noise=one_ulp_up(tt(ci["x_T"])), clip_denoised=False,
                                    model_kwargs={"y": y, "obs_x0": tt(ci["x0"]), "obs_mask": obs_mask}).cpu().numpy()
    sens = rel_l2(moved, final)
    print(json_line({"case": "unet_recon_chain", "precision": precision, "vs_reference": rel_l2(final, g["final"]), "one_ulp_of_x_T": sens}))
    assert rel_l2(final, g["final"]) <= 25.0 * sens + 2e-6, (rel_l2(final, g["final"]), sens)


def unet_long_setup(cases, name, precision):
    """The released U-Net geometry on the native engine + the call the reference made for tests/golden/<name>.npz
    (make_golden_unet_long.py): p_sample_loop through ClassifierFreeSampleModel(MDM_UNET), keyframe conditioning (obs_x0 /
    obs_mask), imputation + reconstruction guidance (weight 20) on every step, injected noise."""
    mu = sub("utils.model_util")
    case = cases.UNET_LONG_CASES[name]
    inp = cases.make_unet_long_inputs(case)
    g = load_golden(name)
    assert np.array_equal(g["fingerprint"], cases.fingerprint(inp)), f"inputs of {name} drifted"
    args = SimpleNamespace(dataset="humanml", arch="unet", keyframe_conditioned=True, dim_mults=case["dim_mults"], cond_mask_prob=0.1)
    model, _ = mu.create_model_and_diffusion(args, None)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert sorted(shapes) == list(g["names"]), "state-dict names differ from the reference's MDM_UNET"
    mu.load_model_wo_clip(model, weights.to_torch(weights.fill_like(shapes, case["weight_seed"])) |
                          {k: v for k, v in model.state_dict().items() if k.endswith(".pe")})
    model = model.to(DEV).eval()
    model.native_precision = precision
    wrapped = sub("model.cfg_sampler").ClassifierFreeSampleModel(model)
    gd, rs = sub("diffusion.gaussian_diffusion"), sub("diffusion.respace")
    diffusion = rs.SpacedDiffusion(rs.space_timesteps(1000, case.get("respacing") or [1000]),
                                   gd.DiffusionConfig(betas=gd.get_named_beta_schedule("cosine", 1000)))
    n = diffusion.num_timesteps
    assert n == cases.unet_long_steps(case)
    if name not in _BIG_NOISE:
        noise = torch.empty((n,) + inp["draw0"].shape, dtype=torch.float32, device=DEV)
        for k in range(n):
            noise[k].copy_(torch.from_numpy(cases.unet_long_draw(case, 1 + k)))
        _BIG_NOISE.clear()
        _BIG_NOISE[name] = noise
    diffusion.injected_noise = _BIG_NOISE[name]
    obs_mask = tt(inp["obs_mask"])
    y = {"mask": tt(inp["len_mask"]), "lengths": tt(inp["lengths"]), "text_embed": tt(inp["enc_text"]),
         "text_scale": tt(inp["text_scale"]), "inpainting_mask": obs_mask, "inpainted_motion": tt(inp["x0"]),
         "imputate": True, "stop_imputation_at": case["stop_imputation_at"], "replacement_distribution": "conditional",
         "reconstruction_guidance": True, "reconstruction_weight": case["recon_weight"], "gradient_schedule": None,
         "stop_recguidance_at": case["stop_recguidance_at"], "diffusion_steps": 1000}
    kw = dict(noise=tt(inp["draw0"]), clip_denoised=False, model_kwargs={"y": y, "obs_x0": tt(inp["x0"]), "obs_mask": obs_mask})
    return case, inp, g, wrapped, diffusion, kw


def one_ulp_up(x):
    return torch.nextafter(x, torch.full_like(x, float("inf")))


@pytest.mark.parametrize("precision", UNET_PRECISIONS)
def test_unet_baseline_batch_guided_chain_vs_reference(cases, precision):
    """VERDICT r5 task 1b: the U-Net at the bench's batch — released geometry (1024 channels), B=32, ragged lengths, all 100 steps
    of p_sample_loop on 'ddim100' (what sample/conditional_synthesis.py calls) with keyframe conditioning, imputation and
    reconstruction guidance on every step (forward AND input-VJP of csrc/unet.hip at M = 64 x 224 rows per evaluation) — vs the
    REAL reference's CPU chain (make_golden_unet_long.py big_unet; model/mdm_unet.py:561-849, gaussian_diffusion.py:405-435).

    What the comparison can mean (profiles/r06_unet_guided_chain_attribution.md): on these random weights the guided chain is
    CHAOTIC over its last ~40 steps — the reference's own fp32 chain ends 1.0e-1 / 3.5e-3 (samples 0 / 7) from the same chain
    run by the reference in float64, and moving x_T by ONE fp32 ulp moves this engine's own result by 2e-3 ... 1.2e-1.  So:
      * the stable window (steps <= 60, where the chain does not amplify): the usual tight bound against the reference;
      * the final samples: the distance from the reference, per sample, against that sample's OWN sensitivity — the distance
        between two runs of this engine whose x_T differ by one ulp.  Rounding differences are amplified like that ulp (measured
        ratio 1-5); a defect worth 100 x the rounding error would be amplified to 100 x the ulp's effect;
      * the two samples the reference also ran in float64: no further from the truth than 10 x the larger of the reference's own
        fp32 distance and the one-ulp sensitivity."""
    name = "big_unet"
    case, inp, g, wrapped, diffusion, kw = unet_long_setup(cases, name, precision)
    at = {int(i): k for k, i in enumerate(g["dump_at"])}
    last, stable = None, 0.0
    for i, out in enumerate(diffusion.p_sample_loop_progressive(wrapped, inp["draw0"].shape, **kw)):
        last = out["sample"]
        if i in at and i < 60:
            stable = max(stable, rel_l2(last[:1].cpu().numpy(), g["dumps"][at[i]]))
    final = last.cpu().numpy()
    assert np.isfinite(final).all()
    assert ok("unet_big_chain.stable_window", stable, 2e-5), stable
    moved = diffusion.p_sample_loop(wrapped, inp["draw0"].shape, **dict(kw, noise=one_ulp_up(kw["noise"]))).cpu().numpy()
    keep = list(case["keep"])
    report = {}
    for i, k in enumerate(keep):
        err, sens = rel_l2(final[k], g["final"][i]), rel_l2(moved[k], final[k])
        report[k] = {"vs_reference": err, "one_ulp_of_x_T": sens}
        assert err <= 10.0 * sens + 1e-5, (k, err, sens)
    for j, r in enumerate(g["f64_rows"]):
        r = int(r)
        mine, ref = rel_l2(final[r], g["final_f64_rows"][j]), rel_l2(g["final"][keep.index(r)], g["final_f64_rows"][j])
        report[r].update(vs_f64=mine, reference_fp32_vs_f64=ref)
        assert mine <= 10.0 * max(ref, report[r]["one_ulp_of_x_T"]) + 1e-5, (r, mine, ref)
    print(json_line({"case": name, "precision": precision, "stable_window": stable, "final": report}))


@pytest.mark.parametrize("precision", UNET_PRECISIONS)
def test_unet_long_chain_drift_vs_reference(cases, precision):
    """VERDICT r5 task 1b: the U-Net's full chain — released geometry, B=2, ALL 1000 ancestral steps, keyframe conditioning +
    imputation + reconstruction guidance (weight 20) on every step — against the reference's fp32 chain AND the same chain run
    by the reference in float64 (make_golden_unet_long.py long_unet; both trajectories dumped every 100 steps).

    Like big_unet the guided chain on random weights amplifies rounding (profiles/r06_unet_guided_chain_attribution.md), so
    the yardstick at every dump is the larger of (a) the reference's own fp32 distance from its float64 chain there and (b) this
    engine's one-ulp-of-x_T sensitivity there: the native chain must be no further from the float64 trajectory than 10 x that
    (+1e-5), and while the reference's own distance is still below 1e-5 (nothing amplified yet) the usual tight bound holds."""
    name = "long_unet"
    case, inp, g, wrapped, diffusion, kw = unet_long_setup(cases, name, precision)
    at = {int(i): k for k, i in enumerate(g["dump_at"])}

    def trajectory(**over):
        snaps, last = {}, None
        for i, out in enumerate(diffusion.p_sample_loop_progressive(wrapped, inp["draw0"].shape, **dict(kw, **over))):
            last = out["sample"]
            if i in at:
                snaps[i] = last[:1].cpu().numpy()
        return snaps, last.cpu().numpy()

    snaps, final = trajectory()
    moved_snaps, moved = trajectory(noise=one_ulp_up(kw["noise"]))
    assert np.isfinite(final).all()
    rows, tight = [], 0.0
    for i in sorted(at):
        ref32, ref64 = g["dumps"][at[i]], g["dumps_f64"][at[i]]
        d_ref, d32, d64 = rel_l2(ref32, ref64), rel_l2(snaps[i], ref32), rel_l2(snaps[i], ref64)
        sens = rel_l2(moved_snaps[i], snaps[i])
        rows.append({"step": i, "vs_fp32": d32, "vs_f64": d64, "reference_fp32_vs_f64": d_ref, "one_ulp_of_x_T": sens})
        if d_ref < 1e-5:
            tight = max(tight, d32)
        assert d64 <= 10.0 * max(d_ref, sens) + 1e-5, rows[-1]
    d_ref = rel_l2(g["final"], g["final_f64"])
    d32, d64, sens = rel_l2(final, g["final"]), rel_l2(final, g["final_f64"]), rel_l2(moved, final)
    print(json_line({"case": name, "precision": precision, "vs_fp32": d32, "vs_f64": d64, "reference_fp32_vs_f64": d_ref,
                     "one_ulp_of_x_T": sens, "tight_window": tight, "trajectory": rows}))
    assert ok("unet_long_chain.tight_window", tight, 2e-5), tight
    assert d64 <= 10.0 * max(d_ref, sens) + 1e-5, (d64, d_ref, sens)
    assert d32 <= 10.0 * max(d_ref, sens) + 1e-5, (d32, d_ref, sens)


def test_keyframes_mask_built_on_device(cases):
    """get_keyframes_mask with device inputs: built on the device (no per-sample host loop), bit-exact vs the reference."""
    eu = sub("utils.editing_util")
    g = load_golden("keyframe_masks")
    kc = cases.KEYFRAME_CASE
    data = torch.zeros(kc["B"], 263, 1, kc["T"], device=DEV)
    lengths = torch.tensor(kc["lengths"], device=DEV)
    for i, (mode, trans, feat, nk) in enumerate(cases.KEYFRAME_MODES):
        np.random.seed(kc["seed"] + i)
        full, joint = eu.get_keyframes_mask(data, lengths, edit_mode=mode, trans_length=trans, feature_mode=feat,
                                            get_joint_mask=True, n_keyframes=nk)
        assert full.device.type == "cuda" and full.is_contiguous()
        assert np.array_equal(np.packbits(full.cpu().numpy()), g[f"full.{i}"]), (mode, trans, feat)
        assert np.array_equal(np.packbits(joint.cpu().numpy()), g[f"joint.{i}"]), (mode, trans, feat)
