"""Error of split-precision product schemes against float64, on one full MDM denoiser evaluation (CPU, numpy).

Replaces every `x @ W.T` of oracle/mdm_oracle.py by an emulation of the scheme (operands rounded to the narrow
type, products and sums in float64 — i.e. ONLY the operand-representation error of the scheme is measured; on the
GPU the products are accumulated in fp32 like the fp32 kernels do).  Justifies the choice of
f16x3 = hi·hi + (hi·lo' + lo'·hi)·2^-11 with lo' = f16((x - hi)·2^11) in csrc/gemm_h3.hpp:

    f32 (numpy)   4.9e-07 rel-L2     f16x3          2.6e-07     f16x3 unscaled lo  1.1e-06
    bf16x3        6.5e-06            bf16x6         2.1e-07     (bf16x6 costs twice the MFMAs of f16x3)
"""
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
for p in (REPO, REPO / "tests" / "golden", REPO / "tests"):
    sys.path.insert(0, str(p))
import cases  # noqa: E402
from oracle import mdm_oracle as mo, weights  # noqa: E402

F32 = np.float32


def bf16(x):
    u = np.ascontiguousarray(x, dtype=F32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(F32)


def f16(x):
    return x.astype(np.float16).astype(F32)


def mm(a, b):
    return a.astype(np.float64) @ b.astype(np.float64)


def make_linear(mode):
    def linear(x, w, b=None):
        if mode == "f32":
            y = x @ w.T
        elif mode == "f64":
            y = mm(x, w.T).astype(F32)
        elif mode == "bf16x3":
            xh, wh = bf16(x), bf16(w)
            xl, wl = bf16(x - xh), bf16(w - wh)
            y = (mm(xh, wh.T) + mm(xh, wl.T) + mm(xl, wh.T)).astype(F32)
        elif mode == "bf16x6":
            xh = bf16(x); r = x - xh; xm = bf16(r); xl = bf16(r - xm)
            wh = bf16(w); r = w - wh; wm = bf16(r); wl = bf16(r - wm)
            y = (mm(xh, wh.T) + mm(xh, wm.T) + mm(xm, wh.T) + mm(xm, wm.T) + mm(xh, wl.T) + mm(xl, wh.T)).astype(F32)
        elif mode == "f16x3":
            xh, wh = f16(x), f16(w)
            xl, wl = f16((x - xh) * F32(2048)), f16((w - wh) * F32(2048))
            y = (mm(xh, wh.T) + (mm(xh, wl.T) + mm(xl, wh.T)) / 2048).astype(F32)
        elif mode == "f16x3_unscaled":
            xh, wh = f16(x), f16(w)
            xl, wl = f16(x - xh), f16(w - wh)
            y = (mm(xh, wh.T) + mm(xh, wl.T) + mm(xl, wh.T)).astype(F32)
        else:
            raise ValueError(mode)
        return y if b is None else (y + b).astype(F32)
    return linear


def main():
    case = cases.CASES["fwd_text"]
    inp = cases.make_inputs(case)
    m = mo.MDMOracle(weights.make_state_dict(case["weight_seed"], text=True))
    res = {}
    for mode in ("f64", "f32", "bf16x3", "bf16x6", "f16x3", "f16x3_unscaled"):
        mo._linear = make_linear(mode)
        res[mode] = m.forward(inp["x"], inp["t"], enc_text=inp.get("enc_text"))
    ref = res["f64"].astype(np.float64)
    for mode, o in res.items():
        d = o.astype(np.float64) - ref
        print(f"{mode:16s} max-abs {np.abs(d).max():.3e}   rel-L2 {np.linalg.norm(d) / np.linalg.norm(ref):.3e}")


if __name__ == "__main__":
    main()
