"""Build recipe of libcondmdi_hip.so (hipcc, gfx950 only) and of the C oracle.

Called by ``__graft_entry__.build()``; safe to call repeatedly (mtime-based, per translation unit).
The shared library is written IN-TREE (``csrc/libcondmdi_hip.so``) so that it travels with the repo
snapshot to the GPU box; it is git-ignored.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
REPO = PKG_DIR.parent
LIB_PATH = CSRC / "libcondmdi_hip.so"
# bench-only instrumentation (ablation switches, cycle stamps, tile-tuning environment knobs) is compiled ONLY into
# this second library (-DCMDI_PROBES), which tools/ load through CMDI_PROBES_LIB=1; the product library has none of it
PROBES_LIB_PATH = CSRC / "libcondmdi_hip_probes.so"

# translation unit -> extra flags
UNITS = {
    "api_engine.hip": [],
    "api_denoiser.hip": [],
    "api_sampler.hip": [],
    "api_hooks.hip": [],
    "gemm_f32.hip": [],
    "gemm_h3.hip": [],
    "gemm_h3p.hip": [],
    # no SLP vectorisation: packed-fp32 VALU (v_pk_add_f32 / v_pk_fma_f32) beside MFMAs costs more than the two scalar
    # instructions it replaces (MI355X guide, "price of one filler beside MFMAs"); same bits either way
    "gemm_h3w.hip": ["-fno-slp-vectorize"],
    "gemm_x6.hip": [],
    "attention_f32.hip": [],
    "attention_h3.hip": [],
    "attention_bwd_f32.hip": [],
    "attention_bwd_h3.hip": [],
    "unet.hip": [],
    "clip_text.hip": [],
    "elementwise.hip": [],
    # reference evaluation order, every op rounded separately (see the header of sampler.hip)
    "sampler.hip": ["-ffp-contract=off"],
    "postprocess.hip": ["-ffp-contract=off"],
}
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-fno-gpu-rdc"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: the CondMDI engine needs ROCm's hipcc to build for gfx950")


def _deps_newer(obj: Path, src: Path) -> bool:
    if not obj.exists():
        return True
    t = obj.stat().st_mtime
    deps = [src, Path(__file__), REPO / "include" / "condmdi.h"] + list(CSRC.glob("*.hpp"))
    return any(d.stat().st_mtime > t for d in deps)


def _compile(unit: str, flags, verbose: bool, probes: bool = False) -> Path:
    src = CSRC / unit
    obj = CSRC / ("build_probes" if probes else "build") / (src.stem + ".o")
    obj.parent.mkdir(exist_ok=True)
    if _deps_newer(obj, src):
        cmd = [_hipcc(), *COMMON, *flags, *(["-DCMDI_PROBES"] if probes else []), "-I", str(CSRC), "-c", str(src),
               "-o", str(obj)]
        if verbose:
            print("[condmdi build]", " ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed for {unit}:\n{res.stdout}\n{res.stderr}")
        if verbose and res.stderr.strip():
            print(res.stderr, file=sys.stderr)
    return obj


def build_native(verbose: bool = False, force: bool = False, probes: bool = False) -> Path:
    """Compile every HIP translation unit for gfx950 and link libcondmdi_hip.so (probes=True: the instrumented
    libcondmdi_hip_probes.so for tools/ instead)."""
    bdir = CSRC / ("build_probes" if probes else "build")
    lib = PROBES_LIB_PATH if probes else LIB_PATH
    if force and bdir.exists():
        shutil.rmtree(bdir)
    with ThreadPoolExecutor(max_workers=min(len(UNITS), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(lambda kv: _compile(kv[0], kv[1], verbose, probes), UNITS.items()))
    if force or not lib.exists() or any(o.stat().st_mtime > lib.stat().st_mtime for o in objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc",
               *map(str, objs), "-o", str(lib)]
        if verbose:
            print("[condmdi build]", " ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    return lib


if __name__ == "__main__":
    print(build_native(verbose=True, force="--force" in sys.argv, probes="--probes" in sys.argv))
