"""Where the persistent in_proj GEMM's extra L2 traffic at M >= 100k comes from (VERDICT r4 task 9).

    python tools/inproj_l2_pmc.py            # driver: three rocprofv3 --pmc passes of itself (--child), prints a table
Child: the in_proj GEMM (N = 1536, K = 512, split-rows output) at C4's M = 100,864 on the tiled kernel (tile 8) and on the
persistent kernel (tile 50), three launches each.  Counters per launch: TCC_HIT_sum / TCC_MISS_sum (L2 requests),
FETCH_SIZE (L2 -> fabric reads, KiB; doubled per the gfx950 correction of MI355X_MICROARCH.md), WRITE_SIZE."""
import csv, importlib, os, shutil, subprocess, sys, tempfile
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
M, N, K = 2 * 256 * 197, 1536, 512

def child():
    import torch
    sys.path.insert(0, str(REPO))
    eng = importlib.import_module("diffusion-motion-inbetweening_amd.engine")
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    a = eng.split_f16(torch.randn(M, K, generator=g).to(dev)); w = eng.split_f16((torch.randn(N, K, generator=g) * 0.05).to(dev))
    b = torch.randn(N, generator=g).to(dev)
    out = torch.empty(M, 2 * N, device=dev, dtype=torch.float16)
    for tile in (8, 50):
        for _ in range(4):
            eng.gemm_h3(a, w, b, tile=tile, epi=0, split_out=True, out=out)
    torch.cuda.synchronize()

def main():
    if "--child" in sys.argv:
        return child()
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    tmp = tempfile.mkdtemp(prefix="cmdi_l2_", dir="/tmp")
    env = dict(os.environ, TMPDIR=tmp)
    vals = {}
    for i, ctrs in enumerate((["TCC_HIT_sum", "TCC_MISS_sum"], ["FETCH_SIZE"], ["WRITE_SIZE"], ["TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum"])):
        out_dir = os.path.join(tmp, f"p{i}")
        r = subprocess.run([exe, "--pmc", *ctrs, "--output-format", "csv", "-d", out_dir, "-o", "p", "--", sys.executable, __file__, "--child"],
                           cwd=tmp, env=env, capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            print(f"pass {ctrs}: rocprofv3 rc={r.returncode}: {r.stderr[-300:]}")
            continue
        for dp, _, fs in os.walk(out_dir):
            for f in fs:
                if f.endswith("counter_collection.csv"):
                    with open(os.path.join(dp, f), newline="") as fh:
                        for row in csv.DictReader(fh):
                            kn = row["Kernel_Name"]
                            if "gemm_h3" not in kn:
                                continue
                            fam = "persistent (gemm_h3p)" if "gemm_h3p" in kn else "tiled (gemm_h3, 128x128)"
                            vals.setdefault((fam, row["Counter_Name"]), []).append(float(row["Counter_Value"]))
    alg = {"A": 4.0 * M * K, "W": 4.0 * N * K, "C": 4.0 * M * N}
    print(f"in_proj at M = {M}: algorithmic MB  A {alg['A'] / 1e6:.0f}  W {alg['W'] / 1e6:.1f}  C {alg['C'] / 1e6:.0f}")
    for fam in ("tiled (gemm_h3, 128x128)", "persistent (gemm_h3p)"):
        g = lambda c: (sum(vals[(fam, c)][1:]) / max(1, len(vals[(fam, c)][1:]))) if (fam, c) in vals else float("nan")   # first launch = warm-up
        hit, miss = g("TCC_HIT_sum"), g("TCC_MISS_sum")
        fetch, write = 2.0 * g("FETCH_SIZE") * 1024, g("WRITE_SIZE") * 1024
        print(f"{fam:>26}: L2 hit rate {hit / (hit + miss):.3f} (hits {hit:.3e}, misses {miss:.3e}) | fabric reads {fetch / 1e6:.0f} MB = {fetch / (alg['A'] + alg['W']):.2f} x (A + W) | "
              f"fabric writes {write / 1e6:.0f} MB = {write / alg['C']:.2f} x C | EA read requests {g('TCC_EA0_RDREQ_sum'):.3e}, write requests {g('TCC_EA0_WRREQ_sum'):.3e}")
    shutil.rmtree(tmp, ignore_errors=True)

if __name__ == "__main__":
    main()
