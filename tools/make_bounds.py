"""tests/bounds.json from a GPU run's measured maxima (VERDICT r3 task 5a).

    python -m pytest tests -m gpu -q          # on the GPU box: writes gpurun_out/tolerances_measured.json
    python tools/make_bounds.py [measured.json] [round tag]   # here: tests/bounds.json + profiles/<tag>_tolerances.json

bound = 4 x the largest value any precision mode / parametrisation of the test produced, rounded UP to two significant
digits, and never above the hand-stated default of the call site (the round-1..3 bounds).  A key that measured exactly 0
(bitwise-equal comparisons) keeps a floor of 1e-12 so that the comparison stays a comparison.
"""
import json
import math
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
FACTOR = 4.0


def round_up(x: float) -> float:
    if x <= 0:
        return 1e-12
    e = math.floor(math.log10(x)) - 1
    return math.ceil(x / 10 ** e - 1e-9) * 10 ** e


def main():
    src = Path(sys.argv[1]) if len(sys.argv) > 1 else REPO / "gpurun_out" / "tolerances_measured.json"
    tag = sys.argv[2] if len(sys.argv) > 2 else "r04"
    measured = json.loads(src.read_text())
    bounds = {}
    for key, rec in sorted(measured.items()):
        b = min(float(rec["default"]), round_up(FACTOR * rec["max"]))
        bounds[key] = {"bound": float(f"{b:.2g}"), "measured": rec["max"], "stated_r3": rec["default"], "n": rec["n"]}
    (REPO / "tests" / "bounds.json").write_text(json.dumps(bounds, indent=1, sort_keys=True) + "\n")
    (REPO / "profiles" / f"{tag}_tolerances.json").write_text(json.dumps(
        {"rule": f"bound = min(stated, {FACTOR:g} x measured max over all precisions, rounded up to 2 digits)",
         "keys": bounds}, indent=1, sort_keys=True) + "\n")
    worst = max(bounds.values(), key=lambda v: v["measured"] / v["bound"])
    print(f"{len(bounds)} keys; tightest margin {worst}")


if __name__ == "__main__":
    main()
