"""tests/bounds.json from a GPU run's measured maxima (VERDICT r3 task 5a; r4 task 2: per precision, and a ratchet).

    python -m pytest tests -m gpu -q          # on the GPU box: writes gpurun_out/tolerances_measured.json
    python tools/make_bounds.py [measured.json] [round tag] [--allow-raise]
                                              # here: tests/bounds.json + profiles/<tag>_tolerances.json

Keys are "<comparison>@<precision>" for tests parametrised by precision (tests/conftest.py::ok), plain otherwise.
bound = 4 x the measured maximum of THAT key, rounded UP to two significant digits, never above the hand-stated default
of the call site (the round-1..3 bounds).  A key that measured exactly 0 (bitwise-equal comparisons) keeps a floor of
1e-12 so that the comparison stays a comparison.

Ratchet: a bound already in tests/bounds.json (for a new per-precision key: the legacy precision-blind bound it replaces)
is never RAISED by a new measurement unless --allow-raise is given; every key whose bound would rise is printed, and
without the flag the old bound stays in force (so the regression shows up as a failing test, not as a wider bound).
A NaN measurement is refused outright.  Keys present in bounds.json but absent from the measurement are kept.
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


def derive(measured: dict, old: dict, allow_raise: bool = False):
    """(new bounds table, [(key, old bound, wanted bound)] of the keys a plain re-derivation would have raised)."""
    bounds, raised = dict(old), []
    for key, rec in sorted(measured.items()):
        if math.isnan(float(rec["max"])):
            raise SystemExit(f"{key}: measured NaN — refusing to derive a bound from it")
        want = float(f"{min(float(rec['default']), round_up(FACTOR * rec['max'])):.2g}")
        prev_key = key if key in old else key.split("@")[0]
        prev = float(old[prev_key]["bound"]) if prev_key in old else None
        b = want
        if prev is not None and want > prev:
            raised.append((key, prev, want))
            if not allow_raise:
                b = prev
        bounds[key] = {"bound": b, "measured": rec["max"], "stated": rec["default"], "n": rec["n"]}
    # a legacy precision-blind key is dropped once every precision of it has its own entry
    for key in [k for k in bounds if "@" not in k]:
        if any(k.startswith(key + "@") for k in bounds) and key not in measured:
            del bounds[key]
    return bounds, raised


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    allow = "--allow-raise" in sys.argv
    src = Path(args[0]) if args else REPO / "gpurun_out" / "tolerances_measured.json"
    tag = args[1] if len(args) > 1 else "r05"
    path = REPO / "tests" / "bounds.json"
    old = json.loads(path.read_text()) if path.exists() else {}
    bounds, raised = derive(json.loads(src.read_text()), old, allow)
    for key, prev, want in raised:
        print(f"{'RAISED' if allow else 'KEPT (would rise)'}: {key}: {prev:g} -> {want:g}")
    if raised and not allow:
        print(f"{len(raised)} bound(s) would rise; kept the old ones (pass --allow-raise to accept, and say why in DESIGN.md)")
    path.write_text(json.dumps(bounds, indent=1, sort_keys=True) + "\n")
    (REPO / "profiles" / f"{tag}_tolerances.json").write_text(json.dumps(
        {"rule": f"bound = min(stated, {FACTOR:g} x measured max of the key (per precision), rounded up to 2 digits); "
                 "never raised without --allow-raise", "keys": bounds}, indent=1, sort_keys=True) + "\n")
    worst = max(bounds.items(), key=lambda kv: kv[1]["measured"] / kv[1]["bound"])
    print(f"{len(bounds)} keys; tightest margin {worst}")


if __name__ == "__main__":
    main()
