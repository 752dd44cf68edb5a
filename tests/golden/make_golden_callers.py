"""tests/golden/caller_<edit|conditional_synthesis|synthesize>.npz: the EXACT arguments the reference's own sample scripts
pass to ``diffusion.p_sample_loop`` (sample/edit.py:133-146, sample/conditional_synthesis.py:214-227,
sample/synthesize.py:136-149) — captured while each script's ``main()`` runs, unchanged, on the reference's OWN modules —
together with the REAL reference sampler's output for that call on the ``[10]`` respacing with an injected noise stream
(tests/helpers/run_reference_caller.py, CALLER_MODE=reference).  Runs only where /root/reference exists (~2 minutes).

    python tests/golden/make_golden_callers.py [name ...]
"""
import shutil
import sys
import tempfile
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent / "helpers"))
sys.path.insert(0, str(HERE))
import caller_setup  # noqa: E402
import cases  # noqa: E402


def main():
    for name in sys.argv[1:] or list(cases.CALLER_CASES):
        case = cases.CALLER_CASES[name]
        with tempfile.TemporaryDirectory() as tmp:
            tmp = Path(tmp)
            caller_setup.write_checkpoint(tmp, case["model_args"], case["weight_seed"])
            res = caller_setup.run_script(tmp, case, "reference", cases.CALLER_SAMPLES)
            (call,) = res["calls"]
            print(name, call, flush=True)
            shutil.copy(tmp / "recorded_call.npz", HERE / f"caller_{name}.npz")


if __name__ == "__main__":
    main()
