"""How close can this checkpoint's activations come to the f16 range of the default precision?  Weight-only bounds
(diffusion-motion-inbetweening_amd/utils/range_certificate.py) for an MDM trans_enc checkpoint: `certified` means no input inside
the assumptions can trip the range guard, i.e. the bf16x6 fallback (1.7-2.1 x the cost) is never taken for plain sampling.

usage: python tools/range_certificate.py <model.pt | synthetic> [--x-bound 16] [--text-l2-bound 32] [--frames 196]
  <model.pt>   a reference checkpoint (torch.load -> state dict, 'model' / 'state_dict' wrappers accepted)
  synthetic    the seeded random weights the tests and bench.py use (oracle-free: tests/golden/cases.py is not needed)
"""
import argparse
import importlib
import json
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("checkpoint")
    ap.add_argument("--x-bound", type=float, default=16.0)
    ap.add_argument("--text-l2-bound", type=float, default=32.0)
    ap.add_argument("--frames", type=int, default=196)
    args = ap.parse_args()
    rc = importlib.import_module("diffusion-motion-inbetweening_amd.utils.range_certificate")
    import torch
    if args.checkpoint == "synthetic":
        mu = importlib.import_module("diffusion-motion-inbetweening_amd.utils.model_util")
        from types import SimpleNamespace
        model, _ = mu.create_model_and_diffusion(SimpleNamespace(dataset="humanml", arch="trans_enc"), None)
        sd = model.state_dict()
    else:
        sd = torch.load(args.checkpoint, map_location="cpu")
        for k in ("model", "state_dict", "model_avg"):
            if isinstance(sd, dict) and k in sd and isinstance(sd[k], dict):
                sd = sd[k]
    sd = {k: v for k, v in sd.items() if not k.startswith("clip_model.")}
    print(json.dumps(rc.trans_enc_range_certificate(sd, x_bound=args.x_bound, text_l2_bound=args.text_l2_bound,
                                                     n_frames=args.frames), indent=1))


if __name__ == "__main__":
    main()
