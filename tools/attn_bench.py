import importlib, sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
eng = importlib.import_module("diffusion-motion-inbetweening_amd.engine")
dev = torch.device("cuda:0")
for n_seq in (64, 512):
    S, H = 197, 4
    qkv = torch.randn(n_seq * S, 3 * H * 128, device=dev)
    for _ in range(3): eng.attention_fwd(qkv, n_seq, S, H)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): eng.attention_fwd(qkv, n_seq, S, H)
    b.record(); torch.cuda.synchronize()
    dt = a.elapsed_time(b) / 20 * 1e-3
    fl = n_seq * 4.0 * S * S * 512
    print(f"n_seq={n_seq}: {dt*1e6:.1f} us  {fl/dt/1e12:.1f} TFLOP/s (algorithmic 4*S^2*d per sequence)")
