"""Per-launch timing of one Depth-Anything step (VISP_TIMING_DETAIL=1: every launch group is its own row), batch 32 at 518x518,
direct launches on one stream. Usage: VISP_TIMING_DETAIL=1 python tools/dpt_launches.py [batch]"""
import os
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
os.environ.setdefault("VISP_TIMING_DETAIL", "1")
from __graft_entry__ import load_package  # noqa: E402

load_package()
import numpy as np  # noqa: E402
import torch  # noqa: E402
from visioncpp_amd import synth, vision  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
W = H = 518
path = Path(tempfile.gettempdir()) / "visp_dpt_launches.gguf"
synth.write_gguf(path, synth.SMALL, seed=0)
dev = vision.Device.init(index=0)
model = vision.Model.load(path, dev, vision.Arch.depth_anything)
imgs = synth.images(min(B, 8), W, H, seed=1)
imgs = np.concatenate([imgs] * ((B + len(imgs) - 1) // len(imgs)))[:B]
rgb = torch.from_numpy(imgs).cuda()
out = torch.empty((B, H, W), dtype=torch.float32, device="cuda")
stream = torch.cuda.Stream().cuda_stream
model.reserve(B, W, H)
for _ in range(2):
    model.compute_batch_device(rgb.data_ptr(), B, W, H, out.data_ptr(), None, stream)
torch.cuda.synchronize()
model.enable_timing(True)
acc = {}
reps = 5
for _ in range(reps):
    model.compute_batch_device(rgb.data_ptr(), B, W, H, out.data_ptr(), None, stream)
    torch.cuda.synchronize()
    for t in model.read_timing():
        a = acc.setdefault(t["name"], dict(t, ms=0.0))
        a["ms"] += t["ms"] / reps
tot = 0.0
for name, a in acc.items():
    base = name.split("#")[0]
    if base in ("block", "attention", "block_qkv0"):
        continue
    tot += a["ms"]
    print(f"{name:22s} {a['ms'] * 1e3:8.1f} us  {a['flops'] / max(a['ms'], 1e-9) / 1e9:8.1f} TFLOP/s  {a['bytes'] / max(a['ms'], 1e-9) / 1e6:8.1f} GB/s")
print(f"non-encoder total {tot:.3f} ms")
