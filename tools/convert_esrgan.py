#!/usr/bin/env python3
"""usage: convert_esrgan.py model.safetensors [-o out.gguf]   RRDBNet (ESRGAN / Real-ESRGAN) -> GGUF, see
vision.cpp_amd/convert.py. `.pth` files are read with torch.load(weights_only=True) (params / params_ema unwrapped)."""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from visioncpp_amd import convert  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("input")
ap.add_argument("--output", "-o", default=None)
a = ap.parse_args()
if a.input.endswith((".safetensors", ".safetensor")):
    sd = convert.load_safetensors(a.input)
else:
    import torch

    sd = torch.load(a.input, map_location="cpu", weights_only=True)
    for wrap in ("params_ema", "params"):
        if wrap in sd:
            sd = sd[wrap]
            break
    sd = {k: v.float().numpy() for k, v in sd.items()}
out = a.output or str(Path(a.input).with_suffix("")) + "-F16.gguf"
print(convert.convert_esrgan(sd, out))
