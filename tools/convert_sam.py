#!/usr/bin/env python3
"""usage: convert_sam.py mobile_sam.safetensors [-o out.gguf]   MobileSAM (TinyViT-5M + SAM decoder) -> GGUF, see
vision.cpp_amd/convert.py. `.pt` files are read with torch.load(weights_only=True): plain state dicts only."""
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

    sd = {k: v.float().numpy() for k, v in torch.load(a.input, map_location="cpu", weights_only=True).items()}
out = a.output or str(Path(a.input).with_suffix("")) + "-F16.gguf"
print(convert.convert_sam(sd, out))
