#!/usr/bin/env python3
"""usage: convert_depth_anything.py model.safetensors [-o out.gguf]   (see vision.cpp_amd/convert.py)"""
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
out = a.output or str(Path(a.input).with_suffix("")) + "-F16.gguf"
print(convert.convert_depth_anything(convert.load_safetensors(a.input), out))
