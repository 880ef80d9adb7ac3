#!/usr/bin/env python3
"""PCIe-inclusive rates of the host entry points (never the bench `value`): images in pageable host memory ->
visp_*_compute_batch_host -> results in host memory."""
import sys
import tempfile
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from visioncpp_amd import synth  # noqa: E402
from visioncpp_amd.vision import Backend, Device, Model  # noqa: E402

dev = Device.init(Backend.gpu)
with tempfile.TemporaryDirectory() as td:
    da = Model.load(synth.write_gguf(Path(td) / "d.gguf", synth.SMALL, 0) and Path(td) / "d.gguf", dev)
    es = Model.load(synth.write_esrgan_gguf(Path(td) / "e.gguf", synth.ESRGAN_X4, 1), dev)
imgs = synth.images(32, 518, 518, seed=1)
da.compute_batch(imgs)
t0 = time.perf_counter()
for _ in range(5):
    da.compute_batch(imgs)
dt = (time.perf_counter() - t0) / 5
print(f"depth-anything host API: {32 / dt:.0f} images/s ({dt * 1e3:.1f} ms per 32 images; 25.8 MB in, 34.3 MB out)")
small = synth.images(16, 256, 256, seed=2)
es.upscale_batch(small)
t0 = time.perf_counter()
for _ in range(5):
    es.upscale_batch(small)
dt = (time.perf_counter() - t0) / 5
print(f"esrgan host API: {16 / dt:.0f} images/s ({dt * 1e3:.1f} ms per 16 images; 3.1 MB in, 67.1 MB out)")
with tempfile.TemporaryDirectory() as td:
    sm = Model.load(synth.write_tinyvit_gguf(Path(td) / "s.gguf", synth.TINYVIT_5M, 3), dev)
big = synth.images(4, 1024, 1024, seed=3)
big = __import__("numpy").concatenate([big] * 4)
sm.sam_encode_batch(big)
t0 = time.perf_counter()
for _ in range(5):
    sm.sam_encode_batch(big)
dt = (time.perf_counter() - t0) / 5
print(f"mobile-sam encoder host API: {16 / dt:.0f} images/s ({dt * 1e3:.1f} ms per 16 images; 50.3 MB in, 67.1 MB out)")
