#!/usr/bin/env python3
"""Latency of the reference's own entry point -- visp_model_compute on ONE 518 x 518 image from host memory (resize + normalise + forward + normalise +
resize back, blocking) -- and of the batch entry at batch 1 with inputs resident in HBM: what a caller that does not batch gets."""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from visioncpp_amd import synth, vision  # noqa: E402

path = Path(tempfile.gettempdir()) / "lat_da.gguf"
synth.write_gguf(path, synth.SMALL, seed=0)
dev = vision.Device.init(vision.Backend.gpu)
model = vision.Model.load(path, dev, vision.Arch.depth_anything)
img = synth.images(1, 518, 518, seed=1)[0]
for _ in range(5):
    model.compute(img)
t0 = time.perf_counter()
n = 50
for _ in range(n):
    model.compute(img)
print(f"visp_model_compute, one 518x518 rgb_u8 image from host memory: {(time.perf_counter() - t0) / n * 1e3:.3f} ms")
rgb = vision.DeviceBuffer.from_numpy(img[None])
out = vision.DeviceBuffer(518 * 518 * 4)
for g in (False, True):
    model.use_graph(g)
    for _ in range(5):
        model.compute_batch_device(rgb.ptr, 1, 518, 518, out.ptr)
    t0 = time.perf_counter()
    for _ in range(n):
        model.compute_batch_device(rgb.ptr, 1, 518, 518, out.ptr)
    print(f"batch entry at batch 1, resident, blocking, hipGraph={int(g)}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms")
for sched in (0, 1):
    model.set_schedule(sched)
    model.use_graph(True)
    for B in (1, 2, 4):
        rgbb = vision.DeviceBuffer.from_numpy(np.repeat(img[None], B, 0))
        outb = vision.DeviceBuffer(B * 518 * 518 * 4)
        for _ in range(5):
            model.compute_batch_device(rgbb.ptr, B, 518, 518, outb.ptr)
        t0 = time.perf_counter()
        for _ in range(n):
            model.compute_batch_device(rgbb.ptr, B, 518, 518, outb.ptr)
        print(f"schedule {sched} (0 = one launch per node group, 1 = model kernels), batch {B}, hipGraph: {(time.perf_counter() - t0) / n * 1e3:.3f} ms")
