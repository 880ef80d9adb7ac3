#!/usr/bin/env python3
"""Depth-Anything-V2-Small at the north-star shape through the graph layer, every way it can be driven (round 4: the model entries of the
library run on this layer too):
  * a graph built by the Python face (vision.cpp_amd/graph.py::depthany_predict, the node list the reference's arch sources build) on an
    f32 image, one stream, eager launches and one hipGraph replay -- with the model-kernel node groups (default) and without
    (set_fused_models(False): one launch per epilogue-fused node);
  * the library's own step (visp_depthany_compute_batch_device: u8 in HBM -> normalised depth, three sub-batch graphs on parallel
    streams inside one hipGraph) -- bench.py's workload.
Prints ms per batch and images/s for each."""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from visioncpp_amd import graph as G  # noqa: E402
from visioncpp_amd import synth, vision  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
W = H = 518
path = Path(tempfile.gettempdir()) / "graph_bench_da.gguf"
synth.write_gguf(path, synth.SMALL, seed=0)
dev = vision.Device.init(vision.Backend.gpu)
imgs = np.concatenate([synth.images(4, W, H, seed=1)] * ((B + 3) // 4))[:B]

mean, std = np.array([0.485, 0.456, 0.406], np.float32), np.array([0.229, 0.224, 0.225], np.float32)
pre = ((imgs.astype(np.float32) / 255.0 - mean) / std).astype(np.float32)


def timed(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e3


depth = None
for fused, hip_graph in ((False, True), (True, False), (True, True)):
    g = G.Graph(dev, G.Weights(path))
    g.set_fused_models(fused)
    xi = g.input((3, W, H, B), G.F32, "image")
    out = G.depthany_predict(G.ModelRef(g), xi, 12, 6)
    t0 = time.perf_counter()
    g.allocate()
    t_alloc = time.perf_counter() - t0
    g.use_hip_graph(hip_graph)
    g.set(xi, pre)
    ms = timed(g.compute)
    s = g.summary()
    print(f"graph ({'model-kernel groups' if fused else 'one launch per node'}, one stream, {'hipGraph replay' if hip_graph else 'eager launches'}): "
          f"{ms:.3f} ms per batch of {B} = {B / ms * 1e3:.0f} img/s; {s['launches']} launches, arena {s['arena_bytes'] / 1e6:.0f} MB, lower + pack + upload {t_alloc:.2f} s", flush=True)
    depth = g.get(out)[..., 0]
    del g

model = vision.Model.load(path, dev)
model.use_graph(True)
rgb = vision.DeviceBuffer.from_numpy(imgs)
outb = vision.DeviceBuffer(B * W * H * 4)
ms = timed(lambda: model.compute_batch_device(rgb.ptr, B, W, H, outb.ptr))
print(f"library step (u8 -> normalised depth, 3 sub-batch graphs on parallel streams, one hipGraph): {ms:.3f} ms per batch of {B} = {B / ms * 1e3:.0f} img/s")
step = outb.to_numpy(np.float32, (B, H, W))
nd = (depth - depth.min(axis=(1, 2), keepdims=True)) / (depth.max(axis=(1, 2), keepdims=True) - depth.min(axis=(1, 2), keepdims=True))
print(f"mean |one-stream graph - library step| on the normalised depth: {np.abs(nd - step).mean():.2e}")
