#!/usr/bin/env python3
"""Per-group timing of the ESRGAN row (BASELINE.json configs[2]: Real-ESRGAN-4x f16, 256^2 -> 1024^2, batch 16).
Usage: python tools/bench_esrgan.py [--batch 16] [--size 256] [--blocks 23] [--group 64] [--iters 5]"""
import argparse
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from visioncpp_amd import synth  # noqa: E402
from visioncpp_amd.vision import Backend, Device, DeviceBuffer, Model  # noqa: E402
from visioncpp_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--blocks", type=int, default=23)
    ap.add_argument("--scale", type=int, default=4)
    ap.add_argument("--group", type=int, default=64)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    cfg = synth.EsrganConfig(num_blocks=a.blocks, scale=a.scale, name="bench")
    dev = Device.init(Backend.gpu)
    with tempfile.TemporaryDirectory() as td:
        m = Model.load(synth.write_esrgan_gguf(Path(td) / "e.gguf", cfg, 1), dev)
    m.set_tile_group(a.group)
    imgs = synth.images(a.batch, a.size, a.size, seed=3)
    din = DeviceBuffer.from_numpy(imgs)
    s = a.scale
    dout = DeviceBuffer(a.batch * a.size * s * a.size * s * 4)
    api = L.get_lib()
    for _ in range(2):
        m.upscale_batch_device(din.ptr, a.batch, a.size, a.size, dout.ptr)
    t0 = time.perf_counter()
    for _ in range(a.iters):
        m.upscale_batch_device(din.ptr, a.batch, a.size, a.size, dout.ptr)
    dt = (time.perf_counter() - t0) / a.iters
    m.enable_timing(True)
    m.upscale_batch_device(din.ptr, a.batch, a.size, a.size, dout.ptr)
    tm = m.read_timing()
    m.enable_timing(False)
    tot_f = sum(t["flops"] for t in tm)
    print(f"batch {a.batch} {a.size}^2 x{s} blocks {a.blocks} group {a.group}: {dt * 1e3:.2f} ms/step, {a.batch / dt:.1f} img/s, "
          f"{tot_f / dt / 1e12:.1f} TFLOP/s over the step ({tot_f / a.batch / 1e9:.1f} GFLOP/img incl. tile overlap)")
    print(f"{'group':<14}{'ms':>9}{'%':>7}{'launch':>8}{'TFLOP/s':>10}{'GB/s':>10}")
    tt = sum(t["ms"] for t in tm)
    for t in tm:
        print(f"{t['name']:<14}{t['ms']:>9.3f}{100 * t['ms'] / tt:>7.1f}{t['launches']:>8}{t['flops'] / t['ms'] / 1e9 if t['ms'] else 0:>10.1f}"
              f"{t['bytes'] / t['ms'] / 1e6 if t['ms'] else 0:>10.1f}")


if __name__ == "__main__":
    main()
