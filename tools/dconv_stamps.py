#!/usr/bin/env python3
"""In-kernel phase stamps of vx_dconv3x3_f16 (diagnostics): cycles per step spent waiting for the LDS-DMA, in the
barrier, issuing DMA, in the MFMA loop, in the epilogue. Shapes = the ESRGAN dense-block convs at 64 tiles of 144^2."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from tests import gpu_util as G  # noqa: E402
from visioncpp_amd import _lib as L  # noqa: E402


def run(cin, cout, B=64, H=144, W=144, x_residual=False, reps=3, nhwc=False, res=False):
    rng = np.random.default_rng(0)
    planes = max(cin // 32, 6)
    x = G.dev((rng.standard_normal((planes, B, H, W, 32)) * 0.5).astype(np.float16))
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    wd = G.dev(G.pack_dconv(w, cin, cout))
    bd = G.dev(np.zeros(cout, np.float32))
    out = G.empty(2 * B * H * W * 32 * 2, zero=False)
    stamps = G.empty(1024 * 8 * 8)
    a = L.DconvArgs()
    a.x, a.x_plane, a.cin = x.ptr, B * H * W * 32, cin
    a.B, a.H, a.W = B, H, W
    a.w, a.bias, a.cout = wd.ptr, bd.ptr, cout
    a.epi, a.act, a.s1, a.s2 = L.DC_F16, 1, 0.2, 1.0
    a.out, a.out_plane = out.ptr, B * H * W * 32
    a.x_residual = int(x_residual)
    if nhwc:  # the DPT maps: NHWC through the pixel / plane strides, optional residual
        a.x_pix, a.x_plane, a.out_pix, a.out_plane, a.res1_pix, a.res1_plane, a.res2_pix, a.res2_plane = cin, 32, cout, 32, cout, 32, cout, 32
        a.act, a.s1 = 0, 1.0
        if res:
            a.res1 = x.ptr
    api = G.api()
    ev0, ev1 = C.c_void_p(), C.c_void_p()
    api.vx_event_create(C.byref(ev0)); api.vx_event_create(C.byref(ev1))
    for _ in range(2):
        L.vx_check(api.vx_dconv3x3_f16(C.byref(a), None))
    api.vx_event_record(ev0, None)
    for _ in range(reps):
        L.vx_check(api.vx_dconv3x3_f16(C.byref(a), None))
    api.vx_event_record(ev1, None)
    ms = C.c_float()
    api.vx_event_elapsed_ms(ev0, ev1, C.byref(ms))
    a.stamps = stamps.ptr
    L.vx_check(api.vx_dconv3x3_f16(C.byref(a), None))
    G.sync()
    st = stamps.to_numpy(np.uint64, (1024, 8))[:256].astype(np.float64)
    steps = st[:, 6]
    per = st[:, :6].sum(0) / steps.sum()
    flops = 2.0 * B * H * W * 9 * cin * cout
    us = ms.value / reps * 1e3
    names = ["dma_wait", "barrier", "dma_issue+cursor", "mfma_loop", "epilogue", "tile_setup"]
    print(f"cin {cin:3d} cout {cout:2d}{' xres' if x_residual else ''}: {us:7.1f} us  {flops / us / 1e6:7.1f} TFLOP/s   steps/block {steps.mean():.1f}  "
          f"cycles/step: " + "  ".join(f"{n} {v:6.0f}" for n, v in zip(names, per)) + f"  total {per.sum():6.0f}")
    G.release()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "dpt":  # the Depth-Anything tail at batch 32
        run(64, 64, B=32, H=148, W=148, nhwc=True, res=True)
        run(64, 64, B=32, H=148, W=148, nhwc=True)
        run(64, 64, B=32, H=74, W=74, nhwc=True, res=True)
        run(64, 32, B=32, H=296, W=296, nhwc=True)
        run(32, 32, B=32, H=518, W=518, nhwc=True)
        sys.exit(0)
    for cin, cout, xr in [(64, 32, False), (96, 32, False), (128, 32, False), (160, 32, False), (192, 64, True), (64, 64, False)]:
        run(cin, cout, x_residual=xr)
