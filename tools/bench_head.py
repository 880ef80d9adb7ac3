#!/usr/bin/env python3
"""Same-device timing of the DPT tail's LDS-ring convs at the Depth-Anything shapes (NHWC, batch 11 and 32 by default):
head.conv1 64->32 @296^2 (plain | resizing 148^2 itself), head.conv2+3 32->32->1 @518^2 (plain | resizing 296^2 itself), the
stand-alone bilinear kernels they replace, and the fusion residual-unit conv 64->64 @148^2. VISP_LIBRARY picks the build."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from tests import gpu_util as G  # noqa: E402
from visioncpp_amd import _lib as L  # noqa: E402


def timed(fn, reps=20):
    api = G.api()
    e0, e1 = C.c_void_p(), C.c_void_p()
    api.vx_event_create(C.byref(e0)); api.vx_event_create(C.byref(e1))
    for _ in range(3):
        fn()
    api.vx_event_record(e0, None)
    for _ in range(reps):
        fn()
    api.vx_event_record(e1, None)
    G.sync()
    ms = C.c_float()
    api.vx_event_elapsed_ms(e0, e1, C.byref(ms))
    return ms.value / reps * 1e3


def conv_args(x, B, H, W, cin, cout, w, bias, out, head=None, bil=None, relu_in=False, res=None):
    d = L.DconvArgs()
    d.x, d.x_pix, d.x_plane, d.cin, d.B, d.H, d.W = x.ptr, cin, 32, cin, B, H, W
    d.w, d.bias, d.cout, d.epi, d.s1, d.s2 = w.ptr, bias.ptr, cout, L.DC_F16, 1.0, 1.0
    d.out, d.out_pix, d.out_plane = out.ptr, cout, 32
    d.res1_pix = d.res2_pix = cout
    d.res1_plane = d.res2_plane = 32
    d.a_relu = int(relu_in)
    if res is not None:
        d.res1 = res.ptr
    if bil:
        d.bil_hs, d.bil_ws = bil
    if head is not None:
        d.epi, d.head_w, d.head_bias, d.head_scale = L.DC_HEAD_F32, head.ptr, 0.05, 1.0
    return d


def run(B):
    rng = np.random.default_rng(0)
    api = G.api()

    def rnd(*shape):
        return G.dev((rng.standard_normal(shape) * 0.5).astype(np.float16))

    def wts(cout, cin):
        return G.dev(G.pack_dconv((rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32), cin, cout))

    rows = []
    # head.conv1: 64 -> 32 at 296^2
    lo, hi = rnd(B, 148, 148, 64), rnd(B, 296, 296, 64)
    w1, b32 = wts(32, 64), G.dev(np.zeros(32, np.float32))
    o1 = G.empty(B * 296 * 296 * 32 * 2, zero=False)
    a = conv_args(hi, B, 296, 296, 64, 32, w1, b32, o1)
    rows.append(("head.conv1 plain", timed(lambda: L.vx_check(api.vx_dconv3x3_f16(C.byref(a), None))), 2.0 * B * 296 * 296 * 9 * 64 * 32))
    rows.append(("bilinear 148->296 x64", timed(lambda: L.vx_check(api.vx_bilinear_ac_f16(lo.ptr, hi.ptr, B, 148, 148, 64, 296, 296, None))), 0))
    if hasattr(api, "vx_dconv_bilinear_supported"):
        ab = conv_args(lo, B, 296, 296, 64, 32, w1, b32, o1, bil=(148, 148))
        rows.append(("head.conv1 resizing", timed(lambda: L.vx_check(api.vx_dconv3x3_f16(C.byref(ab), None))), 2.0 * B * 296 * 296 * 9 * 64 * 32))
    G.release()
    # head.conv2 + conv3: 32 -> 32 -> 1 at 518^2
    lo, hi = rnd(B, 296, 296, 32), rnd(B, 518, 518, 32)
    w2 = wts(32, 32)
    hw = G.dev(np.abs(rng.standard_normal(32)).astype(np.float32))
    od = G.empty(B * 518 * 518 * 4, zero=False)
    a2 = conv_args(hi, B, 518, 518, 32, 32, w2, b32, od, head=hw)
    rows.append(("head.conv2+3 plain", timed(lambda: L.vx_check(api.vx_dconv3x3_f16(C.byref(a2), None))), 2.0 * B * 518 * 518 * 32 * (9 * 32 + 1)))
    rows.append(("bilinear 296->518 x32", timed(lambda: L.vx_check(api.vx_bilinear_ac_f16(lo.ptr, hi.ptr, B, 296, 296, 32, 518, 518, None))), 0))
    if hasattr(api, "vx_dconv_bilinear_supported"):
        a2b = conv_args(lo, B, 518, 518, 32, 32, w2, b32, od, head=hw, bil=(296, 296))
        rows.append(("head.conv2+3 resizing", timed(lambda: L.vx_check(api.vx_dconv3x3_f16(C.byref(a2b), None))), 2.0 * B * 518 * 518 * 32 * (9 * 32 + 1)))
    G.release()
    # fusion residual unit convs: 64 -> 64 at 148^2 (relu on load + relu; + residual) and 74^2
    for hw_ in (148, 74, 37):
        x, r = rnd(B, hw_, hw_, 64), rnd(B, hw_, hw_, 64)
        w3, b64 = wts(64, 64), G.dev(np.zeros(64, np.float32))
        o3 = G.empty(B * hw_ * hw_ * 64 * 2, zero=False)
        a3 = conv_args(x, B, hw_, hw_, 64, 64, w3, b64, o3, relu_in=True)
        a3.act = 2
        rows.append((f"rcu conv1 64->64 @{hw_}", timed(lambda: L.vx_check(api.vx_dconv3x3_f16(C.byref(a3), None))), 2.0 * B * hw_ * hw_ * 9 * 64 * 64))
        a4 = conv_args(x, B, hw_, hw_, 64, 64, w3, b64, o3, res=r)
        rows.append((f"rcu conv2 64->64 @{hw_} + res", timed(lambda: L.vx_check(api.vx_dconv3x3_f16(C.byref(a4), None))), 2.0 * B * hw_ * hw_ * 9 * 64 * 64))
        G.release()
    for name, us, fl in rows:
        print(f"B {B:2d} {name:30s} {us:8.1f} us" + (f" {fl / us / 1e6:7.0f} TF" if fl else ""))


if __name__ == "__main__":
    for B in ([int(v) for v in sys.argv[1:]] or [11, 32]):
        run(B)
