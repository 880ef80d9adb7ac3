#!/usr/bin/env python3
"""Same-device timing of the two 3x3 conv kernels on the DPT shapes of the Depth-Anything path (batch 32):
conv3x3_halo_kernel (NHWC, kernels_conv.hip) vs dconv3x3_kernel (planar, kernels_dconv.hip)."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from tests import gpu_util as G  # noqa: E402
from visioncpp_amd import _lib as L  # noqa: E402


def timed(fn, reps=10):
    api = G.api()
    e0, e1 = C.c_void_p(), C.c_void_p()
    api.vx_event_create(C.byref(e0)); api.vx_event_create(C.byref(e1))
    for _ in range(3):
        fn()
    api.vx_event_record(e0, None)
    for _ in range(reps):
        fn()
    api.vx_event_record(e1, None)
    ms = C.c_float()
    api.vx_event_elapsed_ms(e0, e1, C.byref(ms))
    return ms.value / reps * 1e3


def run(B, H, W, cin, cout):
    rng = np.random.default_rng(0)
    M = B * H * W
    flops = 2.0 * M * 9 * cin * cout
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    # halo kernel, NHWC
    x = G.dev((rng.standard_normal((B, H, W, cin)) * 0.5).astype(np.float16))
    wp = G.dev(G.pad_weight(np.ascontiguousarray(w.transpose(0, 2, 3, 1)).reshape(cout, -1)))
    bd = G.dev(np.zeros(cout, np.float32))
    out = G.empty(M * cout * 2, zero=False)
    a = L.GemmArgs()
    a.A, a.W, a.bias, a.M, a.N, a.K = x.ptr, wp.ptr, bd.ptr, M, cout, -(-9 * cin // 64) * 64
    a.conv_kh = a.conv_kw = 3; a.conv_stride = 1; a.conv_pad = 1
    a.conv_H, a.conv_W, a.conv_Cin, a.conv_OH, a.conv_OW = H, W, cin, H, W
    a.epi, a.out, a.ldo = L.EPI_F16, out.ptr, cout
    t_halo = timed(lambda: L.vx_check(G.api().vx_conv3x3_f16(C.byref(a), None)))
    # dconv, planar
    xp = G.dev((rng.standard_normal((cin // 32, B, H, W, 32)) * 0.5).astype(np.float16))
    wd = G.dev(G.pack_dconv(w, cin, cout))
    outp = G.empty(M * cout * 2, zero=False)
    d = L.DconvArgs()
    d.x, d.x_plane, d.cin, d.B, d.H, d.W = xp.ptr, M * 32, cin, B, H, W
    d.w, d.bias, d.cout, d.epi, d.s1, d.s2 = wd.ptr, bd.ptr, cout, L.DC_F16, 1.0, 1.0
    d.out, d.out_plane = outp.ptr, M * 32
    t_d = timed(lambda: L.vx_check(G.api().vx_dconv3x3_f16(C.byref(d), None)))
    print(f"B {B} {H}x{W} {cin}->{cout}: halo {t_halo:7.1f} us {flops / t_halo / 1e6:6.0f} TF   dconv {t_d:7.1f} us {flops / t_d / 1e6:6.0f} TF   x{t_halo / t_d:.2f}")
    G.release()


if __name__ == "__main__":
    for shape in [(32, 148, 148, 64, 64), (32, 74, 74, 64, 64), (32, 296, 296, 64, 32), (32, 518, 518, 32, 32), (32, 148, 148, 64, 32)]:
        run(*shape)
