#!/usr/bin/env python3
"""Launches the two dominant ESRGAN conv shapes a few times each, nothing else -- the target of the rocprofv3
--pmc passes (FETCH_SIZE / WRITE_SIZE / SQ counters need one pass each and per-kernel attribution by name):
  dconv3x3_kernel<64,...>  = dense-block conv5 (cin 192 -> 64, x_residual), 64 tiles of 144x144
  dconv3x3_kernel<32,...>  = dense-block conv4 (cin 160 -> 32, LeakyReLU),  64 tiles of 144x144"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from tests import gpu_util as G  # noqa: E402
from visioncpp_amd import _lib as L  # noqa: E402


def launch(cin, cout, xres, reps, B=64, H=144, W=144):
    rng = np.random.default_rng(0)
    x = G.dev((rng.standard_normal((6, B, H, W, 32)) * 0.5).astype(np.float16))
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    wd, bd = G.dev(G.pack_dconv(w, cin, cout)), G.dev(np.zeros(cout, np.float32))
    out = G.empty(2 * B * H * W * 32 * 2, zero=False)
    a = L.DconvArgs()
    a.x, a.x_plane, a.cin = x.ptr, B * H * W * 32, cin
    a.B, a.H, a.W = B, H, W
    a.w, a.bias, a.cout = wd.ptr, bd.ptr, cout
    a.epi, a.act, a.s1, a.s2 = L.DC_F16, 1, 0.2, 1.0
    a.out, a.out_plane = out.ptr, B * H * W * 32
    a.x_residual = int(xres)
    for _ in range(reps):
        L.vx_check(G.api().vx_dconv3x3_f16(C.byref(a), None))
    G.sync()
    G.release()


if __name__ == "__main__":
    launch(192, 64, True, 6)
    launch(160, 32, False, 6)
