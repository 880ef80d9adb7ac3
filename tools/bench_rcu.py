#!/usr/bin/env python3
"""Launch time of the fused residual-unit kernel (kernels_rcu.hip) at the DPT fusion stages' map sizes, sub-batch (11) and whole batch (32):
   python tools/bench_rcu.py      (compare with the two LDS-ring conv launches + projection of tools/dpt_launches.py)"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from tests import gpu_util as G  # noqa: E402
from visioncpp_amd import _lib as L  # noqa: E402


def run(B, hw, proj, res):
    api = G.api()
    rng = np.random.default_rng(0)
    keep = [G.dev(rng.standard_normal((B, hw, hw, 64)).astype(np.float16)), G.dev(G.pack_rcu(rng.standard_normal((64, 3, 3, 64)) / 24)), G.dev(np.zeros(64, np.float32)),
            G.dev(G.pack_rcu(rng.standard_normal((64, 3, 3, 64)) / 24)), G.dev(np.zeros(64, np.float32)), G.dev(G.pack_rcu(rng.standard_normal((64, 1, 1, 64)) / 8)),
            G.dev(rng.standard_normal((B, hw, hw, 64)).astype(np.float16))]
    out = G.empty(B * hw * hw * 64 * 2, zero=False)
    a = L.RcuArgs()
    a.x, a.w1, a.b1, a.w2, a.b2 = (k.ptr for k in keep[:5])
    if proj:
        a.wp, a.bp = keep[5].ptr, keep[2].ptr
    if res:
        a.res2 = keep[6].ptr
    a.out, a.B, a.H, a.W = out.ptr, B, hw, hw
    ev0, ev1 = C.c_void_p(), C.c_void_p()
    api.vx_event_create(C.byref(ev0)); api.vx_event_create(C.byref(ev1))
    for _ in range(3):
        L.vx_check(api.vx_rcu_fused_f16(C.byref(a), None))
    api.vx_event_record(ev0, None)
    n = 20
    for _ in range(n):
        L.vx_check(api.vx_rcu_fused_f16(C.byref(a), None))
    api.vx_event_record(ev1, None)
    G.sync()
    ms = C.c_float()
    api.vx_event_elapsed_ms(ev0, ev1, C.byref(ms))
    fl = 2.0 * B * hw * hw * 64 * 576 * 2 + (2.0 * B * hw * hw * 64 * 64 if proj else 0)
    print(f"rcu fused B={B:2d} {hw:3d}x{hw:<3d} proj={int(proj)} res={int(res)}: {ms.value / n * 1e3:7.1f} us  {fl / (ms.value / n * 1e-3) / 1e12:7.1f} TFLOP/s (useful)")
    G.release()


for B in (11, 32):
    for hw in (19, 37, 74):
        run(B, hw, False, True)
        run(B, hw, True, False)
