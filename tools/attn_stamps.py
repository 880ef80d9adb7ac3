#!/usr/bin/env python3
"""In-kernel phase stamps of the attention kernel (diagnostics; the stamped instantiation is never on the product path): per wave, the
cycles of its lifetime spent (a) waiting for the K/V tile + in the tile barrier + issuing the next tile's DMA, (b) in scores +
exponentials + row sum, (c) in the PV product, at the north-star shape (32 x 6 heads x 1370 tokens) and at the sub-batch shape."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from tests import gpu_util as G  # noqa: E402
from visioncpp_amd import _lib as L  # noqa: E402


def run(B, H=6, T=1370, NW=int(__import__("os").environ.get("VISP_ATTN_WAVES", "8"))):
    api = G.api()
    rng = np.random.default_rng(0)
    q, k, v = (G.dev((rng.standard_normal((B, H, T, 64)) * (0.18 if i == 0 else 1.0)).astype(np.float16)) for i in range(3))
    out = G.empty(B * T * H * 64 * 2, zero=False)
    qpb = 32 * NW
    blocks = ((T + qpb - 1) // qpb) * B * H
    ev0, ev1 = C.c_void_p(), C.c_void_p()
    api.vx_event_create(C.byref(ev0)); api.vx_event_create(C.byref(ev1))
    for _ in range(3):
        L.vx_check(api.vx_attention_f16(q.ptr, k.ptr, v.ptr, out.ptr, B, H, T, None))
    api.vx_event_record(ev0, None)
    for _ in range(10):
        L.vx_check(api.vx_attention_f16(q.ptr, k.ptr, v.ptr, out.ptr, B, H, T, None))
    api.vx_event_record(ev1, None)
    ms = C.c_float()
    api.vx_event_elapsed_ms(ev0, ev1, C.byref(ms))
    stamps = G.empty(blocks * NW * 4 * 8)
    api.vx_attention_set_stamps(stamps.ptr)
    api.vx_event_record(ev0, None)
    L.vx_check(api.vx_attention_f16(q.ptr, k.ptr, v.ptr, out.ptr, B, H, T, None))
    api.vx_event_record(ev1, None)
    ms2 = C.c_float()
    api.vx_event_elapsed_ms(ev0, ev1, C.byref(ms2))
    api.vx_attention_set_stamps(None)
    G.sync()
    st = stamps.to_numpy(np.uint64, (blocks, NW, 4)).astype(np.float64)
    active = st[..., 1] > 0  # waves with queries
    tiles = (T + 63) // 64
    a = st[active]
    idle = st[~active]
    print(f"B={B}: {ms.value / 10 * 1e3:.1f} us per launch ({blocks} blocks), stamped launch {ms2.value * 1e3:.1f} us; {int(active.sum())} waves with queries, {int((~active).sum())} without")
    print(f"  per wave and 64-key tile (s_memtime ticks = 100 MHz x ... see lifetime): wait+barrier+dma {a[:, 0].mean() / tiles:8.1f}  scores+softmax {a[:, 1].mean() / tiles:8.1f}  pv {a[:, 2].mean() / tiles:8.1f}"
          f"  | lifetime {a[:, 3].mean():9.0f} (min {a[:, 3].min():.0f}, max {a[:, 3].max():.0f}); shares: wait {a[:, 0].sum() / a[:, 3].sum():.3f} softmax {a[:, 1].sum() / a[:, 3].sum():.3f} pv {a[:, 2].sum() / a[:, 3].sum():.3f}")
    if idle.size:
        print(f"  waves without queries: lifetime {idle[:, 3].mean():.0f}")
    G.release()


for B in (32, 11):
    run(B)
