#!/usr/bin/env python3
"""In-kernel phase stamps of the DPT head kernel (kernels_headconv.hip; diagnostics): per wave and tile, ticks spent computing the
coordinate tables, waiting (DMA + barriers), issuing the next patch + interpolating the halo, in the MFMA loop + epilogue; at the
north-star shape (32 x 296^2 x 32 -> 518^2)."""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from tests import gpu_util as G  # noqa: E402
from visioncpp_amd import _lib as L  # noqa: E402

B, hs, ws, H, W = 32, 296, 296, 518, 518
lib = G.api()
rng = np.random.default_rng(0)
x = G.dev((rng.standard_normal((B, hs, ws, 32)) * 0.7).astype(np.float16))
rows = np.zeros((32, 320), np.float16)
rows[:, :288] = (rng.standard_normal((32, 288)) / 17).astype(np.float16)
frag = np.empty(lib.vx_headconv_frag_bytes() // 2, np.float16)
L.vx_check(lib.vx_headconv_pack(rows.ctypes.data, 320, frag.ctypes.data))
fd, bd, wd = G.dev(frag), G.dev(np.zeros(32, np.float32)), G.dev(np.ones(32, np.float32))
out = G.empty(B * H * W * 4, zero=False)
ev0, ev1 = C.c_void_p(), C.c_void_p()
lib.vx_event_create(C.byref(ev0)); lib.vx_event_create(C.byref(ev1))


def launch():
    L.vx_check(lib.vx_headconv_bil_f16(x.ptr, fd.ptr, bd.ptr, wd.ptr, 0.0, 1.0, out.ptr, B, H, W, hs, ws, None))


for _ in range(3):
    launch()
lib.vx_event_record(ev0, None)
for _ in range(10):
    launch()
lib.vx_event_record(ev1, None)
ms = C.c_float()
lib.vx_event_elapsed_ms(ev0, ev1, C.byref(ms))
bpc = int(os.environ.get("VISP_HEADCONV_BPC", "2"))
blocks = 256 * bpc
st = G.empty(blocks * 4 * 8 * 8)
lib.vx_headconv_set_stamps(st.ptr)
launch()
lib.vx_headconv_set_stamps(None)
G.sync()
s = st.to_numpy(np.uint64, (blocks, 4, 8)).astype(np.float64)
tiles = s[..., 4].sum()
flops = 2.0 * B * H * W * 32 * 289
print(f"blocks per CU {bpc}: {ms.value / 10 * 1e3:.1f} us per launch = {flops / (ms.value / 10 * 1e-3) / 1e12:.0f} TFLOP/s; per wave and tile (ticks): tables {s[..., 0].sum() / tiles:.0f}  wait+barriers {s[..., 1].sum() / tiles:.0f}  "
      f"dma issue+interp {s[..., 2].sum() / tiles:.0f}  mfma+epilogue {s[..., 3].sum() / tiles:.0f}  | tiles per block {s[:, 0, 4].mean():.1f}, lifetime {s[..., 5].mean():.0f} ticks (min {s[..., 5].min():.0f}, max {s[..., 5].max():.0f})")
