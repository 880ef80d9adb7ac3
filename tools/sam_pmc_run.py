#!/usr/bin/env python3
"""Launches the dominant MobileSAM-encoder kernels at the shapes of one batch-16 step, nothing else -- the target of the
rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE need one pass each; attribution by kernel name):
  tv_dwconv3x3_kernel<1,8>   = MBConv depthwise 3x3 + GELU on [16, 256, 256, 256]        (algorithmic 2 x 537 MB)
  mbconv_dw_pw_kernel        = depthwise + GELU + conv3 + residual + GELU, same map      (algorithmic 537 + 2 x 134 MB)
  window_attention_kernel<2> = stage-1 attention, 16 x 361 windows of 49 tokens, 4 heads (qkv 217 MB in, 72 MB out + bias)
  window_attention_kernel<7> = stage-2 attention, 16 x 25 windows of 196 tokens, 5 heads"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from tests import gpu_util as G  # noqa: E402
from visioncpp_amd import _lib as L  # noqa: E402


def depthwise(reps, B=16, H=256, W=256, C=256):
    rng = np.random.default_rng(0)
    x = G.dev((rng.standard_normal((B, H, W, C)) * 0.5).astype(np.float16))
    w = G.dev((rng.standard_normal((9, C)) / 3).astype(np.float16))
    b = G.dev(np.zeros(C, np.float32))
    y = G.empty(B * H * W * C * 2, zero=False)
    for _ in range(reps):
        L.vx_check(G.api().vx_dwconv3x3_f16(x.ptr, w.ptr, b.ptr, y.ptr, B, H, W, C, 1, 1, None))
    G.sync()
    G.release()


def fused_mbconv(reps, B=16, H=256, W=256, C=256, Co=64):
    rng = np.random.default_rng(2)
    h = G.dev((rng.standard_normal((B, H, W, C)) * 0.5).astype(np.float16))
    x = G.dev((rng.standard_normal((B, H, W, Co)) * 0.5).astype(np.float16))
    w2 = G.dev((rng.standard_normal((9, C)) / 3).astype(np.float16))
    w3 = G.dev((rng.standard_normal((Co, C)) / 16).astype(np.float16))
    b2, b3 = G.dev(np.zeros(C, np.float32)), G.dev(np.zeros(Co, np.float32))
    y = G.empty(B * H * W * Co * 2, zero=False)
    for _ in range(reps):
        L.vx_check(G.api().vx_mbconv_dw_pw_f16(h.ptr, w2.ptr, b2.ptr, w3.ptr, b3.ptr, x.ptr, y.ptr, B, H, W, C, Co, None))
    G.sync()
    G.release()


def attention(reps, n_windows, N, heads):
    rng = np.random.default_rng(1)
    qkv = G.dev((rng.standard_normal((n_windows * N, heads * 96)) * 0.5).astype(np.float16))
    bias = rng.standard_normal((heads, N, N)).astype(np.float32)
    packed = np.zeros(G.api().vx_window_attention_bias_bytes(N, heads) // 2, np.uint16)
    L.vx_check(G.api().vx_window_attention_pack_bias(bias.ctypes.data, N, heads, packed.ctypes.data))
    out = G.empty(n_windows * N * heads * 32 * 2, zero=False)
    pb = G.dev(packed)
    for _ in range(reps):
        L.vx_check(G.api().vx_window_attention_f16(qkv.ptr, pb.ptr, out.ptr, n_windows, N, heads, None))
    G.sync()
    G.release()


if __name__ == "__main__":
    depthwise(6)
    fused_mbconv(6)
    attention(6, 16 * 361, 49, 4)
    attention(6, 16 * 25, 196, 5)
