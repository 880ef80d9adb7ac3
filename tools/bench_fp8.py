#!/usr/bin/env python3
"""The opt-in e4m3 GEMM (csrc/kernels_gemm_fp8.hip, v_mfma_scale_f32_16x16x128_f8f6f4) next to the f16 GEMM on the MLP shapes of the TinyViT-5M
stages at BASELINE.json configs[4]'s per-GPU batch (128 images of 1024 x 1024): TFLOP/s of each, the cost of quantising the activations, and the
error of the e4m3 result against the f16 one on random operands. (What e4m3 does to MobileSAM's masks is tests/test_fp8_decision.py: rejected.)"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
from bench_kernels import api, stream, timeit, L, DeviceBuffer  # noqa: E402


def case(name, M, N, K, act):
    rng = np.random.default_rng(0)
    Kp, Np = (K + 127) // 128 * 128, (N + 127) // 128 * 128
    x = rng.standard_normal((M, K)).astype(np.float16)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = (0.1 * rng.standard_normal(Np)).astype(np.float32)
    wq, ws = np.zeros((Np, Kp), np.uint8), np.ones(Np, np.float32)
    L.vx_check(api.vx_quantize_rows_e4m3_host(w.ctypes.data, N, K, Kp, wq.ctypes.data, ws.ctypes.data))
    xd = DeviceBuffer.from_numpy(x)
    qd, sd = DeviceBuffer(M * Kp), DeviceBuffer(M * 4)
    wqd, wsd, bd = DeviceBuffer.from_numpy(wq), DeviceBuffer.from_numpy(ws), DeviceBuffer.from_numpy(b)
    out8 = DeviceBuffer(M * N * 2)
    a = L.GemmFp8Args()
    a.A, a.a_scale, a.W, a.w_scale, a.bias = qd.ptr, sd.ptr, wqd.ptr, wsd.ptr, bd.ptr
    a.M, a.N, a.Kp, a.n_valid, a.out, a.ldo, a.act, a.res = M, Np, Kp, N, out8.ptr, N, act, None
    t_q = timeit(lambda: L.vx_check(api.vx_quantize_rows_e4m3(xd.ptr, K, qd.ptr, sd.ptr, M, K, Kp, stream)))
    t_8 = timeit(lambda: L.vx_check(api.vx_gemm_fp8(C.byref(a), stream)))
    # the f16 GEMM of the library on the same problem (K padded to 64, N to its tile)
    K16 = (K + 63) // 64 * 64
    x16 = np.zeros((M, K16), np.float16); x16[:, :K] = x
    w16 = np.zeros((Np, K16), np.float16); w16[:N, :K] = w.astype(np.float16)
    xd16, wd16 = DeviceBuffer.from_numpy(x16), DeviceBuffer.from_numpy(w16)
    out16 = DeviceBuffer(M * N * 2)
    g = L.GemmArgs()
    g.A, g.lda, g.W, g.bias, g.M, g.N, g.K = xd16.ptr, K16, wd16.ptr, bd.ptr, M, Np, K16
    g.epi, g.out, g.ldo, g.n_valid = (L.EPI_F16_GELU if act else L.EPI_F16), out16.ptr, N, N
    t_16 = timeit(lambda: L.vx_check(api.vx_gemm_f16(C.byref(g), stream)))
    L.vx_check(api.vx_stream_sync(stream))
    y8, y16 = out8.to_numpy(np.float16, (M, N)).astype(np.float32), out16.to_numpy(np.float16, (M, N)).astype(np.float32)
    flops = 2.0 * M * N * K
    err = float(np.abs(y8 - y16).mean() / np.abs(y16).mean())
    print(f"{name:34s} M={M} N={N} K={K}: e4m3 {t_8 * 1e3:7.1f} us = {flops / t_8 / 1e9:6.0f} TFLOP/s (+ {t_q * 1e3:6.1f} us to quantise the activations) | "
          f"f16 {t_16 * 1e3:7.1f} us = {flops / t_16 / 1e9:6.0f} TFLOP/s | mean |e4m3 - f16| / mean |f16| = {err:.3f}", flush=True)


if __name__ == "__main__":
    B = 128  # images per GPU in configs[4]
    for rnd in range(2):
        case("stage 3 fc1 + gelu (C = 320)", B * 32 * 32, 1280, 320, 1)
        case("stage 3 fc2", B * 32 * 32, 320, 1280, 0)
        case("stage 2 fc1 + gelu (C = 160)", B * 64 * 64, 640, 160, 1)
        case("stage 2 fc2", B * 64 * 64, 160, 640, 0)
        case("square 4096 x 4096 x 4096", 4096, 4096, 4096, 0)
