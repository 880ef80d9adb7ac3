#!/usr/bin/env python3
"""Calibration only (NOT product code): what the vendor libraries reach on this path's shapes on the same box --
torch.mm (hipBLASLt/rocBLAS) for the encoder GEMMs, F.scaled_dot_product_attention for the MHSA."""
import torch
import torch.nn.functional as F

dev = "cuda"
M = 32 * 1370


def t(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name, N, K in [("qkv", 1152, 384), ("fc1", 1536, 384), ("fc2", 384, 1536), ("out", 384, 384)]:
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K, device=dev, dtype=torch.float16)
    ms = t(lambda: torch.mm(a, w.t()))
    print(f"torch.mm {name:4s} M={M} N={N} K={K}: {ms * 1e3:7.1f} us {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s (plain GEMM, no epilogue)")
q, k, v = (torch.randn(32, 6, 1370, 64, device=dev, dtype=torch.float16) for _ in range(3))
ms = t(lambda: F.scaled_dot_product_attention(q, k, v))
print(f"sdpa B=32 H=6 T=1370 d=64: {ms * 1e3:7.1f} us {4.0 * 32 * 6 * 1370 * 1370 * 64 / ms / 1e9:7.1f} TFLOP/s")
