#!/usr/bin/env python3
"""Kernel micro-benchmarks on one MI355X (HIP events on a dedicated stream, interleaved rounds in one
process). Shapes are the Depth-Anything-V2-Small encoder at batch 32 (M = 32*1370 tokens)."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from visioncpp_amd import _lib as L  # noqa: E402
from visioncpp_amd.vision import DeviceBuffer  # noqa: E402

api = L.get_lib()
L.vx_check(api.vx_set_device(0))
stream = C.c_void_p()
L.vx_check(api.vx_stream_create(C.byref(stream)))


def timeit(fn, iters=5 if '--quick' in sys.argv else 20, warm=1 if '--quick' in sys.argv else 3):
    for _ in range(warm):
        fn()
    e0, e1 = C.c_void_p(), C.c_void_p()
    api.vx_event_create(C.byref(e0)); api.vx_event_create(C.byref(e1))
    api.vx_event_record(e0, stream)
    for _ in range(iters):
        fn()
    api.vx_event_record(e1, stream)
    ms = C.c_float()
    L.vx_check(api.vx_event_elapsed_ms(e0, e1, C.byref(ms)))
    return ms.value / iters


def gemm_case(name, M, N, K, epi, stages, n_valid=0):
    rng = np.random.default_rng(0)
    a = DeviceBuffer.from_numpy(rng.standard_normal((M, K)).astype(np.float16))
    w = DeviceBuffer.from_numpy((rng.standard_normal((N, K)) * K ** -0.5).astype(np.float16))
    b = DeviceBuffer.from_numpy(rng.standard_normal(N).astype(np.float32))
    lam = DeviceBuffer.from_numpy(np.full(N, 0.1, np.float32))
    out = DeviceBuffer(M * max(N, 384) * 4)
    q, k, v = (DeviceBuffer(M * 384 * 2) for _ in range(3))
    g = L.GemmArgs()
    g.A, g.lda, g.W, g.bias, g.M, g.N, g.K = a.ptr, K, w.ptr, b.ptr, M, N, K
    g.epi, g.out, g.ldo, g.lambda_ = epi, out.ptr, N, lam.ptr
    g.q, g.k, g.vt, g.qkv_T, g.qkv_H, g.q_scale = q.ptr, k.ptr, v.ptr, 1370, 6, 0.125
    g.stages = stages
    g.n_valid = n_valid
    ms = timeit(lambda: L.vx_check(api.vx_gemm_f16(C.byref(g), stream)))
    print(f"{name:10s} M={M} N={N} K={K} stages={stages}: {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s", flush=True)
    return ms


def attn_case(B, H, T):
    rng = np.random.default_rng(1)
    q, k, v = (DeviceBuffer.from_numpy((rng.standard_normal((B, H, T, 64)) * 0.5).astype(np.float16)) for _ in range(3))
    o = DeviceBuffer(B * T * H * 64 * 2)
    ms = timeit(lambda: L.vx_check(api.vx_attention_f16(q.ptr, k.ptr, v.ptr, o.ptr, B, H, T, stream)))
    print(f"attention  B={B} H={H} T={T}: {ms * 1e3:8.1f} us  {4.0 * B * H * T * T * 64 / ms / 1e9:7.1f} TFLOP/s", flush=True)


def gemm_stamps(name, M, N, K, epi):
    """Per-block phase anatomy from in-kernel s_memtime stamps (diagnostic)."""
    rng = np.random.default_rng(0)
    a = DeviceBuffer.from_numpy(rng.standard_normal((M, K)).astype(np.float16))
    w = DeviceBuffer.from_numpy((rng.standard_normal((N, K)) * K ** -0.5).astype(np.float16))
    b = DeviceBuffer.from_numpy(rng.standard_normal(N).astype(np.float32))
    lam = DeviceBuffer.from_numpy(np.full(N, 0.1, np.float32))
    out = DeviceBuffer(M * max(N, 384) * 4)
    q, k, v = (DeviceBuffer(M * 384 * 2) for _ in range(3))
    nblk = -(-M // 128) * (N // 128)
    st = DeviceBuffer(nblk * 64)
    st.zero()
    g = L.GemmArgs()
    g.A, g.lda, g.W, g.bias, g.M, g.N, g.K = a.ptr, K, w.ptr, b.ptr, M, N, K
    g.epi, g.out, g.ldo, g.lambda_ = epi, out.ptr, N, lam.ptr
    g.q, g.k, g.vt, g.qkv_T, g.qkv_H, g.q_scale = q.ptr, k.ptr, v.ptr, 1370, 6, 0.125
    for _ in range(300):  # let the clock settle under sustained load before stamping
        L.vx_check(api.vx_gemm_f16(C.byref(g), stream))
    g.debug_stamps = st.ptr
    L.vx_check(api.vx_gemm_f16(C.byref(g), stream))
    L.vx_check(api.vx_stream_sync(stream))
    t = st.to_numpy(np.uint64, (nblk, 8)).astype(np.int64)
    t0 = t[:, 0].min()
    names = ["setup", "first-tile wait", "k-loop", "epilogue phase 1", "epilogue phase 2"]
    print(f"--- {name} M={M} N={N} K={K}: {nblk} blocks; kernel span {(t[:, 5].max() - t0)} ticks")
    for i, nm in enumerate(names):
        d = t[:, i + 1] - t[:, i]
        print(f"   {nm:18s} median {np.median(d):9.0f}  p10 {np.percentile(d, 10):9.0f}  p90 {np.percentile(d, 90):9.0f}")
    rt = (t[:, 7] - t[:, 6]).astype(np.float64)  # 100 MHz ticks
    ok = rt > 0
    mhz = (t[ok, 5] - t[ok, 0]) / rt[ok] * 100.0
    print(f"   in-kernel shader clock (s_memtime / s_memrealtime): median {np.median(mhz):.0f} MHz  p10 {np.percentile(mhz, 10):.0f}  p90 {np.percentile(mhz, 90):.0f}")
    life = t[:, 5] - t[:, 0]
    print(f"   {'block lifetime':18s} median {np.median(life):9.0f}  p10 {np.percentile(life, 10):9.0f}  p90 {np.percentile(life, 90):9.0f}")
    starts = np.sort(t[:, 0] - t0)
    print("   block start ticks (every 10th percentile):", [int(np.percentile(starts, p)) for p in range(0, 101, 10)])


if __name__ == "__main__":
    M = 32 * 1370
    if "--stamps" in sys.argv:
        gemm_stamps("qkv", M, 1152, 384, L.EPI_QKV)
        gemm_stamps("fc1_gelu", M, 1536, 384, L.EPI_F16_GELU)
        gemm_stamps("fc2_resid", M, 384, 1536, L.EPI_RESID_F32)
        gemm_stamps("out_resid", M, 384, 384, L.EPI_RESID_F32)
        sys.exit(0)
    if "--writes" in sys.argv:  # is the f16 epilogue store-bound? same math, 1/16 of the stores
        for _ in range(2):
            gemm_case("fc1 full", M, 1536, 384, L.EPI_F16_GELU, 0)
            gemm_case("fc1 nv=8", M, 1536, 384, L.EPI_F16_GELU, 0, n_valid=8)
            gemm_case("f16 full", M, 1536, 384, L.EPI_F16, 0)
            gemm_case("f16 nv=8", M, 1536, 384, L.EPI_F16, 0, n_valid=8)
        sys.exit(0)
    if "--quick" in sys.argv:  # one pass, default variants (for rocprofv3 counter runs)
        gemm_case("qkv", M, 1152, 384, L.EPI_QKV, 0)
        gemm_case("fc1_gelu", M, 1536, 384, L.EPI_F16_GELU, 0)
        gemm_case("fc2_resid", M, 384, 1536, L.EPI_RESID_F32, 0)
        gemm_case("out_resid", M, 384, 384, L.EPI_RESID_F32, 0)
        attn_case(32, 6, 1370)
        sys.exit(0)
    for rnd in range(2):
        for st in (1, 64):
            gemm_case("qkv", M, 1152, 384, L.EPI_QKV, st)
            gemm_case("fc1_gelu", M, 1536, 384, L.EPI_F16_GELU, st)
            gemm_case("fc2_resid", M, 384, 1536, L.EPI_RESID_F32, st)
            gemm_case("out_resid", M, 384, 384, L.EPI_RESID_F32, st)
    attn_case(32, 6, 1370)
