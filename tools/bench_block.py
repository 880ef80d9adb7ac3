#!/usr/bin/env python3
"""Micro-benchmark of the token-stationary DINOv2 block kernel (vx_dino_block16_f16) at the north-star shape
(M = 32 x 1370 tokens), next to the launches it replaces (LayerNorm + QKV / out-proj / fc1 / fc2 GEMMs)."""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
from bench_kernels import api, gemm_case, stream, timeit, L, DeviceBuffer  # noqa: E402

D, HID, H, T = 384, 1536, 6, 1370


def block_case(B, mlp=True, tap=False, qkv=True, fn="vx_dino_block16_f16"):
    M = B * T
    rng = np.random.default_rng(0)
    f16 = lambda *s, sc=1.0: np.ascontiguousarray((rng.standard_normal(s) * sc).astype(np.float16))  # noqa: E731
    wo, w1, w2, wq = f16(D, D, sc=D ** -0.5), f16(HID, D, sc=D ** -0.5), f16(D, HID, sc=HID ** -0.5), f16(3 * D, D, sc=D ** -0.5)
    pm = np.zeros(api.vx_dino_block_mlp_bytes() // 2, np.uint16)
    pq = np.zeros(api.vx_dino_block_qkv_bytes() // 2, np.uint16)
    stem = fn[:-len("_f16")]
    L.vx_check(getattr(api, stem + "_pack_mlp")(wo.ctypes.data, w1.ctypes.data, w2.ctypes.data, pm.ctypes.data))
    L.vx_check(getattr(api, stem + "_pack_qkv")(wq.ctypes.data, pq.ctypes.data))
    vm = np.concatenate([rng.standard_normal(384) * .1, np.full(384, .1), np.ones(384), np.zeros(384), rng.standard_normal(1536) * .1,
                         rng.standard_normal(384) * .1, np.full(384, .1)]).astype(np.float32)
    vq = np.concatenate([np.ones(384), np.zeros(384), rng.standard_normal(1152) * .1]).astype(np.float32)
    vt = np.concatenate([np.ones(384), np.zeros(384)]).astype(np.float32)
    bufs = [DeviceBuffer.from_numpy(a) for a in (pm, pq, vm, vq, vt)]
    x = DeviceBuffer.from_numpy(rng.standard_normal((M, D)).astype(np.float32))
    att = DeviceBuffer.from_numpy(f16(M, D))
    feat = DeviceBuffer(M * D * 2)
    q, k, v = (DeviceBuffer(M * D * 2) for _ in range(3))
    a = L.DinoBlockArgs()
    a.x, a.M, a.T, a.H, a.q_scale, a.eps = x.ptr, M, T, H, 0.125, 1e-6
    if mlp:
        a.att, a.w_mlp, a.vec_mlp = att.ptr, bufs[0].ptr, bufs[2].ptr
    if tap:
        a.feat, a.vec_tap = feat.ptr, bufs[4].ptr
    if qkv:
        a.q, a.k, a.v, a.w_qkv, a.vec_qkv = q.ptr, k.ptr, v.ptr, bufs[1].ptr, bufs[3].ptr
    launch = getattr(api, fn)
    ms = timeit(lambda: L.vx_check(launch(C.byref(a), stream)))
    flops = 2.0 * M * D * ((D + 2 * HID) * mlp + 3 * D * qkv)
    print(f"{fn} B={B} M={M} mlp={int(mlp)} tap={int(tap)} qkv={int(qkv)}: {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.1f} TFLOP/s  ({-(-M // 128)} workgroups)", flush=True)
    return ms


if __name__ == "__main__" and "--pmc" in sys.argv:
    # target of the rocprofv3 --pmc passes (profiles/r02_pmc/): the north-star launch shapes once each, nothing else
    from bench_kernels import attn_case
    # (batch 32 runs as three sub-batches of 11 / 11 / 10 images on parallel streams: these are the shapes of its launches)
    block_case(11, fn="vx_dino_block16_f16")
    block_case(11, mlp=False, fn="vx_dino_block16_f16")
    attn_case(11, 6, 1370)
    block_case(32, fn="vx_dino_block16_f16")
    attn_case(32, 6, 1370)
    sys.exit(0)

if __name__ == "__main__" and "--l2" in sys.argv:
    # target of the L2 counter passes (tools/pmc_block_l2.sh): the block kernel at 118 / 247 / 343 workgroups, nothing else
    for B in (11, 23, 32):
        block_case(B, fn="vx_dino_block16_f16")
    sys.exit(0)

if __name__ == "__main__" and "--only16" in sys.argv:
    for B in (23, 32, 11):
        block_case(B, fn="vx_dino_block16_f16")
    sys.exit(0)

if __name__ == "__main__" and "--stamps" not in sys.argv:
    for rnd in range(2):
        for f16 in ("vx_dino_block16_f16",):
            block_case(23, fn=f16)
            block_case(32, fn=f16)
            block_case(11, fn=f16)
            block_case(32, mlp=False, fn=f16)
            block_case(32, qkv=False, tap=True, fn=f16)
    if "--gemms" in sys.argv:
        M = 32 * T
        gemm_case("qkv", M, 1152, 384, L.EPI_QKV, 0)
        gemm_case("fc1_gelu", M, 1536, 384, L.EPI_F16_GELU, 0)
        gemm_case("fc2_resid", M, 384, 1536, L.EPI_RESID_F32, 0)
        gemm_case("out_resid", M, 384, 384, L.EPI_RESID_F32, 0)


def block_stamps(B, fn="vx_dino_block16_f16"):
    """Per-workgroup phase anatomy from in-kernel s_memtime stamps (diagnostic)."""
    M = B * T
    rng = np.random.default_rng(0)
    f16 = lambda *s, sc=1.0: np.ascontiguousarray((rng.standard_normal(s) * sc).astype(np.float16))  # noqa: E731
    wo, w1, w2, wq = f16(D, D, sc=D ** -0.5), f16(HID, D, sc=D ** -0.5), f16(D, HID, sc=HID ** -0.5), f16(3 * D, D, sc=D ** -0.5)
    pm = np.zeros(api.vx_dino_block_mlp_bytes() // 2, np.uint16)
    pq = np.zeros(api.vx_dino_block_qkv_bytes() // 2, np.uint16)
    stem = fn[:-len("_f16")]
    L.vx_check(getattr(api, stem + "_pack_mlp")(wo.ctypes.data, w1.ctypes.data, w2.ctypes.data, pm.ctypes.data))
    L.vx_check(getattr(api, stem + "_pack_qkv")(wq.ctypes.data, pq.ctypes.data))
    vm = np.concatenate([np.zeros(384), np.full(384, .1), np.ones(384), np.zeros(384), np.zeros(1536), np.zeros(384), np.full(384, .1)]).astype(np.float32)
    vq = np.concatenate([np.ones(384), np.zeros(384), np.zeros(1152)]).astype(np.float32)
    bufs = [DeviceBuffer.from_numpy(a) for a in (pm, pq, vm, vq)]
    x = DeviceBuffer.from_numpy(rng.standard_normal((M, D)).astype(np.float32))
    att = DeviceBuffer.from_numpy(f16(M, D))
    q, k, v = (DeviceBuffer(M * D * 2) for _ in range(3))
    nblk = -(-M // 128)
    st = DeviceBuffer(nblk * 128)
    st.zero()
    a = L.DinoBlockArgs()
    a.x, a.M, a.T, a.H, a.q_scale, a.eps = x.ptr, M, T, H, 0.125, 1e-6
    a.att, a.w_mlp, a.vec_mlp = att.ptr, bufs[0].ptr, bufs[2].ptr
    a.q, a.k, a.v, a.w_qkv, a.vec_qkv = q.ptr, k.ptr, v.ptr, bufs[1].ptr, bufs[3].ptr
    launch = getattr(api, fn)
    for _ in range(50):
        L.vx_check(launch(C.byref(a), stream))
    a.stamps = st.ptr
    L.vx_check(launch(C.byref(a), stream))
    L.vx_check(api.vx_stream_sync(stream))
    t = st.to_numpy(np.uint64, (nblk, 16)).astype(np.int64)
    names = ["prologue (att, vectors, slabs 0-3)", "to first tile", "out-proj 12 tiles", "LN2", "MLP 96 slabs", "fc2 epilogue (x re-read, write)",
             "LN stats", "(tap)", "LN1 -> frags", "QKV 36 tiles"]
    print(f"--- {fn} B={B}: {nblk} workgroups; span {t[:, 10].max() - t[:, 0].min()} ticks")
    for i, nm in enumerate(names):
        d = t[:, i + 1] - t[:, i]
        print(f"   {nm:36s} median {np.median(d):9.0f}  p10 {np.percentile(d, 10):9.0f}  p90 {np.percentile(d, 90):9.0f}")
    life = t[:, 10] - t[:, 0]
    print(f"   {'workgroup lifetime':36s} median {np.median(life):9.0f}  p10 {np.percentile(life, 10):9.0f}  p90 {np.percentile(life, 90):9.0f}")
    if t[:, 13].max() > 0:  # -DVISP_BLOCK16_DBG=1024: waits of the two waves of SIMD 0 at the boundaries
        for nm, col in (("wave 0", 11), ("wave 4", 12)):
            vm, bar = t[:, col] >> 32, t[:, col] & 0xffffffff
            print(f"   {nm}: {int(np.median(t[:, 13]))} boundaries; waiting for its copies: median {np.median(vm):9.0f} cycles in total, at the barrier: {np.median(bar):9.0f}")
    rt = (t[:, 15] - t[:, 14]).astype(np.float64)
    ok = rt > 0
    print(f"   shader clock: median {np.median(life[ok] / rt[ok] * 100.0):.0f} MHz")
    starts = np.sort(t[:, 0] - t[:, 0].min())
    print("   start ticks (deciles):", [int(np.percentile(starts, p)) for p in range(0, 101, 10)])


if __name__ == "__main__" and "--stamps16" in sys.argv:
    block_stamps(int(os.environ.get("STAMPS_B", "11")), "vx_dino_block16_f16")
    sys.exit(0)

if __name__ == "__main__" and "--stamps" in sys.argv:
    block_stamps(11)
    block_stamps(23)
