"""Helpers for the -m gpu parity tests: numpy <-> device buffers and kernel launch wrappers
over the vx_* C ABI (include/visp_hip_kernels.h)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from visioncpp_amd import _lib as L
from visioncpp_amd.vision import DeviceBuffer


def api():
    return L.get_lib()


_live: list = []  # buffers stay allocated until release(): launches are asynchronous


def dev(a: np.ndarray) -> DeviceBuffer:
    b = DeviceBuffer.from_numpy(a)
    _live.append(b)
    return b


def empty(nbytes: int, zero=True) -> DeviceBuffer:
    b = DeviceBuffer(nbytes)
    if zero:
        b.zero()
    _live.append(b)
    return b


def release():
    sync()
    _live.clear()


def sync():
    L.vx_check(api().vx_stream_sync(None))


def pad_weight(w: np.ndarray, n_align=32, k_align=64):
    """[N, K] float -> f16 [Np, Kp] zero padded (what csrc/depthany.cpp's packer produces)."""
    n, k = w.shape
    npad, kpad = -(-n // n_align) * n_align, -(-k // k_align) * k_align
    out = np.zeros((npad, kpad), np.float16)
    out[:n, :k] = w.astype(np.float16)
    return out


def pad_vec(b: np.ndarray | None, n: int):
    if b is None:
        return None
    out = np.zeros(n, np.float32)
    out[: b.size] = b
    return out


def gemm(a_dev, w: np.ndarray, bias, M, epi, *, lda=None, out=None, ldo=None, keep=None, **kw):
    """Launches vx_gemm_f16. w: padded f16 [N, K]. Returns nothing; caller reads `out`."""
    keep = keep if keep is not None else []
    wd = dev(w)
    bd = dev(bias.astype(np.float32)) if bias is not None else None
    keep += [wd, bd]
    g = L.GemmArgs()
    g.A = a_dev.ptr
    g.lda = lda if lda is not None else w.shape[1]
    g.W = wd.ptr
    g.bias = bd.ptr if bd else None
    g.M, g.N, g.K = M, w.shape[0], w.shape[1]
    g.epi = epi
    g.out = out.ptr if out is not None else None
    g.ldo = ldo if ldo is not None else w.shape[0]
    for k, v in kw.items():
        if k.startswith("_"):
            continue
        if isinstance(v, DeviceBuffer):
            keep.append(v)
            v = v.ptr
        setattr(g, k, v)
    fn = api().vx_conv3x3_f16 if kw.get("_halo") else api().vx_gemm_f16
    L.vx_check(fn(C.byref(g), None))
    sync()
    return keep


def rel_err(got: np.ndarray, want: np.ndarray) -> float:
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-30))
