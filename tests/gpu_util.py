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


# ---- ESRGAN dense-block conv (vx_dconv3x3_f16) -----------------------------------------------------------------

def pack_dconv(w: np.ndarray, cin_pad: int | None = None, cout_pad: int | None = None) -> np.ndarray:
    """torch conv weight [Cout, Cin, 3, 3] -> f16 [cin/32][9][cout][32] with the four 16-byte groups of each
    (tap, n) row at position g ^ ((n >> 2) & 3) -- the layout include/visp_hip_kernels.h documents."""
    co, ci = w.shape[:2]
    cip = cin_pad or -(-ci // 32) * 32
    cop = cout_pad or -(-co // 32) * 32
    wp = np.zeros((cop, cip, 3, 3), np.float16)
    wp[:co, :ci] = w.astype(np.float16)
    t = wp.reshape(cop, cip // 32, 4, 8, 9).transpose(1, 4, 0, 2, 3)  # [chunk][tap][n][g][8]
    out = np.empty_like(t)
    n = np.arange(cop)
    for g in range(4):
        out[:, :, n, g ^ ((n >> 2) & 3)] = t[:, :, n, g]
    return np.ascontiguousarray(out)


def to_planes(x: np.ndarray) -> np.ndarray:
    """[B,H,W,C] -> [C/32][B,H,W,32] (the planar activation layout of vx_dconv3x3_f16)."""
    B, H, W, C = x.shape
    return np.ascontiguousarray(x.reshape(B, H, W, C // 32, 32).transpose(3, 0, 1, 2, 4))


def from_planes(p: np.ndarray) -> np.ndarray:
    P, B, H, W, _ = p.shape
    return np.ascontiguousarray(p.transpose(1, 2, 3, 0, 4).reshape(B, H, W, P * 32))


def dconv(x_dev, n_planes, cin, B, H, W, w: np.ndarray, bias, *, up2=False, act=0, out=None, out_planes=None, out_plane0=0,
          res1=None, s1=1.0, res2=None, s2=1.0, rgb=False, cin_pad=None, x_residual=False, nhwc=None, a_relu=False, head=None, bil=None):
    """Launches vx_dconv3x3_f16 on planar buffers (x_dev: n_planes planes of [B,H(/2),W(/2),32]); the output goes to
    planes out_plane0.. of `out` (out_planes planes of [B,H,W,32]). Returns all planes of the output buffer as
    [B,H,W,32*planes] f16, or f32 [B,H,W,3] for the rgb head. res1/res2: planar device buffers [cout/32][B,H,W,32]."""
    cout = w.shape[0]
    cop = -(-cout // 32) * 32
    wd = dev(pack_dconv(w, cin_pad or cin, cop))
    bd = dev(pad_vec(bias, cop)) if bias is not None else None
    src_px = B * (H // 2 if up2 else H) * (W // 2 if up2 else W) * 32
    dst_px = B * H * W * 32
    if rgb:
        ob = out or empty(B * H * W * 3 * 4)
    else:
        out_planes = out_planes or cop // 32
        ob = out or empty(out_planes * dst_px * 2)
    a = L.DconvArgs()
    a.x, a.x_plane, a.cin, a.up2 = x_dev.ptr, src_px, cin, int(up2)
    a.B, a.H, a.W = B, H, W
    a.w, a.bias, a.cout = wd.ptr, (bd.ptr if bd else None), cop
    a.epi, a.act = (L.DC_RGB_F32 if rgb else L.DC_F16), act
    a.s1, a.res1, a.res1_plane = s1, (res1.ptr if res1 else None), dst_px
    a.s2, a.res2, a.res2_plane = s2, (res2.ptr if res2 else None), dst_px
    a.out, a.out_plane = ob.ptr + out_plane0 * dst_px * 2, dst_px
    a.x_residual = int(x_residual)
    a.a_relu = int(a_relu)
    if nhwc:  # (channels of the input map, channels of the output/residual maps): NHWC buffers instead of planes
        cx, co = nhwc
        a.x_pix, a.x_plane = cx, 32
        a.out_pix, a.out_plane, a.res1_pix, a.res1_plane, a.res2_pix, a.res2_plane = co, 32, co, 32, co, 32
    if bil is not None:  # x_dev is the low-resolution map [B, hs, ws, cin]: bilinear (align_corners) resize in the halo loader
        a.bil_hs, a.bil_ws = bil
    if head is not None:
        w3, b3, scale = head
        hw = dev(np.asarray(w3, np.float32))
        a.epi, a.head_w, a.head_bias, a.head_scale = L.DC_HEAD_F32, hw.ptr, b3, scale
        ob = empty(B * H * W * 4)
        a.out = ob.ptr
    L.vx_check(api().vx_dconv3x3_f16(C.byref(a), None))
    sync()
    if head is not None:
        return ob.to_numpy(np.float32, (B, H, W))
    if nhwc:
        return ob.to_numpy(np.float16, (B, H, W, nhwc[1]))
    if rgb:
        return ob.to_numpy(np.float32, (B, H, W, 3))
    return from_planes(ob.to_numpy(np.float16, (out_planes, B, H, W, 32)))


def pack_rcu(w: np.ndarray) -> np.ndarray:
    """[64, kh, kw, 64] float -> f16 [taps][64 n][64 c] with the 16-byte groups of row n at position g ^ ((n >> 1) & 7) (kernels_rcu.hip's slab image)."""
    n, kh, kw, c = w.shape
    assert n == 64 and c == 64
    t = w.astype(np.float16).transpose(1, 2, 0, 3).reshape(kh * kw, 64, 8, 8)  # [tap][n][group][8]
    out = np.empty_like(t)
    for row in range(64):
        for g in range(8):
            out[:, row, g ^ ((row >> 1) & 7)] = t[:, row, g]
    return np.ascontiguousarray(out.reshape(kh * kw, 64, 64))
