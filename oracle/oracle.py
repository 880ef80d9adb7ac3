"""ctypes front-end of the CPU ORACLE (oracle/libvisp_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg. The product (vision.cpp_amd) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_LIB = _DIR / "libvisp_oracle.so"

F32, F16, I32 = 0, 1, 26
RGBA_U8, BGRA_U8, ARGB_U8, RGB_U8, ALPHA_U8, RGBA_F32, RGB_F32, ALPHA_F32 = range(8)
LAYOUT_UNKNOWN, LAYOUT_WHCN, LAYOUT_CWHN = 0, 1, 2
GELU_GGML_F16_LUT, GELU_TANH_F32, GELU_ERF_F32 = 0, 1, 2

_NP_TYPE = {np.dtype(np.float32): F32, np.dtype(np.float16): F16, np.dtype(np.int32): I32}


class Tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("type", C.c_int32), ("ne", C.c_int64 * 4)]


class Params(C.Structure):
    _fields_ = [
        ("patch_size", C.c_int), ("embed_dim", C.c_int), ("n_layers", C.c_int), ("n_heads", C.c_int),
        ("image_size", C.c_int), ("image_multiple", C.c_int), ("feature_layers", C.c_int * 4),
        ("max_depth", C.c_float), ("gelu_mode", C.c_int),
    ]


class Capture(C.Structure):
    _fields_ = [("name", C.c_char_p), ("dst", C.POINTER(C.c_float)), ("capacity", C.c_int64), ("written", C.c_int64)]


def build(force: bool = False) -> Path:
    src = [_DIR / "visp_oracle.c", _DIR / "visp_oracle.h"]
    if force or not _LIB.exists() or any(s.stat().st_mtime > _LIB.stat().st_mtime for s in src):
        subprocess.run(["make", "-B", "-C", str(_DIR)], check=True, capture_output=True)
    return _LIB


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not _LIB.exists():
            build()
        L = C.CDLL(str(_LIB))
        fp, u8p, u16p = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_uint16)
        L.vo_last_error.restype = C.c_char_p
        L.vo_model_create.restype = C.c_void_p
        L.vo_model_create.argtypes = [C.POINTER(Tensor), C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int]
        L.vo_model_destroy.argtypes = [C.c_void_p]
        L.vo_model_tensor.restype = fp
        L.vo_model_tensor.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]
        L.vo_depthany_predict.argtypes = [C.c_void_p, C.POINTER(Params), fp, C.c_int, C.c_int, fp, C.POINTER(Capture), C.c_int]
        L.vo_depthany_compute.argtypes = [C.c_void_p, C.POINTER(Params), u8p, C.c_int, C.c_int, fp, fp]
        L.vo_dino_layer.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, fp, C.c_int64, C.c_int64]
        L.vo_f16_to_f32.argtypes = [u16p, fp, C.c_int64]
        L.vo_f32_to_f16.argtypes = [fp, u16p, C.c_int64]
        L.vo_image_u8_to_f32.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, fp, C.c_int, C.c_int, C.c_int, fp, fp, C.c_int, C.c_int]
        L.vo_image_f32_to_u8.argtypes = [fp, C.c_int, C.c_int, C.c_int, u8p, C.c_int, C.c_float, C.c_float]
        L.vo_image_normalize.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]
        L.vo_image_scale.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.vo_depthany_image_extent.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.vo_transfer_tensor.argtypes = [C.POINTER(Tensor), C.c_int, C.c_void_p, C.POINTER(C.c_int64)]
        L.vo_linear.argtypes = [fp, C.c_int64, C.c_int64, fp, fp, C.c_int64, fp]
        L.vo_layer_norm.argtypes = [fp, C.c_int64, C.c_int64, fp, fp, C.c_float, fp]
        L.vo_gelu.argtypes = [fp, fp, C.c_int64, C.c_int]
        L.vo_attention.argtypes = [fp, fp, fp, C.c_int64, C.c_int, C.c_int, C.c_float, fp]
        L.vo_conv2d_nhwc.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp]
        L.vo_conv_transpose2d_nhwc.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, fp]
        L.vo_interpolate_bilinear_nhwc.argtypes = [fp] + [C.c_int] * 7 + [fp]
        L.vo_interpolate_bicubic_nhwc.argtypes = [fp] + [C.c_int] * 7 + [fp]
        L.vo_set_num_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def _fp(a: np.ndarray | None):
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _check(ok: int):
    if not ok:
        raise RuntimeError(lib().vo_last_error().decode())


def num_threads() -> int:
    return lib().vo_num_threads()


def set_num_threads(n: int):
    lib().vo_set_num_threads(n)


# ---- scalar / image ---------------------------------------------------------------------

def f16_to_f32(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a).view(np.uint16)
    out = np.empty(a.shape, np.float32)
    lib().vo_f16_to_f32(a.ctypes.data_as(C.POINTER(C.c_uint16)), _fp(out), a.size)
    return out


def f32_to_f16(a: np.ndarray) -> np.ndarray:
    a = _f32(a)
    out = np.empty(a.shape, np.uint16)
    lib().vo_f32_to_f16(_fp(a), out.ctypes.data_as(C.POINTER(C.c_uint16)), a.size)
    return out.view(np.float16)


_CH = {RGBA_U8: 4, BGRA_U8: 4, ARGB_U8: 4, RGB_U8: 3, ALPHA_U8: 1, RGBA_F32: 4, RGB_F32: 3, ALPHA_F32: 1}


def image_u8_to_f32(src: np.ndarray, sformat: int, dformat: int, offset=(0, 0, 0, 0), scale=(1, 1, 1, 1),
                    dst_extent=None, tile_offset=(0, 0)) -> np.ndarray:
    src = np.ascontiguousarray(src, dtype=np.uint8)
    h, w = src.shape[:2]
    dw, dh = dst_extent if dst_extent else (w, h)
    out = np.empty((dh, dw, _CH[dformat]), np.float32)
    off, sc = _f32(offset), _f32(scale)
    _check(lib().vo_image_u8_to_f32(src.ctypes.data_as(C.POINTER(C.c_uint8)), w, h, w * _CH[sformat], sformat,
                                    _fp(out), dw, dh, dformat, _fp(off), _fp(sc), tile_offset[0], tile_offset[1]))
    return out


def image_f32_to_u8(src: np.ndarray, sformat: int, dformat: int, scale=1.0, offset=0.0) -> np.ndarray:
    src = _f32(src)
    h, w = src.shape[:2]
    out = np.empty((h, w, _CH[dformat]), np.uint8)
    _check(lib().vo_image_f32_to_u8(_fp(src), w, h, sformat, out.ctypes.data_as(C.POINTER(C.c_uint8)), dformat, scale, offset))
    return out


def image_normalize(src: np.ndarray, mn=0.0, mx=1.0) -> np.ndarray:
    src = _f32(src)
    h, w = src.shape[:2]
    ch = 1 if src.ndim == 2 else src.shape[2]
    out = np.empty_like(src)
    lib().vo_image_normalize(_fp(src), _fp(out), w, h, ch, mn, mx)
    return out


def image_scale(src: np.ndarray, fmt: int, ow: int, oh: int) -> np.ndarray:
    """image.cpp:328-356 (stb_image_resize semantics). src: [h, w, ch] (or [h, w]) u8 / f32 in `fmt`."""
    src = np.ascontiguousarray(src)
    h, w = src.shape[:2]
    out = np.empty((oh, ow) + src.shape[2:], src.dtype)
    _check(lib().vo_image_scale(src.ctypes.data, w, h, fmt, out.ctypes.data, ow, oh))
    return out


def depthany_image_extent(w: int, h: int, image_size=518, image_multiple=14):
    ow, oh = C.c_int(), C.c_int()
    lib().vo_depthany_image_extent(w, h, image_size, image_multiple, C.byref(ow), C.byref(oh))
    return ow.value, oh.value


# ---- tensors ------------------------------------------------------------------------------

def _mk_tensor(name: str, arr: np.ndarray, keep: list) -> Tensor:
    """numpy array in torch order (slowest axis first) -> vo_tensor in ggml order."""
    a = np.ascontiguousarray(arr)
    if a.dtype not in _NP_TYPE:
        raise TypeError(f"unsupported dtype {a.dtype} for {name}")
    shape = list(a.shape)[::-1]
    assert len(shape) <= 4
    shape += [1] * (4 - len(shape))
    nm = name.encode()
    keep += [a, nm]
    return Tensor(nm, a.ctypes.data, _NP_TYPE[a.dtype], (C.c_int64 * 4)(*shape))


def transfer_tensor(arr: np.ndarray, whcn_to_cwhn: bool) -> np.ndarray:
    keep: list = []
    t = _mk_tensor("t", arr, keep)
    out = np.empty(arr.size, np.int32 if arr.dtype == np.int32 else np.float32)
    ne = (C.c_int64 * 4)()
    _check(lib().vo_transfer_tensor(C.byref(t), int(whcn_to_cwhn), out.ctypes.data, ne))
    return out.reshape(tuple(ne)[::-1])


# ---- primitives (torch-order numpy in / out) ------------------------------------------------

def linear(x, w, b=None):
    x, w = _f32(x), _f32(w)
    b = None if b is None else _f32(b)
    M, K, N = int(np.prod(x.shape[:-1])), x.shape[-1], w.shape[0]
    y = np.empty(x.shape[:-1] + (N,), np.float32)
    lib().vo_linear(_fp(x), M, K, _fp(w), _fp(b), N, _fp(y))
    return y


def layer_norm(x, w, b, eps=1e-5):
    x, w, b = _f32(x), _f32(w), _f32(b)
    y = np.empty_like(x)
    lib().vo_layer_norm(_fp(x), int(np.prod(x.shape[:-1])), x.shape[-1], _fp(w), _fp(b), eps, _fp(y))
    return y


def gelu(x, mode=GELU_GGML_F16_LUT):
    x = _f32(x)
    y = np.empty_like(x)
    lib().vo_gelu(_fp(x), _fp(y), x.size, mode)
    return y


def attention(q, k, v, n_heads, scale):
    """q,k,v: [N, C]; returns [N, C]."""
    q, k, v = _f32(q), _f32(k), _f32(v)
    N, Cc = q.shape
    o = np.empty_like(q)
    lib().vo_attention(_fp(q), _fp(k), _fp(v), N, n_heads, Cc // n_heads, scale, _fp(o))
    return o


def conv2d_nhwc(x, w, b=None, stride=1, pad=0):
    """x [B,H,W,Cin], w [Cout,kh,kw,Cin]"""
    x, w = _f32(x), _f32(w)
    b = None if b is None else _f32(b)
    B, H, W, Cin = x.shape
    Cout, kh, kw, _ = w.shape
    OH, OW = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    y = np.empty((B, OH, OW, Cout), np.float32)
    lib().vo_conv2d_nhwc(_fp(x), B, H, W, Cin, _fp(w), _fp(b), Cout, kh, kw, stride, pad, _fp(y))
    return y


def conv_transpose2d_nhwc(x, w, b=None, stride=1):
    """x [B,H,W,Cin], w torch layout [Cin,Cout,kh,kw]"""
    x, w = _f32(x), _f32(w)
    b = None if b is None else _f32(b)
    B, H, W, Cin = x.shape
    _, Cout, kh, kw = w.shape
    y = np.empty((B, (H - 1) * stride + kh, (W - 1) * stride + kw, Cout), np.float32)
    lib().vo_conv_transpose2d_nhwc(_fp(x), B, H, W, Cin, _fp(w), _fp(b), Cout, kh, kw, stride, _fp(y))
    return y


def interpolate_nhwc(x, size, mode="bilinear", align_corners=False):
    x = _f32(x)
    B, H, W, Cc = x.shape
    OH, OW = size
    y = np.empty((B, OH, OW, Cc), np.float32)
    fn = lib().vo_interpolate_bilinear_nhwc if mode == "bilinear" else lib().vo_interpolate_bicubic_nhwc
    fn(_fp(x), B, H, W, Cc, OH, OW, int(align_corners), _fp(y))
    return y


# ---- model ------------------------------------------------------------------------------------

class Model:
    """Oracle-side model: tensors as they sit in the GGUF (numpy, torch axis order)."""

    def __init__(self, tensors: dict[str, np.ndarray], conv2d_weights: list[int], layout: str = "whcn"):
        self._keep: list = []
        arr = (Tensor * len(tensors))(*[_mk_tensor(k, v, self._keep) for k, v in tensors.items()])
        idx = (C.c_int32 * max(1, len(conv2d_weights)))(*conv2d_weights)
        lay = {"whcn": LAYOUT_WHCN, "cwhn": LAYOUT_CWHN}.get(layout, LAYOUT_UNKNOWN)
        self._h = lib().vo_model_create(arr, len(tensors), idx, len(conv2d_weights), lay)
        if not self._h:
            raise RuntimeError(lib().vo_last_error().decode())

    def __del__(self):
        if getattr(self, "_h", None):
            lib().vo_model_destroy(self._h)
            self._h = None

    def tensor(self, name: str) -> np.ndarray:
        ne = (C.c_int64 * 4)()
        p = lib().vo_model_tensor(self._h, name.encode(), ne)
        if not p:
            raise KeyError(name)
        shape = tuple(ne)[::-1]
        return np.ctypeslib.as_array(p, shape=(int(np.prod(shape)),)).reshape(shape).copy()

    def predict(self, params: Params, image_f32: np.ndarray, captures: dict[str, int] | None = None):
        """image_f32: [h, w, 3] normalised. Returns raw depth [h, w] (+ dict of captures)."""
        img = _f32(image_f32)
        h, w = img.shape[:2]
        out = np.empty((h, w), np.float32)
        caps = captures or {}
        bufs = {k: np.empty(n, np.float32) for k, n in caps.items()}
        names = [k.encode() for k in caps]
        carr = (Capture * max(1, len(caps)))(*[Capture(nm, _fp(bufs[k]), bufs[k].size, 0) for nm, k in zip(names, caps)])
        _check(lib().vo_depthany_predict(self._h, C.byref(params), _fp(img), w, h, _fp(out), carr, len(caps)))
        if captures is None:
            return out
        res = {}
        for i, k in enumerate(caps):
            if carr[i].written > bufs[k].size:
                raise RuntimeError(f"capture {k}: needs {carr[i].written} floats")
            res[k] = bufs[k][: carr[i].written].copy()
        return out, res

    def compute(self, params: Params, rgb_u8: np.ndarray):
        """rgb_u8: [h, w, 3]. Returns (normalised [h,w], raw [h,w])."""
        img = np.ascontiguousarray(rgb_u8, dtype=np.uint8)
        h, w = img.shape[:2]
        out, raw = np.empty((h, w), np.float32), np.empty((h, w), np.float32)
        _check(lib().vo_depthany_compute(self._h, C.byref(params), img.ctypes.data_as(C.POINTER(C.c_uint8)), w, h, _fp(out), _fp(raw)))
        return out, raw

    def dino_layer(self, prefix: str, x: np.ndarray, n_heads: int, gelu_mode=GELU_GGML_F16_LUT):
        x = _f32(x).copy()
        _check(lib().vo_dino_layer(self._h, prefix.encode(), n_heads, gelu_mode, _fp(x), x.shape[0], x.shape[1]))
        return x


def make_params(patch_size=14, embed_dim=384, n_layers=12, n_heads=6, image_size=518, image_multiple=14,
                feature_layers=(2, 5, 8, 11), max_depth=1.0, gelu_mode=GELU_GGML_F16_LUT) -> Params:
    return Params(patch_size, embed_dim, n_layers, n_heads, image_size, image_multiple,
                  (C.c_int * 4)(*feature_layers), max_depth, gelu_mode)


# ---- ESRGAN (reference src/visp/arch/esrgan.cpp, vision.cpp:208-253, image.cpp:612-693) -------------

class EsrganParams(C.Structure):
    _fields_ = [("scale", C.c_int), ("n_blocks", C.c_int)]


class TileLayout(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("image_w", "image_h", "overlap_x", "overlap_y", "n_x", "n_y", "tile_w", "tile_h")]


def _esr_lib():
    L = lib()
    if not getattr(L, "_esr_ready", False):
        fp, u8p = C.POINTER(C.c_float), C.POINTER(C.c_uint8)
        L.vo_esrgan_generate.argtypes = [C.c_void_p, C.POINTER(EsrganParams), fp, C.c_int, C.c_int, fp, C.POINTER(Capture), C.c_int]
        L.vo_esrgan_rdb.argtypes = [C.c_void_p, C.c_char_p, fp, C.c_int, C.c_int, C.c_int]
        L.vo_tile_layout_init.argtypes = [C.POINTER(TileLayout), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.vo_tile_scale.argtypes = [C.POINTER(TileLayout), C.c_int, C.POINTER(TileLayout)]
        L.vo_tile_merge.argtypes = [fp, fp, C.c_int, C.c_int, C.POINTER(TileLayout)]
        L.vo_esrgan_compute.argtypes = [C.c_void_p, C.POINTER(EsrganParams), u8p, C.c_int, C.c_int, C.c_int, u8p]
        L._esr_ready = True
    return L


def tile_layout(w: int, h: int, max_tile_size: int, overlap: int, align: int = 16) -> TileLayout:
    t = TileLayout()
    _esr_lib().vo_tile_layout_init(C.byref(t), w, h, max_tile_size, overlap, align)
    return t


def tile_scale(t: TileLayout, scale: int) -> TileLayout:
    r = TileLayout()
    _esr_lib().vo_tile_scale(C.byref(t), scale, C.byref(r))
    return r


def tile_merge(tile: np.ndarray, dst: np.ndarray, cx: int, cy: int, layout: TileLayout):
    """tile [tile_h, tile_w, 3] f32 blended into dst [image_h, image_w, 3] f32 (in place)."""
    t = _f32(tile)
    assert dst.dtype == np.float32 and dst.flags.c_contiguous
    _esr_lib().vo_tile_merge(_fp(t), _fp(dst), cx, cy, C.byref(layout))


def esrgan_generate(model: "Model", scale: int, n_blocks: int, x: np.ndarray, captures: dict[str, int] | None = None):
    x = _f32(x)
    h, w = x.shape[:2]
    out = np.empty((h * scale, w * scale, 3), np.float32)
    p = EsrganParams(scale, n_blocks)
    caps = captures or {}
    bufs = {k: np.empty(n, np.float32) for k, n in caps.items()}
    names = [k.encode() for k in caps]
    carr = (Capture * max(1, len(caps)))(*[Capture(nm, _fp(bufs[k]), bufs[k].size, 0) for nm, k in zip(names, caps)])
    _check(_esr_lib().vo_esrgan_generate(model._h, C.byref(p), _fp(x), w, h, _fp(out), carr, len(caps)))
    if captures is None:
        return out
    return out, {k: bufs[k][: carr[i].written].copy() for i, k in enumerate(caps)}


def esrgan_rdb(model: "Model", prefix: str, x: np.ndarray) -> np.ndarray:
    x = _f32(x).copy()
    h, w, nf = x.shape
    _check(_esr_lib().vo_esrgan_rdb(model._h, prefix.encode(), _fp(x), w, h, nf))
    return x


def esrgan_compute(model: "Model", scale: int, n_blocks: int, img_u8: np.ndarray, fmt: int = RGB_U8) -> np.ndarray:
    img = np.ascontiguousarray(img_u8, dtype=np.uint8)
    h, w = img.shape[:2]
    out = np.empty((h * scale, w * scale, 4), np.uint8)
    p = EsrganParams(scale, n_blocks)
    _check(_esr_lib().vo_esrgan_compute(model._h, C.byref(p), img.ctypes.data_as(C.POINTER(C.c_uint8)), w, h, fmt,
                                        out.ctypes.data_as(C.POINTER(C.c_uint8))))
    return out


# ---- TinyViT image encoder of MobileSAM (reference src/visp/arch/mobile-sam.cpp:20-215) --------------------------------

class TinyVitLayer(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("resolution", "embed_dim", "depth", "num_heads", "window_size", "downsample")]


class TinyVitParams(C.Structure):
    _fields_ = [("img_size", C.c_int), ("layers", TinyVitLayer * 4)]


def _tv_lib():
    L = lib()
    if not getattr(L, "_tv_ready", False):
        fp = C.POINTER(C.c_float)
        L.vo_tinyvit_encode.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(TinyVitParams), fp, fp, C.POINTER(Capture), C.c_int]
        L.vo_tinyvit_block.argtypes = [C.c_void_p, C.c_char_p, fp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.vo_attention_rel_bias.argtypes = [C.c_void_p, C.c_char_p, fp, C.c_int, C.c_int, C.c_int, C.c_int, fp]
        L.vo_conv2d_depthwise_nhwc.argtypes = [fp, C.c_int, C.c_int, C.c_int, fp, fp, C.c_int, C.c_int, C.c_int, fp]
        L.vo_tinyvit_set_gelu_modes.argtypes = [C.c_int, C.c_int]
        L._tv_ready = True
    return L


def tinyvit_set_gelu_modes(mbconv_mode: int = GELU_GGML_F16_LUT, other_mode: int = GELU_GGML_F16_LUT):
    _tv_lib().vo_tinyvit_set_gelu_modes(mbconv_mode, other_mode)


def tinyvit_params(img_size: int, layers) -> TinyVitParams:
    p = TinyVitParams()
    p.img_size = img_size
    for i, l in enumerate(layers):
        p.layers[i] = TinyVitLayer(*l)
    return p


def tinyvit_encode(model: "Model", params: TinyVitParams, image: np.ndarray, prefix: str = "enc", captures: dict[str, int] | None = None):
    """image: normalised rgb f32 [img, img, 3] -> [res, res, 256]"""
    x = _f32(image)
    res = params.layers[3].resolution
    out = np.empty((res, res, 256), np.float32)
    caps = captures or {}
    bufs = {k: np.empty(n, np.float32) for k, n in caps.items()}
    names = [k.encode() for k in caps]
    carr = (Capture * max(1, len(caps)))(*[Capture(nm, _fp(bufs[k]), bufs[k].size, 0) for nm, k in zip(names, caps)])
    _check(_tv_lib().vo_tinyvit_encode(model._h, prefix.encode(), C.byref(params), _fp(x), _fp(out), carr, len(caps)))
    if captures is None:
        return out
    return out, {k: bufs[k][: carr[i].written].copy() for i, k in enumerate(caps)}


def tinyvit_block(model: "Model", prefix: str, x: np.ndarray, res: int, heads: int, window: int) -> np.ndarray:
    x = _f32(x).copy()
    assert x.shape[0] == res * res
    _check(_tv_lib().vo_tinyvit_block(model._h, prefix.encode(), _fp(x), res, x.shape[1], heads, window))
    return x


def attention_rel_bias(model: "Model", prefix: str, x: np.ndarray, heads: int) -> np.ndarray:
    x = _f32(x)
    n_win, N, dim = x.shape
    y = np.empty_like(x)
    _check(_tv_lib().vo_attention_rel_bias(model._h, prefix.encode(), _fp(x), n_win, N, dim, heads, _fp(y)))
    return y


def conv2d_depthwise_nhwc(x: np.ndarray, w: np.ndarray, b: np.ndarray | None, stride: int, pad: int) -> np.ndarray:
    """x [H,W,C]; w [k,k,C] (tap-major, channel contiguous)"""
    x, w = _f32(x), _f32(w)
    H, W_, Cc = x.shape
    k = w.shape[0]
    OH, OW = (H + 2 * pad - k) // stride + 1, (W_ + 2 * pad - k) // stride + 1
    y = np.empty((OH, OW, Cc), np.float32)
    _tv_lib().vo_conv2d_depthwise_nhwc(_fp(x), H, W_, Cc, _fp(w), _fp(_f32(b)) if b is not None else None, k, stride, pad, _fp(y))
    return y


# ---- MobileSAM prompt encoder + mask decoder (sam_compute) ----------------------------------------------------------

def _sam_lib():
    L = lib()
    if not getattr(L, "_sam_ready", False):
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
        L.vo_sam_process_prompt.argtypes = [ip, C.c_int, C.c_int, C.c_int, C.c_int, fp]
        L.vo_sam_process_prompt.restype = None
        L.vo_sam_embed_prompt.argtypes = [C.c_void_p, fp, C.c_int, fp, ip]
        L.vo_sam_predict_masks.argtypes = [C.c_void_p, fp, C.c_int, C.c_int, fp, C.c_int, fp, fp]
        L.vo_sam_process_mask.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.vo_sam_process_mask.restype = None
        L.vo_sam_compute.argtypes = [C.c_void_p, fp, C.c_int, C.c_int, C.c_int, C.c_int, ip, C.c_int, C.c_void_p, fp, fp]
        L._sam_ready = True
    return L


def sam_process_prompt(prompt, image_w: int, image_h: int, image_size: int = 1024) -> np.ndarray:
    """pixel point (x, y) or box (x0, y0, x1, y1) of the original image -> [-1, 1] coordinates"""
    p = (C.c_int * len(prompt))(*[int(v) for v in prompt])
    out = np.zeros(4, np.float32)
    _sam_lib().vo_sam_process_prompt(p, len(prompt), image_w, image_h, image_size, _fp(out))
    return out


def sam_embed_prompt(model: "Model", coords: np.ndarray, is_box: bool) -> np.ndarray:
    c = _f32(coords)
    out, dim = np.zeros((2, 512), np.float32), C.c_int()
    _check(_sam_lib().vo_sam_embed_prompt(model._h, _fp(c), int(is_box), _fp(out), C.byref(dim)))
    return out.reshape(-1)[: 2 * dim.value].reshape(2, dim.value).copy()


def sam_predict_masks(model: "Model", embed: np.ndarray, sparse: np.ndarray):
    """embed NHWC [res, res, dim], sparse [n, dim] -> masks [4, 4 res, 4 res], iou [4]"""
    e, s = _f32(embed), _f32(sparse)
    res, dim = e.shape[0], e.shape[2]
    masks, iou = np.empty((4, 4 * res, 4 * res), np.float32), np.empty(4, np.float32)
    _check(_sam_lib().vo_sam_predict_masks(model._h, _fp(e), res, dim, _fp(s), s.shape[0], _fp(masks), _fp(iou)))
    return masks, iou


def sam_process_mask(mask: np.ndarray, target_w: int, target_h: int, image_size: int = 1024) -> np.ndarray:
    mk = _f32(mask)
    out = np.empty((target_h, target_w), np.uint8)
    _sam_lib().vo_sam_process_mask(_fp(mk), mk.shape[0], image_size, target_w, target_h, out.ctypes.data)
    return out


def sam_compute(model: "Model", embed: np.ndarray, image_w: int, image_h: int, prompt, return_all: bool = False):
    """sam_compute after sam_encode: alpha_u8 mask [image_h, image_w] (+ iou [4], masks [4, 4 res, 4 res])"""
    e = _f32(embed)
    res, dim = e.shape[0], e.shape[2]
    p = (C.c_int * len(prompt))(*[int(v) for v in prompt])
    out = np.empty((image_h, image_w), np.uint8)
    iou, masks = np.empty(4, np.float32), np.empty((4, 4 * res, 4 * res), np.float32)
    _check(_sam_lib().vo_sam_compute(model._h, _fp(e), res, dim, image_w, image_h, p, len(prompt), out.ctypes.data, _fp(iou), _fp(masks)))
    return (out, iou, masks) if return_all else out


# ---- SWIN transformer encoder, the BiRefNet backbone (reference src/visp/arch/swin.cpp) ------------------------------

class SwinParams(C.Structure):
    _fields_ = [("embed_dim", C.c_int), ("window_size", C.c_int), ("depths", C.c_int * 4), ("n_heads", C.c_int * 4)]


def swin_params(embed_dim=96, window_size=7, depths=(2, 2, 6, 2), n_heads=(3, 6, 12, 24)) -> SwinParams:
    """Defaults = swin_t_params (swin.cpp:266-275)."""
    return SwinParams(embed_dim, window_size, (C.c_int * 4)(*depths), (C.c_int * 4)(*n_heads))


def _swin_lib():
    L = lib()
    if not getattr(L, "_swin_ready", False):
        fp = C.POINTER(C.c_float)
        L.vo_swin_rel_pos_index.argtypes = [C.c_int, C.POINTER(C.c_int32)]
        L.vo_swin_attention_mask.argtypes = [C.c_int, C.c_int, C.c_int, fp]
        L.vo_swin_block.argtypes = [C.c_void_p, C.c_char_p, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp]
        L.vo_swin_patch_merging.argtypes = [C.c_void_p, C.c_char_p, fp, C.c_int, C.c_int, C.c_int, C.POINTER(fp), C.POINTER(C.c_int)]
        L.vo_swin_encode.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(SwinParams), fp, C.c_int, C.c_int, fp * 4, (C.c_int * 3) * 4,
                                     C.POINTER(Capture), C.c_int]
        L.vo_free.argtypes = [C.c_void_p]
        L.vo_swin_set_mask_mode.argtypes = [C.c_int]
        L.vo_swin_get_mask_mode.restype = C.c_int
        L._swin_ready = True
    return L


def swin_rel_pos_index(window: int) -> np.ndarray:
    out = np.empty(window ** 4, np.int32)
    _swin_lib().vo_swin_rel_pos_index(window, out.ctypes.data_as(C.POINTER(C.c_int32)))
    return out


def swin_attention_mask(w: int, h: int, window: int) -> np.ndarray:
    """[n_windows, ws^2, ws^2] of 0 / -inf (swin.cpp:165-213)."""
    nw = ((w + window - 1) // window) * ((h + window - 1) // window)
    out = np.empty((nw, window * window, window * window), np.float32)
    _swin_lib().vo_swin_attention_mask(w, h, window, _fp(out))
    return out


def swin_set_mask_mode(shifted_only: bool) -> None:
    """False (default) = the reference as written: the layer's shift mask acts in every block (swin.cpp:128-139, 226-237);
    True = shifted blocks only (the reference's torch twin / HuggingFace Swin, which tests/golden/swin_mini.npz was made with)."""
    _swin_lib().vo_swin_set_mask_mode(int(bool(shifted_only)))


def swin_get_mask_mode() -> bool:
    return bool(_swin_lib().vo_swin_get_mask_mode())


def swin_block(model: "Model", prefix: str, x: np.ndarray, w: int, h: int, heads: int, window: int, shift: int,
               masked: bool | None = None) -> np.ndarray:
    """x: tokens [h*w, C] (row = y*w + x) -> same shape. masked: apply the layer's shift mask (None = as the current mask mode
    would: always for shifted blocks, for unshifted ones only in the reference-as-written mode)."""
    x = _f32(x).copy()
    assert x.shape[0] == w * h
    if masked is None:
        masked = shift > 0 or not swin_get_mask_mode()
    mask = swin_attention_mask(w, h, window) if masked else None
    _check(_swin_lib().vo_swin_block(model._h, prefix.encode(), _fp(x), w, h, x.shape[1], heads, window, shift, _fp(mask)))
    return x


def swin_patch_merging(model: "Model", prefix: str, x: np.ndarray, w: int, h: int) -> np.ndarray:
    x = _f32(x)
    fp = C.POINTER(C.c_float)
    out, co = fp(), C.c_int()
    _check(_swin_lib().vo_swin_patch_merging(model._h, prefix.encode(), _fp(x), w, h, x.shape[1], C.byref(out), C.byref(co)))
    n = (w // 2) * (h // 2)
    res = np.ctypeslib.as_array(out, shape=(n * co.value,)).reshape(n, co.value).copy()
    _swin_lib().vo_free(out)
    return res


def swin_encode(model: "Model", params: SwinParams, image: np.ndarray, prefix: str = "bb", captures: dict[str, int] | None = None):
    """image: normalised rgb f32 [H, W, 3] -> list of four normed stage outputs [h_i, w_i, C_i] (swin.cpp:237-262)."""
    img = _f32(image)
    H, W = img.shape[:2]
    fp = C.POINTER(C.c_float)
    outs = (fp * 4)()
    dims = ((C.c_int * 3) * 4)()
    caps = captures or {}
    bufs = {k: np.empty(n, np.float32) for k, n in caps.items()}
    names = [k.encode() for k in caps]
    carr = (Capture * max(1, len(caps)))(*[Capture(nm, _fp(bufs[k]), bufs[k].size, 0) for nm, k in zip(names, caps)])
    _check(_swin_lib().vo_swin_encode(model._h, prefix.encode(), C.byref(params), _fp(img), W, H, outs, dims, carr, len(caps)))
    res = []
    for i in range(4):
        w, h, c = dims[i][0], dims[i][1], dims[i][2]
        res.append(np.ctypeslib.as_array(outs[i], shape=(h * w * c,)).reshape(h, w, c).copy())
        _swin_lib().vo_free(outs[i])
    if captures is None:
        return res
    return res, {k: bufs[k][: carr[i].written].copy() for i, k in enumerate(caps)}


# ---- BiRefNet (reference src/visp/arch/birefnet.cpp); the deformable convolution is parity-unpinned (see visp_oracle.c) ----

def _bf_lib():
    L = _swin_lib()
    if not getattr(L, "_bf_ready", False):
        fp = C.POINTER(C.c_float)
        L.vo_deform_conv2d_nhwc.argtypes = [fp, C.c_int, C.c_int, C.c_int, fp, C.c_int, C.c_int, C.c_int, fp, fp, C.c_int, C.c_int, fp]
        L.vo_birefnet_encode.argtypes = [C.c_void_p, C.POINTER(SwinParams), fp, C.c_int, C.c_int, fp * 4, (C.c_int * 3) * 4]
        L.vo_birefnet_predict.argtypes = [C.c_void_p, C.POINTER(SwinParams), fp, C.c_int, C.c_int, fp, C.POINTER(Capture), C.c_int]
        L.vo_image_to_patches.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp]
        L._bf_ready = True
    return L


def deform_conv2d_nhwc(x, w, offset, mask=None, stride=1, pad=0):
    """x [H, W, Cin], w [Cout, kh, kw, Cin], offset [OH, OW, 2*kh*kw] (dy, dx per tap), mask [OH, OW, kh*kw] -> [OH, OW, Cout]."""
    x, w, offset = _f32(x), _f32(w), _f32(offset)
    mask = _f32(mask) if mask is not None else None
    H, W, Cin = x.shape
    Cout, kh, kw, _ = w.shape
    OH, OW = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    y = np.empty((OH, OW, Cout), np.float32)
    _bf_lib().vo_deform_conv2d_nhwc(_fp(x), H, W, Cin, _fp(w), Cout, kh, kw, _fp(offset), _fp(mask), stride, pad, _fp(y))
    return y


def birefnet_encode(model: "Model", params: SwinParams, image: np.ndarray):
    """normalised rgb f32 [H, W, 3] -> the four concatenated encoder features [h_i, w_i, C_i] (birefnet.cpp:43-73)."""
    img = _f32(image)
    H, W = img.shape[:2]
    fp = C.POINTER(C.c_float)
    outs, dims = (fp * 4)(), ((C.c_int * 3) * 4)()
    _check(_bf_lib().vo_birefnet_encode(model._h, C.byref(params), _fp(img), W, H, outs, dims))
    res = []
    for i in range(4):
        w, h, c = dims[i][0], dims[i][1], dims[i][2]
        res.append(np.ctypeslib.as_array(outs[i], shape=(h * w * c,)).reshape(h, w, c).copy())
        _bf_lib().vo_free(outs[i])
    return res


def birefnet_predict(model: "Model", params: SwinParams, image: np.ndarray, captures: dict[str, int] | None = None):
    """normalised rgb f32 [H, W, 3] -> sigmoid mask [H, W] (birefnet_predict, birefnet.cpp:252-260)."""
    img = _f32(image)
    H, W = img.shape[:2]
    out = np.empty((H, W), np.float32)
    caps = captures or {}
    bufs = {k: np.empty(n, np.float32) for k, n in caps.items()}
    names = [k.encode() for k in caps]
    carr = (Capture * max(1, len(caps)))(*[Capture(nm, _fp(bufs[k]), bufs[k].size, 0) for nm, k in zip(names, caps)])
    _check(_bf_lib().vo_birefnet_predict(model._h, C.byref(params), _fp(img), W, H, _fp(out), carr, len(caps)))
    if captures is None:
        return out
    return out, {k: bufs[k][: carr[i].written].copy() for i, k in enumerate(caps)}


def image_to_patches(image: np.ndarray, w: int, h: int) -> np.ndarray:
    """[IH, IW, C] -> [h, w, gw*gh*C] (birefnet.cpp:158-167)."""
    img = _f32(image)
    IH, IW, Cc = img.shape
    out = np.empty((h, w, (IW // w) * (IH // h) * Cc), np.float32)
    _bf_lib().vo_image_to_patches(_fp(img), IW, IH, Cc, w, h, _fp(out))
    return out
