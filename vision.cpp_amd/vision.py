"""Python face of the backend: the reference's bindings/python/visioncpp/vision.py classes
(Backend, Device, Arch, Model.load/compute) on top of the same C symbols, plus the batched
extension (`Model.compute_batch`, device-resident `compute_batch_device`)."""
from __future__ import annotations

import ctypes
from ctypes import byref, c_int32, c_int64, c_size_t, c_void_p
from enum import Enum
from pathlib import Path

import numpy as np

from . import _lib as lib
from ._lib import check, get_lib, vx_check


class ImageFormat(Enum):
    rgba_u8 = 0
    bgra_u8 = 1
    argb_u8 = 2
    rgb_u8 = 3
    alpha_u8 = 4
    rgba_f32 = 5
    rgb_f32 = 6
    alpha_f32 = 7


class Backend(Enum):
    auto = 0
    cpu = 1
    gpu = 2
    vulkan = gpu | 1 << 8


class Arch(Enum):
    sam = 0
    birefnet = 1
    depth_anything = 2
    migan = 3
    esrgan = 4
    unknown = 5


class Device:
    @staticmethod
    def init(backend: Backend = Backend.auto, index: int | None = None):
        api = get_lib()
        handle = c_void_p()
        if index is None:
            check(api.visp_device_init(backend.value, byref(handle)))
        else:
            check(api.visp_hip_device_init(index, byref(handle)))
        return Device(api, handle)

    def __init__(self, api, handle):
        self._api, self._handle = api, handle

    @property
    def type(self) -> Backend:
        return Backend(self._api.visp_device_type(self._handle))

    @property
    def name(self) -> str:
        return self._api.visp_device_name(self._handle).decode()

    @property
    def description(self) -> str:
        return self._api.visp_device_description(self._handle).decode()

    def __del__(self):
        if getattr(self, "_handle", None):
            self._api.visp_device_destroy(self._handle)
            self._handle = None


_CHANNELS = {0: 4, 1: 4, 2: 4, 3: 3, 4: 1, 5: 4, 6: 3, 7: 1}


class Model:
    @classmethod
    def load(cls, path, device: Device, arch: Arch = Arch.unknown, no_upload: bool = False):
        api = get_lib()
        handle = c_void_p()
        p = lib.path_to_char_p(path)
        if arch is Arch.unknown:
            v = c_int32()
            check(api.visp_model_detect_family(p, byref(v)))
            arch = Arch(v.value)
        check(api.visp_model_load_ex(p, device._handle, arch.value, 1 if no_upload else 0, byref(handle)))
        return cls(api, handle, arch, device)

    def __init__(self, api, handle, arch: Arch, device: Device):
        self.arch, self._api, self._handle, self._device = arch, api, handle, device

    def __del__(self):
        if getattr(self, "_handle", None):
            self._api.visp_model_destroy(self._handle, self.arch.value)
            self._handle = None

    # ---- reference API: one image of any size/format -> u8 depth map (c-api.cpp:230-251)
    def compute(self, image: np.ndarray, format: ImageFormat = ImageFormat.rgb_u8, args: list[int] | None = None) -> np.ndarray:
        img = np.ascontiguousarray(image, dtype=np.uint8)
        h, w = img.shape[:2]
        view = lib.ImageView(w, h, w * _CHANNELS[format.value], format.value, img.ctypes.data)
        views = (lib.ImageView * 1)(view)
        out_view, out_data = lib.ImageView(), c_void_p()
        a = list(args or [])
        check(self._api.visp_model_compute(self._handle, self.arch.value, views, 1, (c_int32 * max(1, len(a)))(*a), len(a), byref(out_view), byref(out_data)))
        return self._take_image(out_view, out_data)

    def _take_image(self, out_view, out_data) -> np.ndarray:
        try:
            n = out_view.height * out_view.stride
            buf = (ctypes.c_uint8 * n).from_address(out_view.data)
            ch = out_view.stride // max(1, out_view.width)
            res = np.frombuffer(buf, np.uint8).reshape(out_view.height, out_view.width, ch).copy()
            if ch == 1:
                res = res[..., 0]
        finally:
            self._api.visp_image_destroy(out_data)
        return res

    def _family_fn(self, name: str):
        prefix = {Arch.esrgan: "visp_esrgan_", Arch.sam: "visp_sam_", Arch.birefnet: "visp_swin_"}.get(self.arch, "visp_depthany_")
        return getattr(self._api, prefix + name)

    # ---- batched extension
    @property
    def info(self) -> lib.DepthAnyInfo:
        i = lib.DepthAnyInfo()
        check(self._api.visp_depthany_get_info(self._handle, byref(i)))
        return i

    def image_extent(self, w: int, h: int):
        ow, oh = c_int32(), c_int32()
        check(self._api.visp_depthany_image_extent(self._handle, w, h, byref(ow), byref(oh)))
        return ow.value, oh.value

    def weights_arena(self):
        p, n = c_void_p(), c_size_t()
        check(self._family_fn("weights_arena")(self._handle, byref(p), byref(n)))
        return p.value, n.value

    def weights_ready(self):
        check(self._family_fn("weights_ready")(self._handle))

    def reserve(self, batch: int, w: int, h: int):
        check(self._api.visp_depthany_reserve(self._handle, batch, w, h))

    def use_graph(self, enable: bool = True):
        check(self._api.visp_depthany_use_graph(self._handle, int(enable)))

    def sam_set_fp8_mlp(self, enable: bool = True):
        """MobileSAM encoder, opt-in: the transformer stages' MLPs on the e4m3 matrix instruction (BASELINE.json configs[4]); several percent of
        embedding error (tests/test_fp8_decision.py rejected it for masks). Never the default."""
        check(self._api.visp_sam_set_fp8_mlp(self._handle, int(enable)))

    def set_schedule(self, schedule: int):
        """-1 = automatic (the default: the token-stationary block kernel where the model has its shape, embed dim 384 / mlp 1536 /
        head dim 64, otherwise GEMM launches); 0 = GEMM launches per op group; 1 = block kernel per layer."""
        check(self._api.visp_depthany_set_schedule(self._handle, int(schedule)))

    def set_split(self, n: int):
        """Sub-batches of one step on parallel streams: 0 = automatic, 1 = none, up to 4 (bit-identical results)."""
        check(self._api.visp_depthany_set_split(self._handle, int(n)))

    def compute_batch(self, images: np.ndarray, return_raw: bool = False):
        """images: uint8 [B, h, w, 3] on the host -> float32 [B, h, w] in [0, 1]."""
        imgs = np.ascontiguousarray(images, dtype=np.uint8)
        b, h, w, c = imgs.shape
        assert c == 3
        out = np.empty((b, h, w), np.float32)
        raw = np.empty((b, h, w), np.float32) if return_raw else None
        check(self._api.visp_depthany_compute_batch_host(self._handle, imgs.ctypes.data, b, w, h, out.ctypes.data,
                                                         raw.ctypes.data if return_raw else None))
        return (out, raw) if return_raw else out

    def compute_batch_device(self, rgb_dev: int, batch: int, w: int, h: int, out_dev: int, raw_dev: int | None = None,
                             stream: int | None = None):
        check(self._api.visp_depthany_compute_batch_device(self._handle, rgb_dev, batch, w, h, out_dev, raw_dev, stream))

    def enable_captures(self, enable: bool = True):
        check(self._family_fn("enable_captures")(self._handle, int(enable)))

    def read_capture(self, name: str) -> np.ndarray:
        n, shape = c_int64(), (c_int64 * 4)()
        f = self._family_fn("read_capture")
        check(f(self._handle, name.encode(), None, 0, byref(n), shape))
        out = np.empty(n.value, np.float32)
        check(f(self._handle, name.encode(), out.ctypes.data, n.value, byref(n), shape))
        dims = [int(d) for d in shape]
        while len(dims) > 1 and dims[-1] == 1:
            dims.pop()
        return out.reshape(dims)

    def enable_timing(self, enable: bool = True):
        check(self._family_fn("enable_timing")(self._handle, int(enable)))

    def read_timing(self):
        arr, n = (lib.Timing * 256)(), c_int32()
        check(self._family_fn("read_timing")(self._handle, arr, 256, byref(n)))
        return [dict(name=arr[i].name.decode(), ms=arr[i].ms, launches=arr[i].launches, flops=arr[i].flops, bytes=arr[i].bytes)
                for i in range(n.value)]


    # ---- BiRefNet (family 1): batched extension
    def swin_set_mask_mode(self, shifted_only: bool):
        """False (default) = the reference as written: the layer's shift mask acts in every block (swin.cpp:128-139, 226-237);
        True = shifted blocks only, as the reference's torch twin / the original Swin."""
        check(self._api.visp_swin_set_mask_mode(self._handle, int(bool(shifted_only))))

    def birefnet_image_extent(self, w: int, h: int):
        ow, oh = c_int32(), c_int32()
        check(self._api.visp_birefnet_image_extent(self._handle, w, h, byref(ow), byref(oh)))
        return ow.value, oh.value

    def segment_batch(self, images: np.ndarray) -> np.ndarray:
        """images: uint8 [B, h, w, 3] at the model extent -> sigmoid masks f32 [B, h, w] (birefnet_predict per image)."""
        imgs = np.ascontiguousarray(images, dtype=np.uint8)
        b, h, w, c = imgs.shape
        assert c == 3
        out = np.empty((b, h, w), np.float32)
        check(self._api.visp_birefnet_compute_batch_host(self._handle, imgs.ctypes.data, b, w, h, out.ctypes.data))
        return out

    def segment_batch_device(self, rgb_dev: int, batch: int, w: int, h: int, mask_dev: int, stream: int | None = None):
        check(self._api.visp_birefnet_compute_batch_device(self._handle, rgb_dev, batch, w, h, mask_dev, stream))

    # ---- ESRGAN (family 4): batched extension
    @property
    def esrgan_info(self) -> lib.EsrganInfo:
        i = lib.EsrganInfo()
        check(self._api.visp_esrgan_get_info(self._handle, byref(i)))
        return i

    def set_tile_group(self, tiles: int):
        check(self._api.visp_esrgan_set_tile_group(self._handle, tiles))

    def upscale_batch(self, images: np.ndarray, format: ImageFormat = ImageFormat.rgb_u8) -> np.ndarray:
        """images: uint8 [B, h, w, C] on the host -> rgba uint8 [B, h*scale, w*scale, 4] (esrgan_compute per image)."""
        imgs = np.ascontiguousarray(images, dtype=np.uint8)
        b, h, w, c = imgs.shape
        assert c == _CHANNELS[format.value]
        s = self.esrgan_info.scale
        out = np.empty((b, h * s, w * s, 4), np.uint8)
        check(self._api.visp_esrgan_compute_batch_host(self._handle, imgs.ctypes.data, b, w, h, format.value, out.ctypes.data))
        return out

    def upscale_batch_device(self, img_dev: int, batch: int, w: int, h: int, out_dev: int, format: ImageFormat = ImageFormat.rgb_u8,
                             stream: int | None = None):
        check(self._api.visp_esrgan_compute_batch_device(self._handle, img_dev, batch, w, h, format.value, out_dev, stream))

    def esrgan_generate(self, tiles: np.ndarray) -> np.ndarray:
        """rgb float32 tiles [n, h, w, 3] -> [n, h*scale, w*scale, 3] (esrgan_generate, no tiling / u8 conversion)."""
        t = np.ascontiguousarray(tiles, dtype=np.float32)
        n, h, w, c = t.shape
        assert c == 3
        s = self.esrgan_info.scale
        out = np.empty((n, h * s, w * s, 3), np.float32)
        check(self._api.visp_esrgan_generate_host(self._handle, t.ctypes.data, n, w, h, out.ctypes.data))
        return out

    # ---- MobileSAM image encoder (family 0)
    def sam_encode(self, image: np.ndarray, format: ImageFormat = ImageFormat.rgb_u8) -> np.ndarray:
        """sam_encode (reference vision.cpp:36-52): one u8 colour image of any extent -> embedding float32 [64, 64, 256]."""
        img = np.ascontiguousarray(image, dtype=np.uint8)
        h, w = img.shape[:2]
        view = lib.ImageView(w, h, w * _CHANNELS[format.value], format.value, img.ctypes.data)
        check(self._api.visp_sam_encode(self._handle, byref(view)))
        shape = (c_int64 * 3)()
        out = np.empty((64, 64, 256), np.float32)
        check(self._api.visp_sam_read_embedding(self._handle, out.ctypes.data, out.size, shape))
        return out.reshape([int(d) for d in shape])

    def sam_compute(self, prompt: list[int]) -> np.ndarray:
        """sam_compute (reference vision.cpp:54-92) on the last sam_encode: point [x, y] or box [x0, y0, x1, y1] -> u8 mask [h, w]."""
        out_view, out_data = lib.ImageView(), c_void_p()
        check(self._api.visp_sam_compute(self._handle, (c_int32 * len(prompt))(*prompt), len(prompt), byref(out_view), byref(out_data)))
        return self._take_image(out_view, out_data)

    def sam_read_masks(self):
        """mask logits [4, 256, 256] and iou predictions [4] of the last sam_compute."""
        masks, iou = np.empty((4, 256, 256), np.float32), (ctypes.c_float * 4)()
        check(self._api.visp_sam_read_masks(self._handle, masks.ctypes.data, masks.size, iou))
        return masks, np.array(list(iou), np.float32)

    def sam_encode_batch(self, images: np.ndarray) -> np.ndarray:
        """images: uint8 [B, 1024, 1024, 3] on the host -> image embeddings float32 [B, 64, 64, 256]."""
        imgs = np.ascontiguousarray(images, dtype=np.uint8)
        b, h, w, c = imgs.shape
        assert (h, w, c) == (1024, 1024, 3)
        out = np.empty((b, 64, 64, 256), np.float32)
        check(self._api.visp_sam_encode_batch_host(self._handle, imgs.ctypes.data, b, out.ctypes.data))
        return out

    def sam_encode_batch_device(self, rgb_dev: int, batch: int, out_dev: int, stream: int | None = None):
        check(self._api.visp_sam_encode_batch_device(self._handle, rgb_dev, batch, out_dev, stream))


class DepthPipeline:
    """Overlapped host pipeline over visp_depthany_pipeline_* (include/visp_c_api.h): submit() returns a ticket at once, wait()
    returns that batch's [batch, h, w] f32 depth maps; up to n_slots - 1 batches may be in flight."""

    def __init__(self, model: "Model", batch: int, w: int, h: int, n_slots: int = 3):
        self._api, self._model = model._api, model
        self.batch, self.w, self.h, self.n_slots = batch, w, h, n_slots
        p = c_void_p()
        check(self._api.visp_depthany_pipeline_create(model._handle, batch, w, h, n_slots, byref(p)))
        self._p = p

    def input_view(self) -> np.ndarray:
        """The next slot's pinned input buffer as a writable [batch, h, w, 3] uint8 array (fill it, then submit(None))."""
        ptr = c_void_p()
        check(self._api.visp_depthany_pipeline_input(self._p, byref(ptr)))
        n = self.batch * self.h * self.w * 3
        return np.frombuffer((ctypes.c_uint8 * n).from_address(ptr.value), np.uint8).reshape(self.batch, self.h, self.w, 3)

    def submit(self, images: np.ndarray | None) -> int:
        t = ctypes.c_int32()
        if images is not None:
            images = np.ascontiguousarray(images, dtype=np.uint8)
            assert images.shape == (self.batch, self.h, self.w, 3)
        check(self._api.visp_depthany_pipeline_submit(self._p, images.ctypes.data if images is not None else None, byref(t)))
        return t.value

    def wait(self, ticket: int, copy: bool = True) -> np.ndarray:
        ptr = c_void_p()
        check(self._api.visp_depthany_pipeline_wait(self._p, ticket, byref(ptr)))
        n = self.batch * self.h * self.w
        a = np.frombuffer((ctypes.c_float * n).from_address(ptr.value), np.float32).reshape(self.batch, self.h, self.w)
        return a.copy() if copy else a

    def close(self):
        if getattr(self, "_p", None):
            self._api.visp_depthany_pipeline_destroy(self._p)
            self._p = None

    def __del__(self):
        self.close()


def image_scale(image: np.ndarray, width: int, height: int, format: ImageFormat = ImageFormat.rgb_u8) -> np.ndarray:
    """The reference's image_scale (src/visp/image.cpp:328-356) on the host: [h, w, ch] u8 or f32 -> [height, width, ch]."""
    api = lib.get_lib()
    is_f32 = format.value >= ImageFormat.rgba_f32.value
    img = np.ascontiguousarray(image, dtype=np.float32 if is_f32 else np.uint8)
    h, w = img.shape[:2]
    ch = _CHANNELS[format.value]
    view = lib.ImageView(w, h, w * ch * img.itemsize, format.value, img.ctypes.data)
    out_view, out_data = lib.ImageView(), c_void_p()
    check(api.visp_image_scale(byref(view), width, height, byref(out_view), byref(out_data)))
    try:
        n = out_view.height * out_view.stride
        buf = (ctypes.c_uint8 * n).from_address(out_view.data)
        res = np.frombuffer(buf, img.dtype).reshape(out_view.height, out_view.width, ch).copy()
    finally:
        api.visp_image_destroy(out_data)
    return res[..., 0] if ch == 1 else res


def esrgan_tile_layout(w: int, h: int, scale: int = 1) -> dict:
    v = (c_int32 * 8)()
    check(get_lib().visp_esrgan_tile_layout(w, h, scale, v))
    return dict(zip(("image_w", "image_h", "overlap_x", "overlap_y", "n_x", "n_y", "tile_w", "tile_h"), [int(x) for x in v]))


class DeviceBuffer:
    """Raw device allocation through the vx_* runtime ABI (tests and bench use it so the hot path
    never depends on torch tensors)."""

    def __init__(self, nbytes: int):
        self._api = get_lib()
        self.nbytes = nbytes
        p = c_void_p()
        vx_check(self._api.vx_malloc(byref(p), nbytes))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, a: np.ndarray):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        vx_check(b._api.vx_memcpy_h2d(b.ptr, a.ctypes.data, a.nbytes, None))
        vx_check(b._api.vx_stream_sync(None))
        return b

    def to_numpy(self, dtype, shape) -> np.ndarray:
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        vx_check(self._api.vx_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes, None))
        return out

    def zero(self):
        vx_check(self._api.vx_memset(self.ptr, 0, self.nbytes, None))
        vx_check(self._api.vx_stream_sync(None))

    def __del__(self):
        if getattr(self, "ptr", None):
            self._api.vx_free(self.ptr)
            self.ptr = None


class SwinEncoder(Model):
    """The SWIN encoder of a birefnet GGUF on its own (reference swin_encode, src/visp/arch/swin.cpp:237-262, behind
    birefnet::encode); the whole BiRefNet (encoder + decoder) is Model.load(..., Arch.birefnet). Outputs are the four normed
    stage maps, f32 [B, h_i, w_i, C_i]."""

    @classmethod
    def load(cls, path, device: Device):
        api = get_lib()
        handle = c_void_p()
        check(api.visp_swin_load(lib.path_to_char_p(path), device._handle, byref(handle)))
        return cls(api, handle, Arch.birefnet, device)

    def output_dims(self, w: int, h: int):
        d = (c_int32 * 12)()
        check(self._api.visp_swin_output_dims(self._handle, w, h, d))
        return [(d[3 * i], d[3 * i + 1], d[3 * i + 2]) for i in range(4)]  # (w_i, h_i, C_i)

    def encode_batch(self, images: np.ndarray):
        """images: uint8 [B, h, w, 3] on the host -> list of four f32 arrays [B, h_i, w_i, C_i]."""
        imgs = np.ascontiguousarray(images, dtype=np.uint8)
        b, h, w, c = imgs.shape
        assert c == 3
        outs = [np.empty((b, hh, ww, cc), np.float32) for ww, hh, cc in self.output_dims(w, h)]
        ptrs = (c_void_p * 4)(*[o.ctypes.data for o in outs])
        check(self._api.visp_swin_encode_batch_host(self._handle, imgs.ctypes.data, b, w, h, ptrs))
        return outs

    def encode_batch_device(self, rgb_dev: int, batch: int, w: int, h: int, outs_dev, stream: int | None = None):
        check(self._api.visp_swin_encode_batch_device(self._handle, rgb_dev, batch, w, h, (c_void_p * 4)(*outs_dev), stream))
