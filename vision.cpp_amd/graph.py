"""Graph layer: Python face of the `visp_graph_*` C entries (include/visp_c_api.h), with the reference's builder vocabulary --
`ModelRef` = `model_ref` (include/visp/ml.h:199-245: weights by name under a prefix, `m["sub"]`, `m[i]`), the `nn.h` functions
(`linear`, `layer_norm`, `conv_2d`, `conv_transpose_2d`, `patch_embed`, `attention`), `slice_` / `concat` / `interpolate`
(ml.cpp:746-788) and the raw element-wise / shape ops the reference's arch code calls on ggml directly. Shapes are ggml's ne
order (ne[0] contiguous); 2D maps are CWHN (`model_build_flag::cwhn`), so the layout helpers of nn.h are identities here.

`depthany_predict` below is Depth-Anything-V2 written against this layer the way the reference's arch code is
(src/visp/arch/dino.cpp, src/visp/arch/depth-anything.cpp): the generic path through the executor, next to the hand-scheduled
step of csrc/depthany.cpp that the benchmark measures."""
from __future__ import annotations

import ctypes
import math
from ctypes import byref, c_float, c_int32, c_int64, c_void_p

import numpy as np

from . import _lib as lib
from ._lib import check, get_lib

F32, F16, U8 = 0, 1, 24
(OP_LINEAR, OP_LAYER_NORM, OP_GELU, OP_RELU, OP_SCALE, OP_ADD, OP_MUL, OP_CONV_2D, OP_CONV_TRANSPOSE_2D, OP_INTERPOLATE, OP_ATTENTION,
 OP_CONCAT, OP_SLICE, OP_RESHAPE, OP_REPEAT, OP_PATCH_EMBED, OP_CONT, OP_IMAGE_U8_TO_F32, OP_IMAGE_NORMALIZE, OP_LEAKY_RELU) = range(2, 22)
SCALE_MODE_NEAREST, SCALE_MODE_BILINEAR, SCALE_MODE_BICUBIC, SCALE_FLAG_ALIGN_CORNERS = 0, 1, 2, 256  # ggml's values (ml.cpp:782-788)
SLICE_ALL = (0, 2**62, 1)


def _ne(shape) -> tuple[int, int, int, int]:
    shape = tuple(int(v) for v in shape)
    return shape + (1,) * (4 - len(shape))


class Tensor:
    """A handle into its graph (`visp::tensor`)."""

    def __init__(self, graph: "Graph", index: int):
        self.graph, self.index = graph, index

    @property
    def ne(self) -> tuple[int, int, int, int]:
        return self.graph.info(self)[1]

    @property
    def dtype(self) -> int:
        return self.graph.info(self)[0]

    @property
    def is_constant(self) -> bool:
        return self.graph.info(self)[2]

    def __repr__(self):
        return f"Tensor({self.index}, ne={self.ne})"


class Weights:
    """`model_weights` (ml.h:126-149): a model's tensors by name. Device images are made by the first graph that uses a tensor and shared
    by every later graph over the same weights."""

    def __init__(self, path=None):
        self._api = get_lib()
        h = c_void_p()
        if path is None:
            check(self._api.visp_weights_create(byref(h)))
        else:
            check(self._api.visp_weights_load(lib.path_to_char_p(path), byref(h)))
        self._handle = h

    def add(self, name: str, array: np.ndarray, dtype: int = F16):
        a = np.ascontiguousarray(array, dtype=np.float32)
        check(self._api.visp_weights_add(self._handle, name.encode(), dtype, (c_int64 * 4)(*_ne(a.shape[::-1])), a.ctypes.data_as(c_void_p)))

    def __del__(self):
        if getattr(self, "_handle", None):
            self._api.visp_weights_destroy(self._handle)
            self._handle = None


class Graph:
    """`compute_graph` over a model's weights. `device=None`: lower and plan only (no GPU needed)."""

    def __init__(self, device=None, weights: Weights | None = None):
        self._api = get_lib()
        self._device = device  # kept alive; bound when the graph is allocated (compute_graph_allocate(graph, backend))
        self._weights = weights
        h = c_void_p()
        check(self._api.visp_graph_create(weights._handle if weights is not None else None, byref(h)))
        self._handle = h

    def __del__(self):
        if getattr(self, "_handle", None):
            self._api.visp_graph_destroy(self._handle)
            self._handle = None

    # ---- weights
    def add_weight(self, name: str, array: np.ndarray, dtype: int = F16) -> Tensor:
        """`array` in torch / numpy index order (slowest first): its reversed shape is the ggml ne."""
        a = np.ascontiguousarray(array, dtype=np.float32)
        ne = (c_int64 * 4)(*_ne(a.shape[::-1]))
        out = c_int32()
        check(self._api.visp_graph_add_weight(self._handle, name.encode(), dtype, ne, a.ctypes.data_as(c_void_p), byref(out)))
        return Tensor(self, out.value)

    def find(self, name: str) -> Tensor | None:
        out = c_int32()
        check(self._api.visp_graph_find_weight(self._handle, name.encode(), byref(out)))
        return Tensor(self, out.value) if out.value >= 0 else None

    # ---- nodes
    def input(self, ne, dtype: int = F32, name: str = "input") -> Tensor:
        out = c_int32()
        check(self._api.visp_graph_input(self._handle, dtype, (c_int64 * 4)(*_ne(ne)), name.encode(), byref(out)))
        return Tensor(self, out.value)

    def op(self, op: int, src, iparams=(), fparams=()) -> Tensor:
        srcs = (c_int32 * len(src))(*[t.index for t in src])
        ip = (c_int64 * max(1, len(iparams)))(*[int(v) for v in iparams])
        fp = (c_float * max(1, len(fparams)))(*[float(v) for v in fparams])
        out = c_int32()
        check(self._api.visp_graph_op(self._handle, op, srcs, len(src), ip, len(iparams), fp, len(fparams), byref(out)))
        return Tensor(self, out.value)

    def set_name(self, t: Tensor, name: str) -> Tensor:
        check(self._api.visp_graph_set_name(self._handle, t.index, name.encode()))
        return t

    def get_tensor(self, name: str) -> Tensor | None:
        out = c_int32()
        check(self._api.visp_graph_get_tensor(self._handle, name.encode(), byref(out)))
        return Tensor(self, out.value) if out.value >= 0 else None

    def output(self, t: Tensor, name: str = "output") -> Tensor:
        check(self._api.visp_graph_output(self._handle, t.index, name.encode()))
        return t

    def info(self, t: Tensor):
        dtype, const = c_int32(), c_int32()
        ne = (c_int64 * 4)()
        check(self._api.visp_graph_tensor_info(self._handle, t.index, byref(dtype), ne, byref(const)))
        return dtype.value, tuple(ne), bool(const.value)

    def read_constant(self, t: Tensor) -> np.ndarray:
        ne = self.info(t)[1]
        out = np.empty(ne[::-1], np.float32)
        check(self._api.visp_graph_read_constant(self._handle, t.index, out.ctypes.data_as(c_void_p), out.size))
        return out

    # ---- execution
    def set_fused_models(self, enable: bool):
        """Before allocate(). True (default): the lowering may map whole node groups onto the kernels written for them; False: one launch per
        epilogue-fused node on f16 activations."""
        check(self._api.visp_graph_set_fused_models(self._handle, int(enable)))

    def allocate(self):
        check(self._api.visp_graph_allocate(self._handle, self._device._handle if self._device is not None else None))

    def use_hip_graph(self, enable: bool = True):
        check(self._api.visp_graph_use_hip_graph(self._handle, 1 if enable else 0))

    def compute(self):
        check(self._api.visp_graph_compute(self._handle))

    def set(self, t: Tensor, array: np.ndarray):
        dtype, ne, _ = self.info(t)
        a = np.ascontiguousarray(array, dtype={F32: np.float32, U8: np.uint8}.get(dtype, np.float16))
        if a.size != int(np.prod(ne)):
            raise ValueError(f"tensor {t} takes {int(np.prod(ne))} elements, got {a.size}")
        check(self._api.visp_graph_tensor_set(self._handle, t.index, a.ctypes.data_as(c_void_p), a.nbytes))

    def get(self, t: Tensor) -> np.ndarray:
        """f32, in numpy order (reversed ne)."""
        ne = self.info(t)[1]
        out = np.empty(ne[::-1], np.float32)
        check(self._api.visp_graph_tensor_get(self._handle, t.index, out.ctypes.data_as(c_void_p), out.nbytes, 1))
        return out

    def describe(self) -> str:
        need = c_int64()
        check(self._api.visp_graph_describe(self._handle, None, 0, byref(need)))
        buf = ctypes.create_string_buffer(need.value)
        check(self._api.visp_graph_describe(self._handle, buf, need.value, None))
        return buf.value.decode()

    def summary(self) -> dict:
        last = self.describe().strip().splitlines()[-1]
        return {k: int(v) for k, v in (kv.split("=") for kv in last.split())}


class ModelRef:
    """`visp::model_ref` (ml.h:199-245): a graph + a name prefix. `m["a"]["b"]`, `m[3]`, `m.weights("weight")`."""

    def __init__(self, graph: Graph, prefix: str = ""):
        self.graph, self.prefix = graph, prefix

    def __getitem__(self, sub) -> "ModelRef":
        sub = str(sub)
        return ModelRef(self.graph, f"{self.prefix}.{sub}" if self.prefix else sub)

    def with_prefix(self, prefix: str) -> "ModelRef":
        return ModelRef(self.graph, prefix)

    def _full(self, name: str) -> str:
        return f"{self.prefix}.{name}" if self.prefix else name

    def find(self, name: str) -> Tensor | None:
        return self.graph.find(self._full(name))

    def weights(self, name: str) -> Tensor:
        t = self.find(name)
        if t is None:
            raise KeyError(f"tensor not found: {self._full(name)}")
        return t


def named(m: ModelRef, t: Tensor) -> Tensor:  # ml.cpp:640-643
    return m.graph.set_name(t, m.prefix)


# ---- nn.h -------------------------------------------------------------------------------------------------------------------

def linear(m: ModelRef, x: Tensor) -> Tensor:  # nn.cpp:6-12
    src = [x, m.weights("weight")]
    if (b := m.find("bias")) is not None:
        src.append(b)
    return m.graph.op(OP_LINEAR, src)


def layer_norm(m: ModelRef, x: Tensor, eps: float = 1e-5) -> Tensor:  # nn.cpp:14-19
    return m.graph.op(OP_LAYER_NORM, [x, m.weights("weight"), m.weights("bias")], fparams=[eps])


def conv_2d(m: ModelRef, x: Tensor, stride: int = 1, pad: int = 0) -> Tensor:  # nn.cpp:72-100
    src = [x, m.weights("weight")]
    if (b := m.find("bias")) is not None:
        src.append(b)
    return m.graph.op(OP_CONV_2D, src, [stride, pad])


def conv_transpose_2d(m: ModelRef, x: Tensor, stride: int) -> Tensor:  # nn.cpp:117-129
    src = [x, m.weights("weight")]
    if (b := m.find("bias")) is not None:
        src.append(b)
    return m.graph.op(OP_CONV_TRANSPOSE_2D, src, [stride])


def patch_embed(m: ModelRef, x: Tensor, patch_size: int) -> Tensor:  # nn.cpp:166-180 (DINOv2: no norm)
    p = m["projection"]
    src = [x, p.weights("weight")]
    if (b := p.find("bias")) is not None:
        src.append(b)
    return m.graph.op(OP_PATCH_EMBED, src, [patch_size])


def attention(m: ModelRef, q: Tensor, k: Tensor, v: Tensor, mask, scale: float, m_out: ModelRef) -> Tensor:  # nn.cpp:210-244
    if mask is not None:
        raise NotImplementedError("attention masks are built for the window-attention families only")
    x = m.graph.op(OP_ATTENTION, [q, k, v], fparams=[scale])
    return linear(m_out, x)


# ---- ml.h tensor operations + the ggml ops the arch code uses directly ----------------------------------------------------------

def slice_(m: ModelRef, x: Tensor, s0=SLICE_ALL, s1=SLICE_ALL, s2=SLICE_ALL, s3=SLICE_ALL) -> Tensor:  # ml.cpp:746-768
    ip = []
    for s in (s0, s1, s2, s3):
        if isinstance(s, int):
            s = (s, s + 1, 1)
        s = tuple(s) + (1,) * (3 - len(s))
        ip += list(s)
    return m.graph.op(OP_SLICE, [x], ip)


def concat(m: ModelRef, tensors, dim: int) -> Tensor:  # ml.cpp:770-780 (n-ary: folded left to right)
    out = tensors[0]
    for t in tensors[1:]:
        out = m.graph.op(OP_CONCAT, [out, t], [dim])
    return out


def interpolate(m: ModelRef, x: Tensor, target, mode: int) -> Tensor:  # ml.cpp:782-788; x is CWHN here
    return m.graph.op(OP_INTERPOLATE, [x], [target[0], target[1], mode])


def add(m, a, b): return m.graph.op(OP_ADD, [a, b])
def mul(m, a, b): return m.graph.op(OP_MUL, [a, b])
def gelu(m, x): return m.graph.op(OP_GELU, [x])
def relu(m, x): return m.graph.op(OP_RELU, [x])
def scale(m, x, s): return m.graph.op(OP_SCALE, [x], fparams=[s])
def reshape(m, x, *ne): return m.graph.op(OP_RESHAPE, [x], list(_ne(ne)))
def repeat(m, x, *ne): return m.graph.op(OP_REPEAT, [x], list(_ne(ne)))
def cont(m, x): return m.graph.op(OP_CONT, [x])
def leaky_relu(m, x, slope): return m.graph.op(OP_LEAKY_RELU, [x], fparams=[slope])  # ggml_leaky_relu


# ---- Depth-Anything-V2 against this layer (the structure of src/visp/arch/dino.cpp + depth-anything.cpp) -----------------------

BILINEAR_AC = SCALE_MODE_BILINEAR | SCALE_FLAG_ALIGN_CORNERS


def dino_interpolate_pos_encoding(m: ModelRef, x: Tensor, w: int, h: int, patch_size: int) -> Tensor:  # dino.cpp:10-30
    pos = m.weights("position_embeddings")
    n_patch, n = x.ne[1] - 1, pos.ne[1] - 1
    if n_patch == n and w == h:
        return pos
    cls_embed = slice_(m, pos, SLICE_ALL, 0)
    patch = slice_(m, pos, SLICE_ALL, (1, n + 1))
    dim = x.ne[0]
    side = int(math.sqrt(n) + 0.01)
    patch = reshape(m, patch, dim, side, side, 1)
    patch = interpolate(m, patch, (w // patch_size, h // patch_size), SCALE_MODE_BICUBIC)
    patch = reshape(m, patch, dim, (w // patch_size) * (h // patch_size), 1)
    return concat(m, [cls_embed, patch], 1)


def dino_prepare_tokens(m: ModelRef, x: Tensor, patch_size: int) -> Tensor:  # dino.cpp:32-46
    c, w, h, n = x.ne
    x = patch_embed(m["patch_embeddings"], x, patch_size)
    x = reshape(m, x, x.ne[0], x.ne[1] * x.ne[2], x.ne[3])
    cls_token = m.weights("cls_token")
    if cls_token.ne[2] != n:
        cls_token = repeat(m, cls_token, cls_token.ne[0], 1, n, 1)
    x = concat(m, [cls_token, x], 1)
    return add(m, x, dino_interpolate_pos_encoding(m, x, w, h, patch_size))


def dino_layer(m: ModelRef, x: Tensor, n_heads: int) -> Tensor:  # dino.cpp:48-90
    c, n, b, _ = x.ne
    att = layer_norm(m["norm1"], x, 1e-6)
    ma = m["attention"]
    q, k, v = (reshape(m, linear(ma["attention"][name], att), c // n_heads, n_heads, n, b) for name in ("query", "key", "value"))
    att = attention(ma, q, k, v, None, 1.0 / math.sqrt(c / n_heads), ma["output.dense"])
    x = add(m, x, mul(m, att, m["layer_scale1"].weights("lambda1")))
    ffn = layer_norm(m["norm2"], x, 1e-6)
    ffn = linear(m["mlp.fc2"], gelu(m, linear(m["mlp.fc1"], ffn)))
    x = add(m, x, mul(m, ffn, m["layer_scale2"].weights("lambda1")))
    return named(m, x)


def dino_get_intermediate_layers(m: ModelRef, x: Tensor, layers, n_layers: int, n_heads: int, patch_size: int):  # dino.cpp:92-110
    x = dino_prepare_tokens(m["embeddings"], x, patch_size)
    outputs = []
    for i in range(n_layers):
        x = dino_layer(m["encoder.layer"][i], x, n_heads)
        if i in layers:
            outputs.append(m.graph.set_name(layer_norm(m["layernorm"], x, 1e-6), f"dino_layer_{i}"))
    return outputs


def dpt_residual_conv(m: ModelRef, x: Tensor) -> Tensor:  # depth-anything.cpp:15-23
    out = conv_2d(m["convolution1"], relu(m, x), 1, 1)
    out = conv_2d(m["convolution2"], relu(m, out), 1, 1)
    return named(m, add(m, x, out))


def dpt_feature_fusion(m: ModelRef, x0: Tensor, x1: Tensor | None, size) -> Tensor:  # depth-anything.cpp:25-42
    x = x0
    if x1 is not None:
        x = add(m, x, dpt_residual_conv(m["residual_layer1"], x1))
    x = dpt_residual_conv(m["residual_layer2"], x)
    w, h = size if size is not None else (x.ne[1] * 2, x.ne[2] * 2)
    x = interpolate(m, x, (w, h), BILINEAR_AC)
    return named(m, conv_2d(m["projection"], x))


def dpt_neck(m: ModelRef, features, patch_w: int, patch_h: int) -> Tensor:  # depth-anything.cpp:44-79
    layer = []
    for i, x in enumerate(features):
        x = slice_(m, x, SLICE_ALL, (1, x.ne[1]))
        x = reshape(m, x, x.ne[0], patch_w, patch_h, x.ne[2])
        r = m["reassemble_stage.layers"][i]
        x = conv_2d(r["projection"], x)
        if i == 0:
            x = conv_transpose_2d(r["resize"], x, 4)
        elif i == 1:
            x = conv_transpose_2d(r["resize"], x, 2)
        elif i == 3:
            x = conv_2d(r["resize"], x, 2, 1)
        layer.append(x)
    layer = [conv_2d(m["convs"][i], layer[i], 1, 1) for i in range(4)]
    f = m["fusion_stage.layers"]
    fused = dpt_feature_fusion(f[0], layer[3], None, layer[2].ne[1:3])
    fused = dpt_feature_fusion(f[1], fused, layer[2], layer[1].ne[1:3])
    fused = dpt_feature_fusion(f[2], fused, layer[1], layer[0].ne[1:3])
    return dpt_feature_fusion(f[3], fused, layer[0], None)


def dpt_head(m: ModelRef, x: Tensor, w: int, h: int, max_depth: float) -> Tensor:  # depth-anything.cpp:81-96
    out = conv_2d(m["conv1"], x, 1, 1)
    out = interpolate(m, out, (w, h), BILINEAR_AC)
    out = relu(m, conv_2d(m["conv2"], out, 1, 1))
    out = relu(m, conv_2d(m["conv3"], out))
    return scale(m, out, max_depth) if max_depth != 1 else out


def depthany_predict(m: ModelRef, image: Tensor, n_layers: int, n_heads: int, patch_size: int = 14, feature_layers=(2, 5, 8, 11),
                     max_depth: float = 1.0) -> Tensor:  # depth-anything.cpp:100-110
    c, w, h, n = image.ne
    features = dino_get_intermediate_layers(m["backbone"], image, feature_layers, n_layers, n_heads, patch_size)
    fused = dpt_neck(m["neck"], features, w // patch_size, h // patch_size)
    depth = dpt_head(m["head"], fused, w, h, max_depth)
    return m.graph.output(depth, "output")


# ---- ESRGAN / Real-ESRGAN against this layer (the structure of src/visp/arch/esrgan.cpp) ---------------------------------------------------

def esrgan_upsample(m: ModelRef, x: Tensor) -> Tensor:  # esrgan.cpp:13-19
    c, w, h, n = x.ne
    x = interpolate(m, x, (w * 2, h * 2), SCALE_MODE_NEAREST)
    x = conv_2d(m, x, 1, 1)
    return named(m, leaky_relu(m, x, 0.2))


def esrgan_conv_block(m: ModelRef, x: Tensor) -> Tensor:  # esrgan.cpp:21-25
    return leaky_relu(m, conv_2d(m[0], x, 1, 1), 0.2)


def esrgan_residual_dense_block(m: ModelRef, x: Tensor) -> Tensor:  # esrgan.cpp:27-41 (channels are dimension 0 of a CWHN tensor)
    x1 = esrgan_conv_block(m["conv1"], x)
    c1 = concat(m, [x, x1], 0)
    x2 = esrgan_conv_block(m["conv2"], c1)
    c2 = concat(m, [c1, x2], 0)
    x3 = esrgan_conv_block(m["conv3"], c2)
    c3 = concat(m, [c2, x3], 0)
    x4 = esrgan_conv_block(m["conv4"], c3)
    c4 = concat(m, [c3, x4], 0)
    x5 = scale(m, conv_2d(m["conv5.0"], c4, 1, 1), 0.2)
    return named(m, add(m, x, x5))


def esrgan_rrdb(m: ModelRef, x: Tensor) -> Tensor:  # esrgan.cpp:43-51
    x_in = x
    for k in (1, 2, 3):
        x = esrgan_residual_dense_block(m[f"RDB{k}"], x)
    return named(m, add(m, scale(m, x, 0.2), x_in))


def esrgan_generate(m: ModelRef, x: Tensor, scale_factor: int, n_blocks: int) -> Tensor:  # esrgan.cpp:55-79; x = the f32 image [3, W, H, N]
    m = m["model"]
    x = conv_2d(m[0], x, 1, 1)
    sub = x
    block = m[1]["sub"]
    for i in range(n_blocks):
        sub = esrgan_rrdb(block[i], sub)
    sub = conv_2d(block[n_blocks], sub, 1, 1)
    x = add(m, x, sub)
    seq = 2
    for _ in range(int(np.log2(scale_factor))):
        x = esrgan_upsample(m[seq + 1], x)
        seq += 3
    x = leaky_relu(m, conv_2d(m[seq], x, 1, 1), 0.2)
    x = conv_2d(m[seq + 2], x, 1, 1)
    return m.graph.output(x, "result")
