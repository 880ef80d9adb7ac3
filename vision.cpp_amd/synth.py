"""Synthetic Depth-Anything-V2 checkpoints and inputs (no network: random-init weights).

`state_dict()` produces tensors with the HuggingFace `transformers` names and torch shapes
(SURVEY.md Appendix C), in state-dict order. `write_gguf()` applies exactly the on-disk
contract of the reference's scripts/convert.py:428-475 (convert_depth_anything with
`--quantize f16`, default layout): patch-embed and reassemble projections stored NHWC,
ConvTranspose kernels untouched, every other conv kernel left OIHW and listed in
`depthanything.conv2d_weights`, cls/pos embeddings kept f32, everything else f16.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

from .gguf import GGUFWriter


@dataclass
class Config:
    embed_dim: int = 384
    n_layers: int = 12
    n_heads: int = 6
    patch_size: int = 14
    image_size: int = 518  # pos-embed grid = (image_size / patch_size)^2
    mlp_ratio: int = 4
    neck_sizes: tuple = (48, 96, 192, 384)
    fusion_size: int = 64
    head_size: int = 32
    feature_layers: tuple = (2, 5, 8, 11)
    name: str = "small"

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size


SMALL = Config()
# tiny config for fast CPU parity runs: 5x5 patches, 2 heads x 16
TINY = Config(embed_dim=32, n_layers=4, n_heads=2, image_size=70, neck_sizes=(8, 16, 32, 32), fusion_size=16,
              head_size=8, feature_layers=(0, 1, 2, 3), name="tiny")
# "mini": real head_dim 64 and 2 heads, 8x8 patches; exercises the same kernels as SMALL at 1/100 the cost
MINI = Config(embed_dim=128, n_layers=4, n_heads=2, image_size=112, neck_sizes=(48, 96, 192, 384),
              fusion_size=64, head_size=32, feature_layers=(0, 1, 2, 3), name="mini")
# "short": the north-star widths (384 / 1536, 6 heads of 64) with 3 layers on 8x8 patches -- the smallest model that runs
# the token-stationary block kernel (csrc/kernels_block16.hip); the last two taps name the same layer
SHORT = Config(embed_dim=384, n_layers=3, n_heads=6, image_size=112, feature_layers=(0, 1, 2, 2), name="short")


def state_dict(cfg: Config = SMALL, seed: int = 0) -> dict[str, np.ndarray]:
    """float32 tensors, HF names, torch shapes, HF state-dict order."""
    rng = np.random.default_rng(seed)
    D, Hd = cfg.embed_dim, cfg.embed_dim * cfg.mlp_ratio
    sd: dict[str, np.ndarray] = {}

    def normal(shape, std):
        return (rng.standard_normal(shape) * std).astype(np.float32)

    def lin(name, out_f, in_f, gain=1.0):
        sd[f"{name}.weight"] = normal((out_f, in_f), gain / np.sqrt(in_f))
        sd[f"{name}.bias"] = normal((out_f,), 0.02)

    def conv(name, out_c, in_c, k, bias=True, gain=1.0):
        sd[f"{name}.weight"] = normal((out_c, in_c, k, k), gain / np.sqrt(in_c * k * k))
        if bias:
            sd[f"{name}.bias"] = normal((out_c,), 0.02)

    def norm(name):
        sd[f"{name}.weight"] = (1.0 + rng.standard_normal(D) * 0.05).astype(np.float32)
        sd[f"{name}.bias"] = normal((D,), 0.02)

    e = "backbone.embeddings"
    sd[f"{e}.cls_token"] = normal((1, 1, D), 0.5)
    sd[f"{e}.mask_token"] = np.zeros((1, D), np.float32)
    sd[f"{e}.position_embeddings"] = normal((1, cfg.grid * cfg.grid + 1, D), 0.5)
    conv(f"{e}.patch_embeddings.projection", D, 3, cfg.patch_size, gain=1.0)
    for i in range(cfg.n_layers):
        p = f"backbone.encoder.layer.{i}"
        norm(f"{p}.norm1")
        lin(f"{p}.attention.attention.query", D, D, gain=1.5)
        lin(f"{p}.attention.attention.key", D, D, gain=1.5)
        lin(f"{p}.attention.attention.value", D, D)
        lin(f"{p}.attention.output.dense", D, D)
        sd[f"{p}.layer_scale1.lambda1"] = (0.3 + rng.standard_normal(D) * 0.05).astype(np.float32)
        norm(f"{p}.norm2")
        lin(f"{p}.mlp.fc1", Hd, D, gain=1.4)
        lin(f"{p}.mlp.fc2", D, Hd, gain=1.4)
        sd[f"{p}.layer_scale2.lambda1"] = (0.3 + rng.standard_normal(D) * 0.05).astype(np.float32)
    norm("backbone.layernorm")

    r = "neck.reassemble_stage.layers"
    for i, c in enumerate(cfg.neck_sizes):
        conv(f"{r}.{i}.projection", c, D, 1)
        if i == 0:  # ConvTranspose2d(c, c, 4, stride 4): weight [Cin, Cout, 4, 4]
            sd[f"{r}.{i}.resize.weight"] = normal((c, c, 4, 4), 1.0 / np.sqrt(c))
            sd[f"{r}.{i}.resize.bias"] = normal((c,), 0.02)
        elif i == 1:  # ConvTranspose2d(c, c, 2, stride 2)
            sd[f"{r}.{i}.resize.weight"] = normal((c, c, 2, 2), 1.0 / np.sqrt(c))
            sd[f"{r}.{i}.resize.bias"] = normal((c,), 0.02)
        elif i == 3:  # Conv2d(c, c, 3, stride 2, pad 1)
            conv(f"{r}.{i}.resize", c, c, 3)
    F = cfg.fusion_size
    for i, c in enumerate(cfg.neck_sizes):
        conv(f"neck.convs.{i}", F, c, 3, bias=False)
    for i in range(4):
        p = f"neck.fusion_stage.layers.{i}"
        conv(f"{p}.projection", F, F, 1)
        for rl in ("residual_layer1", "residual_layer2"):
            conv(f"{p}.{rl}.convolution1", F, F, 3, gain=1.2)
            conv(f"{p}.{rl}.convolution2", F, F, 3, gain=0.7)
    conv("head.conv1", cfg.head_size, F, 3)
    conv("head.conv2", cfg.head_size, cfg.head_size, 3, gain=1.4)
    conv("head.conv3", 1, cfg.head_size, 1, gain=1.0)
    # non-negative 1x1 weights + positive bias: the final ReLU stays active, so the min-max
    # normalised depth map is well conditioned (a real checkpoint behaves the same way)
    sd["head.conv3.weight"] = np.abs(sd["head.conv3.weight"])
    sd["head.conv3.bias"] = np.array([0.1], np.float32)
    return sd


def _is_conv_2d(name: str, t: np.ndarray) -> bool:  # scripts/convert.py:111-117
    return t.ndim == 4 and t.shape[2] == t.shape[3] and t.shape[2] in (1, 3, 4, 7, 14) and name.endswith("weight")


def gguf_tensors(sd: dict[str, np.ndarray], layout: str = "whcn"):
    """Applies convert_depth_anything's per-tensor rules (scripts/convert.py:460-475).
    layout "whcn" = the converter's default (--layout nchw): other conv kernels stay OIHW and are listed in
    conv2d_weights; "cwhn" (--layout nhwc): Writer.convert_tensor_2d permutes them to OHWI at once, nothing is listed.

    Returns (ordered dict name -> array as stored, conv2d_weights index list)."""
    out: dict[str, np.ndarray] = {}
    conv2d: list[int] = []
    for name, t in sd.items():
        if _is_conv_2d(name, t):
            if "patch_embeddings" in name or ("projection" in name and "fusion" not in name):
                t = np.ascontiguousarray(t.transpose(0, 2, 3, 1))  # conv_2d_to_nhwc: Cout H W Cin
            elif "0.resize" in name or "1.resize" in name:
                pass  # ConvTranspose2d, layout untouched
            elif layout == "cwhn":
                t = np.ascontiguousarray(t.transpose(0, 2, 3, 1))
            else:
                conv2d.append(len(out))  # Writer.convert_tensor_2d with layout nchw
        if "position_embeddings" in name or "cls_token" in name:
            out[name] = t.astype(np.float32)
        else:
            out[name] = t.astype(np.float16)
    return out, conv2d


def write_gguf(path: str | Path, cfg: Config = SMALL, seed: int = 0, sd: dict[str, np.ndarray] | None = None, layout: str = "whcn") -> Path:
    sd = sd if sd is not None else state_dict(cfg, seed)
    tensors, conv2d = gguf_tensors(sd, layout)
    w = GGUFWriter(path, "depthanything")
    w.add_string("depthanything.tensor_data_layout", layout)  # set_tensor_layout_default(nchw) unless --layout nhwc
    w.add_int32("dino.patch_size", cfg.patch_size)
    w.add_int32("dino.embed_dim", cfg.embed_dim)
    w.add_int32("depthanything.image_size", cfg.image_size)
    w.add_int32("dino.n_heads", cfg.n_heads)
    w.add_int32("dino.n_layers", cfg.n_layers)
    w.add_array_i32("depthanything.feature_layers", cfg.feature_layers)
    w.add_uint32("general.quantization_version", 2)
    w.add_uint32("general.file_type", 1)  # f16
    if conv2d:  # Writer.add_conv2d_weight_indices: only when something is listed
        w.add_array_i32("depthanything.conv2d_weights", conv2d)
    for name, t in tensors.items():
        w.add_tensor(name, t)
    w.write()
    return Path(path)


def images(n: int, w: int = 518, h: int = 518, seed: int = 1234) -> np.ndarray:
    """n synthetic rgb_u8 images [n, h, w, 3]: low-frequency structure + noise, so that the
    min-max normalised depth output is well conditioned."""
    out = np.empty((n, h, w, 3), np.uint8)
    yy, xx = np.meshgrid(np.linspace(0, 1, h, dtype=np.float32), np.linspace(0, 1, w, dtype=np.float32), indexing="ij")
    for i in range(n):
        rng = np.random.default_rng(seed + i)
        img = np.zeros((h, w, 3), np.float32)
        for c in range(3):
            for _ in range(4):
                fx, fy = rng.uniform(0.5, 4.0, 2)
                ph = rng.uniform(0, 2 * np.pi, 2)
                img[..., c] += rng.uniform(0.1, 0.3) * np.sin(2 * np.pi * fx * xx + ph[0]) * np.cos(2 * np.pi * fy * yy + ph[1])
        img = 0.5 + img + rng.uniform(-0.08, 0.08, img.shape).astype(np.float32)
        out[i] = np.clip(img * 255.0, 0, 255).astype(np.uint8)
    return out


# ---- ESRGAN / Real-ESRGAN (RRDBNet, spandrel key layout; reference tests/test_esrgan.py:150-212) -------------

@dataclass
class EsrganConfig:
    num_filters: int = 64
    num_blocks: int = 23
    scale: int = 4
    gc: int = 32
    name: str = "x4"


ESRGAN_X4 = EsrganConfig()                                           # RealESRGAN_x4 (BASELINE.json configs[2])
ESRGAN_TINY = EsrganConfig(num_filters=64, num_blocks=2, scale=2, name="tiny")


def esrgan_state_dict(cfg: EsrganConfig = ESRGAN_X4, seed: int = 0) -> dict[str, np.ndarray]:
    """float32 tensors with the names spandrel gives RRDBNet (what scripts/convert.py:524-527 writes)."""
    rng = np.random.default_rng(seed)
    sd: dict[str, np.ndarray] = {}
    nf, gc = cfg.num_filters, cfg.gc

    def conv(name, cout, cin, gain=1.0):
        sd[f"{name}.weight"] = (rng.standard_normal((cout, cin, 3, 3)) * gain / np.sqrt(cin * 9)).astype(np.float32)
        sd[f"{name}.bias"] = (rng.standard_normal(cout) * 0.02).astype(np.float32)

    conv("model.0", nf, 3)
    for i in range(cfg.num_blocks):
        for r in (1, 2, 3):
            p = f"model.1.sub.{i}.RDB{r}"
            for k in range(4):
                conv(f"{p}.conv{k + 1}.0", gc, nf + k * gc, gain=1.2)
            conv(f"{p}.conv5.0", nf, nf + 4 * gc, gain=1.5)
            if r == 3:
                # random RRDBs grow the signal by ~1.2x each (x + 0.2*(x + ...)); trained nets do not. Cancel the
                # identity part of the last dense block so 23 blocks stay O(1) in f16.
                sd[f"{p}.conv5.0.weight"][np.arange(nf), np.arange(nf), 1, 1] -= 5.0
    conv(f"model.1.sub.{cfg.num_blocks}", nf, nf, gain=0.7)
    seq = 2
    s = cfg.scale
    while s > 1:
        conv(f"model.{seq + 1}", nf, nf, gain=1.2)
        seq += 3
        s >>= 1
    conv(f"model.{seq}", nf, nf, gain=1.2)
    conv(f"model.{seq + 2}", 3, nf, gain=0.25)
    sd[f"model.{seq + 2}.bias"] = np.array([0.45, 0.5, 0.55], np.float32)  # keeps the output inside [0, 1] mostly
    return sd


def esrgan_gguf_tensors(sd: dict[str, np.ndarray], layout: str = "whcn"):
    """scripts/convert.py:504-527 (convert_esrgan, --quantize f16): with the default layout every conv kernel stays
    OIHW and is listed in esrgan.conv2d_weights; with --layout nhwc ("cwhn") kernels are stored OHWI and nothing is
    listed; all float tensors -> f16."""
    out: dict[str, np.ndarray] = {}
    conv2d: list[int] = []
    for name, t in sd.items():
        if _is_conv_2d(name, t):
            if layout == "cwhn":
                t = np.ascontiguousarray(t.transpose(0, 2, 3, 1))
            else:
                conv2d.append(len(out))
        out[name] = t.astype(np.float16)
    return out, conv2d


def write_esrgan_gguf(path: str | Path, cfg: EsrganConfig = ESRGAN_X4, seed: int = 0, sd: dict[str, np.ndarray] | None = None,
                      layout: str = "whcn") -> Path:
    sd = sd if sd is not None else esrgan_state_dict(cfg, seed)
    tensors, conv2d = esrgan_gguf_tensors(sd, layout)
    w = GGUFWriter(path, "esrgan")
    w.add_string("esrgan.tensor_data_layout", layout)
    w.add_int32("esrgan.scale", cfg.scale)
    w.add_int32("esrgan.block_count", cfg.num_blocks)
    w.add_int32("esrgan.filter_count", cfg.num_filters)
    w.add_uint32("general.quantization_version", 2)
    w.add_uint32("general.file_type", 1)
    if conv2d:
        w.add_array_i32("esrgan.conv2d_weights", conv2d)
    for name, t in tensors.items():
        w.add_tensor(name, t)
    w.write()
    return Path(path)


# ---- TinyViT image encoder of MobileSAM (reference tests/test_mobile_sam.py:18-765, scripts/convert.py:204-262) -------

@dataclass
class TinyVitConfig:
    img_size: int = 1024
    embed_dims: tuple = (64, 128, 160, 320)
    depths: tuple = (2, 2, 6, 2)
    num_heads: tuple = (2, 4, 5, 10)
    window_sizes: tuple = (7, 7, 14, 7)
    name: str = "5m"

    def layers(self):
        """(resolution, embed_dim, depth, num_heads, window_size, downsample) per layer, mobile-sam.h:31-36"""
        r0 = self.img_size // 4
        res = [r0, r0 // 2, r0 // 4, r0 // 4]
        return [(res[i], self.embed_dims[i], self.depths[i], self.num_heads[i], self.window_sizes[i], int(i < 3)) for i in range(4)]


TINYVIT_5M = TinyVitConfig()


def attention_bias_idxs(ws: int) -> np.ndarray:
    """build_attention_bias_indices (scripts/convert.py:250-262)"""
    pts = [(a, b) for a in range(ws) for b in range(ws)]
    offs: dict = {}
    idx = []
    for p1 in pts:
        for p2 in pts:
            o = (abs(p1[0] - p2[0]), abs(p1[1] - p2[1]))
            if o not in offs:
                offs[o] = len(offs)
            idx.append(offs[o])
    return np.array(idx, np.int64).reshape(len(pts), len(pts))


def tinyvit_state_dict(cfg: TinyVitConfig = TINYVIT_5M, seed: int = 0) -> dict[str, np.ndarray]:
    """float32 tensors under the names of the reference's torch TinyViT (BatchNorm NOT fused, as in a checkpoint)."""
    rng = np.random.default_rng(seed)
    sd: dict[str, np.ndarray] = {}

    def conv_bn(name, cout, cin, k, groups=1, gain=1.0):
        fan = (cin // groups) * k * k
        sd[f"{name}.c.weight"] = (rng.standard_normal((cout, cin // groups, k, k)) * gain / np.sqrt(fan)).astype(np.float32)
        sd[f"{name}.bn.weight"] = (1 + 0.1 * rng.standard_normal(cout)).astype(np.float32)
        sd[f"{name}.bn.bias"] = (0.05 * rng.standard_normal(cout)).astype(np.float32)
        sd[f"{name}.bn.running_mean"] = (0.05 * rng.standard_normal(cout)).astype(np.float32)
        sd[f"{name}.bn.running_var"] = (1 + 0.2 * rng.random(cout)).astype(np.float32)

    def linear(name, n, k, gain=1.0):
        sd[f"{name}.weight"] = (rng.standard_normal((n, k)) * gain / np.sqrt(k)).astype(np.float32)
        sd[f"{name}.bias"] = (0.02 * rng.standard_normal(n)).astype(np.float32)

    def norm(name, c):
        sd[f"{name}.weight"] = (1 + 0.05 * rng.standard_normal(c)).astype(np.float32)
        sd[f"{name}.bias"] = (0.05 * rng.standard_normal(c)).astype(np.float32)

    def merging(name, dim, out):
        conv_bn(f"{name}.conv1", out, dim, 1)
        conv_bn(f"{name}.conv2", out, out, 3, groups=out)
        conv_bn(f"{name}.conv3", out, out, 1)

    e = cfg.embed_dims
    conv_bn("patch_embed.seq.0", e[0] // 2, 3, 3, gain=1.5)
    conv_bn("patch_embed.seq.2", e[0], e[0] // 2, 3, gain=1.5)
    for i in range(cfg.depths[0]):
        p = f"layers.0.blocks.{i}"
        conv_bn(f"{p}.conv1", 4 * e[0], e[0], 1)
        conv_bn(f"{p}.conv2", 4 * e[0], 4 * e[0], 3, groups=4 * e[0])
        conv_bn(f"{p}.conv3", e[0], 4 * e[0], 1, gain=0.5)
    merging("layers.0.downsample", e[0], e[1])
    for l in range(1, 4):
        dim, heads, ws = e[l], cfg.num_heads[l], cfg.window_sizes[l]
        for i in range(cfg.depths[l]):
            p = f"layers.{l}.blocks.{i}"
            sd[f"{p}.attn.attention_biases"] = (0.5 * rng.standard_normal((heads, ws * ws))).astype(np.float32)
            norm(f"{p}.attn.norm", dim)
            linear(f"{p}.attn.qkv", 3 * dim, dim)
            linear(f"{p}.attn.proj", dim, dim, gain=0.4)
            norm(f"{p}.mlp.norm", dim)
            linear(f"{p}.mlp.fc1", 4 * dim, dim)
            linear(f"{p}.mlp.fc2", dim, 4 * dim, gain=0.4)
            conv_bn(f"{p}.local_conv", dim, dim, 3, groups=dim, gain=0.9)
        if l < 3:
            merging(f"layers.{l}.downsample", dim, e[l + 1])
    sd["neck.0.weight"] = (rng.standard_normal((256, e[3], 1, 1)) / np.sqrt(e[3])).astype(np.float32)
    norm("neck.1", 256)
    sd["neck.2.weight"] = (rng.standard_normal((256, 256, 3, 3)) / np.sqrt(256 * 9)).astype(np.float32)
    norm("neck.3", 256)
    return sd


def tinyvit_gguf_tensors(sd: dict[str, np.ndarray], prefix: str = "enc."):
    """convert_sam's per-tensor rules for the image encoder (scripts/convert.py:204-247, 157-188), default whcn layout:
    BatchNorm fused into '<conv>.c.weight' / '.c.bias' (eps 1e-5); fused kernels stay OIHW and are listed in
    conv2d_weights -- except `local_conv`, always written NHWC (depthwise: H W 1 C) and not listed; attention_biases
    gathered to [heads, N, N] as 'attention_biases_indexed'; neck convs listed; floats -> f16."""
    out: dict[str, np.ndarray] = {}
    conv2d: list[int] = []
    for key, t in sd.items():
        name = prefix + key
        if key.endswith("attention_biases"):
            ws = int(round(np.sqrt(t.shape[1])))
            out[name + "_indexed"] = np.ascontiguousarray(t[:, attention_bias_idxs(ws)]).astype(np.float16)
            continue
        if key.endswith(".c.weight"):
            base = key[: -len("c.weight")]
            g = sd[base + "bn.weight"] / np.sqrt(sd[base + "bn.running_var"] + np.float32(1e-5))
            w = (t * g[:, None, None, None]).astype(np.float32)
            b = ((0 - sd[base + "bn.running_mean"]) * g + sd[base + "bn.bias"]).astype(np.float32)
            if "local_conv" in key:
                w = np.ascontiguousarray(w.transpose(2, 3, 1, 0))  # conv_2d_to_nhwc, depthwise: H W 1 C
            else:
                conv2d.append(len(out))
            out[name] = w.astype(np.float16)
            out[name.replace("weight", "bias")] = b.astype(np.float16)
            continue
        if ".bn." in key:
            continue
        if key in ("neck.0.weight", "neck.2.weight"):
            conv2d.append(len(out))
        out[name] = t.astype(np.float16)
    return out, conv2d


def write_tinyvit_gguf(path: str | Path, cfg: TinyVitConfig = TINYVIT_5M, seed: int = 0, sd: dict[str, np.ndarray] | None = None) -> Path:
    """A 'mobile-sam' GGUF holding the image encoder only (scripts/convert.py:204-247, 593-596 for the metadata)."""
    sd = sd if sd is not None else tinyvit_state_dict(cfg, seed)
    tensors, conv2d = tinyvit_gguf_tensors(sd)
    w = GGUFWriter(path, "mobile-sam")
    w.add_string("mobile-sam.tensor_data_layout", "whcn")
    w.add_uint32("general.quantization_version", 2)
    w.add_uint32("general.file_type", 1)
    if conv2d:
        w.add_array_i32("mobile-sam.conv2d_weights", conv2d)
    for name, t in tensors.items():
        w.add_tensor(name, t)
    w.write()
    return Path(path)


# ---- MobileSAM prompt encoder + mask decoder (reference tests/test_mobile_sam.py:796-1470, scripts/convert.py:204-282) ----

def sam_decoder_state_dict(seed: int = 0, dim: int = 256, heads: int = 8, mlp_dim: int = 2048, depth: int = 2) -> dict[str, np.ndarray]:
    """float32 tensors under the names of a SAM checkpoint (prompt_encoder.*, mask_decoder.*); mask_downscaling is never read
    by the reference's graph and is left out."""
    rng = np.random.default_rng(seed)
    sd: dict[str, np.ndarray] = {}

    def linear(name, n, k, gain=1.0):
        sd[f"{name}.weight"] = (rng.standard_normal((n, k)) * gain / np.sqrt(k)).astype(np.float32)
        sd[f"{name}.bias"] = (0.02 * rng.standard_normal(n)).astype(np.float32)

    def norm(name, c):
        sd[f"{name}.weight"] = (1 + 0.05 * rng.standard_normal(c)).astype(np.float32)
        sd[f"{name}.bias"] = (0.05 * rng.standard_normal(c)).astype(np.float32)

    def attention(name, internal):
        for pj in ("q_proj", "k_proj", "v_proj"):
            linear(f"{name}.{pj}", internal, dim)
        linear(f"{name}.out_proj", dim, internal, gain=0.7)

    pe = "prompt_encoder"
    sd[f"{pe}.pe_layer.positional_encoding_gaussian_matrix"] = rng.standard_normal((2, dim // 2)).astype(np.float32)
    for i in range(4):
        sd[f"{pe}.point_embeddings.{i}.weight"] = (0.5 * rng.standard_normal((1, dim))).astype(np.float32)
    sd[f"{pe}.not_a_point_embed.weight"] = (0.5 * rng.standard_normal((1, dim))).astype(np.float32)
    sd[f"{pe}.no_mask_embed.weight"] = (0.2 * rng.standard_normal((1, dim))).astype(np.float32)
    md = "mask_decoder"
    for i in range(depth):
        p = f"{md}.transformer.layers.{i}"
        attention(f"{p}.self_attn", dim)
        norm(f"{p}.norm1", dim)
        attention(f"{p}.cross_attn_token_to_image", dim // 2)
        norm(f"{p}.norm2", dim)
        linear(f"{p}.mlp.lin1", mlp_dim, dim)
        linear(f"{p}.mlp.lin2", dim, mlp_dim, gain=0.7)
        norm(f"{p}.norm3", dim)
        norm(f"{p}.norm4", dim)
        attention(f"{p}.cross_attn_image_to_token", dim // 2)
    attention(f"{md}.transformer.final_attn_token_to_image", dim // 2)
    norm(f"{md}.transformer.norm_final_attn", dim)
    sd[f"{md}.iou_token.weight"] = rng.standard_normal((1, dim)).astype(np.float32)
    sd[f"{md}.mask_tokens.weight"] = rng.standard_normal((4, dim)).astype(np.float32)
    sd[f"{md}.output_upscaling.0.weight"] = (rng.standard_normal((dim, dim // 4, 2, 2)) / np.sqrt(dim)).astype(np.float32)
    sd[f"{md}.output_upscaling.0.bias"] = (0.02 * rng.standard_normal(dim // 4)).astype(np.float32)
    norm(f"{md}.output_upscaling.1", dim // 4)
    sd[f"{md}.output_upscaling.3.weight"] = (rng.standard_normal((dim // 4, dim // 8, 2, 2)) / np.sqrt(dim // 4)).astype(np.float32)
    sd[f"{md}.output_upscaling.3.bias"] = (0.02 * rng.standard_normal(dim // 8)).astype(np.float32)
    for i in range(4):
        p = f"{md}.output_hypernetworks_mlps.{i}.layers"
        linear(f"{p}.0", dim, dim)
        linear(f"{p}.1", dim, dim)
        linear(f"{p}.2", dim // 8, dim)
    p = f"{md}.iou_prediction_head.layers"
    linear(f"{p}.0", 256, dim)
    linear(f"{p}.1", 256, 256)
    linear(f"{p}.2", 4, 256)
    return sd


def sam_dense_positional_embedding(gaussian: np.ndarray, size: int = 64) -> np.ndarray:
    """build_dense_positional_embeddings (scripts/convert.py:265-282): [size, size, 2F] float32, (h, w, c) order."""
    g = np.asarray(gaussian, np.float32)
    centre = ((np.arange(size, dtype=np.float32) + np.float32(1) - np.float32(0.5)) / np.float32(size)).astype(np.float32)
    coords = np.stack(np.broadcast_arrays(centre[None, :], centre[:, None]), axis=-1).astype(np.float32)  # (x, y) per (h, w)
    coords = np.float32(2) * coords - np.float32(1)
    proj = (np.float32(2 * np.pi) * (coords @ g)).astype(np.float32)
    return np.concatenate([np.sin(proj), np.cos(proj)], axis=-1).astype(np.float32)


def sam_decoder_gguf_tensors(sd: dict[str, np.ndarray]) -> dict[str, np.ndarray]:
    """convert_sam's rules for everything outside the image encoder (scripts/convert.py:210-247): mask_decoder. -> dec.,
    _token_to_image / _image_to_token -> _t2i / _i2t, the dense positional embedding precomputed (f32), iou / mask tokens f32,
    everything else f16."""
    out: dict[str, np.ndarray] = {}
    for key, t in sd.items():
        name = key.replace("mask_decoder.", "dec.").replace("_image_to_token.", "_i2t.").replace("_token_to_image.", "_t2i.")
        if key == "prompt_encoder.pe_layer.positional_encoding_gaussian_matrix":
            out["dec.dense_positional_embedding"] = sam_dense_positional_embedding(t)
        if name in ("dec.iou_token.weight", "dec.mask_tokens.weight"):
            out[name] = t.astype(np.float32)
            continue
        out[name] = t.astype(np.float16)
    return out


def mobile_sam_gguf_tensors(enc_sd: dict[str, np.ndarray], dec_sd: dict[str, np.ndarray]):
    tensors, conv2d = tinyvit_gguf_tensors(enc_sd)
    tensors.update(sam_decoder_gguf_tensors(dec_sd))
    return tensors, conv2d


def write_mobile_sam_gguf(path: str | Path, cfg: TinyVitConfig = TINYVIT_5M, seed: int = 0, enc_sd=None, dec_sd=None) -> Path:
    """The whole MobileSAM file: image encoder + prompt encoder + mask decoder."""
    enc_sd = enc_sd if enc_sd is not None else tinyvit_state_dict(cfg, seed)
    dec_sd = dec_sd if dec_sd is not None else sam_decoder_state_dict(seed + 1000)
    tensors, conv2d = mobile_sam_gguf_tensors(enc_sd, dec_sd)
    w = GGUFWriter(path, "mobile-sam")
    w.add_string("mobile-sam.tensor_data_layout", "whcn")
    w.add_uint32("general.quantization_version", 2)
    w.add_uint32("general.file_type", 1)
    if conv2d:
        w.add_array_i32("mobile-sam.conv2d_weights", conv2d)
    for name, t in tensors.items():
        w.add_tensor(name, t)
    w.write()
    return Path(path)


# ---- SWIN transformer encoder, the BiRefNet backbone (reference src/visp/arch/swin.cpp, scripts/convert.py:358-419) ----

@dataclass(frozen=True)
class SwinConfig:
    """swin_t_params (swin.cpp:266-275) by default; small instances for the parity tests keep head_dim = 32."""
    embed_dim: int = 96
    window_size: int = 7
    depths: tuple = (2, 2, 6, 2)
    n_heads: tuple = (3, 6, 12, 24)
    mlp_ratio: int = 4
    image_size: int = 1024
    name: str = "swin_t"


SWIN_T = SwinConfig()
SWIN_MINI = SwinConfig(embed_dim=32, depths=(2, 2, 2, 2), n_heads=(1, 2, 4, 8), image_size=224, name="swin_mini")


def swin_state_dict(cfg: SwinConfig = SWIN_T, seed: int = 0, prefix: str = "bb.") -> dict[str, np.ndarray]:
    """float32 tensors under the names of a BiRefNet checkpoint's backbone (timm-style Swin), torch shapes. The
    relative_position_index buffers are left out: the converter drops them (convert.py:391-392)."""
    rng = np.random.default_rng(seed)
    sd: dict[str, np.ndarray] = {}

    def normal(shape, std):
        return (rng.standard_normal(shape) * std).astype(np.float32)

    def lin(name, out_f, in_f, gain=1.0, bias=True):
        sd[f"{name}.weight"] = normal((out_f, in_f), gain / np.sqrt(in_f))
        if bias:
            sd[f"{name}.bias"] = normal((out_f,), 0.02)

    def norm(name, c):
        sd[f"{name}.weight"] = (1.0 + rng.standard_normal(c) * 0.05).astype(np.float32)
        sd[f"{name}.bias"] = normal((c,), 0.02)

    C = cfg.embed_dim
    sd[f"{prefix}patch_embed.proj.weight"] = normal((C, 3, 4, 4), 1.0 / np.sqrt(48))
    sd[f"{prefix}patch_embed.proj.bias"] = normal((C,), 0.02)
    norm(f"{prefix}patch_embed.norm", C)
    ws = cfg.window_size
    for l in range(4):
        c = C << l
        for b in range(cfg.depths[l]):
            p = f"{prefix}layers.{l}.blocks.{b}"
            norm(f"{p}.norm1", c)
            lin(f"{p}.attn.qkv", 3 * c, c, gain=1.5)
            sd[f"{p}.attn.relative_position_bias_table"] = normal(((2 * ws - 1) ** 2, cfg.n_heads[l]), 0.5)
            lin(f"{p}.attn.proj", c, c)
            norm(f"{p}.norm2", c)
            lin(f"{p}.mlp.fc1", cfg.mlp_ratio * c, c, gain=1.4)
            lin(f"{p}.mlp.fc2", c, cfg.mlp_ratio * c, gain=1.4)
        if l < 3:
            norm(f"{prefix}layers.{l}.downsample.norm", 4 * c)
            lin(f"{prefix}layers.{l}.downsample.reduction", 2 * c, 4 * c, bias=False)
    for l in range(4):
        norm(f"{prefix}norm{l}", C << l)
    return sd


def swin_gguf_tensors(sd: dict[str, np.ndarray]):
    """convert_birefnet's rules for the backbone tensors (convert.py:413-419): the patch_embed kernel is always stored NHWC
    (never listed in conv2d_weights), everything is written f16."""
    out: dict[str, np.ndarray] = {}
    for name, t in sd.items():
        if t.ndim == 4 and "patch_embed" in name:
            t = np.ascontiguousarray(t.transpose(0, 2, 3, 1))
        out[name] = t.astype(np.float16)
    return out, []


def write_swin_gguf(path: str | Path, cfg: SwinConfig = SWIN_T, seed: int = 0, sd: dict[str, np.ndarray] | None = None) -> Path:
    """A 'birefnet' GGUF holding the backbone only (metadata of convert.py:358-380)."""
    sd = sd if sd is not None else swin_state_dict(cfg, seed)
    tensors, _ = swin_gguf_tensors(sd)
    w = GGUFWriter(path, "birefnet")
    w.add_string("birefnet.tensor_data_layout", "whcn")
    w.add_string("swin.config", "tiny" if cfg.embed_dim == 96 else ("large" if cfg.embed_dim == 192 else cfg.name))
    w.add_int32("swin.embed_dim", cfg.embed_dim)
    if cfg.embed_dim not in (96, 192):  # test instances: the layer table travels in the file (not a reference key)
        w.add_int32("swin.window_size", cfg.window_size)
        w.add_array_i32("swin.depths", cfg.depths)
        w.add_array_i32("swin.n_heads", cfg.n_heads)
    w.add_int32("birefnet.image_size", cfg.image_size)
    w.add_int32("birefnet.image_multiple", 128)
    w.add_uint32("general.quantization_version", 2)
    w.add_uint32("general.file_type", 1)
    for name, t in tensors.items():
        w.add_tensor(name, t)
    w.write()
    return Path(path)


# ---- BiRefNet: SWIN backbone + squeeze block + decoder (reference src/visp/arch/birefnet.cpp, tests/test_birefnet.py:1025-1260) ----

def birefnet_state_dict(cfg: SwinConfig = SWIN_T, seed: int = 0) -> dict[str, np.ndarray]:
    """float32 tensors under the names convert_birefnet writes (convert.py:381-419: decoder_block -> block, atrous_conv /
    regular_conv -> conv, offset_conv -> offset, modulator_conv -> modulator), with every BatchNorm already fused the way the
    converter fuses it: conv + bn -> conv weight / bias (conv_in, conv_out, dec_att.conv1, global_avg_pool.1, gdt_convs_N.0),
    the ASPP branch norms -> bn.weight / bn.bias (mul + add). Channel table of the reference's Decoder for backbone width C0:
    lateral channels [16, 8, 4, 2] C0, squeeze 30 C0 -> 16 C0, decoder-block inter channels 64, ASPP planes 256."""
    sd = swin_state_dict(cfg, seed)
    rng = np.random.default_rng(seed + 7919)
    C0 = cfg.embed_dim
    assert C0 % 4 == 0

    def normal(shape, std):
        return (rng.standard_normal(shape) * std).astype(np.float32)

    def conv(name, out_c, in_c, k, bias=True, gain=1.0):
        sd[f"{name}.weight"] = normal((out_c, in_c, k, k), gain / np.sqrt(in_c * k * k))
        if bias:
            sd[f"{name}.bias"] = normal((out_c,), 0.05)

    def deform(name, in_c, out_c, k):  # DeformableConv2d: offsets about half a pixel, modulation around 1
        conv(f"{name}.conv.offset", 2 * k * k, in_c, k, gain=0.6)
        conv(f"{name}.conv.modulator", k * k, in_c, k, gain=1.0)
        conv(f"{name}.conv.conv", out_c, in_c, k, bias=False, gain=1.2)
        sd[f"{name}.bn.weight"] = (1.0 + rng.standard_normal(out_c) * 0.1).astype(np.float32)
        sd[f"{name}.bn.bias"] = normal((out_c,), 0.1)

    def dec_block(name, in_c, out_c, inter=64, planes=256):
        conv(f"{name}.conv_in", inter, in_c, 3, gain=1.3)
        deform(f"{name}.dec_att.aspp1", inter, planes, 1)
        for i, k in enumerate((1, 3, 7)):
            deform(f"{name}.dec_att.aspp_deforms.{i}", inter, planes, k)
        conv(f"{name}.dec_att.global_avg_pool.1", planes, inter, 1, gain=1.3)
        conv(f"{name}.dec_att.conv1", inter, 5 * planes, 1, gain=1.3)
        conv(f"{name}.conv_out", out_c, inter, 3, gain=1.3)

    ch = [16 * C0, 8 * C0, 4 * C0, 2 * C0]
    dec_block("squeeze_module.0", 30 * C0, ch[0])
    d = "decoder."
    ipt_out = {5: ch[0] // 8, 4: ch[0] // 8, 3: ch[1] // 8, 2: ch[2] // 8, 1: ch[3] // 8}
    for lvl, grid in ((5, 32), (4, 16), (3, 8), (2, 4), (1, 1)):
        conv(f"{d}ipt_blk{lvl}.conv1", 64, 3 * grid * grid, 3)
        conv(f"{d}ipt_blk{lvl}.conv_out", ipt_out[lvl], 64, 3)
    dec_block(f"{d}block4", ch[0] + ipt_out[5], ch[1])
    dec_block(f"{d}block3", ch[1] + ipt_out[4], ch[2])
    dec_block(f"{d}block2", ch[2] + ipt_out[3], ch[3])
    dec_block(f"{d}block1", ch[3] + ipt_out[2], ch[3] // 2)
    conv(f"{d}conv_out1.0", 1, ch[3] // 2 + ipt_out[1], 1)
    for lvl, c in ((4, ch[1]), (3, ch[2]), (2, ch[3])):
        conv(f"{d}lateral_block{lvl}.conv", c, c, 1)
        conv(f"{d}gdt_convs_{lvl}.0", 16, c, 3, gain=1.3)
        conv(f"{d}gdt_convs_attn_{lvl}.0", 1, 16, 1)
    return sd


def birefnet_gguf_tensors(sd: dict[str, np.ndarray]):
    """convert_birefnet's storage rules (convert.py:413-419): the SWIN patch_embed kernel NHWC, every other conv kernel stays
    OIHW and its tensor index is listed in conv2d_weights (--layout nchw, the converter's default); everything f16."""
    out: dict[str, np.ndarray] = {}
    conv2d: list[int] = []
    for name, t in sd.items():
        if t.ndim == 4:
            if "patch_embed" in name:
                t = np.ascontiguousarray(t.transpose(0, 2, 3, 1))
            else:
                conv2d.append(len(out))
        out[name] = t.astype(np.float16)
    return out, conv2d


def write_birefnet_gguf(path: str | Path, cfg: SwinConfig = SWIN_T, seed: int = 0, sd: dict[str, np.ndarray] | None = None) -> Path:
    sd = sd if sd is not None else birefnet_state_dict(cfg, seed)
    tensors, conv2d = birefnet_gguf_tensors(sd)
    w = GGUFWriter(path, "birefnet")
    w.add_string("birefnet.tensor_data_layout", "whcn")
    w.add_string("swin.config", "tiny" if cfg.embed_dim == 96 else ("large" if cfg.embed_dim == 192 else cfg.name))
    w.add_int32("swin.embed_dim", cfg.embed_dim)
    if cfg.embed_dim not in (96, 192):
        w.add_int32("swin.window_size", cfg.window_size)
        w.add_array_i32("swin.depths", cfg.depths)
        w.add_array_i32("swin.n_heads", cfg.n_heads)
    w.add_int32("birefnet.image_size", cfg.image_size)
    w.add_int32("birefnet.image_multiple", 128)
    w.add_uint32("general.quantization_version", 2)
    w.add_uint32("general.file_type", 1)
    if conv2d:
        w.add_array_i32("birefnet.conv2d_weights", conv2d)
    for name, t in tensors.items():
        w.add_tensor(name, t)
    w.write()
    return Path(path)
