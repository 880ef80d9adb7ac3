#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ (run in the CPU container).

Sources of truth (none of them is this repo's code):
  * torch.nn.functional ops, with the inputs/shapes the reference's tests/test_primitives.py
    uses to pin ggml's ops (linear :20-28, layer_norm :112-121, interpolate :165-184,
    conv_transpose2d :66-91) -> ops.npz
  * HuggingFace transformers' DepthAnythingForDepthEstimation built from a LOCAL config with
    weights from vision.cpp_amd/synth.py (seeded; rounded through f16 as the GGUF stores
    them), GELU = tanh approximation (what ggml_gelu computes), f32 math
    -> depthany_tiny.npz (every module boundary), depthany_mini.npz, depthany_small.npz
       (518x518 output, strided sample + statistics)
Nothing from /root/reference is read or copied. Weights are regenerated from the seed at
test time; a checksum of the state dict is stored to catch RNG drift.
"""
import hashlib
import os
import sys
from functools import reduce
from pathlib import Path

os.environ["HF_HUB_OFFLINE"] = "1"
import numpy as np
import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
from visioncpp_amd import synth  # noqa: E402

OUT = Path(__file__).resolve().parent
torch.manual_seed(0)
torch.set_num_threads(8)


def input_tensor(*shape):  # reference tests/workbench.py:260-262
    end = reduce(lambda x, y: x * y, shape, 1)
    return torch.arange(0, end).reshape(*shape) / end


def make_ops():
    d = {}
    g = torch.Generator().manual_seed(1)
    # linear (test_primitives.py:20-28)
    x = torch.rand(2, 5, generator=g)
    w = torch.rand(3, 5, generator=g)
    b = torch.tensor([7, 21, -5]).float()
    d["linear_x"], d["linear_w"], d["linear_b"] = x, w, b
    d["linear_y"] = F.linear(x, w, b)
    # layer_norm (:112-121), eps 1e-5 and the 1e-6 DINO uses
    x = torch.rand(4, 5, 20, generator=g)
    w = torch.rand(20, generator=g)
    b = torch.rand(20, generator=g)
    d["ln_x"], d["ln_w"], d["ln_b"] = x, w, b
    d["ln_y_1e5"] = F.layer_norm(x, [20], w, b, eps=1e-5)
    d["ln_y_1e6"] = F.layer_norm(x, [20], w, b, eps=1e-6)
    # interpolate (:165-184): arange inputs, NCHW
    for size, (bb, c, h, wd) in {"one": (1, 2, 1, 3), "small": (1, 3, 2, 3), "large": (4, 19, 20, 30)}.items():
        x = torch.arange(bb * c * h * wd).reshape(bb, c, h, wd).float()
        for scale in (0.6, 2.0):
            target = (round(h * scale), round(wd * scale))
            for mode in ("bilinear", "bicubic"):
                for ac in (True, False):
                    y = F.interpolate(x, size=target, mode=mode, align_corners=ac)
                    d[f"interp_{size}_{scale}_{mode}_{int(ac)}"] = y
    # conv_transpose2d (:66-91)
    for name, (k, s) in {"3x3": (3, 1), "5x5": (5, 1), "stride2": (3, 2)}.items():
        x = input_tensor(2, 11, 4, 5)
        w = input_tensor(11, 2, k, k)
        d[f"convT_{name}"] = F.conv_transpose2d(x, w, None, stride=s)
    # the two shapes Depth-Anything uses (k == stride), with bias
    for name, (c, k) in {"k4s4": (6, 4), "k2s2": (10, 2)}.items():
        x = torch.rand(2, c, 5, 7, generator=g) - 0.5
        w = torch.rand(c, c, k, k, generator=g) - 0.5
        b = torch.rand(c, generator=g)
        d[f"convT_{name}_x"], d[f"convT_{name}_w"], d[f"convT_{name}_b"] = x, w, b
        d[f"convT_{name}_y"] = F.conv_transpose2d(x, w, b, stride=k)
    # conv2d: 3x3 s1 p1, 3x3 s2 p1, 1x1, 14x14 s14 (the four forms on the path)
    for name, (ci, co, k, s, p, h, wd) in {"3x3": (5, 7, 3, 1, 1, 9, 11), "3x3s2": (6, 4, 3, 2, 1, 9, 11),
                                            "1x1": (8, 3, 1, 1, 0, 4, 5), "patch": (3, 6, 14, 14, 0, 28, 42)}.items():
        x = torch.rand(2, ci, h, wd, generator=g) - 0.5
        w = torch.rand(co, ci, k, k, generator=g) - 0.5
        b = torch.rand(co, generator=g)
        d[f"conv_{name}_x"], d[f"conv_{name}_w"], d[f"conv_{name}_b"] = x, w, b
        d[f"conv_{name}_y"] = F.conv2d(x, w, b, stride=s, padding=p)
    # attention: softmax(q k^T / sqrt(hd)) v, 2 heads x 8, 13 tokens
    q, k, v = (torch.rand(13, 16, generator=g) - 0.5 for _ in range(3))
    qh, kh, vh = (t.reshape(13, 2, 8).permute(1, 0, 2) for t in (q, k, v))
    o = F.scaled_dot_product_attention(qh, kh, vh).permute(1, 0, 2).reshape(13, 16)
    d["attn_q"], d["attn_k"], d["attn_v"], d["attn_o"] = q, k, v, o
    # gelu tanh
    x = torch.linspace(-12, 12, 4001)
    d["gelu_x"] = x
    d["gelu_tanh"] = F.gelu(x, approximate="tanh")
    np.savez_compressed(OUT / "ops.npz", **{k: v.numpy() for k, v in d.items()})
    print("ops.npz:", len(d), "arrays")


def sd_checksum(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(v.astype(np.float16)).tobytes())
    return h.hexdigest()


def build_hf(cfg: synth.Config, seed: int):
    from transformers import DepthAnythingConfig, DepthAnythingForDepthEstimation, Dinov2Config

    bc = Dinov2Config(hidden_size=cfg.embed_dim, num_hidden_layers=cfg.n_layers, num_attention_heads=cfg.n_heads,
                      image_size=cfg.image_size, patch_size=cfg.patch_size, mlp_ratio=cfg.mlp_ratio,
                      out_features=[f"stage{i + 1}" for i in cfg.feature_layers], apply_layernorm=True,
                      reshape_hidden_states=False, hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-6)
    hc = DepthAnythingConfig(backbone_config=bc, reassemble_hidden_size=cfg.embed_dim, patch_size=cfg.patch_size,
                             neck_hidden_sizes=list(cfg.neck_sizes), fusion_hidden_size=cfg.fusion_size,
                             head_hidden_size=cfg.head_size, reassemble_factors=[4, 2, 1, 0.5])
    model = DepthAnythingForDepthEstimation(hc).eval()
    sd = synth.state_dict(cfg, seed)
    hf_keys = list(model.state_dict().keys())
    assert hf_keys == list(sd.keys()), "synth.state_dict order/names differ from transformers"
    # weights as the GGUF stores them: f16 except cls/pos (scripts/convert.py:471-473), then f32 math
    tsd = {}
    for k, v in sd.items():
        keep32 = "position_embeddings" in k or "cls_token" in k
        tsd[k] = torch.from_numpy(v if keep32 else v.astype(np.float16).astype(np.float32))
    model.load_state_dict(tsd, strict=True)
    return model, sd


def preprocess(img_u8: np.ndarray) -> torch.Tensor:
    """(u8/255 - mean)/std in f32 (reference depth-anything.cpp:130-140), -> NCHW."""
    mean = np.array([0.485, 0.456, 0.406], np.float32)
    std = np.array([0.229, 0.224, 0.225], np.float32)
    x = (img_u8.astype(np.float32) / np.float32(255.0) + (-mean)) * (np.float32(1.0) / std)
    return torch.from_numpy(x).permute(2, 0, 1)[None].contiguous()


def run_with_captures(model, x):
    caps = {}

    def nhwc(t):
        return t.detach().permute(0, 2, 3, 1).contiguous().numpy()[0]

    hooks = []
    bb = model.backbone
    hooks.append(bb.embeddings.register_forward_hook(lambda m, i, o: caps.__setitem__("tokens", o.detach().numpy()[0])))
    for li, layer in enumerate(bb.encoder.layer):
        def hk(m, i, o, li=li):
            t = o[0] if isinstance(o, tuple) else o
            caps[f"layer_{li}"] = t.detach().numpy()[0]
        hooks.append(layer.register_forward_hook(hk))
    for i, l in enumerate(model.neck.reassemble_stage.layers):
        hooks.append(l.register_forward_hook(lambda m, inp, o, i=i: caps.__setitem__(f"reassemble_{i}", nhwc(o))))
    for i, l in enumerate(model.neck.convs):
        hooks.append(l.register_forward_hook(lambda m, inp, o, i=i: caps.__setitem__(f"neck_conv_{i}", nhwc(o))))
    for i, l in enumerate(model.neck.fusion_stage.layers):
        hooks.append(l.register_forward_hook(lambda m, inp, o, i=i: caps.__setitem__(f"fusion_{i}", nhwc(o))))
    hooks.append(model.head.conv1.register_forward_hook(lambda m, inp, o: caps.__setitem__("head_conv1", nhwc(o))))
    with torch.no_grad():
        out = model(pixel_values=x)
        fm = model.backbone(x).feature_maps
    for h in hooks:
        h.remove()
    depth = out.predicted_depth.numpy()[0]
    return depth, caps, [f.numpy()[0] for f in fm]


def make_model_fixture(cfg: synth.Config, seed: int, w: int, h: int, img_seed: int, full: bool):
    model, sd = build_hf(cfg, seed)
    img = synth.images(1, w, h, seed=img_seed)[0]
    depth, caps, fm = run_with_captures(model, preprocess(img))
    d = {"image_seed": np.int64(img_seed), "weights_seed": np.int64(seed), "extent": np.array([w, h]),
         "sd_sha256": np.frombuffer(sd_checksum(sd).encode(), np.uint8)}
    if full:
        d["depth"] = depth
        for k, v in caps.items():
            d[k] = v
        for f, li in zip(fm, cfg.feature_layers):
            d[f"dino_layer_{li}"] = f
    else:
        st = 7 if w > 200 else 1
        d["depth_sample"] = depth[::st, ::st].copy()
        d["depth_stats"] = np.array([depth.min(), depth.max(), depth.mean(), depth.std()], np.float64)
        d["tokens_sample"] = caps["tokens"][::37, ::8].copy()
        for li in cfg.feature_layers:
            d[f"layer_{li}_sample"] = caps[f"layer_{li}"][::37, ::8].copy()
        d["fusion_3_sample"] = caps["fusion_3"][::8, ::8, ::4].copy()
        d["head_conv1_sample"] = caps["head_conv1"][::8, ::8, ::4].copy()
    name = f"depthany_{cfg.name}{'' if (w, h) == (cfg.image_size, cfg.image_size) else f'_{w}x{h}'}.npz"
    np.savez_compressed(OUT / name, **d)
    print(name, "depth min/max/mean", depth.min(), depth.max(), depth.mean(), "frac zero", (depth == 0).mean())


if __name__ == "__main__":
    make_ops()
    make_model_fixture(synth.TINY, seed=3, w=70, h=70, img_seed=11, full=True)
    make_model_fixture(synth.MINI, seed=4, w=112, h=112, img_seed=12, full=False)
    make_model_fixture(synth.SMALL, seed=0, w=518, h=518, img_seed=1234, full=False)
