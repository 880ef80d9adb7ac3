"""Checkpoint -> GGUF conversion for Depth-Anything-V2 (the `transformers` / .safetensors layout), following
the reference's scripts/convert.py:428-475 (convert_depth_anything) and :46-98 (Writer) rule by rule, so that a
file written here is interchangeable with one written by the reference's converter:

  * KV: general.architecture = "depthanything", depthanything.tensor_data_layout = "whcn" (default layout),
    dino.patch_size / dino.embed_dim from the patch-embedding kernel shape, depthanything.image_size = 518,
    dino.n_heads / dino.n_layers / depthanything.feature_layers from the embed dim (384 S, 768 B, 1024 L),
    general.file_type (1 = f16), depthanything.conv2d_weights (indices of OIHW kernels);
  * tensors in state-dict order, names unchanged (< 64 chars); patch-embed and reassemble `projection` kernels
    permuted to NHWC, ConvTranspose (`0.resize`, `1.resize`) untouched, other 2-D conv kernels left OIHW and
    listed in conv2d_weights; cls_token / position_embeddings kept f32, everything else cast to f16.
The third-party `gguf` package the reference uses is not required (own writer, vision.cpp_amd/gguf.py)."""
from __future__ import annotations

from pathlib import Path

import numpy as np

from .gguf import GGUFWriter
from .synth import gguf_tensors

_VARIANTS = {384: (6, 12, [2, 5, 8, 11]), 768: (12, 12, [2, 5, 8, 11]), 1024: (16, 24, [4, 11, 17, 23])}


def convert_depth_anything(state_dict: dict[str, np.ndarray], out_path: str | Path, image_size: int = 518) -> Path:
    if "pretrained.cls_token" in state_dict:
        raise ValueError("The converter is written for the transformers (.safetensors) version of the model; "
                         "the original weights (.pth) are not supported")  # convert.py:437-440
    key = "backbone.embeddings.patch_embeddings.projection.weight"
    if key not in state_dict:
        raise ValueError(f"not a Depth-Anything checkpoint: {key} missing")
    shape = state_dict[key].shape
    embed_dim, patch = int(shape[0]), int(shape[2])
    w = GGUFWriter(out_path, "depthanything")
    w.add_string("depthanything.tensor_data_layout", "whcn")
    w.add_int32("dino.patch_size", patch)
    w.add_int32("dino.embed_dim", embed_dim)
    w.add_int32("depthanything.image_size", image_size)
    if embed_dim in _VARIANTS:
        heads, layers, feats = _VARIANTS[embed_dim]
    else:  # non-standard (test-size) models: head_dim 64, taps at the four quarter points
        layers = 1 + max(int(k.split(".")[3]) for k in state_dict if k.startswith("backbone.encoder.layer."))
        heads, feats = max(1, embed_dim // 64), [layers * (i + 1) // 4 - 1 for i in range(4)]
    w.add_int32("dino.n_heads", heads)
    w.add_int32("dino.n_layers", layers)
    w.add_array_i32("depthanything.feature_layers", feats)
    tensors, conv2d = gguf_tensors({k: np.asarray(v, dtype=np.float32) for k, v in state_dict.items()})
    w.add_uint32("general.quantization_version", 2)
    w.add_uint32("general.file_type", 1)
    w.add_array_i32("depthanything.conv2d_weights", conv2d)
    for name, t in tensors.items():
        w.add_tensor(name, t)
    w.write()
    return Path(out_path)


def load_safetensors(path: str | Path) -> dict[str, np.ndarray]:
    from safetensors.numpy import load_file  # plain tensor container, nothing is executed from the file

    return load_file(str(path))
