#!/usr/bin/env python3
"""Golden vectors for the SWIN encoder (BiRefNet backbone). The reference's torch twin of this
module (tests/test_birefnet.py) cannot be imported here (timm and torchvision are absent), so the pin is HuggingFace
`transformers` SwinBackbone -- an independent implementation of the same architecture (pad after norm1, cyclic shift, shifted
window mask, relative position bias table, patch merging order, one LayerNorm per stage output) -- built from a local config
with seeded weights. Only data is stored; weights and the input are regenerated from seeds (vision.cpp_amd/synth.py), rounded
to f16 as in the GGUF. Sizes are chosen so that every stage map is larger than the window (transformers shrinks window and
shift when a map is not, the reference pads instead) and every stage needs padding: 256 x 288 -> maps 64x72, 32x36, 16x18, 8x9."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from transformers import SwinBackbone, SwinConfig  # noqa: E402
from visioncpp_amd import synth  # noqa: E402

OUT = Path(__file__).resolve().parent
torch.set_num_threads(8)


def f16r(a):
    return a.astype(np.float16).astype(np.float32)


def to_hf(sd, cfg, prefix="bb."):
    """BiRefNet / timm names -> transformers SwinBackbone names (qkv split into q, k, v)."""
    out = {}
    for k, v in sd.items():
        k = k[len(prefix):]
        v = torch.from_numpy(f16r(v))
        if k.startswith("patch_embed.proj"):
            out["swin.embeddings.patch_embeddings.projection" + k[len("patch_embed.proj"):]] = v
        elif k.startswith("patch_embed.norm"):
            out["swin.embeddings.norm" + k[len("patch_embed.norm"):]] = v
        elif k.startswith("norm"):
            stage = int(k[4]) + 1
            out[f"hidden_states_norms.stage{stage}" + k[5:]] = v
        else:
            k = "swin.encoder." + k
            k = k.replace(".norm1.", ".layernorm_before.").replace(".norm2.", ".layernorm_after.")
            k = k.replace(".attn.proj.", ".attention.o_proj.")
            k = k.replace(".attn.relative_position_bias_table", ".attention.relative_position_bias.relative_position_bias_table")
            if ".attn.qkv." in k:
                c = v.shape[0] // 3
                for i, n in enumerate(("q_proj", "k_proj", "v_proj")):
                    out[k.replace(".attn.qkv.", f".attention.{n}.")] = v[i * c:(i + 1) * c].clone()
                continue
            out[k] = v
    return out


def swin_input(w, h, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([np.sin(xx * 0.05 + c) + np.cos(yy * 0.07 - c) for c in range(3)], -1) * 0.8 + rng.standard_normal((h, w, 3)) * 0.5
    return img.astype(np.float32)


if __name__ == "__main__":
    cfg = synth.SWIN_MINI
    W, H = 256, 288
    sd = synth.swin_state_dict(cfg, seed=3)
    hf_cfg = SwinConfig(image_size=224, patch_size=4, num_channels=3, embed_dim=cfg.embed_dim, depths=list(cfg.depths), num_heads=list(cfg.n_heads),
                        window_size=cfg.window_size, mlp_ratio=float(cfg.mlp_ratio), qkv_bias=True, hidden_act="gelu", layer_norm_eps=1e-5,
                        hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, drop_path_rate=0.0, use_absolute_embeddings=False,
                        out_features=["stage1", "stage2", "stage3", "stage4"])
    net = SwinBackbone(hf_cfg)
    state = to_hf(sd, cfg)
    res = net.load_state_dict(state, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert all("relative_position_index" in k or k.startswith("swin.layernorm.") for k in res.missing_keys), res.missing_keys  # swin.layernorm only feeds last_hidden_state
    net.eval()
    img = swin_input(W, H, seed=17)
    with torch.no_grad():
        out = net(torch.from_numpy(img.transpose(2, 0, 1)[None]), output_hidden_states=True)
    feats = {f"stage{i}": o[0].permute(1, 2, 0).contiguous().numpy() for i, o in enumerate(out.feature_maps)}  # NHWC
    # block-level vectors: the first stage's two blocks (unshifted + shifted, padded 64x72 -> 70x77) on their own
    with torch.no_grad():
        emb, dims = net.swin.embeddings(torch.from_numpy(img.transpose(2, 0, 1)[None]))
        layer0 = net.swin.encoder.layers[0]
        b0 = layer0.blocks[0](emb, dims)[0]
        b1 = layer0.blocks[1](b0, dims)[0]
        merged = layer0.downsample(b1, dims)
    # token subsets keep the fixture small: rows [::step] of the [tokens, C] views (steps stored next to the data)
    steps = {"patch_embed": 9, "block0": 5, "block1": 5, "merged0": 3, "stage0": 3, "stage1": 2, "stage2": 1, "stage3": 1}
    full = {"patch_embed": emb[0].numpy(), "block0": b0[0].numpy(), "block1": b1[0].numpy(), "merged0": merged[0].numpy()}
    full.update({k: v.reshape(-1, v.shape[-1]) for k, v in feats.items()})
    np.savez_compressed(OUT / "swin_mini.npz", W=W, H=H, input_seed=17, weight_seed=3, **{k: v[::steps[k]] for k, v in full.items()},
                        **{f"step_{k}": s for k, s in steps.items()})
    print({k: v.shape for k, v in feats.items()}, emb.shape, b0.shape, merged.shape)
