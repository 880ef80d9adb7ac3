"""Pins the TinyViT (MobileSAM image encoder) part of the oracle against outputs of the reference's own torch modules
(tests/golden/make_golden_tinyvit.py imports reference tests/test_mobile_sam.py TinyViT in the CPU container): the full
5M configuration at 1024x1024, every stage boundary. Tolerance as the reference's own module tests use for this family
(test_mobile_sam.py: rtol 1e-3, atol 0.02 -- ggml's GELU is the tanh form through an f16 table, torch's Mlp uses erf)."""
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle as O
from visioncpp_amd import synth

GOLD = Path(__file__).parent / "golden"


@pytest.fixture(scope="module")
def enc():
    g = np.load(GOLD / "tinyvit_5m.npz")
    cfg = synth.TINYVIT_5M
    tensors, conv2d = synth.tinyvit_gguf_tensors(synth.tinyvit_state_dict(cfg, int(g["weights_seed"])))
    return g, cfg, O.Model(tensors, conv2d, "whcn")


def test_gguf_contract(enc):
    g, cfg, m = enc
    tensors, conv2d = synth.tinyvit_gguf_tensors(synth.tinyvit_state_dict(cfg, 2))
    names = list(tensors)
    assert "enc.patch_embed.seq.0.c.weight" in names and "enc.patch_embed.seq.0.c.bias" in names
    assert not any(".bn." in n for n in names)                                   # BatchNorm fused away (convert.py:157-188)
    assert tensors["enc.layers.2.blocks.0.attn.attention_biases_indexed"].shape == (5, 196, 196)
    assert tensors["enc.layers.1.blocks.0.local_conv.c.weight"].shape == (3, 3, 1, 128)     # always NHWC (convert.py:224-229)
    listed = {names[i] for i in conv2d}
    assert "enc.neck.2.weight" in listed and "enc.layers.1.blocks.0.local_conv.c.weight" not in listed


def _run(enc):
    g, cfg, m = enc
    img = synth.images(1, cfg.img_size, cfg.img_size, seed=int(g["image_seed"]))[0].astype(np.float32) / np.float32(255.0)
    x = (img - np.array([0.485, 0.456, 0.406], np.float32)) / np.array([0.229, 0.224, 0.225], np.float32)
    layers = cfg.layers()
    caps = {"patch_embed": 256 * 256 * 64, **{f"layer_{i}": layers[min(i + 1, 3)][0] ** 2 * layers[min(i + 1, 3)][1] for i in range(4)}}
    return O.tinyvit_encode(m, O.tinyvit_params(cfg.img_size, layers), x, captures=caps)


def test_structure_matches_reference_torch_exactly(enc):
    """With the GELU forms of the torch twin (tanh inside MBConv, exact elsewhere) the restatement reproduces the
    reference's torch TinyViT at every stage boundary to float rounding: every index, padding and fusion rule agrees."""
    g = enc[0]
    O.tinyvit_set_gelu_modes(O.GELU_TANH_F32, O.GELU_ERF_F32)
    try:
        y, c = _run(enc)
    finally:
        O.tinyvit_set_gelu_modes()
    np.testing.assert_allclose(c["patch_embed"].reshape(256, 256, 64)[::16, ::16], g["patch_embed_sample"], rtol=1e-4, atol=1e-4)
    for i in range(4):
        got = c[f"layer_{i}"].reshape(-1, g[f"layer_{i}_sample"].shape[1])[::37]
        np.testing.assert_allclose(got, g[f"layer_{i}_sample"], rtol=1e-3, atol=2e-4, err_msg=f"layer_{i}")
    # the neck's LayerNorm2d uses eps 1e-6 in torch and layer_norm's default 1e-5 in the reference C++ (mobile-sam.cpp:201-205)
    np.testing.assert_allclose(y[::4, ::4], g["result_sample"], rtol=1e-3, atol=2e-3)


def test_reference_gelu_form_stays_within_its_module_tolerance(enc):
    """The reference's own form (ggml_gelu: tanh approximation through an f16 table, also in the Mlp) drifts from the
    torch twin by what its module tests allow per module (atol 0.02), a little more after 12 stacked blocks."""
    g = enc[0]
    y, c = _run(enc)
    np.testing.assert_allclose(c["patch_embed"].reshape(256, 256, 64)[::16, ::16], g["patch_embed_sample"], rtol=1e-3, atol=0.02)
    for i in range(4):
        got = c[f"layer_{i}"].reshape(-1, g[f"layer_{i}_sample"].shape[1])[::37]
        err = np.abs(got - g[f"layer_{i}_sample"])
        assert err.mean() < 5e-3 and err.max() < 0.08, (i, err.mean(), err.max())
    err = np.abs(y[::4, ::4] - g["result_sample"])
    assert err.mean() < 0.01 and err.max() < 0.15, (err.mean(), err.max())
