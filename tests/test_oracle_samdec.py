"""Pins the MobileSAM prompt-encoder / mask-decoder part of the oracle against outputs of the reference's own torch modules
(tests/golden/make_golden_samdec.py imports reference tests/test_mobile_sam.py PromptEncoder, TwoWayTransformer, MaskDecoder in
the CPU container): full-size configuration, one point prompt and one box prompt. Tolerances are the reference's own for
these modules (test_mobile_sam.py:1426-1470: rtol 1e-2 / atol 1e-2 on masks, rtol 1e-2 on iou)."""
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle as O
from visioncpp_amd import synth

GOLD = Path(__file__).parent / "golden"


@pytest.fixture(scope="module")
def dec():
    g = np.load(GOLD / "samdec.npz")
    tensors = synth.sam_decoder_gguf_tensors(synth.sam_decoder_state_dict(int(g["weights_seed"])))
    embed = np.random.default_rng(int(g["embed_seed"])).standard_normal((64, 64, 256)).astype(np.float32)
    return g, O.Model(tensors, [], "whcn"), embed


def test_gguf_contract_of_the_decoder():
    t = synth.sam_decoder_gguf_tensors(synth.sam_decoder_state_dict(0))
    assert t["dec.iou_token.weight"].dtype == np.float32 and t["dec.mask_tokens.weight"].dtype == np.float32   # convert.py:243-245
    assert t["dec.dense_positional_embedding"].dtype == np.float32 and t["dec.dense_positional_embedding"].shape == (64, 64, 256)
    assert "dec.transformer.layers.1.cross_attn_i2t.out_proj.weight" in t and "dec.transformer.final_attn_t2i.q_proj.bias" in t
    assert not any("mask_decoder" in k or "token_to_image" in k for k in t)
    assert t["dec.output_upscaling.0.weight"].shape == (256, 64, 2, 2) and t["dec.output_upscaling.0.weight"].dtype == np.float16


def test_prompt_coordinates():
    """sam_process_point / sam_process_box (mobile-sam.cpp:213-236): pixel centre, longest side -> 1024, then [-1, 1]."""
    np.testing.assert_allclose(O.sam_process_prompt([0, 0], 1024, 1024), [2 * 0.5 / 1024 - 1, 2 * 0.5 / 1024 - 1, 0, 0], atol=1e-7)
    np.testing.assert_allclose(O.sam_process_prompt([100, 50, 300, 200], 512, 384), [2 * 200.5 / 1024 - 1, 2 * 100.5 / 1024 - 1, 2 * 600.5 / 1024 - 1, 2 * 400.5 / 1024 - 1], atol=1e-6)


def test_dense_positional_embedding_matches_the_reference_module(dec):
    g, m, _ = dec
    pe = synth.sam_dense_positional_embedding(synth.sam_decoder_state_dict(int(g["weights_seed"]))["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"].astype(np.float16).astype(np.float32))
    np.testing.assert_allclose(pe[::8, ::8], g["dense_pe_sample"], atol=2e-5)


@pytest.mark.parametrize("name,prompt", [("point", [700, 300]), ("box", [200, 120, 640, 900])])
def test_prompt_encoder_and_mask_decoder_match_reference_torch(dec, name, prompt):
    g, m, embed = dec
    coords = O.sam_process_prompt(prompt, 1024, 1024)
    sparse = O.sam_embed_prompt(m, coords, name == "box")
    np.testing.assert_allclose(sparse, g[f"{name}_sparse"], rtol=1e-3, atol=2e-4)   # sin / cos of O(10) arguments in f32
    masks, iou = O.sam_predict_masks(m, embed, sparse)
    np.testing.assert_allclose(masks[:, ::4, ::4], g[f"{name}_masks_sample"], rtol=1e-2, atol=1e-2)
    err = np.abs(masks[:, ::4, ::4] - g[f"{name}_masks_sample"])
    assert err.mean() < 2e-3, err.mean()   # far inside the reference's bound: same f16-rounded weights, f32 math on both sides
    np.testing.assert_allclose(iou, g[f"{name}_iou"], rtol=1e-2, atol=1e-3)


def test_process_mask_and_compute(dec):
    """sam_process_mask (mobile-sam.cpp:556-583): 256 -> 1024 bilinear, crop to the scaled extent, resize + threshold; and
    sam_compute_impl's choice of the best of the FIRST THREE masks (vision.cpp:78-83)."""
    g, m, embed = dec
    out, iou, masks = O.sam_compute(m, embed, 640, 480, [320, 200], return_all=True)
    assert out.shape == (480, 640) and set(np.unique(out)) <= {0, 255}
    idx = int(np.argmax(iou[:3]))
    assert np.array_equal(out, O.sam_process_mask(masks[idx], 640, 480))
    # a constant-sign mask stays constant; a half-plane keeps its edge where the scaled image puts it
    assert (O.sam_process_mask(np.full((256, 256), 3.0, np.float32), 100, 80) == 255).all()
    half = np.where(np.arange(256)[None, :] < 128, 1.0, -1.0).astype(np.float32) * np.ones((256, 1), np.float32)
    pm = O.sam_process_mask(half, 200, 100)   # longest side 200 -> 1024: the x midpoint of the mask maps to x = 100
    assert (pm[:, :99] == 255).all() and (pm[:, 101:] == 0).all()
    with pytest.raises(RuntimeError, match="must be 2 or 4"):
        O.sam_compute(m, embed, 64, 64, [1, 2, 3])
