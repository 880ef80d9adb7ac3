"""CPU suite: the oracle's SWIN encoder (oracle/visp_oracle.c vo_swin_*, restating reference src/visp/arch/swin.cpp) pinned
against tests/golden/swin_mini.npz -- HuggingFace transformers' SwinBackbone with the same seeded, f16-rounded weights
(generator: tests/golden/make_golden_swin.py; the reference's own torch twin in tests/test_birefnet.py needs timm and
torchvision, which are not importable here) -- plus the index / mask helpers against independent numpy restatements of what
the reference's tests compare them with (tests/test_birefnet.py:440-482: the original Swin's 3-slice image mask)."""
import sys
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle
from visioncpp_amd import synth

sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))


def _input(w, h, seed):  # the generator's input (tests/golden/make_golden_swin.py swin_input), restated so the fixture holds no image
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([np.sin(xx * 0.05 + c) + np.cos(yy * 0.07 - c) for c in range(3)], -1) * 0.8 + rng.standard_normal((h, w, 3)) * 0.5
    return img.astype(np.float32)


def _model(cfg, seed):
    tensors, conv_idx = synth.swin_gguf_tensors(synth.swin_state_dict(cfg, seed))
    return oracle.Model(tensors, conv_idx)


def test_relative_position_index_matches_the_original_swin_formula():
    for ws in (2, 3, 7):
        coords = np.stack(np.meshgrid(np.arange(ws), np.arange(ws), indexing="ij")).reshape(2, -1)
        rel = (coords[:, :, None] - coords[:, None, :]).transpose(1, 2, 0) + (ws - 1)
        want = rel[..., 0] * (2 * ws - 1) + rel[..., 1]                      # [query, key]
        np.testing.assert_array_equal(oracle.swin_rel_pos_index(ws).reshape(ws * ws, ws * ws), want)


@pytest.mark.parametrize("w,h,ws", [(18, 18, 6), (20, 13, 7), (8, 9, 7), (64, 72, 7)])
def test_attention_mask_matches_the_three_slice_image_mask(w, h, ws):
    """BasicLayer.attention_mask of the original Swin: label the padded map by the 3 x 3 slices (0..-ws, -ws..-shift, -shift..),
    partition, compare labels pairwise; the reference writes -inf where they differ (test_birefnet.py:440-460)."""
    shift = ws // 2
    hp, wp = -(-h // ws) * ws, -(-w // ws) * ws
    img = np.zeros((hp, wp), np.int32)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[hs, wsl] = cnt
            cnt += 1
    win = img.reshape(hp // ws, ws, wp // ws, ws).transpose(0, 2, 1, 3).reshape(-1, ws * ws)
    want = np.where(win[:, None, :] != win[:, :, None], -np.inf, 0.0).astype(np.float32)
    np.testing.assert_array_equal(oracle.swin_attention_mask(w, h, ws), want)


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(golden_dir / "swin_mini.npz")


def test_swin_encoder_matches_transformers_fixture(golden):
    cfg = synth.SWIN_MINI
    W, H = int(golden["W"]), int(golden["H"])
    model = _model(cfg, int(golden["weight_seed"]))
    img = _input(W, H, int(golden["input_seed"]))
    P = oracle.swin_params(cfg.embed_dim, cfg.window_size, cfg.depths, cfg.n_heads)
    oracle.tinyvit_set_gelu_modes(other_mode=oracle.GELU_ERF_F32)  # transformers' "gelu" is the exact form
    oracle.swin_set_mask_mode(True)  # transformers (like the reference's torch twin) masks the shifted blocks only
    try:
        T = (W // 4) * (H // 4)
        feats, caps = oracle.swin_encode(model, P, img, "bb", captures={"patch_embed": T * 32, "block_0_0": T * 32, "block_0_1": T * 32})
        merged = oracle.swin_patch_merging(model, "bb.layers.0.downsample", caps["block_0_1"].reshape(T, 32), W // 4, H // 4)
    finally:
        oracle.tinyvit_set_gelu_modes()
        oracle.swin_set_mask_mode(False)
    for name, got in (("patch_embed", caps["patch_embed"]), ("block0", caps["block_0_0"]), ("block1", caps["block_0_1"])):
        want = golden[name]
        got = got.reshape(T, 32)[:: int(golden[f"step_{name}"])]
        assert np.abs(got - want).max() < 2e-4 * max(1.0, np.abs(want).max()), name
    want = golden["merged0"]
    assert np.abs(merged[:: int(golden["step_merged0"])] - want).max() < 2e-4 * np.abs(want).max()
    dims = [(H // 4 >> i, W // 4 >> i, 32 << i) for i in range(4)]
    for i, f in enumerate(feats):
        assert f.shape == dims[i]
        want = golden[f"stage{i}"]
        got = f.reshape(-1, f.shape[-1])[:: int(golden[f"step_stage{i}"])]
        assert np.abs(got - want).max() < 1e-3 * np.abs(want).max(), i  # rtol of the reference's tensors_match


def test_swin_block_properties():
    """Properties the fixture does not show: an unshifted block on a map that needs no padding treats windows independently
    (changing one window leaves the others untouched); a shifted block mixes across the shift; errors are reported."""
    cfg = synth.SWIN_MINI
    model = _model(cfg, 1)
    rng = np.random.default_rng(0)
    w = h = 14
    x = rng.standard_normal((w * h, 32)).astype(np.float32)
    y0 = oracle.swin_block(model, "bb.layers.0.blocks.0", x, w, h, 1, 7, 0)
    x2 = x.copy().reshape(h, w, 32)
    x2[:7, :7] += rng.standard_normal((7, 7, 32)).astype(np.float32)  # (a constant offset would vanish in norm1)
    y1 = oracle.swin_block(model, "bb.layers.0.blocks.0", x2.reshape(-1, 32), w, h, 1, 7, 0).reshape(h, w, 32)
    y0 = y0.reshape(h, w, 32)
    np.testing.assert_array_equal(y1[7:, :], y0[7:, :])
    np.testing.assert_array_equal(y1[:7, 7:], y0[:7, 7:])
    assert np.abs(y1[:7, :7] - y0[:7, :7]).max() > 0.1
    ys = oracle.swin_block(model, "bb.layers.0.blocks.1", x2.reshape(-1, 32), w, h, 1, 7, 3).reshape(h, w, 32)
    yb = oracle.swin_block(model, "bb.layers.0.blocks.1", x, w, h, 1, 7, 3).reshape(h, w, 32)
    assert np.abs(ys[7:10, 7:10] - yb[7:10, 7:10]).max() > 1e-3  # the shifted window spans rows/cols 3..9
    with pytest.raises(RuntimeError, match="even spatial"):
        oracle.swin_patch_merging(model, "bb.layers.0.downsample", x[: 7 * 14], 7, 14)
    with pytest.raises(RuntimeError, match="multiple of the patch size"):
        oracle.swin_encode(model, oracle.swin_params(32, 7, cfg.depths, cfg.n_heads), np.zeros((30, 32, 3), np.float32))


def test_unshifted_block_mask_semantics_follow_the_reference_as_written():
    """The reference's swin::layer hands the layer's attn_mask to every block and swin::block forwards it unconditionally
    (swin.cpp:128-139, 226-237): the -inf edge mask of compute_attention_mask also acts in UNSHIFTED blocks, in the windows of
    the last row / column. Its torch twin (tests/test_birefnet.py:249-255) masks shifted blocks only. The default here is the
    reference as written; this pins what that means on a 14x14 map (2x2 windows of 7, shift zone = last 3 rows / columns):
    with the mask, the tokens of the shift zone and the rest of an edge window do not see each other."""
    cfg = synth.SWIN_MINI
    model = _model(cfg, 1)
    rng = np.random.default_rng(5)
    w = h = 14
    x = rng.standard_normal((w * h, 32)).astype(np.float32)
    assert oracle.swin_get_mask_mode() is False  # default: reference as written
    pfx = "bb.layers.0.blocks.0"
    y_ref = oracle.swin_block(model, pfx, x, w, h, 1, 7, 0).reshape(h, w, 32)                 # default = masked
    y_msk = oracle.swin_block(model, pfx, x, w, h, 1, 7, 0, masked=True).reshape(h, w, 32)
    y_tor = oracle.swin_block(model, pfx, x, w, h, 1, 7, 0, masked=False).reshape(h, w, 32)   # torch twin
    np.testing.assert_array_equal(y_ref, y_msk)
    # the interior window (0, 0) carries no mask: identical either way; every edge window differs
    np.testing.assert_array_equal(y_msk[:7, :7], y_tor[:7, :7])
    for sl in ((slice(0, 7), slice(7, 14)), (slice(7, 14), slice(0, 7)), (slice(7, 14), slice(7, 14))):
        assert np.abs(y_msk[sl] - y_tor[sl]).max() > 1e-3
    # masked: perturbing the non-shift part (rows 7..10) of window (1, 0) leaves its shift-zone rows 11..13 bit-identical;
    # unmasked they change
    x2 = x.copy().reshape(h, w, 32)
    x2[7:11, 0:7] += rng.standard_normal((4, 7, 32)).astype(np.float32)
    y2_msk = oracle.swin_block(model, pfx, x2.reshape(-1, 32), w, h, 1, 7, 0, masked=True).reshape(h, w, 32)
    y2_tor = oracle.swin_block(model, pfx, x2.reshape(-1, 32), w, h, 1, 7, 0, masked=False).reshape(h, w, 32)
    np.testing.assert_array_equal(y2_msk[11:14, 0:7], y_msk[11:14, 0:7])
    assert np.abs(y2_tor[11:14, 0:7] - y_tor[11:14, 0:7]).max() > 1e-4
    # and swin_encode composes blocks per the mode: stage outputs differ between the two modes on a map with edge windows
    P = oracle.swin_params(cfg.embed_dim, cfg.window_size, cfg.depths, cfg.n_heads)
    img = rng.standard_normal((64, 64, 3)).astype(np.float32)
    a = oracle.swin_encode(model, P, img)
    oracle.swin_set_mask_mode(True)
    try:
        b = oracle.swin_encode(model, P, img)
    finally:
        oracle.swin_set_mask_mode(False)
    assert np.abs(a[0] - b[0]).max() > 1e-4
