"""Pins the ESRGAN part of the oracle:
  - tile_layout / tile_merge against the literal vectors of the reference's tests/test-image.cpp:303-358,
  - RDB and whole RRDBNet against fixtures generated from the reference's own torch modules
    (tests/golden/make_golden_esrgan.py imports reference tests/test_esrgan.py:70-212 in the CPU container).
"""
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle as O
from visioncpp_amd import synth

GOLD = Path(__file__).parent / "golden"


def _model(cfg, seed):
    sd = synth.esrgan_state_dict(cfg, seed)
    tensors, conv2d = synth.esrgan_gguf_tensors(sd)
    return O.Model(tensors, conv2d, "whcn")


def test_tile_merge_reference_vectors():
    """Literal vectors of tests/test-image.cpp:303-341 (tile_merge): four constant 5x5 tiles into 8x8."""
    layout = O.tile_layout(8, 8, 6, 2, 1)
    assert (layout.n_x, layout.n_y, layout.tile_w, layout.tile_h) == (2, 2, 5, 5)
    dst = np.zeros((8, 8, 3), np.float32)
    for t, (cx, cy) in enumerate([(0, 0), (1, 0), (0, 1), (1, 1)]):
        O.tile_merge(np.full((5, 5, 3), float(t), np.float32), dst, cx, cy, layout)
    e00, e10 = (4 * 0 + 2 * 1 + 2 * 2 + 1 * 3) / 9, (2 * 0 + 4 * 1 + 1 * 2 + 2 * 3) / 9
    e01, e11 = (2 * 0 + 1 * 1 + 4 * 2 + 2 * 3) / 9, (1 * 0 + 2 * 1 + 2 * 2 + 4 * 3) / 9
    a, b = 1 / 3, 2 / 3
    exp = np.array([
        [0, 0, 0, a, b, 1, 1, 1], [0, 0, 0, a, b, 1, 1, 1], [0, 0, 0, a, b, 1, 1, 1],
        [b, b, b, e00, e10, 5 / 3, 5 / 3, 5 / 3], [4 / 3, 4 / 3, 4 / 3, e01, e11, 7 / 3, 7 / 3, 7 / 3],
        [2, 2, 2, 7 / 3, 8 / 3, 3, 3, 3], [2, 2, 2, 7 / 3, 8 / 3, 3, 3, 3], [2, 2, 2, 7 / 3, 8 / 3, 3, 3, 3]], np.float32)
    np.testing.assert_allclose(dst, np.repeat(exp[..., None], 3, -1), atol=1e-5)   # CHECK_IMAGES_EQUAL tolerance


def test_tile_merge_blending_reference():
    """tests/test-image.cpp:343-360 (tile_merge_blending): constant-one tiles blend to exactly one."""
    layout = O.tile_layout(22, 19, 10, 3, 2)
    dst = np.zeros((19, 22, 3), np.float32)
    tile = np.ones((layout.tile_h, layout.tile_w, 3), np.float32)
    for ty in range(layout.n_y):
        for tx in range(layout.n_x):
            O.tile_merge(tile, dst, tx, ty, layout)
    assert (dst == 1.0).all()


@pytest.mark.parametrize("w,h,max_tile,overlap,align", [(256, 256, 224, 16, 16), (1000, 700, 224, 16, 16), (224, 224, 224, 16, 16),
                                                         (100, 50, 224, 16, 16), (640, 480, 128, 8, 16)])
def test_tile_layout_covers_image(w, h, max_tile, overlap, align):
    t = O.tile_layout(w, h, max_tile, overlap, align)
    assert t.tile_w <= max(max_tile, align) + align and t.tile_w % align == 0 and t.tile_h % align == 0
    assert t.n_x * (t.tile_w - t.overlap_x) + t.overlap_x >= w
    assert t.n_y * (t.tile_h - t.overlap_y) + t.overlap_y >= h
    s = O.tile_scale(t, 4)
    assert (s.image_w, s.tile_w, s.overlap_x, s.n_x) == (4 * w, 4 * t.tile_w, 4 * t.overlap_x, t.n_x)
    # a smooth image split into tiles and merged again is reproduced exactly where weights sum to one
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([xx / w, yy / h, (xx + yy) / (w + h)], -1).astype(np.float32)
    dst = np.zeros_like(img)
    for ty in range(t.n_y):
        for tx in range(t.n_x):
            x0, y0 = tx * (t.tile_w - t.overlap_x), ty * (t.tile_h - t.overlap_y)
            ys = np.clip(np.arange(y0, y0 + t.tile_h), 0, h - 1)
            xs = np.clip(np.arange(x0, x0 + t.tile_w), 0, w - 1)
            O.tile_merge(img[np.ix_(ys, xs)], dst, tx, ty, t)
    np.testing.assert_allclose(dst, img, atol=2e-6)


@pytest.mark.parametrize("name,cfg", [("tiny", synth.ESRGAN_TINY), ("x4_64", synth.ESRGAN_X4)])
def test_rrdbnet_matches_reference_torch(name, cfg):
    g = np.load(GOLD / f"esrgan_{name}.npz")
    m = _model(cfg, int(g["weights_seed"]))
    size = int(g["size"])
    img = synth.images(1, size, size, seed=int(g["image_seed"]))[0]
    x = img.astype(np.float32) / np.float32(255.0)
    y = O.esrgan_generate(m, cfg.scale, cfg.num_blocks, x)
    st = g["result_sample"].shape[0]
    stride = y.shape[0] // st
    np.testing.assert_allclose(y[::stride, ::stride], g["result_sample"], atol=2e-4, rtol=1e-4)
    stats = np.array([y.min(), y.max(), y.mean(), y.std()], np.float64)
    np.testing.assert_allclose(stats, g["result_stats"], atol=1e-4)
    # one dense block (reference test_residual_dense_block)
    ry = O.esrgan_rdb(m, "model.1.sub.0.RDB1", g["rdb_x"])
    np.testing.assert_allclose(ry, g["rdb_y"], atol=2e-5, rtol=1e-5)


def test_esrgan_compute_tiles_equal_untiled():
    """vision.cpp:220-253: tiled compute. With an image of one tile the pipeline is generate + u8 conversion."""
    cfg = synth.ESRGAN_TINY
    m = _model(cfg, 7)
    img = synth.images(1, 48, 48, seed=3)[0]
    out = O.esrgan_compute(m, cfg.scale, cfg.num_blocks, img, O.RGB_U8)
    assert out.shape == (96, 96, 4) and (out[..., 3] == 255).all()
    y = O.esrgan_generate(m, cfg.scale, cfg.num_blocks, img.astype(np.float32) / np.float32(255.0))
    ref = (np.clip(y, 0, 1) * np.float32(255.0)).astype(np.uint8)
    assert np.abs(out[..., :3].astype(int) - ref.astype(int)).max() <= 1
    # multi-tile: 300x260 image -> 2x2 tiles of 224 max; interior far from seams equals per-tile conv up to blend
    img2 = synth.images(1, 300, 260, seed=4)[0]
    out2 = O.esrgan_compute(m, cfg.scale, cfg.num_blocks, img2, O.RGB_U8)
    assert out2.shape == (600, 520, 4) or out2.shape == (520, 600, 4)


def test_cwhn_layout_file_gives_the_same_network():
    """A GGUF written with --layout nhwc (kernels OHWI, no conv2d_weights list, tensor_data_layout = cwhn) must load
    to the same f32 tensors as the default whcn file (model_transfer permutes only listed kernels, ml.cpp:449-516)."""
    cfg = synth.ESRGAN_TINY
    sd = synth.esrgan_state_dict(cfg, 7)
    tw, cw = synth.esrgan_gguf_tensors(sd, "whcn")
    tc, cc = synth.esrgan_gguf_tensors(sd, "cwhn")
    assert cc == [] and len(cw) == sum(1 for v in sd.values() if v.ndim == 4)
    x = synth.images(1, 20, 16, seed=1)[0].astype(np.float32) / np.float32(255.0)
    a = O.esrgan_generate(O.Model(tw, cw, "whcn"), cfg.scale, cfg.num_blocks, x)
    b = O.esrgan_generate(O.Model(tc, cc, "cwhn"), cfg.scale, cfg.num_blocks, x)
    assert np.array_equal(a, b)
