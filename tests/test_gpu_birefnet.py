"""-m gpu: BiRefNet (csrc/birefnet.cpp, kernels_birefnet.hip; reference src/visp/arch/birefnet.cpp) against the CPU oracle's
restatement (tests/test_oracle_birefnet.py; the deformable convolution is parity-unpinned there, so what is checked here is that
the device computes what the oracle computes), through the C ABI: the glue kernels on their own, the deformable convolution as
the product composes it (offset/modulator GEMM -> sampling kernel -> weight GEMM), every decoder level of a small configuration,
visp_model_compute on an image that is not at the model extent, and the error paths."""
import dataclasses

import numpy as np
import pytest

from oracle import oracle
from visioncpp_amd import _lib as L
from visioncpp_amd import synth, vision

pytestmark = pytest.mark.gpu

from gpu_util import api, dev, empty, gemm, pad_weight, rel_err, release, sync  # noqa: E402

MEAN = np.array([0.485, 0.456, 0.406], np.float32)
STD = np.array([0.229, 0.224, 0.225], np.float32)


@pytest.fixture(scope="module")
def device():
    assert api().vx_device_count() > 0, "no HIP device visible: the product path has no CPU fallback"
    return vision.Device.init(vision.Backend.gpu)


@pytest.fixture(autouse=True)
def _release_buffers():
    yield
    release()


def _pre(img_u8):  # birefnet_process_input (birefnet.cpp:259-270)
    return ((img_u8.astype(np.float32) / 255.0 - MEAN) / STD).astype(np.float32)


def test_preprocess_half_and_patches():
    imgs = synth.images(2, 96, 64, seed=3)  # W = 96, H = 64
    B, H, W = 2, 64, 96
    out = empty(B * (H // 2) * (W // 2) * 8 * 2)
    L.vx_check(api().vx_bf_preprocess_half(dev(imgs).ptr, out.ptr, B, H, W, None))
    sync()
    got = out.to_numpy(np.float16, (B, H // 2, W // 2, 8)).astype(np.float32)
    want = np.stack([oracle.interpolate_nhwc(_pre(imgs[b])[None], (H // 2, W // 2), "bilinear", True)[0] for b in range(B)])
    assert np.abs(got[..., :3] + got[..., 3:6] - want).max() < 2e-6 * 8, "value + residue reproduces the f32 pixel"
    assert np.all(got[..., 6:] == 0)
    h, w = 8, 12  # grid 8 x 8
    pt = empty(B * h * w * 64 * 3 * 2)
    L.vx_check(api().vx_bf_patches(dev(imgs).ptr, pt.ptr, B, H, W, h, w, None))
    sync()
    gotp = pt.to_numpy(np.float16, (B, h, w, 192)).astype(np.float32)
    for b in range(B):
        assert np.abs(gotp[b] - oracle.image_to_patches(_pre(imgs[b]), w, h)).max() < 2e-3
    assert api().vx_bf_patches(dev(imgs).ptr, pt.ptr, B, H, W, 7, 12, None) == 0 and b"must divide" in api().vx_last_error()


@pytest.mark.parametrize("h,w,oh,ow,Cc", [(8, 12, 16, 24, 64), (16, 24, 2, 3, 40), (5, 7, 5, 7, 8), (9, 4, 64, 64, 16)])
def test_strided_bilinear_resize(h, w, oh, ow, Cc):
    rng = np.random.default_rng(h)
    B, lds, ldd, off = 2, Cc + 16, Cc + 24, 8
    src = rng.standard_normal((B, h, w, lds)).astype(np.float16)
    dst = empty(B * oh * ow * ldd * 2)
    base = dst.ptr + off * 2
    L.vx_check(api().vx_bf_resize_f16(dev(src).ptr + 16 * 2, lds, base, ldd, B, h, w, Cc, oh, ow, None))
    sync()
    got = dst.to_numpy(np.float16, (B, oh, ow, ldd)).astype(np.float32)
    want = oracle.interpolate_nhwc(src[..., 16:16 + Cc].astype(np.float32), (oh, ow), "bilinear", True)
    assert rel_err(got[..., off:off + Cc], want) < 2e-3
    assert np.all(got[..., :off] == 0) and np.all(got[..., off + Cc:] == 0), "only the channel slice is written"


@pytest.mark.parametrize("k", [1, 3, 7])
def test_deformable_conv_as_composed_on_the_device(k):
    """offsets | modulator logits -> vx_bf_deform_cols_f16 -> GEMM with the kernel weights, against vo_deform_conv2d_nhwc with the
    same f16-rounded operands (mask = 2 sigmoid(logits), birefnet.cpp:83-92)."""
    rng = np.random.default_rng(k)
    B, h, w, Cc, Cout, taps = 2, 10, 13, 64, 32, k * k
    x = rng.standard_normal((B, h, w, Cc)).astype(np.float16)
    ldom = -(-3 * taps // 8) * 8
    om = np.zeros((B, h, w, ldom), np.float16)
    om[..., :2 * taps] = (rng.standard_normal((B, h, w, 2 * taps)) * 1.2).astype(np.float16)
    om[..., 2 * taps:3 * taps] = rng.standard_normal((B, h, w, taps)).astype(np.float16)
    om[0, 0, 0, :2] = [-30, 4]
    wgt = (rng.standard_normal((Cout, k, k, Cc)) / np.sqrt(taps * Cc)).astype(np.float16)
    cols = empty(B * h * w * taps * Cc * 2)
    L.vx_check(api().vx_bf_deform_cols_f16(dev(x).ptr, dev(om).ptr, ldom, cols.ptr, B, h, w, Cc, k, None))
    out = empty(B * h * w * Cout * 2)
    gemm(cols, pad_weight(wgt.reshape(Cout, -1).astype(np.float32)), None, B * h * w, L.EPI_F16, lda=taps * Cc, out=out, ldo=Cout, n_valid=Cout)
    got = out.to_numpy(np.float16, (B, h, w, Cout)).astype(np.float32)
    for b in range(B):
        off = om[b, ..., :2 * taps].astype(np.float32)
        mask = 2.0 / (1.0 + np.exp(-om[b, ..., 2 * taps:3 * taps].astype(np.float32)))
        want = oracle.deform_conv2d_nhwc(x[b].astype(np.float32), wgt.astype(np.float32), off, mask, 1, k // 2)
        assert rel_err(got[b], want) < 6e-3, (k, b)


def test_mean_broadcast_mul_sigmoid():
    rng = np.random.default_rng(0)
    B, n, Cc = 3, 5000, 72
    x = rng.standard_normal((B, n, Cc)).astype(np.float16)
    y = empty(B * Cc * 2)
    L.vx_check(api().vx_bf_mean_f16(dev(x).ptr, Cc, y.ptr, empty(B * Cc * 4).ptr, B, n, Cc, None))
    sync()
    assert np.abs(y.to_numpy(np.float16, (B, Cc)).astype(np.float32) - x.astype(np.float32).mean(1)).max() < 2e-3
    # a fixed summation order (no float atomics): the same bits every launch, whatever else the device is doing
    first, xd = y.to_numpy(np.uint16, (B, Cc)).copy(), dev(x)
    for _ in range(5):
        L.vx_check(api().vx_bf_mean_f16(xd.ptr, Cc, y.ptr, empty(B * Cc * 4, zero=False).ptr, B, n, Cc, None))
        sync()
        assert np.array_equal(y.to_numpy(np.uint16, (B, Cc)), first)
    g = rng.standard_normal((B, Cc)).astype(np.float16)
    dst = empty(B * n * (Cc + 8) * 2)
    L.vx_check(api().vx_bf_broadcast_f16(dev(g).ptr, Cc, dst.ptr + 16, Cc + 8, B, n, Cc, None))
    sync()
    d = dst.to_numpy(np.float16, (B, n, Cc + 8))
    assert np.array_equal(d[..., 8:], np.broadcast_to(g[:, None], (B, n, Cc))) and np.all(d[..., :8] == 0)
    a = np.zeros((B * n, 8), np.float16)
    a[:, 0] = rng.standard_normal(B * n).astype(np.float16)
    yy = dev(x.copy())
    L.vx_check(api().vx_bf_mul_sigmoid_f16(yy.ptr, Cc, dev(a).ptr, 8, B * n, Cc, None))
    sync()
    want = x.reshape(-1, Cc).astype(np.float32) / (1.0 + np.exp(-a[:, :1].astype(np.float32)))
    assert rel_err(yy.to_numpy(np.float16, (B * n, Cc)).astype(np.float32), want) < 2e-3
    o = empty(B * n * 4)
    L.vx_check(api().vx_bf_sigmoid_out_f32(dev(a).ptr, 8, o.ptr, B * n, None))
    sync()
    assert np.abs(o.to_numpy(np.float32, (B * n,)) - 1.0 / (1.0 + np.exp(-a[:, 0].astype(np.float32)))).max() < 1e-6


@pytest.fixture(scope="module")
def mini(device, tmp_path_factory):
    cfg = dataclasses.replace(synth.SWIN_MINI, image_size=256)
    path = synth.write_birefnet_gguf(tmp_path_factory.mktemp("bf") / "birefnet_mini.gguf", cfg, seed=6)
    tensors, conv_idx = synth.birefnet_gguf_tensors(synth.birefnet_state_dict(cfg, 6))
    return (vision.Model.load(path, device), oracle.Model(tensors, conv_idx),
            oracle.swin_params(cfg.embed_dim, cfg.window_size, cfg.depths, cfg.n_heads), path)


def test_birefnet_every_level(mini):
    """192 x 128 image (W x H), batch 2: encoder features, squeeze block, the four decoder levels and the mask against the oracle."""
    model, om, P, _ = mini
    W, H = 192, 128
    imgs = synth.images(2, W, H, seed=21)
    model.enable_captures(True)
    masks = model.segment_batch(imgs)
    assert masks.shape == (2, H, W)
    sizes = {"feature_0": 32 * 48 * 64, "feature_1": 16 * 24 * 128, "feature_2": 8 * 12 * 256, "feature_3": 4 * 6 * 960, "squeeze": 4 * 6 * 512,
             "p4": 4 * 6 * 256, "p3": 8 * 12 * 128, "p2": 16 * 24 * 64, "p1": 32 * 48 * 32}
    tol = {"feature_0": 1e-2, "feature_1": 1e-2, "feature_2": 1e-2, "feature_3": 1.5e-2, "squeeze": 2e-2, "p4": 3e-2, "p3": 3e-2, "p2": 3e-2, "p1": 3e-2}
    for b in range(2):
        want, caps = oracle.birefnet_predict(om, P, _pre(imgs[b]), captures=sizes)
        for name in sizes:
            got = model.read_capture(name)[b].reshape(-1)
            assert rel_err(got, caps[name]) < tol[name], (b, name, rel_err(got, caps[name]))
            assert np.abs(got - caps[name]).mean() < 4e-3 * np.abs(caps[name]).max(), (b, name)
        assert np.abs(masks[b] - want).max() < 2e-2 and np.abs(masks[b] - want).mean() < 2e-3, (b, np.abs(masks[b] - want).max())
    model.enable_captures(False)
    again = model.segment_batch(imgs[::-1].copy())
    np.testing.assert_array_equal(again[::-1], masks)  # images are independent units; repeated launches are bit-identical


def test_birefnet_lite_configuration(device, tmp_path):
    """BiRefNet-lite's shape (swin_t backbone: features 192 / 384 / 768 / 2880 channels, decoder table of
    tests/test_birefnet.py:1130-1175) on one 256 x 256 image."""
    cfg = dataclasses.replace(synth.SWIN_T, image_size=256)
    sd = synth.birefnet_state_dict(cfg, 9)
    model = vision.Model.load(synth.write_birefnet_gguf(tmp_path / "lite.gguf", cfg, sd=sd), device)
    tensors, conv_idx = synth.birefnet_gguf_tensors(sd)
    om = oracle.Model(tensors, conv_idx)
    P = oracle.swin_params(cfg.embed_dim, cfg.window_size, cfg.depths, cfg.n_heads)
    img = synth.images(1, 256, 256, seed=13)
    model.enable_captures(True)
    mask = model.segment_batch(img)[0]
    want, caps = oracle.birefnet_predict(om, P, _pre(img[0]), captures={"feature_3": 8 * 8 * 2880, "squeeze": 8 * 8 * 1536, "p1": 64 * 64 * 96})
    for name, tol in (("feature_3", 2e-2), ("squeeze", 3e-2), ("p1", 4e-2)):
        got = model.read_capture(name)[0].reshape(-1)
        assert rel_err(got, caps[name]) < tol, (name, rel_err(got, caps[name]))
    assert np.abs(mask - want).max() < 3e-2 and np.abs(mask - want).mean() < 3e-3, np.abs(mask - want).max()


def test_birefnet_lite_1024_batch_8(device, tmp_path):
    """BASELINE.json configs[3] at its per-GPU share: BiRefNet-lite (swin_t backbone), 1024 x 1024, batch 8 (64 images over 8 GPUs).
    Size-independent properties over the whole batch -- finite, a sigmoid's range, images independent of their batch position (bit
    for bit: a duplicated image gives a duplicated mask, a single-image launch gives the same mask) -- and one image against the
    oracle's birefnet_predict at full size (edge windows, window padding 256 -> 259 on the 1/4 map, all four decoder levels)."""
    cfg = dataclasses.replace(synth.SWIN_T, image_size=1024)
    sd = synth.birefnet_state_dict(cfg, 9)
    model = vision.Model.load(synth.write_birefnet_gguf(tmp_path / "lite1024.gguf", cfg, sd=sd), device)
    imgs = synth.images(8, 1024, 1024, seed=64)
    imgs[5] = imgs[1]
    masks = model.segment_batch(imgs)
    assert masks.shape == (8, 1024, 1024) and np.isfinite(masks).all() and masks.min() >= 0.0 and masks.max() <= 1.0
    assert masks.std() > 1e-4  # (not a constant image)
    np.testing.assert_array_equal(masks[5], masks[1])
    np.testing.assert_array_equal(model.segment_batch(imgs[2:3])[0], masks[2])
    np.testing.assert_array_equal(model.segment_batch(imgs[::-1].copy())[::-1], masks)
    tensors, conv_idx = synth.birefnet_gguf_tensors(sd)
    om = oracle.Model(tensors, conv_idx)
    P = oracle.swin_params(cfg.embed_dim, cfg.window_size, cfg.depths, cfg.n_heads)
    want = oracle.birefnet_predict(om, P, _pre(imgs[2]))
    d = np.abs(masks[2] - want)
    assert d.max() < 3e-2 and d.mean() < 3e-3, (d.max(), d.mean())


def test_reference_c_api_compute(mini):
    """visp_model_compute for family 1 = birefnet_compute (vision.cpp:108-132): a 300 x 200 bgra image is scaled to the model
    extent (256 x 256) with image_scale, segmented, scaled back and returned as alpha_u8."""
    model, om, P, _ = mini
    rgb = synth.images(1, 300, 200, seed=8)[0]
    bgra = np.concatenate([rgb[..., ::-1], np.full(rgb.shape[:2] + (1,), 255, np.uint8)], -1)
    got = model.compute(bgra, vision.ImageFormat.bgra_u8)
    assert got.shape == (200, 300) and got.dtype == np.uint8
    assert model.birefnet_image_extent(300, 200) == (256, 256)
    scaled = oracle.image_scale(rgb, oracle.RGB_U8, 256, 256)
    mask = oracle.birefnet_predict(om, P, _pre(scaled))
    want = oracle.image_f32_to_u8(oracle.image_scale(mask, oracle.ALPHA_F32, 300, 200), oracle.ALPHA_F32, oracle.ALPHA_U8)
    d = np.abs(got.astype(np.int32) - want.reshape(200, 300).astype(np.int32))
    assert d.max() <= 6 and d.mean() < 0.6, (d.max(), d.mean())


def test_birefnet_errors(mini, device, tmp_path):
    model, _, _, path = mini
    with pytest.raises(L.Error, match="multiple of 64"):
        model.segment_batch(np.zeros((1, 96, 128, 3), np.uint8))
    with pytest.raises(L.Error, match="8-bit colour"):
        model.compute(np.zeros((64, 64), np.uint8), vision.ImageFormat.alpha_u8)
    enc = vision.SwinEncoder.load(path, device)  # the encoder-only handle of the same file is not a birefnet model
    with pytest.raises(L.Error, match="swin encoder"):
        L.check(api().visp_birefnet_compute_batch_host(enc._handle, None, 1, 64, 64, None))
    cfg = dataclasses.replace(synth.SWIN_MINI, image_size=256)
    sd = dict(synth.birefnet_state_dict(cfg, 6))
    del sd["decoder.gdt_convs_attn_3.0.weight"]
    with pytest.raises(L.Error, match="not found"):
        vision.Model.load(synth.write_birefnet_gguf(tmp_path / "bad.gguf", cfg, sd=sd), device)
    # swin encoder entry points accept the full model
    outs = vision.SwinEncoder(model._api, model._handle, vision.Arch.birefnet, model._device)
    try:
        assert outs.output_dims(128, 128)[0] == (32, 32, 32)
    finally:
        outs._handle = None  # borrowed handle: the fixture's Model owns it
