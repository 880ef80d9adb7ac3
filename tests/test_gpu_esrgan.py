"""GPU parity of the ESRGAN row: dense-block conv kernel, tile pre/post-processing and (below) the whole
RRDBNet through the C ABI, each against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _release():
    yield
    from tests import gpu_util as G
    G.release()


def _conv_ref(x, w, b, act=0):
    """x [B,H,W,Cin] f32, w torch [Cout,Cin,3,3] (f16-rounded), oracle conv + optional LeakyReLU 0.2."""
    B, H, W, Cin = x.shape
    wf = np.ascontiguousarray(w.astype(np.float16).astype(np.float32).transpose(0, 2, 3, 1))
    y = O.conv2d_nhwc(x, wf, b, stride=1, pad=1)
    if act:
        y = np.where(y > 0, y, y * np.float32(0.2))
    return y


@pytest.mark.parametrize("cin,cout,H,W,B", [(64, 32, 16, 32, 1), (96, 32, 24, 40, 2), (128, 32, 33, 47, 1), (160, 32, 16, 16, 3),
                                            (192, 64, 48, 48, 2), (64, 64, 37, 70, 1), (32, 64, 20, 36, 1), (64, 32, 144, 144, 1)])
def test_dconv_channel_prefix(cin, cout, H, W, B):
    from tests import gpu_util as G
    rng = np.random.default_rng(cin * 7 + cout)
    C = 192
    buf = (rng.standard_normal((B, H, W, C)) * 0.5).astype(np.float16)
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    ref = _conv_ref(buf[..., :cin].astype(np.float32), w, b, act=1)
    xd = G.dev(G.to_planes(buf))
    # output into plane(s) of a second six-plane buffer; the other planes must stay untouched
    canary = np.full((B, H, W, C), 7.0, np.float16)
    od = G.dev(G.to_planes(canary))
    p0 = 2 if cout == 32 else 0
    got = G.dconv(xd, 6, cin, B, H, W, w, b, act=1, out=od, out_planes=6, out_plane0=p0)
    off = p0 * 32
    np.testing.assert_allclose(got[..., off:off + cout].astype(np.float32), ref, atol=4e-3, rtol=4e-3)
    mask = np.ones(C, bool)
    mask[off:off + cout] = False
    assert (got[..., mask] == np.float16(7.0)).all()


def test_dconv_scaled_residuals():
    """conv5 of a dense block at the end of an RRDB: (conv*0.2 + x)*0.2 + rrdb_in (esrgan.cpp:38-40, 49-50)."""
    from tests import gpu_util as G
    rng = np.random.default_rng(3)
    B, H, W = 2, 40, 52
    buf = (rng.standard_normal((B, H, W, 192)) * 0.5).astype(np.float16)
    r2 = (rng.standard_normal((B, H, W, 64)) * 0.5).astype(np.float16)
    w = (rng.standard_normal((64, 192, 3, 3)) / np.sqrt(192 * 9)).astype(np.float32)
    b = (rng.standard_normal(64) * 0.1).astype(np.float32)
    y = _conv_ref(buf.astype(np.float32), w, b)
    x64 = buf[..., :64].astype(np.float32)
    xd, x64d, r2d = G.dev(G.to_planes(buf)), G.dev(G.to_planes(buf[..., :64])), G.dev(G.to_planes(r2))
    got1 = G.dconv(xd, 6, 192, B, H, W, w, b, res1=x64d, s1=0.2)
    np.testing.assert_allclose(got1.astype(np.float32), y * 0.2 + x64, atol=3e-3, rtol=3e-3)
    got2 = G.dconv(xd, 6, 192, B, H, W, w, b, res1=x64d, s1=0.2, res2=r2d, s2=0.2)
    ref2 = (y * 0.2 + x64) * 0.2 + r2.astype(np.float32)
    np.testing.assert_allclose(got2.astype(np.float32), ref2, atol=3e-3, rtol=3e-3)
    # the same through the identity fold (x taken from the halo in LDS instead of a second read)
    got3 = G.dconv(xd, 6, 192, B, H, W, w, b, x_residual=True, s1=0.2)
    np.testing.assert_allclose(got3.astype(np.float32), y * 0.2 + x64, atol=3e-3, rtol=3e-3)
    got4 = G.dconv(xd, 6, 192, B, H, W, w, b, x_residual=True, s1=0.2, res2=r2d, s2=0.2)
    np.testing.assert_allclose(got4.astype(np.float32), ref2, atol=3e-3, rtol=3e-3)


def test_dconv_upsample_and_rgb_head():
    from tests import gpu_util as G
    rng = np.random.default_rng(4)
    B, h, w_ = 2, 18, 26
    x = (rng.standard_normal((B, h, w_, 64)) * 0.5).astype(np.float16)
    w = (rng.standard_normal((64, 64, 3, 3)) / 24).astype(np.float32)
    b = (rng.standard_normal(64) * 0.1).astype(np.float32)
    up = np.repeat(np.repeat(x.astype(np.float32), 2, axis=1), 2, axis=2)   # nearest x2 (esrgan.cpp:13-16)
    ref = _conv_ref(up, w, b, act=1)
    got = G.dconv(G.dev(G.to_planes(x)), 2, 64, B, 2 * h, 2 * w_, w, b, up2=True, act=1)
    np.testing.assert_allclose(got.astype(np.float32), ref, atol=4e-3, rtol=4e-3)
    w3 = (rng.standard_normal((3, 64, 3, 3)) / 24).astype(np.float32)
    b3 = np.array([0.4, 0.5, 0.6], np.float32)
    ref3 = _conv_ref(x.astype(np.float32), w3, b3)
    got3 = G.dconv(G.dev(G.to_planes(x)), 2, 64, B, h, w_, w3, b3, rgb=True)
    np.testing.assert_allclose(got3, ref3, atol=2e-3, rtol=2e-3)


@pytest.mark.parametrize("cin,cout,H,W", [(64, 64, 74, 74), (64, 32, 40, 100), (32, 32, 33, 70)])
def test_dconv_nhwc_relu_forms(cin, cout, H, W):
    """The DPT uses of the kernel (depth-anything.cpp:15-23, 81-94): NHWC maps through pixel / plane strides,
    conv(relu(x)) + ReLU (residual unit conv 1), conv(x) + two residuals (conv 2), and the fused depth head."""
    from tests import gpu_util as G
    rng = np.random.default_rng(cin + cout + H)
    B = 2
    x = (rng.standard_normal((B, H, W, cin)) * 0.5).astype(np.float16)
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    xd = G.dev(x)
    got = G.dconv(xd, 0, cin, B, H, W, w, b, act=2, a_relu=True, nhwc=(cin, cout), out=G.empty(B * H * W * cout * 2))
    ref = np.maximum(_conv_ref(np.maximum(x.astype(np.float32), 0), w, b), 0)
    np.testing.assert_allclose(got.astype(np.float32), ref, atol=4e-3, rtol=4e-3)
    r1 = (rng.standard_normal((B, H, W, cout)) * 0.5).astype(np.float16)
    r2 = (rng.standard_normal((B, H, W, cout)) * 0.5).astype(np.float16)
    got = G.dconv(xd, 0, cin, B, H, W, w, b, nhwc=(cin, cout), res1=G.dev(r1), res2=G.dev(r2), out=G.empty(B * H * W * cout * 2))
    ref = _conv_ref(x.astype(np.float32), w, b) + r1.astype(np.float32) + r2.astype(np.float32)
    np.testing.assert_allclose(got.astype(np.float32), ref, atol=6e-3, rtol=4e-3)
    if cout == 32:
        w3, b3 = np.abs(rng.standard_normal(32) * 0.3).astype(np.float32), 0.05
        got = G.dconv(xd, 0, cin, B, H, W, w, b, nhwc=(cin, cout), head=(w3, b3, 1.5))
        h2 = np.maximum(_conv_ref(x.astype(np.float32), w, b), 0)
        want = np.maximum(h2 @ w3 + b3, 0) * 1.5
        np.testing.assert_allclose(got, want, atol=3e-3, rtol=3e-3)


@pytest.mark.parametrize("cin,hs,ws,H,W,B", [(64, 148, 148, 296, 296, 1), (32, 296, 296, 518, 518, 1), (32, 40, 60, 70, 105, 2), (64, 19, 37, 37, 74, 3),
                                             (32, 100, 28, 175, 50, 2)])
def test_dconv_bilinear_input(cin, hs, ws, H, W, B):
    """Round 3: ggml_interpolate(BILINEAR | ALIGN_CORNERS) fused into the consumer conv (depth-anything.cpp:36-38 -> head.conv1,
    :84-85 -> head.conv2; ml.cpp:782-788): the conv of the resized map, resized in the halo loader, against
    conv(oracle.interpolate(x)) -- both head forms (f16 map out, fused depth head) and both tile shapes (widths with a
    remainder <= 16 take the 32x16 strip), several batch images so that persistent blocks cross image boundaries."""
    from tests import gpu_util as G
    rng = np.random.default_rng(cin + hs + W)
    assert G.api().vx_dconv_bilinear_supported(32, H, W, hs, ws) == 1
    x = (rng.standard_normal((B, hs, ws, cin)) * 0.5).astype(np.float16)
    w = (rng.standard_normal((32, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    b = (rng.standard_normal(32) * 0.1).astype(np.float32)
    up = O.interpolate_nhwc(x.astype(np.float32), (H, W), "bilinear", True)
    ref = _conv_ref(up, w, b)
    xd = G.dev(x)
    got = G.dconv(xd, 0, cin, B, H, W, w, b, nhwc=(cin, 32), bil=(hs, ws), out=G.empty(B * H * W * 32 * 2))
    # the interpolation runs on packed f16 (4 products, 3 roundings per value) where the stand-alone kernel rounds once
    np.testing.assert_allclose(got.astype(np.float32), ref, atol=5e-3, rtol=4e-3)
    assert np.abs(got.astype(np.float32) - ref).mean() < 4e-4
    w3, b3 = np.abs(rng.standard_normal(32) * 0.3).astype(np.float32), 0.05
    got = G.dconv(xd, 0, cin, B, H, W, w, b, nhwc=(cin, 32), bil=(hs, ws), head=(w3, b3, 1.5))
    want = np.maximum(np.maximum(ref, 0) @ w3 + b3, 0) * 1.5
    np.testing.assert_allclose(got, want, atol=4e-3, rtol=3e-3)


def test_dconv_bilinear_limits():
    from tests import gpu_util as G
    ok = G.api().vx_dconv_bilinear_supported
    assert ok(32, 518, 518, 296, 296) == 1 and ok(32, 296, 296, 148, 148) == 1 and ok(32, 518, 700, 296, 400) == 1
    assert ok(64, 296, 296, 148, 148) == 0      # Cout = 64 keeps the DMA halo
    assert ok(32, 100, 100, 80, 80) == 0        # scale 0.8: the 18 x 34 halo does not fit a 12 x 21 source patch
    assert ok(32, 1500, 1500, 750, 750) == 0    # H + W beyond the row / column table
    x = G.dev(np.zeros((1, 80, 80, 32), np.float16))
    w = np.zeros((32, 32, 3, 3), np.float32)
    with pytest.raises(Exception, match="interpolating loader"):
        G.dconv(x, 0, 32, 1, 100, 100, w, None, nhwc=(32, 32), bil=(80, 80), out=G.empty(100 * 100 * 32 * 2))


@pytest.mark.parametrize("w,h,fmt", [(256, 256, O.RGB_U8), (300, 260, O.RGBA_U8), (100, 50, O.BGRA_U8), (64, 64, O.ARGB_U8)])
def test_tiles_in_out(w, h, fmt):
    """vx_esrgan_tiles_in == image_u8_to_f32 per tile (clamped reads); vx_esrgan_tiles_out == tile_merge of all
    tiles in order + f32->u8, bit for bit against the oracle."""
    from tests import gpu_util as G
    from visioncpp_amd import _lib as L
    import ctypes as C
    rng = np.random.default_rng(w + h)
    B, ch = 2, O._CH[fmt]
    img = rng.integers(0, 256, (B, h, w, ch), dtype=np.uint8)
    t = O.tile_layout(w, h, 224, 16, 16)
    lt = L.TileLayout(*[getattr(t, n) for n, _ in L.TileLayout._fields_])
    nt = t.n_x * t.n_y
    out = G.empty(B * nt * t.tile_h * t.tile_w * 32 * 2)
    L.vx_check(G.api().vx_esrgan_tiles_in(G.dev(img).ptr, B, w, h, fmt, C.byref(lt), out.ptr, None))
    G.sync()
    got = out.to_numpy(np.float16, (B, nt, t.tile_h, t.tile_w, 32)).astype(np.float32)
    for b in range(B):
        for ti in range(nt):
            cx, cy = ti % t.n_x, ti // t.n_x
            off = (cx * (t.tile_w - t.overlap_x), cy * (t.tile_h - t.overlap_y))
            ref = O.image_u8_to_f32(img[b], fmt, O.RGB_F32, dst_extent=(t.tile_w, t.tile_h), tile_offset=off)
            np.testing.assert_allclose(got[b, ti, ..., 0:3] + got[b, ti, ..., 3:6], ref, atol=2e-7)
            assert (got[b, ti, ..., 6:] == 0).all()
    # merge: random f32 tiles at scale 2
    ts = O.tile_scale(t, 2)
    lts = L.TileLayout(*[getattr(ts, n) for n, _ in L.TileLayout._fields_])
    tiles = rng.random((B, nt, ts.tile_h, ts.tile_w, 3), dtype=np.float32) * 1.2 - 0.1
    of, ou = G.empty(B * ts.image_h * ts.image_w * 12), G.empty(B * ts.image_h * ts.image_w * 4)
    L.vx_check(G.api().vx_esrgan_tiles_out(G.dev(tiles).ptr, B, C.byref(lts), of.ptr, ou.ptr, None))
    G.sync()
    gf = of.to_numpy(np.float32, (B, ts.image_h, ts.image_w, 3))
    gu = ou.to_numpy(np.uint8, (B, ts.image_h, ts.image_w, 4))
    for b in range(B):
        dst = np.zeros((ts.image_h, ts.image_w, 3), np.float32)
        for ti in range(nt):
            O.tile_merge(tiles[b, ti], dst, ti % t.n_x, ti // t.n_x, ts)
        assert np.array_equal(gf[b], dst)
        ref_u8 = O.image_f32_to_u8(dst, O.RGB_F32, O.RGBA_U8)
        assert np.array_equal(gu[b], ref_u8)


# ---- whole network through the C ABI ---------------------------------------------------------------------------

from pathlib import Path  # noqa: E402

GOLD = Path(__file__).parent / "golden"


@pytest.fixture(scope="module")
def device():
    from visioncpp_amd.vision import Backend, Device
    return Device.init(Backend.gpu)


def _load(tmp_path_factory, device, cfg, seed):
    from visioncpp_amd import synth
    from visioncpp_amd.vision import Arch, Model
    path = tmp_path_factory.mktemp("esrgan") / f"esrgan_{cfg.name}.gguf"
    synth.write_esrgan_gguf(path, cfg, seed)
    m = Model.load(path, device)
    assert m.arch is Arch.esrgan
    sd = synth.esrgan_state_dict(cfg, seed)
    tensors, conv2d = synth.esrgan_gguf_tensors(sd)
    return m, O.Model(tensors, conv2d, "whcn")


@pytest.fixture(scope="module")
def tiny(tmp_path_factory, device):
    from visioncpp_amd import synth
    return _load(tmp_path_factory, device, synth.ESRGAN_TINY, 7) + (synth.ESRGAN_TINY,)


@pytest.fixture(scope="module")
def x4(tmp_path_factory, device):
    from visioncpp_amd import synth
    return _load(tmp_path_factory, device, synth.ESRGAN_X4, 1) + (synth.ESRGAN_X4,)


def test_generate_tiny_vs_oracle(tiny):
    from visioncpp_amd import synth
    m, om, cfg = tiny
    info = m.esrgan_info
    assert (info.scale, info.n_blocks, info.n_filters, info.growth) == (2, 2, 64, 32)
    imgs = synth.images(3, 40, 56, seed=11).astype(np.float32) / np.float32(255.0)   # [3, 56, 40, 3]
    got = m.esrgan_generate(imgs)
    for i in range(3):
        ref = O.esrgan_generate(om, cfg.scale, cfg.num_blocks, imgs[i])
        assert np.abs(got[i] - ref).mean() < 1e-3 and np.abs(got[i] - ref).max() < 8e-3


def test_generate_matches_reference_torch_fixture(tiny, x4):
    """The fixtures hold outputs of the reference's own torch RRDBNet (tests/golden/make_golden_esrgan.py)."""
    from visioncpp_amd import synth
    for (m, om, cfg), name in ((tiny, "tiny"), (x4, "x4_64")):
        g = np.load(GOLD / f"esrgan_{name}.npz")
        size = int(g["size"])
        img = synth.images(1, size, size, seed=int(g["image_seed"])).astype(np.float32) / np.float32(255.0)
        y = m.esrgan_generate(img)[0]
        stride = y.shape[0] // g["result_sample"].shape[0]
        err = np.abs(y[::stride, ::stride] - g["result_sample"])
        assert err.mean() < 1e-3 and err.max() < 1e-2, (name, err.mean(), err.max())   # fp16 tolerance: pixel MAE < 1e-3


def test_upscale_batch_vs_oracle_compute(tiny):
    """esrgan_compute (vision.cpp:220-253): tiling 224/16, merge, rgba_u8 -- batched GPU path vs the oracle per image."""
    from visioncpp_amd import synth
    from visioncpp_amd.vision import ImageFormat
    m, om, cfg = tiny
    for (w, h, fmt, ofmt) in ((300, 260, ImageFormat.rgb_u8, O.RGB_U8), (96, 80, ImageFormat.bgra_u8, O.BGRA_U8),
                              (640, 481, ImageFormat.rgb_u8, O.RGB_U8)):   # 3 x 3 tiles of 224 x 176: two lanes, several groups
        ch = 3 if fmt is ImageFormat.rgb_u8 else 4
        rng = np.random.default_rng(w)
        base = synth.images(2, w, h, seed=w)
        imgs = base if ch == 3 else np.concatenate([base, rng.integers(0, 256, (2, h, w, 1), dtype=np.uint8)], -1)
        got = m.upscale_batch(imgs, fmt)
        assert got.shape == (2, h * cfg.scale, w * cfg.scale, 4) and (got[..., 3] == 255).all()
        for i in range(2):
            ref = O.esrgan_compute(om, cfg.scale, cfg.num_blocks, imgs[i], ofmt)
            d = np.abs(got[i].astype(np.int32) - ref.astype(np.int32))
            assert d.max() <= 2 and (d > 0).mean() < 0.05, (w, h, d.max(), (d > 0).mean())


def test_upscale_x4_batch_properties(x4):
    """Full 23-block Real-ESRGAN-x4 shape at the BASELINE extent (256^2 -> 1024^2, 2x2 tiles of 144): one image
    against the oracle; a batch is independent of its neighbours and of the tile grouping."""
    from visioncpp_amd import synth
    m, om, cfg = x4
    imgs = synth.images(3, 256, 256, seed=5)
    out = m.upscale_batch(imgs)
    assert out.shape == (3, 1024, 1024, 4)
    ref = O.esrgan_compute(om, cfg.scale, cfg.num_blocks, imgs[1], O.RGB_U8)
    d = np.abs(out[1].astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= 3 and d.mean() < 0.25, (d.max(), d.mean())
    single = m.upscale_batch(imgs[1:2])
    assert np.array_equal(single[0], out[1])
    m.set_tile_group(5)
    regroup = m.upscale_batch(imgs)
    m.set_tile_group(64)
    assert np.array_equal(regroup, out)


def test_upscale_x4_baseline_batch_16(x4):
    """BASELINE.json configs[2] at its full size: Real-ESRGAN-4x, 256x256 -> 1024x1024, batch 16 (64 tiles of 144 through the
    23-block net in one call). One image against the oracle, size-independent properties on the rest: every image equals
    its single-image run, duplicated inputs give identical outputs, alpha is opaque."""
    from visioncpp_amd import synth
    m, om, cfg = x4
    imgs = synth.images(16, 256, 256, seed=16)
    imgs[9] = imgs[2]
    out = m.upscale_batch(imgs)
    assert out.shape == (16, 1024, 1024, 4) and (out[..., 3] == 255).all()
    assert np.array_equal(out[9], out[2])
    ref = O.esrgan_compute(om, cfg.scale, cfg.num_blocks, imgs[15], O.RGB_U8)
    d = np.abs(out[15].astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= 3 and d.mean() < 0.25, (d.max(), d.mean())
    for i in (0, 7):
        assert np.array_equal(m.upscale_batch(imgs[i:i + 1])[0], out[i])


def test_c_api_compute_esrgan(tiny):
    """visp_model_compute family 4 (c-api.cpp:103-106): one image view in, rgba_u8 image at extent*scale out."""
    from visioncpp_amd import synth
    from visioncpp_amd.vision import ImageFormat
    m, om, cfg = tiny
    img = synth.images(1, 70, 50, seed=9)[0]
    res = m.compute(img, ImageFormat.rgb_u8)
    assert res.shape == (100, 140, 4)
    assert np.array_equal(res, m.upscale_batch(img[None])[0])


def test_esrgan_errors(tiny, device, tmp_path):
    from visioncpp_amd import synth
    from visioncpp_amd import _lib as L
    from visioncpp_amd.vision import Arch, ImageFormat, Model
    import ctypes as C
    m, _, _ = tiny
    with pytest.raises(L.Error, match="8-bit colour"):
        m.upscale_batch(np.zeros((1, 8, 8, 1), np.uint8), ImageFormat.alpha_u8)
    info = L.DepthAnyInfo()
    assert L.get_lib().visp_depthany_get_info(m._handle, C.byref(info)) == 0   # wrong family handle is refused
    assert b"not a depth_anything model" in L.get_lib().visp_get_last_error()
    bad = tmp_path / "nf48.gguf"
    synth.write_esrgan_gguf(bad, synth.EsrganConfig(num_filters=48, num_blocks=1, scale=2, gc=16, name="bad"), 0)
    with pytest.raises(L.Error, match="not built in this backend"):
        Model.load(bad, device, Arch.esrgan)


@pytest.mark.parametrize("scale,w,h", [(1, 50, 34), (8, 40, 24), (2, 225, 17), (4, 17, 230)])
def test_scales_and_odd_extents(tmp_path_factory, device, scale, w, h):
    """log2(scale) up-sampling stages (esrgan.cpp:69-73: none for x1, three for x8) and extents that are not multiples of
    anything: tile sizes come out as 64x48, 48x32, 128x32, 32x128 (strip tiles, partial tiles, 1- and 2-tile layouts)."""
    from visioncpp_amd import synth
    cfg = synth.EsrganConfig(num_blocks=1, scale=scale, name=f"s{scale}")
    m, om = _load(tmp_path_factory, device, cfg, 11)
    img = synth.images(2, w, h, seed=w + h)
    got = m.upscale_batch(img)
    assert got.shape == (2, h * scale, w * scale, 4)
    for i in range(2):
        ref = O.esrgan_compute(om, scale, 1, img[i], O.RGB_U8)
        d = np.abs(got[i].astype(np.int32) - ref.astype(np.int32))
        assert d.max() <= 2 and (d > 0).mean() < 0.05, (scale, w, h, d.max(), (d > 0).mean())


def test_c_api_strided_view(tiny):
    """visp_model_compute with a row stride larger than width * channels (image_view.stride, image.h:37-41)."""
    import ctypes as C
    from visioncpp_amd import _lib as L
    m, om, cfg = tiny
    rng = np.random.default_rng(2)
    w, h, stride = 45, 30, 45 * 3 + 13
    buf = rng.integers(0, 256, (h, stride), dtype=np.uint8)
    img = np.ascontiguousarray(np.stack([buf[y, :w * 3].reshape(w, 3) for y in range(h)]))
    view = L.ImageView(w, h, stride, 3, buf.ctypes.data)   # rgb_u8
    out_view, out_data = L.ImageView(), C.c_void_p()
    L.check(L.get_lib().visp_model_compute(m._handle, 4, (L.ImageView * 1)(view), 1, (C.c_int32 * 1)(), 0, C.byref(out_view), C.byref(out_data)))
    try:
        n = out_view.height * out_view.stride
        res = np.frombuffer((C.c_uint8 * n).from_address(out_view.data), np.uint8).reshape(out_view.height, out_view.width, 4).copy()
    finally:
        L.get_lib().visp_image_destroy(out_data)
    assert (out_view.width, out_view.height, out_view.format) == (w * cfg.scale, h * cfg.scale, 0)   # rgba_u8
    assert np.array_equal(res, m.upscale_batch(img[None])[0])


def test_cwhn_layout_files_load_identically(tmp_path_factory, device):
    """--layout nhwc GGUFs (tensor_data_layout = cwhn, kernels OHWI, no conv2d_weights list) of both families pack to
    the same weight arena as the default files: bit-identical outputs."""
    from visioncpp_amd import synth
    from visioncpp_amd.vision import Model
    d = tmp_path_factory.mktemp("layouts")
    cfg = synth.ESRGAN_TINY
    img = synth.images(1, 60, 44, seed=2)
    outs = [Model.load(synth.write_esrgan_gguf(d / f"e_{lay}.gguf", cfg, 7, layout=lay), device).upscale_batch(img) for lay in ("whcn", "cwhn")]
    assert np.array_equal(outs[0], outs[1])
    imgs = synth.images(1, 112, 112, seed=3)
    outs = [Model.load(synth.write_gguf(d / f"d_{lay}.gguf", synth.MINI, 4, layout=lay) and d / f"d_{lay}.gguf", device).compute_batch(imgs)
            for lay in ("whcn", "cwhn")]
    assert np.array_equal(outs[0], outs[1])


def test_large_image_many_tiles(tmp_path_factory, device):
    """1300 x 1000 at x4: 6 x 5 tiles of 240 x 224 (30 tiles, 860k up-sampled pixels each). The executor must cap the
    tile group so that a group's planes stay addressable (32-bit buffer descriptors), and the result must not depend on
    the grouping."""
    from visioncpp_amd import synth
    cfg = synth.EsrganConfig(num_blocks=1, scale=4, name="big")
    m, _ = _load(tmp_path_factory, device, cfg, 3)
    img = synth.images(1, 1300, 1000, seed=8)
    a = m.upscale_batch(img)
    assert a.shape == (1, 4000, 5200, 4) and (a[..., 3] == 255).all() and a[..., :3].std() > 1
    m.set_tile_group(4)
    b = m.upscale_batch(img)
    assert np.array_equal(a, b)


def test_weight_arena_replication(device, tmp_path):
    """The N > 1 load path without the collective: a header-only model (what ranks != 0 load) lays out the same arena -- the graph lowering's operand images
    in lowering order --, receives the bytes of a fully loaded model's arena and must then produce bit-identical images."""
    from visioncpp_amd import _lib as L
    from visioncpp_amd import synth
    from visioncpp_amd.vision import Model
    path = tmp_path / "e.gguf"
    synth.write_esrgan_gguf(path, synth.ESRGAN_TINY, 7)
    full = Model.load(path, device)
    empty = Model.load(path, device, no_upload=True)
    imgs = synth.images(2, 48, 40, seed=5)
    with pytest.raises(L.Error, match="not been uploaded"):
        empty.upscale_batch(imgs)
    (src, n), (dst, n2) = full.weights_arena(), empty.weights_arena()
    assert n == n2 and n > 0
    L.vx_check(L.get_lib().vx_memcpy_d2d(dst, src, n, None))
    L.vx_check(L.get_lib().vx_stream_sync(None))
    empty.weights_ready()
    np.testing.assert_array_equal(empty.upscale_batch(imgs), full.upscale_batch(imgs))
