"""-m gpu: the SWIN encoder (BiRefNet backbone; csrc/swin.cpp, kernels_swin.hip, the masked window attention) against the CPU
oracle's restatement of reference src/visp/arch/swin.cpp (pinned in tests/test_oracle_swin.py), through the C ABI: the row
kernels on their own against numpy, every block boundary of a small configuration on a non-square image that needs window
padding at every stage, the SWIN-T configuration, batch independence and the error paths. Activations are f16 on the device
(f32 in the oracle): tolerances are relative to each tensor's largest magnitude."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle
from visioncpp_amd import _lib as L
from visioncpp_amd import synth, vision

pytestmark = pytest.mark.gpu

from gpu_util import api, dev, empty, rel_err, release, sync  # noqa: E402

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


def _windows(x, ws, shift):
    """numpy pad + roll(-shift) + window_partition of [B, H, W, C] -> ([B*nwy*nwx*N, C], validity of each row)."""
    B, H, W, Cc = x.shape
    hp, wp = -(-H // ws) * ws, -(-W // ws) * ws
    p = np.zeros((B, hp, wp, Cc), x.dtype)
    p[:, :H, :W] = x
    valid = np.zeros((B, hp, wp), bool)
    valid[:, :H, :W] = True
    p, valid = np.roll(p, (-shift, -shift), (1, 2)), np.roll(valid, (-shift, -shift), (1, 2))
    part = lambda a: a.reshape(B, hp // ws, ws, wp // ws, ws, -1).transpose(0, 1, 3, 2, 4, 5).reshape(-1, a.shape[-1] if a.ndim == 4 else 1)  # noqa: E731
    return part(p), part(valid[..., None])[:, 0]


@pytest.mark.parametrize("H,W,Cc,ws,shift", [(9, 8, 256, 7, 3), (18, 16, 96, 7, 0), (20, 27, 768, 7, 3), (14, 14, 32, 7, 3)])
def test_layernorm_window_partition_kernel(H, W, Cc, ws, shift):
    rng = np.random.default_rng(H * W)
    B = 2
    x = (rng.standard_normal((B, H, W, Cc)) * 2 + rng.standard_normal((B, H, W, 1))).astype(np.float16)
    w, b = (1 + rng.standard_normal(Cc) * 0.1).astype(np.float32), (rng.standard_normal(Cc) * 0.1).astype(np.float32)
    ln = oracle.layer_norm(x.astype(np.float32).reshape(-1, Cc), w, b, 1e-5).reshape(B, H, W, Cc)
    want, valid = _windows(ln, ws, shift)
    y = empty(want.size * 2)
    L.vx_check(api().vx_swin_layernorm_f16(dev(x).ptr, dev(w).ptr, dev(b).ptr, y.ptr, want.shape[0], Cc, 1e-5, H, W, ws, shift, 0, None))
    sync()
    got = y.to_numpy(np.float16, want.shape).astype(np.float32)
    assert np.all(got[~valid] == 0), "padded window tokens must be exact zeros (the reference pads after norm1)"
    assert rel_err(got[valid], want[valid]) < 2e-3
    # plain rows with f32 output (the per-stage output norms)
    y32 = empty(B * H * W * Cc * 4)
    L.vx_check(api().vx_swin_layernorm_f16(dev(x).ptr, dev(w).ptr, dev(b).ptr, y32.ptr, B * H * W, Cc, 1e-5, 0, 0, 0, 0, 1, None))
    sync()
    assert rel_err(y32.to_numpy(np.float32, (B * H * W, Cc)), ln.reshape(-1, Cc)) < 1e-5


@pytest.mark.parametrize("H,W,Cc,ws,shift", [(9, 8, 64, 7, 3), (16, 18, 96, 7, 0), (28, 21, 192, 7, 3)])
def test_window_reverse_add_kernel(H, W, Cc, ws, shift):
    rng = np.random.default_rng(3)
    B = 2
    x = rng.standard_normal((B, H, W, Cc)).astype(np.float16)
    hp, wp = -(-H // ws) * ws, -(-W // ws) * ws
    rows = B * (hp // ws) * (wp // ws) * ws * ws
    a = rng.standard_normal((rows, Cc)).astype(np.float16)
    # window_reverse, roll(+shift), crop
    full = a.reshape(B, hp // ws, wp // ws, ws, ws, Cc).transpose(0, 1, 3, 2, 4, 5).reshape(B, hp, wp, Cc)
    full = np.roll(full, (shift, shift), (1, 2))[:, :H, :W]
    want = (full.astype(np.float32) + x.astype(np.float32)).astype(np.float16)
    y = empty(x.size * 2)
    L.vx_check(api().vx_swin_window_reverse_add_f16(dev(a).ptr, dev(x).ptr, y.ptr, B, H, W, Cc, ws, shift, None))
    sync()
    np.testing.assert_array_equal(y.to_numpy(np.float16, x.shape), want)


@pytest.mark.parametrize("H,W,Cc", [(8, 6, 32), (18, 16, 96), (4, 10, 384)])
def test_merge_layernorm_kernel(H, W, Cc):
    rng = np.random.default_rng(H)
    B = 2
    x = rng.standard_normal((B, H, W, Cc)).astype(np.float16)
    w, b = (1 + rng.standard_normal(4 * Cc) * 0.1).astype(np.float32), (rng.standard_normal(4 * Cc) * 0.1).astype(np.float32)
    xf = x.astype(np.float32)
    cat = np.concatenate([xf[:, 0::2, 0::2], xf[:, 1::2, 0::2], xf[:, 0::2, 1::2], xf[:, 1::2, 1::2]], -1).reshape(-1, 4 * Cc)  # swin.cpp:146-153
    want = oracle.layer_norm(cat, w, b, 1e-5)
    y = empty(want.size * 2)
    L.vx_check(api().vx_swin_merge_layernorm_f16(dev(x).ptr, dev(w).ptr, dev(b).ptr, y.ptr, B, H, W, Cc, 1e-5, None))
    sync()
    assert rel_err(y.to_numpy(np.float16, want.shape).astype(np.float32), want) < 2e-3
    assert api().vx_swin_merge_layernorm_f16(dev(x).ptr, dev(w).ptr, dev(b).ptr, y.ptr, B, H - 1, W, Cc, 1e-5, None) == 0
    assert b"even spatial" in api().vx_last_error()


@pytest.mark.parametrize("heads,nwx,nwy", [(3, 3, 2), (1, 1, 1), (6, 2, 3)])
def test_masked_window_attention_kernel(heads, nwx, nwy):
    """bias table -> four packed class images; shifted windows of the last row / column use the masks of swin.cpp:165-213."""
    ws, N, hd = 7, 49, 32
    rng = np.random.default_rng(heads)
    B = 2
    n_win = B * nwx * nwy
    Cc = heads * hd
    qkv = (rng.standard_normal((n_win * N, heads, 3, hd)) * 0.7).astype(np.float16)
    table = (rng.standard_normal(((2 * ws - 1) ** 2, heads)) * 0.5).astype(np.float32)
    packed = np.zeros(4 * api().vx_window_attention_bias_bytes(N, heads) // 2, np.uint16)
    L.vx_check(api().vx_swin_attention_pack_bias(table.ctypes.data, ws, heads, packed.ctypes.data))
    idx = oracle.swin_rel_pos_index(ws).reshape(N, N)
    bias = table.astype(np.float16).astype(np.float32)[idx].transpose(2, 0, 1)  # [heads, query, key]
    for shifted in (False, True):
        out = empty(n_win * N * Cc * 2)
        L.vx_check(api().vx_window_attention_masked_f16(dev(qkv).ptr, dev(packed).ptr, out.ptr, n_win, N, heads, nwx if shifted else 0, nwy if shifted else 0, None))
        sync()
        got = out.to_numpy(np.float16, (n_win, N, heads, hd)).astype(np.float32)
        mask = oracle.swin_attention_mask(nwx * ws, nwy * ws, ws) if shifted else np.zeros((nwx * nwy, N, N), np.float32)
        q = qkv[:, :, 0].astype(np.float32).reshape(n_win, N, heads, hd)
        k = qkv[:, :, 1].astype(np.float32).reshape(n_win, N, heads, hd)
        v = qkv[:, :, 2].astype(np.float32).reshape(n_win, N, heads, hd)
        s = np.einsum("wihd,wjhd->whij", q, k) / np.sqrt(hd) + bias[None] + mask[np.arange(n_win) % (nwx * nwy)][:, None]
        s = s - s.max(-1, keepdims=True)
        p = np.exp(s)
        p /= p.sum(-1, keepdims=True)
        want = np.einsum("whij,wjhd->wihd", p, v)
        assert rel_err(got, want) < 4e-3, shifted


def _load(device, tmp_path, cfg, seed):
    path = synth.write_swin_gguf(tmp_path / f"{cfg.name}.gguf", cfg, seed=seed)
    tensors, conv_idx = synth.swin_gguf_tensors(synth.swin_state_dict(cfg, seed))
    return vision.SwinEncoder.load(path, device), oracle.Model(tensors, conv_idx), oracle.swin_params(cfg.embed_dim, cfg.window_size, cfg.depths, cfg.n_heads)


@pytest.mark.parametrize("shifted_only", [False, True])
def test_swin_mini_every_block_boundary(device, tmp_path, shifted_only):
    """256 x 288 image: maps 64x72, 32x36, 16x18, 8x9 -- window padding and shifted-window masks at every stage. Both mask
    semantics: the reference as written (default: the layer's mask acts in every block, swin.cpp:128-139, 226-237) and the
    torch twin's (shifted blocks only), each against the oracle in the same mode -- and the two must differ."""
    cfg = synth.SWIN_MINI
    enc, om, P = _load(device, tmp_path, cfg, 5)
    enc.swin_set_mask_mode(shifted_only)
    oracle.swin_set_mask_mode(shifted_only)
    try:
        _check_every_block_boundary(enc, om, P, cfg, shifted_only)
    finally:
        oracle.swin_set_mask_mode(False)


def _check_every_block_boundary(enc, om, P, cfg, shifted_only):
    W, H = 256, 288
    imgs = synth.images(2, W, H, seed=11)
    assert enc.output_dims(W, H) == [(64, 72, 32), (32, 36, 64), (16, 18, 128), (8, 9, 256)]
    enc.enable_captures(True)
    outs = enc.encode_batch(imgs)
    names = ["patch_embed"] + [f"block_{l}_{b}" for l in range(4) for b in range(cfg.depths[l])]
    sizes = {"patch_embed": 64 * 72 * 32}
    for l in range(4):
        for b in range(cfg.depths[l]):
            sizes[f"block_{l}_{b}"] = (64 >> l) * (72 >> l) * (32 << l)
    for bi in range(2):
        want, caps = oracle.swin_encode(om, P, _pre(imgs[bi]), "bb", captures=sizes)
        for n in names:
            got = enc.read_capture(n)[bi].reshape(-1)
            assert rel_err(got, caps[n]) < 6e-3, (bi, n)
        for i in range(4):
            assert outs[i].shape[1:] == want[i].shape
            assert rel_err(outs[i][bi], want[i]) < 1e-2, (bi, i)
            assert np.abs(outs[i][bi] - want[i]).mean() < 2e-3 * np.abs(want[i]).max(), (bi, i)
    enc.enable_captures(False)
    # images are independent units; a repeated launch is bit-identical
    again = enc.encode_batch(imgs[::-1].copy())
    for i in range(4):
        np.testing.assert_array_equal(again[i][::-1], outs[i])
    # the other semantics give a different first stage (edge windows of the unshifted blocks)
    enc.swin_set_mask_mode(not shifted_only)
    other = enc.encode_batch(imgs)
    enc.swin_set_mask_mode(shifted_only)
    assert np.abs(other[0] - outs[0]).max() > 1e-3


def test_swin_t_configuration(device, tmp_path):
    """swin_t_params (swin.cpp:266-275): embed 96, depths 2/2/6/2, heads 3/6/12/24, at 256 x 256 (maps 64, 32, 16, 8)."""
    cfg = synth.SWIN_T
    enc, om, P = _load(device, tmp_path, cfg, 2)
    imgs = synth.images(1, 256, 256, seed=3)
    outs = enc.encode_batch(imgs)
    want = oracle.swin_encode(om, P, _pre(imgs[0]))
    for i in range(4):
        assert outs[i].shape == (1,) + want[i].shape
        assert rel_err(outs[i][0], want[i]) < 2e-2, i
        assert np.abs(outs[i][0] - want[i]).mean() < 3e-3 * np.abs(want[i]).max(), i


def test_swin_t_1024_batch_8(device, tmp_path):
    """The backbone of configs[3] at full size: swin_t, 1024 x 1024, batch 8 -- maps 256 / 128 / 64 / 32, window 7, so every stage
    pads its windows and the edge windows carry the shift mask. Batch properties bit for bit, one image against the oracle."""
    cfg = synth.SWIN_T
    enc, om, P = _load(device, tmp_path, cfg, 2)
    imgs = synth.images(8, 1024, 1024, seed=31)
    imgs[6] = imgs[0]
    outs = enc.encode_batch(imgs)
    assert [o.shape for o in outs] == [(8, 256, 256, 96), (8, 128, 128, 192), (8, 64, 64, 384), (8, 32, 32, 768)]
    for o in outs:
        assert np.isfinite(o).all()
        np.testing.assert_array_equal(o[6], o[0])
    one = enc.encode_batch(imgs[3:4])
    for i in range(4):
        np.testing.assert_array_equal(one[i][0], outs[i][3])
    want = oracle.swin_encode(om, P, _pre(imgs[3]))
    for i in range(4):
        assert rel_err(outs[i][3], want[i]) < 2e-2, i
        assert np.abs(outs[i][3] - want[i]).mean() < 3e-3 * np.abs(want[i]).max(), i


def test_swin_errors(device, tmp_path):
    cfg = synth.SWIN_MINI
    path = synth.write_swin_gguf(tmp_path / "m.gguf", cfg, seed=1)
    enc = vision.SwinEncoder.load(path, device)
    with pytest.raises(L.Error, match="multiple of 32"):
        enc.encode_batch(np.zeros((1, 100, 128, 3), np.uint8))
    # a backbone-only file is not a whole BiRefNet model: the family loader asks for the decoder's tensors
    with pytest.raises(L.Error, match="not found"):
        vision.Model.load(path, device, vision.Arch.birefnet)
    da = synth.write_gguf(tmp_path / "d.gguf", synth.TINY, seed=0)
    with pytest.raises(L.Error, match="Architecture expected to be 'birefnet'"):
        vision.SwinEncoder.load(da, device)
    bad = dict(synth.swin_state_dict(cfg, 1))
    del bad["bb.layers.2.blocks.1.mlp.fc1.weight"]
    with pytest.raises(L.Error, match="not found"):
        vision.SwinEncoder.load(synth.write_swin_gguf(tmp_path / "bad.gguf", cfg, sd=bad), device)
