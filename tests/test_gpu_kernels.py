"""-m gpu: every HIP kernel behind the vx_* C ABI against the CPU oracle (oracle/) on the same
seeded inputs. f16 operands / f32 accumulation: tolerance is stated per test as the maximum
absolute error relative to the largest reference magnitude."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle
from visioncpp_amd import _lib as L

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _device():
    api = L.get_lib()
    assert api.vx_device_count() > 0, "no HIP device visible: the product path has no CPU fallback"
    L.vx_check(api.vx_set_device(0))
    name, arch = C.create_string_buffer(256), C.create_string_buffer(64)
    L.vx_check(api.vx_device_info(0, name, 256, arch, 64, None, None, None))
    assert arch.value.decode().startswith("gfx950"), arch.value
    yield


@pytest.fixture(autouse=True)
def _release_buffers():
    yield
    release()


from gpu_util import api, dev, empty, gemm, pack_rcu, pad_vec, pad_weight, rel_err, release, sync  # noqa: E402

F16_TOL = 4e-3  # one f16 rounding of the output (2^-11) plus f16-rounded operands over K <= 1536


def _rand(rng, *shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


def _h(a):  # round through f16 (what the device operands hold)
    return a.astype(np.float16).astype(np.float32)


@pytest.mark.parametrize("M,N,K", [(300, 384, 384), (1370 * 2, 1536, 384), (257, 384, 1536), (131, 64, 64), (128, 32, 320), (77, 96, 128)])
@pytest.mark.parametrize("epi", ["f16", "gelu", "relu"])
def test_gemm_f16_epilogues(M, N, K, epi):
    rng = np.random.default_rng(M + N + K)
    a, w, b = _h(_rand(rng, M, K)), _h(_rand(rng, N, K, scale=K ** -0.5)), _rand(rng, N, scale=0.1)
    wp = pad_weight(w)
    out = empty(M * wp.shape[0] * 2)
    code = {"f16": L.EPI_F16, "gelu": L.EPI_F16_GELU, "relu": L.EPI_F16_RELU}[epi]
    gemm(dev(a.astype(np.float16)), wp, pad_vec(b, wp.shape[0]), M, code, lda=K, out=out, ldo=wp.shape[0], n_valid=N)
    got = out.to_numpy(np.float16, (M, wp.shape[0]))[:, :N].astype(np.float32)
    want = oracle.linear(a, w, b)
    if epi == "gelu":
        want = oracle.gelu(want, oracle.GELU_TANH_F32)
    elif epi == "relu":
        want = np.maximum(want, 0)
    assert rel_err(got, want) < F16_TOL
    if wp.shape[0] > N:  # columns >= n_valid must stay untouched (zero-initialised buffer)
        assert not out.to_numpy(np.float16, (M, wp.shape[0]))[:, N:].any()


def test_gemm_row_remap_skips_cls_token():
    """a_group remap: neck projections read rows 1..P of each image's T = P+1 tokens (depth-anything.cpp:50)."""
    rng = np.random.default_rng(5)
    B, P, K, N = 3, 50, 128, 64
    feats = _h(_rand(rng, B * (P + 1), K))
    w, b = _h(_rand(rng, N, K, scale=K ** -0.5)), _rand(rng, N)
    out = empty(B * P * N * 2)
    gemm(dev(feats.astype(np.float16)), pad_weight(w), b, B * P, L.EPI_F16, lda=K, out=out, ldo=N, a_group=P,
         a_group_stride=P + 1, a_row_off=1)
    got = out.to_numpy(np.float16, (B, P, N)).astype(np.float32)
    want = oracle.linear(feats.reshape(B, P + 1, K)[:, 1:], w, b)
    assert rel_err(got, want) < F16_TOL


@pytest.mark.parametrize("M,N,K", [(1370, 384, 384), (200, 128, 512)])
def test_gemm_residual_layerscale(M, N, K):
    """x += lambda * (A W^T + b): attention out-proj / fc2 epilogue (dino.cpp:48-50, 80-87)."""
    rng = np.random.default_rng(7)
    a, w, b = _h(_rand(rng, M, K)), _h(_rand(rng, N, K, scale=K ** -0.5)), _rand(rng, N, scale=0.1)
    lam, x = _rand(rng, N, scale=0.3), _rand(rng, M, N)
    xd = dev(x)
    gemm(dev(a.astype(np.float16)), pad_weight(w), b, M, L.EPI_RESID_F32, lda=K, out=xd, ldo=N, lambda_=dev(lam))
    got = xd.to_numpy(np.float32, (M, N))
    want = x + oracle.linear(a, w, b) * lam
    assert rel_err(got, want) < 1e-3  # f32 output: only operand rounding + accumulation order


def test_gemm_tokens_epilogue():
    """patch-embed GEMM writes rows 1.. of each image and adds the position embedding (dino.cpp:32-46)."""
    rng = np.random.default_rng(8)
    B, P, K, N = 2, 37, 588, 128
    a, w, b = _h(_rand(rng, B * P, K)), _h(_rand(rng, N, K, scale=K ** -0.5)), _rand(rng, N)
    pos = _rand(rng, P + 1, N)
    wp = pad_weight(w, 128)
    ap = np.zeros((B * P, wp.shape[1]), np.float16)
    ap[:, :K] = a
    x = empty(B * (P + 1) * N * 4)
    gemm(dev(ap), wp, b, B * P, L.EPI_TOKENS, lda=wp.shape[1], out=x, ldo=N, pos=dev(pos), tokens_P=P)
    got = x.to_numpy(np.float32, (B, P + 1, N))
    want = oracle.linear(a, w, b).reshape(B, P, N) + pos[1:]
    assert rel_err(got[:, 1:], want) < 1e-3
    assert not got[:, 0].any()  # cls rows are written by vx_write_cls_rows, not by the GEMM


def test_gemm_qkv_scatter():
    rng = np.random.default_rng(9)
    B, T, H = 2, 70, 2
    Cc = H * 64
    a = _h(_rand(rng, B * T, Cc))
    w, b = _h(_rand(rng, 3 * Cc, Cc, scale=Cc ** -0.5)), _rand(rng, 3 * Cc, scale=0.1)
    q, k, v = (empty(B * H * T * 64 * 2) for _ in range(3))
    gemm(dev(a.astype(np.float16)), pad_weight(w), b, B * T, L.EPI_QKV, lda=Cc, q=q, k=k, vt=v, qkv_T=T, qkv_H=H, q_scale=0.125)
    y = oracle.linear(a, w, b).reshape(B, T, 3, H, 64)
    gq, gk, gv = (t.to_numpy(np.float16, (B, H, T, 64)).astype(np.float32) for t in (q, k, v))
    assert rel_err(gq, y[:, :, 0].transpose(0, 2, 1, 3) * 0.125) < F16_TOL
    assert rel_err(gk, y[:, :, 1].transpose(0, 2, 1, 3)) < F16_TOL
    assert rel_err(gv, y[:, :, 2].transpose(0, 2, 1, 3)) < F16_TOL


@pytest.mark.parametrize("s,c", [(4, 48), (2, 96)])
def test_conv_transpose_as_gemm_pixel_shuffle(s, c):
    """conv_transpose_2d with k == stride (nn.cpp:117-129) vs the oracle's torch-semantics convT."""
    rng = np.random.default_rng(s)
    B, Hh, Ww = 2, 5, 7
    x = _h(_rand(rng, B, Hh, Ww, c))
    w, b = _h(_rand(rng, c, c, s, s, scale=c ** -0.5)), _rand(rng, c, scale=0.1)  # torch [Cin, Cout, kh, kw]
    kp = -(-c // 64) * 64
    rows = np.zeros((s * s * c, kp), np.float32)
    rows[:, :c] = w.transpose(2, 3, 1, 0).reshape(s * s * c, c)  # n = (dy*s+dx)*Cout + co, k = ci
    xp = np.zeros((B * Hh * Ww, kp), np.float16)
    xp[:, :c] = x.reshape(-1, c)
    out = empty(B * Hh * s * Ww * s * c * 2)
    gemm(dev(xp), pad_weight(rows), np.tile(b, s * s), B * Hh * Ww, L.EPI_PIXSHUF, lda=kp, out=out, ldo=c, ps_s=s, ps_Cout=c,
         ps_H=Hh, ps_W=Ww, n_valid=s * s * c)
    got = out.to_numpy(np.float16, (B, Hh * s, Ww * s, c)).astype(np.float32)
    want = oracle.conv_transpose2d_nhwc(x, w, b, s)
    assert rel_err(got, want) < F16_TOL


@pytest.mark.parametrize("cin,cout,stride,hw", [(64, 64, 1, (19, 23)), (48, 64, 1, (20, 12)), (32, 32, 1, (30, 17)), (128, 128, 2, (37, 37)),
                                                 (96, 64, 1, (9, 9))])
@pytest.mark.parametrize("mode", ["plain", "rcu1", "rcu2"])
def test_conv3x3_implicit_gemm(cin, cout, stride, hw, mode):
    """3x3 pad-1 NHWC conv (nn.cpp:72-100) and the fused residual-unit forms (depth-anything.cpp:15-30)."""
    if mode != "plain" and (cin != cout or stride != 1):
        pytest.skip("residual forms are same-shape convs")
    rng = np.random.default_rng(cin + cout + stride)
    B, (Hh, Ww) = 2, hw
    x = _h(_rand(rng, B, Hh, Ww, cin))
    w, b = _h(_rand(rng, cout, 3, 3, cin, scale=(9 * cin) ** -0.5)), _rand(rng, cout, scale=0.1)
    OH, OW = (Hh + 2 - 3) // stride + 1, (Ww + 2 - 3) // stride + 1
    out = empty(B * OH * OW * cout * 2)
    kw = dict(conv_kh=3, conv_kw=3, conv_stride=stride, conv_pad=1, conv_H=Hh, conv_W=Ww, conv_Cin=cin, conv_OH=OH, conv_OW=OW,
              n_valid=cout)
    wp = pad_weight(w.reshape(cout, -1))
    xin = x
    if mode == "plain":
        gemm(dev(x.astype(np.float16)), wp, pad_vec(b, wp.shape[0]), B * OH * OW, L.EPI_F16, out=out, ldo=cout, **kw)
        want = oracle.conv2d_nhwc(x, w, b, stride, 1)
    elif mode == "rcu1":  # relu on load, relu on store
        gemm(dev(x.astype(np.float16)), wp, pad_vec(b, wp.shape[0]), B * OH * OW, L.EPI_F16_RELU, out=out, ldo=cout, a_relu=1, **kw)
        want = np.maximum(oracle.conv2d_nhwc(np.maximum(xin, 0), w, b, 1, 1), 0)
    else:  # conv + two residual addends
        r1, r2 = _h(_rand(rng, B, OH, OW, cout)), _h(_rand(rng, B, OH, OW, cout))
        gemm(dev(x.astype(np.float16)), wp, pad_vec(b, wp.shape[0]), B * OH * OW, L.EPI_F16_ADD, out=out, ldo=cout,
             res1=dev(r1.astype(np.float16)), res2=dev(r2.astype(np.float16)), **kw)
        want = oracle.conv2d_nhwc(x, w, b, 1, 1) + r1 + r2
    got = out.to_numpy(np.float16, (B, OH, OW, cout)).astype(np.float32)
    assert rel_err(got, want) < F16_TOL


LOG2E = 1.4426950408889634


@pytest.mark.parametrize("B,H,T", [(1, 1, 64), (2, 2, 65), (1, 3, 50), (1, 2, 257), (2, 6, 1370)])
def test_attention(B, H, T):
    """fused MHSA, head_dim 64 (nn.cpp:210-244) vs the oracle's softmax(q k^T * scale) v."""
    rng = np.random.default_rng(T)
    Cc = H * 64
    q, k, v = (_h(_rand(rng, B, T, Cc)) for _ in range(3))
    scale = 0.125
    qs = q * (scale * LOG2E)  # the QKV epilogue stores q pre-scaled by log2(e) / sqrt(64) (VX_ATTN_Q_SCALE): exp2-domain kernel
    qd = dev(qs.reshape(B, T, H, 64).transpose(0, 2, 1, 3).astype(np.float16))
    kd = dev(k.reshape(B, T, H, 64).transpose(0, 2, 1, 3).astype(np.float16))
    vd = dev(v.reshape(B, T, H, 64).transpose(0, 2, 1, 3).astype(np.float16))
    out = empty(B * T * Cc * 2)
    L.vx_check(api().vx_attention_f16(qd.ptr, kd.ptr, vd.ptr, out.ptr, B, H, T, None))
    sync()
    got = out.to_numpy(np.float16, (B, T, Cc)).astype(np.float32)
    want = np.stack([oracle.attention(q[i], k[i], v[i], H, scale) for i in range(B)])
    # P is rounded to f16 before the PV product: 2^-11 relative per term
    assert rel_err(got, want) < 5e-3


def test_attention_online_softmax_rescale_branch():
    """Forces the running max to jump late (a spiked key in the last tile), so every earlier
    tile's partial sums must be rescaled (guide rule 26: a data-dependent branch needs its own test)."""
    rng = np.random.default_rng(3)
    B, H, T = 1, 1, 300
    q, k, v = (_h(_rand(rng, B, T, 64)) for _ in range(3))
    k[0, 290] = q[0, 7] * 4.0  # key 290 dominates query 7 only
    k = _h(k)
    out = empty(T * 64 * 2)
    L.vx_check(api().vx_attention_f16(dev((q * (0.125 * LOG2E)).astype(np.float16)).ptr, dev(k.astype(np.float16)).ptr,
                                      dev(v.astype(np.float16)).ptr, out.ptr, B, H, T, None))
    sync()
    got = out.to_numpy(np.float16, (T, 64)).astype(np.float32)
    want = oracle.attention(q[0], k[0], v[0], 1, 0.125)
    assert rel_err(got, want) < 5e-3


def _attn(q, k, v, T):
    out = empty(T * 64 * 2)
    L.vx_check(api().vx_attention_f16(dev((q * (0.125 * LOG2E)).astype(np.float16)).ptr, dev(k.astype(np.float16)).ptr,
                                      dev(v.astype(np.float16)).ptr, out.ptr, 1, 1, T, None))
    sync()
    return out.to_numpy(np.float16, (T, 64)).astype(np.float32)


def test_attention_lagging_reference_fast_path_and_its_fallback():
    """Round 3: tiles 1 .. n-2 exponentiate against the reference point the wave already has (no maximum, no subtract) and fall
    back to the full path when a lane's partial row sum exceeds the limit (guide rule 26: force the branch, full-tensor
    independent reference, threshold sweep).
      * query 7: key 200 (tile 3 of 10) beats everything before it by ~40 in the exp2 domain -> the fallback MUST fire mid-stream
        and rescale O and l of tiles 0-2;
      * query 9: key 330 (tile 5) beats the earlier maximum by ~6 -> stays on the fast path with P up to ~2^6 (no rescale);
      * query 11: key 70 (tile 1) by ~9.5 -> P up to ~2^9.5, just under the limit, or the fallback; either must be right.
    Limits 0 (every tile on the full path) / default / 2^14 (the fast path where f16 can still hold P) must agree to rounding."""
    rng = np.random.default_rng(17)
    T = 640
    q, k, v = (_h(_rand(rng, 1, T, 64)) for _ in range(3))
    k[0, 200] = q[0, 7] * 4.0
    k[0, 330] = q[0, 9] * (8.0 / float((q[0, 9] ** 2).sum() * 0.125))     # score 8 -> 11.5 in the exp2 domain
    k[0, 70] = q[0, 11] * (10.5 / float((q[0, 11] ** 2).sum() * 0.125))   # score 10.5 -> 15.1
    k = _h(k)
    want = oracle.attention(q[0], k[0], v[0], 1, 0.125)
    try:
        got = {}
        for limit in (0.0, -1.0, 16384.0):
            api().vx_attention_set_fast_limit(limit)
            got[limit] = _attn(q[0], k[0], v[0], T)
            assert np.isfinite(got[limit]).all(), limit
            assert rel_err(got[limit], want) < 5e-3, limit
            for row in (7, 9, 11):
                assert np.abs(got[limit][row] - want[row]).max() < 5e-3 * max(1.0, np.abs(want[row]).max()), (limit, row)
        # same arithmetic up to the reference point: the three agree far inside the f16 output rounding
        assert np.abs(got[0.0] - got[-1.0]).max() < 2e-3 * np.abs(want).max()
        assert np.abs(got[16384.0] - got[-1.0]).max() < 2e-3 * np.abs(want).max()
    finally:
        api().vx_attention_set_fast_limit(-1.0)


def test_attention_row_sum_of_a_peaked_row():
    """The softmax denominator of a peaked row: one key scores ~9 log2 units above 1369 others (a real checkpoint's cls / register-like
    rows). The tile row sums are packed-f16 add trees (VISP_ATTN_RS = 1), accumulated in f32: the small P's must not be swallowed next to
    the large one. V marks the peak key in channel 0 and the small keys in channel 1, so the output IS the two softmax masses; both the
    fast path and the forced full path are held to 1.5e-3 of the f64 softmax (the f16 output rounding alone is 5e-4)."""
    rng = np.random.default_rng(41)
    T, peak = 1370, 777
    q = _h(_rand(rng, 1, T, 64) * 0.05)
    k = _h(_rand(rng, 1, T, 64) * 0.05)
    q[0, :, 0], k[0, :, 0] = 4.0, 0.0
    k[0, peak, 0] = 9.0 / LOG2E / (4.0 * 0.125)       # + 9 in the exp2 domain for every query
    q, k = _h(q), _h(k)
    v = np.zeros((1, T, 64), np.float32)
    v[0, :, 1] = 1.0
    v[0, peak, 0], v[0, peak, 1] = 1.0, 0.0
    s = (q[0].astype(np.float64) @ k[0].astype(np.float64).T) * 0.125
    p = np.exp(s - s.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    want = p @ v[0].astype(np.float64)
    assert 0.15 < want[:, 0].mean() < 0.45            # the peak holds a fraction, the 1369 small keys the rest
    try:
        for limit in (-1.0, 0.0):
            api().vx_attention_set_fast_limit(limit)
            got = _attn(q[0], k[0], v[0], T)
            assert np.abs(got[:, :2] - want[:, :2]).max() < 1.5e-3, limit
    finally:
        api().vx_attention_set_fast_limit(-1.0)


def test_attention_reference_point_follows_a_slowly_rising_maximum():
    """Scores that climb by ~3 (exp2 domain) per 64-key tile: no single tile trips the limit at once, the lag accumulates until
    one does; and a row whose scores keep FALLING after the first tile (P shrinks to f16 subnormals against the stale
    reference, as in any online softmax anchored at the running maximum)."""
    rng = np.random.default_rng(23)
    T = 1370
    q, k, v = (_h(_rand(rng, 1, T, 64) * 0.3) for _ in range(3))
    ramp = np.arange(T, dtype=np.float32) / 64.0 * 3.0 / LOG2E      # natural-log units, added through one feature
    q[0, :, 0] = 8.0
    k[0, :, 0] = ramp / (8.0 * 0.125)
    k[0, 700:, 0] = k[0, 699, 0] - (np.arange(T - 700, dtype=np.float32) / 64.0 * 2.0 / LOG2E)
    q, k = _h(q), _h(k)
    want = oracle.attention(q[0], k[0], v[0], 1, 0.125)
    got = _attn(q[0], k[0], v[0], T)
    assert np.isfinite(got).all()
    assert rel_err(got, want) < 5e-3


def test_attention_all_scores_far_below_zero():
    """Every score of a row is about -40 in the exp2 domain: the running maximum must follow it down (an online
    softmax anchored at zero would underflow every P in f16 and divide by a zero row sum)."""
    rng = np.random.default_rng(5)
    T = 200
    q = np.zeros((1, T, 64), np.float32)
    k = np.zeros((1, T, 64), np.float32)
    q[..., 0], k[..., 0] = 16.0, -14.0          # q.k * 0.125 = -28 -> -40.4 in the exp2 domain
    q[..., 1:] = _h(_rand(rng, 1, T, 63) * 0.3)
    k[..., 1:] = _h(_rand(rng, 1, T, 63) * 0.3)
    v = _h(_rand(rng, 1, T, 64))
    out = empty(T * 64 * 2)
    L.vx_check(api().vx_attention_f16(dev((q * (0.125 * LOG2E)).astype(np.float16)).ptr, dev(k.astype(np.float16)).ptr,
                                      dev(v.astype(np.float16)).ptr, out.ptr, 1, 1, T, None))
    sync()
    got = out.to_numpy(np.float16, (1, T, 64)).astype(np.float32)
    assert np.isfinite(got).all()
    assert rel_err(got, oracle.attention(q[0], k[0], v[0], 1, 0.125)[None]) < 5e-3


@pytest.mark.parametrize("M,Cc", [(1370, 384), (77, 128), (33, 768), (10, 96)])
def test_layernorm(M, Cc):
    rng = np.random.default_rng(Cc)
    x = _rand(rng, M, Cc, scale=3.0) + 1.5
    w, b = 1 + _rand(rng, Cc, scale=0.1), _rand(rng, Cc, scale=0.1)
    y = empty(M * Cc * 2)
    L.vx_check(api().vx_layernorm_f32_f16(dev(x).ptr, dev(w).ptr, dev(b).ptr, y.ptr, M, Cc, 1e-6, None))
    sync()
    got = y.to_numpy(np.float16, (M, Cc)).astype(np.float32)
    want = oracle.layer_norm(x, w, b, 1e-6)
    assert rel_err(got, want) < 1e-3  # f32 math, one f16 rounding of the output


def test_preprocess_patches_and_f32():
    rng = np.random.default_rng(1)
    B, Hh, Ww, ps = 2, 28, 42, 14
    img = rng.integers(0, 256, (B, Hh, Ww, 3), dtype=np.uint8)
    mean = (C.c_float * 3)(0.485, 0.456, 0.406)
    inv = (C.c_float * 3)(*[np.float32(1.0) / np.float32(s) for s in (0.229, 0.224, 0.225)])
    Kp = 640
    pat = empty(B * 2 * 3 * Kp * 2)
    L.vx_check(api().vx_preprocess_patches(dev(img).ptr, pat.ptr, B, Hh, Ww, ps, Kp, mean, inv, None))
    f32 = empty(B * Hh * Ww * 3 * 4)
    L.vx_check(api().vx_preprocess_f32(dev(img).ptr, f32.ptr, B, Hh, Ww, mean, inv, None))
    sync()
    want = np.stack([oracle.image_u8_to_f32(img[i], oracle.RGB_U8, oracle.RGB_F32, (-0.485, -0.456, -0.406, 0),
                                            (1 / 0.229, 1 / 0.224, 1 / 0.225, 1)) for i in range(B)])
    got32 = f32.to_numpy(np.float32, (B, Hh, Ww, 3))
    np.testing.assert_allclose(got32, want, rtol=0, atol=1e-6)
    gp = pat.to_numpy(np.float16, (B, 2, 3, Kp)).astype(np.float32)
    wp = want.reshape(B, 2, ps, 3, ps, 3).transpose(0, 1, 3, 2, 4, 5).reshape(B, 2, 3, ps * ps * 3)  # k = (ky, kx, c)
    assert rel_err(gp[..., :588], wp) < 1e-3
    assert not gp[..., 588:].any()


@pytest.mark.parametrize("shape,target", [((2, 19, 19, 64), (37, 37)), ((1, 37, 23, 64), (74, 46)), ((2, 40, 40, 32), (70, 70)), ((1, 5, 7, 8), (9, 3))])
def test_bilinear_align_corners(shape, target):
    rng = np.random.default_rng(shape[1])
    x = _h(_rand(rng, *shape))
    B, Hh, Ww, Cc = shape
    y = empty(B * target[0] * target[1] * Cc * 2)
    L.vx_check(api().vx_bilinear_ac_f16(dev(x.astype(np.float16)).ptr, y.ptr, B, Hh, Ww, Cc, target[0], target[1], None))
    sync()
    got = y.to_numpy(np.float16, (B, *target, Cc)).astype(np.float32)
    want = oracle.interpolate_nhwc(x, target, "bilinear", True)
    assert rel_err(got, want) < 1e-3


def test_head_out_and_minmax_normalize():
    rng = np.random.default_rng(2)
    B, n, Cc = 3, 518 * 37, 32
    x = np.maximum(_h(_rand(rng, B * n, Cc)), 0)
    w, bias = np.abs(_rand(rng, Cc, scale=0.2)), 0.1
    depth, out, mm = empty(B * n * 4), empty(B * n * 4), empty(B * 8)
    L.vx_check(api().vx_head_out_f32(dev(x.astype(np.float16)).ptr, dev(w).ptr, bias, 1.0, depth.ptr, B * n, Cc, None))
    L.vx_check(api().vx_minmax_normalize(depth.ptr, out.ptr, mm.ptr, B, n, None))
    sync()
    d = depth.to_numpy(np.float32, (B, n))
    want_d = np.maximum(x @ w + bias, 0).reshape(B, n)
    assert rel_err(d, want_d) < 1e-5
    got = out.to_numpy(np.float32, (B, n))
    want = np.stack([oracle.image_normalize(d[i].reshape(1, n)).ravel() for i in range(B)])
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)
    assert got.min() == 0.0 and abs(got.max() - 1.0) < 1e-6  # exact 0: mul and add are not contracted
    u8 = empty(B * n)
    L.vx_check(api().vx_f32_to_u8(out.ptr, u8.ptr, B * n, None))
    sync()
    np.testing.assert_array_equal(u8.to_numpy(np.uint8, (B, n)), oracle.image_f32_to_u8(got.reshape(B, n, 1), oracle.ALPHA_F32, oracle.ALPHA_U8).reshape(B, n))


def test_constant_image_normalizes_to_zero():
    """delta < 1e-5 => scale 1 (image.cpp:560-563): a constant depth map maps to 0, not NaN."""
    d = np.full((1, 1000), 3.25, np.float32)
    out, mm = empty(4000), empty(8)
    L.vx_check(api().vx_minmax_normalize(dev(d).ptr, out.ptr, mm.ptr, 1, 1000, None))
    sync()
    assert not out.to_numpy(np.float32, (1000,)).any()


@pytest.mark.parametrize("stages", [1, 2, 4, 8, 16, 32, 64])
def test_gemm_tuning_variants(stages):
    """The LDS-ring depth / block-tile knob (vx_gemm_args.stages) must not change results."""
    rng = np.random.default_rng(stages)
    M, N, K = 1370 * 2 + 77, 384, 384
    a, w, b = _h(_rand(rng, M, K)), _h(_rand(rng, N, K, scale=K ** -0.5)), _rand(rng, N, scale=0.1)
    out = empty(M * N * 2)
    gemm(dev(a.astype(np.float16)), pad_weight(w), b, M, L.EPI_F16_GELU, lda=K, out=out, ldo=N, stages=stages)
    want = oracle.gelu(oracle.linear(a, w, b), oracle.GELU_TANH_F32)
    assert rel_err(out.to_numpy(np.float16, (M, N)).astype(np.float32), want) < F16_TOL
    lam, x = _rand(rng, N, scale=0.3), _rand(rng, M, N)
    xd = dev(x)
    gemm(dev(a.astype(np.float16)), pad_weight(w), b, M, L.EPI_RESID_F32, lda=K, out=xd, ldo=N, lambda_=dev(lam), stages=stages)
    assert rel_err(xd.to_numpy(np.float32, (M, N)), x + oracle.linear(a, w, b) * lam) < 1e-3


def test_conv3x3_fused_head_output():
    """head.conv2 + ReLU + head.conv3 + ReLU in one kernel (depth-anything.cpp:87-94) vs oracle conv + 1x1."""
    rng = np.random.default_rng(11)
    B, Hh, Ww, Cc = 2, 29, 31, 32
    x = _h(_rand(rng, B, Hh, Ww, Cc))
    w, b = _h(_rand(rng, 32, 3, 3, Cc, scale=(9 * Cc) ** -0.5)), _rand(rng, 32, scale=0.1)
    w3, b3 = np.abs(_rand(rng, 32, scale=0.3)), 0.05
    out = empty(B * Hh * Ww * 4)
    gemm(dev(x.astype(np.float16)), pad_weight(w.reshape(32, -1)), b, B * Hh * Ww, L.EPI_HEAD_OUT, out=out, ldo=1,
         conv_kh=3, conv_kw=3, conv_stride=1, conv_pad=1, conv_H=Hh, conv_W=Ww, conv_Cin=Cc, conv_OH=Hh, conv_OW=Ww,
         lambda_=dev(w3), head_bias=b3, head_scale=2.0)
    got = out.to_numpy(np.float32, (B, Hh, Ww))
    h2 = _h(np.maximum(oracle.conv2d_nhwc(x, w, b, 1, 1), 0))  # the fused kernel rounds conv2's output to f16 too
    want = np.maximum(h2 @ w3 + b3, 0) * 2.0
    assert rel_err(got, want) < 2e-3


@pytest.mark.parametrize("cin,cout", [(64, 64), (64, 32), (32, 32), (32, 64)])
@pytest.mark.parametrize("hw", [(8, 32), (37, 50), (21, 100)])
@pytest.mark.parametrize("mode", ["plain", "rcu1", "rcu2", "head"])
def test_conv3x3_halo_kernel(cin, cout, hw, mode):
    """Halo-in-LDS 3x3 conv (kernels_conv.hip) vs the oracle, incl. ragged tiles at the right/bottom edges."""
    if mode == "head" and cout != 32:
        pytest.skip("fused head output is a 32-channel epilogue")
    if mode in ("rcu1", "rcu2") and cin != cout:
        pytest.skip("residual forms are same-shape convs")
    rng = np.random.default_rng(cin * 7 + cout + hw[0])
    B, (Hh, Ww) = 2, hw
    x = _h(_rand(rng, B, Hh, Ww, cin))
    w, b = _h(_rand(rng, cout, 3, 3, cin, scale=(9 * cin) ** -0.5)), _rand(rng, cout, scale=0.1)
    kw = dict(conv_kh=3, conv_kw=3, conv_stride=1, conv_pad=1, conv_H=Hh, conv_W=Ww, conv_Cin=cin, conv_OH=Hh, conv_OW=Ww, _halo=True)
    wp = pad_weight(w.reshape(cout, -1))
    xd = dev(x.astype(np.float16))
    M = B * Hh * Ww
    if mode == "head":
        w3, b3 = np.abs(_rand(rng, 32, scale=0.3)), 0.05
        out = empty(M * 4)
        gemm(xd, wp, b, M, L.EPI_HEAD_OUT, out=out, ldo=1, lambda_=dev(w3), head_bias=b3, head_scale=1.5, **kw)
        h2 = _h(np.maximum(oracle.conv2d_nhwc(x, w, b, 1, 1), 0))
        want = np.maximum(h2 @ w3 + b3, 0) * 1.5
        assert rel_err(out.to_numpy(np.float32, (B, Hh, Ww)), want) < 2e-3
        return
    out = empty(M * cout * 2)
    if mode == "plain":
        gemm(xd, wp, b, M, L.EPI_F16, out=out, ldo=cout, **kw)
        want = oracle.conv2d_nhwc(x, w, b, 1, 1)
    elif mode == "rcu1":
        gemm(xd, wp, b, M, L.EPI_F16_RELU, out=out, ldo=cout, a_relu=1, **kw)
        want = np.maximum(oracle.conv2d_nhwc(np.maximum(x, 0), w, b, 1, 1), 0)
    else:
        r1, r2 = _h(_rand(rng, B, Hh, Ww, cout)), _h(_rand(rng, B, Hh, Ww, cout))
        gemm(xd, wp, b, M, L.EPI_F16_ADD, out=out, ldo=cout, res1=dev(r1.astype(np.float16)), res2=dev(r2.astype(np.float16)), **kw)
        want = oracle.conv2d_nhwc(x, w, b, 1, 1) + r1 + r2
    assert rel_err(out.to_numpy(np.float16, (B, Hh, Ww, cout)).astype(np.float32), want) < F16_TOL


@pytest.mark.parametrize("cin,cout,stride,hw,epi", [(384, 64, 1, (19, 19), "plain"), (192, 64, 1, (37, 37), "relu"), (384, 384, 2, (37, 37), "plain"),
                                                      (256, 64, 1, (9, 13), "add")])
def test_conv_split_k(cin, cout, stride, hw, epi):
    """Deep reductions on few output tiles (the DPT neck convs on the 37 x 37 / 19 x 19 maps, depth-anything.cpp:66-69, 62-63): the k-loop
    cut into ranges (vx_gemm_args.k_splits), partial sums added in a fixed order by a second launch. Same result as the oracle,
    bit-identical between two launches, and the heuristic picks a split for exactly these shapes."""
    rng = np.random.default_rng(cin + cout + hw[0])
    B, (Hh, Ww) = 2, hw
    x = _h(_rand(rng, B, Hh, Ww, cin))
    w, b = _h(_rand(rng, cout, 3, 3, cin, scale=(9 * cin) ** -0.5)), _rand(rng, cout, scale=0.1)
    OH, OW = (Hh + 2 - 3) // stride + 1, (Ww + 2 - 3) // stride + 1
    M = B * OH * OW
    wp = pad_weight(w.reshape(cout, -1))
    ks = api().vx_gemm_pick_k_splits(M, wp.shape[0], wp.shape[1])
    assert ks > 1, "the heuristic is meant to split these shapes"
    assert api().vx_gemm_pick_k_splits(32 * 148 * 148, 64, 448) == 1 and api().vx_gemm_pick_k_splits(M, wp.shape[0], 576) == 1
    part = empty(ks * M * wp.shape[0] * 4)
    kw = dict(conv_kh=3, conv_kw=3, conv_stride=stride, conv_pad=1, conv_H=Hh, conv_W=Ww, conv_Cin=cin, conv_OH=OH, conv_OW=OW, n_valid=cout,
              k_splits=ks, k_partial=part)
    xd = dev(x.astype(np.float16))
    outs = []
    r1 = _h(_rand(rng, B, OH, OW, cout))
    for _ in range(2):
        out = empty(M * cout * 2)
        if epi == "plain":
            gemm(xd, wp, pad_vec(b, wp.shape[0]), M, L.EPI_F16, out=out, ldo=cout, **kw)
        elif epi == "relu":
            gemm(xd, wp, pad_vec(b, wp.shape[0]), M, L.EPI_F16_RELU, out=out, ldo=cout, **kw)
        else:
            gemm(xd, wp, pad_vec(b, wp.shape[0]), M, L.EPI_F16_ADD, out=out, ldo=cout, res1=dev(r1.astype(np.float16)), **kw)
        outs.append(out.to_numpy(np.float16, (B, OH, OW, cout)))
    np.testing.assert_array_equal(outs[0], outs[1])
    want = oracle.conv2d_nhwc(x, w, b, stride, 1)
    want = np.maximum(want, 0) if epi == "relu" else (want + r1 if epi == "add" else want)
    assert rel_err(outs[0].astype(np.float32), want) < F16_TOL
    # misuse is reported
    g = L.GemmArgs()
    g.M, g.N, g.K, g.k_splits, g.A, g.W = 128, 64, 128, 4, xd.ptr, xd.ptr
    assert api().vx_gemm_f16(C.byref(g), None) == 0 and b"k_partial" in api().vx_last_error()


@pytest.mark.parametrize("B,hs,ws,H,W", [(2, 37, 37, 64, 64), (1, 296, 296, 518, 518), (3, 26, 40, 45, 70), (1, 10, 11, 17, 19)])
def test_headconv_resize_conv_relu_conv_relu(B, hs, ws, H, W):
    """kernels_headconv.hip: bilinear (align_corners) resize + conv 3x3 32 -> 32 + ReLU + conv 1x1 -> 1 + ReLU + scale in one kernel
    (depth-anything.cpp:84-94) vs torch fp32 on the same f16 operands (the resized map rounded to f16, as the kernel holds it in LDS).
    Shapes: whole tiles, the north star's 296 -> 518 (edge tiles in both directions), a ragged non-square map, a map smaller than a tile."""
    import torch
    import torch.nn.functional as F

    lib = api()
    assert lib.vx_headconv_supported(32, 32, H, W, hs, ws) == 1
    rng = np.random.default_rng(B * 1000 + H)
    x = (rng.standard_normal((B, hs, ws, 32)) * 0.7).astype(np.float16)
    w2 = (rng.standard_normal((32, 3, 3, 32)) / np.sqrt(288)).astype(np.float16)  # OHWI rows, k = (ky, kx, c)
    b2 = (0.1 * rng.standard_normal(32)).astype(np.float32)
    w3 = (rng.standard_normal(32) / 4).astype(np.float32)
    b3, scale = 0.05, 2.5
    rows = np.zeros((32, 320), np.float16)
    rows[:, :288] = w2.reshape(32, 288)
    frag = np.empty(lib.vx_headconv_frag_bytes() // 2, np.float16)
    L.vx_check(lib.vx_headconv_pack(rows.ctypes.data, 320, frag.ctypes.data))
    xd, fd, bd, wd = dev(x), dev(frag), dev(b2), dev(w3)
    out = empty(B * H * W * 4)
    L.vx_check(lib.vx_headconv_bil_f16(xd.ptr, fd.ptr, bd.ptr, wd.ptr, b3, scale, out.ptr, B, H, W, hs, ws, None))
    sync()
    got = out.to_numpy(np.float32, (B, H, W))
    up = F.interpolate(torch.from_numpy(x.astype(np.float32)).permute(0, 3, 1, 2), size=(H, W), mode="bilinear", align_corners=True)
    up = up.half().float()
    c2 = torch.relu(F.conv2d(up, torch.from_numpy(w2.astype(np.float32)).permute(0, 3, 1, 2), torch.from_numpy(b2), padding=1))
    want = scale * torch.relu((c2 * torch.from_numpy(w3).view(1, 32, 1, 1)).sum(1) + b3)
    want = want.numpy()
    assert np.isfinite(got).all()
    err = np.abs(got - want)
    assert err.max() < 6e-3 * max(1.0, float(np.abs(want).max())), (float(err.max()), float(np.abs(want).max()))
    assert err.mean() < 6e-4
    release()


def test_headconv_limits():
    lib = api()
    assert lib.vx_headconv_supported(32, 32, 518, 518, 296, 296) == 1
    assert lib.vx_headconv_supported(64, 32, 518, 518, 296, 296) == 0   # the kernel holds a 32-channel 3x3 kernel in registers
    assert lib.vx_headconv_supported(32, 32, 518, 518, 400, 400) == 0   # scale 0.77: the halo's source does not fit the 12 x 21 patch
    assert lib.vx_headconv_supported(32, 32, 700, 518, 400, 296) == 1   # the 640 x 480 request's model extent
    x = empty(64)
    assert lib.vx_headconv_bil_f16(x.ptr, x.ptr, x.ptr, x.ptr, 0.0, 1.0, x.ptr, 1, 518, 518, 400, 400, None) == 0
    assert b"source patch" in lib.vx_last_error()
    release()


def _rcu_oracle(x, w1, b1, w2, b2, res2=None, wp=None, bp=None):
    """dpt::residual_conv (depth-anything.cpp:15-23) [+ feature_fusion's x0 (:28-31)] [+ the 1x1 out_conv (:39)], intermediates rounded to f16
    where the device stores f16."""
    mid = _h(np.maximum(oracle.conv2d_nhwc(np.maximum(x, 0), w1, b1, 1, 1), 0))
    y = _h(_h(oracle.conv2d_nhwc(mid, w2, b2, 1, 1)) + x)
    if res2 is not None:
        y = _h(y + res2)
    if wp is not None:
        y = oracle.conv2d_nhwc(y, wp, bp, 1, 0)
    return y


@pytest.mark.parametrize("hw", [(19, 19), (37, 37), (74, 74), (50, 37), (8, 5), (1, 1), (96, 21)])
@pytest.mark.parametrize("mode", ["rcu", "rcu+res", "rcu+proj", "rcu+res+proj"])
def test_rcu_fused(hw, mode):
    """One launch per residual unit (kernels_rcu.hip): relu -> conv3x3 -> relu -> conv3x3 -> + x [+ x0] [-> conv1x1], the intermediate map in LDS,
    against the oracle's convs -- map extents that tile evenly, raggedly (edge tiles beyond the map), and maps smaller than one tile."""
    rng = np.random.default_rng(hw[0] * 131 + hw[1] + len(mode))
    B, (Hh, Ww), Cc = 3, hw, 64
    x = _h(_rand(rng, B, Hh, Ww, Cc))
    w1, b1 = _h(_rand(rng, Cc, 3, 3, Cc, scale=(9 * Cc) ** -0.5)), _rand(rng, Cc, scale=0.1)
    w2, b2 = _h(_rand(rng, Cc, 3, 3, Cc, scale=(9 * Cc) ** -0.5)), _rand(rng, Cc, scale=0.1)
    res2 = _h(_rand(rng, B, Hh, Ww, Cc)) if "res" in mode else None
    wp, bp = (_h(_rand(rng, Cc, 1, 1, Cc, scale=Cc ** -0.5)), _rand(rng, Cc, scale=0.1)) if "proj" in mode else (None, None)
    assert api().vx_rcu_supported(Hh, Ww) == 1
    keep = [dev(x.astype(np.float16)), dev(pack_rcu(w1)), dev(b1), dev(pack_rcu(w2)), dev(b2)]
    a = L.RcuArgs()
    a.x, a.w1, a.b1, a.w2, a.b2 = (k.ptr for k in keep)
    if res2 is not None:
        keep.append(dev(res2.astype(np.float16)))
        a.res2 = keep[-1].ptr
    if wp is not None:
        keep += [dev(pack_rcu(wp)), dev(bp)]
        a.wp, a.bp = keep[-2].ptr, keep[-1].ptr
    out = empty(B * Hh * Ww * Cc * 2)
    a.out, a.B, a.H, a.W = out.ptr, B, Hh, Ww
    L.vx_check(api().vx_rcu_fused_f16(C.byref(a), None))
    sync()
    got = out.to_numpy(np.float16, (B, Hh, Ww, Cc)).astype(np.float32)
    want = _rcu_oracle(x, w1, b1, w2, b2, res2, wp, bp)
    assert rel_err(got, want) < F16_TOL


def test_rcu_fused_limits():
    assert api().vx_rcu_supported(148, 148) == 0 and api().vx_rcu_supported(0, 5) == 0
    a = L.RcuArgs()
    a.B, a.H, a.W = 1, 148, 148
    assert api().vx_rcu_fused_f16(C.byref(a), None) == 0 and b"not built" in api().vx_last_error()
