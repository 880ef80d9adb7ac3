"""GPU parity of the TinyViT row (MobileSAM image encoder): the non-GEMM kernels against the CPU oracle, then (below)
the whole encoder through the C ABI against the oracle and the reference-torch fixture."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _release():
    yield
    from tests import gpu_util as G
    G.release()


def _h(a):
    return a.astype(np.float16).astype(np.float32)


@pytest.mark.parametrize("H,W,Cc,stride,gelu", [(16, 16, 64, 1, 1), (37, 21, 160, 1, 0), (32, 32, 128, 2, 1), (15, 9, 320, 2, 0)])
def test_depthwise_conv(H, W, Cc, stride, gelu):
    from tests import gpu_util as G
    from visioncpp_amd import _lib as L
    rng = np.random.default_rng(H * W + Cc)
    B = 2
    x = _h(rng.standard_normal((B, H, W, Cc)))
    w = _h(rng.standard_normal((3, 3, Cc)) / 3)
    b = rng.standard_normal(Cc).astype(np.float32) * 0.1
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = G.empty(B * OH * OW * Cc * 2)
    L.vx_check(G.api().vx_dwconv3x3_f16(G.dev(x.astype(np.float16)).ptr, G.dev(w.astype(np.float16)).ptr, G.dev(b).ptr, out.ptr, B, H, W, Cc, stride, gelu, None))
    G.sync()
    got = out.to_numpy(np.float16, (B, OH, OW, Cc)).astype(np.float32)
    for i in range(B):
        ref = O.conv2d_depthwise_nhwc(x[i], w, b, stride, 1)
        if gelu:
            ref = O.gelu(ref, O.GELU_TANH_F32)
        np.testing.assert_allclose(got[i], ref, atol=4e-3, rtol=4e-3)


@pytest.mark.parametrize("res,ws,Cc", [(14, 7, 128), (10, 4, 160), (16, 14, 320)])
def test_layernorm_window_partition(res, ws, Cc):
    """LayerNorm with the window partition folded into the output row order; padded positions hold the bias vector."""
    from tests import gpu_util as G
    from visioncpp_amd import _lib as L
    rng = np.random.default_rng(res + Cc)
    B = 2
    x = _h(rng.standard_normal((B, res, res, Cc)) * 2 + 0.3)
    w, b = (1 + 0.1 * rng.standard_normal(Cc)).astype(np.float32), (0.1 * rng.standard_normal(Cc)).astype(np.float32)
    nw = -(-res // ws)
    rows = B * nw * nw * ws * ws
    out = G.empty(rows * Cc * 2)
    L.vx_check(G.api().vx_layernorm_f16(G.dev(x.astype(np.float16)).ptr, G.dev(w).ptr, G.dev(b).ptr, out.ptr, rows, Cc, 1e-5, res, ws, 0, None))
    G.sync()
    got = out.to_numpy(np.float16, (B, nw, nw, ws, ws, Cc)).astype(np.float32)
    xp = np.zeros((B, nw * ws, nw * ws, Cc), np.float32)
    xp[:, :res, :res] = x
    ref = O.layer_norm(xp.reshape(-1, Cc), w, b, 1e-5).reshape(B, nw, ws, nw, ws, Cc).transpose(0, 1, 3, 2, 4, 5)
    np.testing.assert_allclose(got, ref, atol=4e-3, rtol=4e-3)
    # plain rows, f32 output
    out32 = G.empty(B * res * res * Cc * 4)
    L.vx_check(G.api().vx_layernorm_f16(G.dev(x.astype(np.float16)).ptr, G.dev(w).ptr, G.dev(b).ptr, out32.ptr, B * res * res, Cc, 1e-5, 0, 0, 1, None))
    G.sync()
    np.testing.assert_allclose(out32.to_numpy(np.float32, (B * res * res, Cc)), O.layer_norm(x.reshape(-1, Cc), w, b, 1e-5), atol=2e-5, rtol=2e-5)


@pytest.mark.parametrize("N,heads,nwin", [(49, 4, 5), (196, 5, 3), (16, 2, 7)])
def test_window_attention(N, heads, nwin):
    from tests import gpu_util as G
    from visioncpp_amd import _lib as L
    rng = np.random.default_rng(N + heads)
    dim = heads * 32
    qkv = _h(rng.standard_normal((nwin, N, 3 * dim)))
    bias = (rng.standard_normal((heads, N, N)) * 0.5).astype(np.float32)
    out = G.empty(nwin * N * dim * 2)
    L.vx_check(G.api().vx_window_attention_f16(G.dev(qkv.astype(np.float16)).ptr, G.dev(bias).ptr, out.ptr, nwin, N, heads, None))
    G.sync()
    got = out.to_numpy(np.float16, (nwin, N, dim)).astype(np.float32)
    q4 = qkv.reshape(nwin, N, heads, 3, 32)
    s = np.einsum("wihc,wjhc->whij", q4[..., 0, :], q4[..., 1, :]) / np.sqrt(32.0) + bias[None]
    p = np.exp(s - s.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    ref = np.einsum("whij,wjhc->wihc", p, q4[..., 2, :]).reshape(nwin, N, dim)
    np.testing.assert_allclose(got, ref, atol=3e-3, rtol=3e-3)


def test_window_reverse_add_and_add_gelu_and_preprocess():
    from tests import gpu_util as G
    from visioncpp_amd import _lib as L
    rng = np.random.default_rng(9)
    B, res, ws, Cc = 2, 10, 4, 64
    nw = 3
    a = _h(rng.standard_normal((B, nw, nw, ws, ws, Cc)))
    x = _h(rng.standard_normal((B, res, res, Cc)))
    out = G.empty(B * res * res * Cc * 2)
    L.vx_check(G.api().vx_window_reverse_add_f16(G.dev(a.astype(np.float16)).ptr, G.dev(x.astype(np.float16)).ptr, out.ptr, B, res, ws, Cc, None))
    G.sync()
    full = a.transpose(0, 1, 3, 2, 4, 5).reshape(B, nw * ws, nw * ws, Cc)[:, :res, :res]
    np.testing.assert_allclose(out.to_numpy(np.float16, (B, res, res, Cc)).astype(np.float32), _h(full + x), atol=2e-3)
    n = B * res * res * Cc
    out2 = G.empty(n * 2)
    L.vx_check(G.api().vx_add_gelu_f16(G.dev(x.astype(np.float16)).ptr, G.dev(full.astype(np.float16)).ptr, out2.ptr, n, None))
    G.sync()
    np.testing.assert_allclose(out2.to_numpy(np.float16, (B, res, res, Cc)).astype(np.float32), O.gelu(_h(full) + x, O.GELU_TANH_F32), atol=3e-3, rtol=3e-3)
    img = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)
    out3 = G.empty(35 * 8 * 2)
    L.vx_check(G.api().vx_tv_preprocess(G.dev(img).ptr, out3.ptr, 35, None))
    G.sync()
    got = out3.to_numpy(np.float16, (5, 7, 8)).astype(np.float32)
    want = (img.astype(np.float32) / np.float32(255.0) - np.array([0.485, 0.456, 0.406], np.float32)) / np.array([0.229, 0.224, 0.225], np.float32)
    np.testing.assert_allclose(got[..., :3] + got[..., 3:6], want, atol=2e-6)
    assert (got[..., 6:] == 0).all()
