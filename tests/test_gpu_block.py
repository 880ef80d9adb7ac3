"""-m gpu: the token-stationary DINOv2 block kernel (csrc/kernels_block16.hip, vx_dino_block16_f16) against the CPU oracle's
linear / layer_norm / gelu on the same seeded inputs, through the vx_* C ABI. Every output the kernel produces is
compared: the residual stream after the attention half (capture), after the MLP half, the tapped LayerNorm rows and the
next layer's head-major q / k / v. Tolerances are relative to the largest reference magnitude of each tensor."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle
from visioncpp_amd import _lib as L

pytestmark = pytest.mark.gpu

from gpu_util import api, dev, empty, rel_err, release, sync  # noqa: E402

D, HID, H = 384, 1536, 6


@pytest.fixture(scope="module", autouse=True)
def _device():
    a = L.get_lib()
    assert a.vx_device_count() > 0, "no HIP device visible: the product path has no CPU fallback"
    L.vx_check(a.vx_set_device(0))
    yield


@pytest.fixture(autouse=True)
def _release_buffers():
    yield
    release()


def _h(a):
    return a.astype(np.float16).astype(np.float32)


def make_weights(seed, lam=0.1):
    rng = np.random.default_rng(seed)
    r = lambda *s, scale=1.0: (rng.standard_normal(s) * scale).astype(np.float32)  # noqa: E731
    w = dict(
        wo=_h(r(D, D, scale=D ** -0.5)), bo=r(D, scale=0.1), lam1=np.full(D, lam, np.float32) + r(D, scale=0.01),
        g2=1 + r(D, scale=0.05), b2=r(D, scale=0.05),
        w1=_h(r(HID, D, scale=D ** -0.5)), b1=r(HID, scale=0.1), w2=_h(r(D, HID, scale=HID ** -0.5)), bfc2=r(D, scale=0.1),
        lam2=np.full(D, lam, np.float32) + r(D, scale=0.01),
        gn=1 + r(D, scale=0.05), bn=r(D, scale=0.05), wqkv=_h(r(3 * D, D, scale=D ** -0.5)), bqkv=r(3 * D, scale=0.1),
        gf=1 + r(D, scale=0.05), bf=r(D, scale=0.05),
    )
    return w


def pack(w, fn=None):
    a = api()
    pack_mlp, pack_qkv = (fn.pack_mlp, fn.pack_qkv) if fn is not None else (a.vx_dino_block16_pack_mlp, a.vx_dino_block16_pack_qkv)
    mlp = np.zeros(a.vx_dino_block_mlp_bytes() // 2, np.uint16)
    qkv = np.zeros(a.vx_dino_block_qkv_bytes() // 2, np.uint16)
    f16 = lambda m: np.ascontiguousarray(m.astype(np.float16))  # noqa: E731
    fold = fn is None or getattr(fn, "folded", True)  # the kernel takes LayerScale folded into Wo / W2 and their biases
    l1, l2 = (w["lam1"], w["lam2"]) if fold else (np.ones(D, np.float32), np.ones(D, np.float32))
    wo, w1, w2, wq = f16(l1[:, None] * w["wo"]), f16(w["w1"]), f16(l2[:, None] * w["w2"]), f16(w["wqkv"])
    L.vx_check(pack_mlp(wo.ctypes.data, w1.ctypes.data, w2.ctypes.data, mlp.ctypes.data))
    L.vx_check(pack_qkv(wq.ctypes.data, qkv.ctypes.data))
    vec_mlp = np.concatenate([l1 * w["bo"], w["lam1"], w["g2"], w["b2"], w["b1"], l2 * w["bfc2"], w["lam2"]]).astype(np.float32)
    vec_qkv = np.concatenate([w["gn"], w["bn"], w["bqkv"]]).astype(np.float32)
    vec_tap = np.concatenate([w["gf"], w["bf"]]).astype(np.float32)
    assert vec_mlp.size == 3840 and vec_qkv.size == 1920 and vec_tap.size == 768
    return dev(mlp), dev(qkv), dev(vec_mlp), dev(vec_qkv), dev(vec_tap)


def reference(w, x, att, eps, mlp, tap, qkv, T, q_scale):
    """dino.cpp:48-90 with the device's rounding points: LayerNorm rows and the hidden activations are rounded to f16
    before the next product (they are MFMA operands), everything else stays f32."""
    out = {}
    if mlp:
        x = x + w["lam1"] * oracle.linear(att, w["wo"], w["bo"])
        out["x1"] = x.copy()
        ln = _h(oracle.layer_norm(x, w["g2"], w["b2"], eps))
        hid = _h(oracle.gelu(oracle.linear(ln, w["w1"], w["b1"]), oracle.GELU_TANH_F32))
        x = x + w["lam2"] * oracle.linear(hid, w["w2"], w["bfc2"])
    out["x"] = x
    if tap:
        out["feat"] = oracle.layer_norm(x, w["gf"], w["bf"], eps)
    if qkv:
        ln = _h(oracle.layer_norm(x, w["gn"], w["bn"], eps))
        y = oracle.linear(ln, w["wqkv"], w["bqkv"])  # [M, 3D]
        B = x.shape[0] // T
        for i, name in enumerate("qkv"):
            t = y[:, i * D:(i + 1) * D].reshape(B, T, H, 64).transpose(0, 2, 1, 3)  # head-major [B, H, T, 64]
            out[name] = t * (q_scale if i == 0 else 1.0)
    return out


@pytest.fixture(params=["vx_dino_block16"], ids=["16-token"])
def block_fn(request):
    """The kernel (kernels_block16.hip: 16 tokens per wave, two waves per SIMD) with its weight packers; the launch carries them along
    for pack(). (The 32-token first form was removed in round 4; its numbers are in profiles/r02_block_kernel_anatomy.txt.)"""
    a = api()
    fn = getattr(a, request.param + "_f16")
    fn.pack_mlp, fn.pack_qkv = getattr(a, request.param + "_pack_mlp"), getattr(a, request.param + "_pack_qkv")
    fn.folded = request.param.endswith("16")
    return fn


@pytest.mark.parametrize("M,T,mlp,tap,qkv", [
    (128, 64, True, True, True),       # one full workgroup
    (300, 75, True, False, True),      # tail workgroup: 44 valid rows, two waves with none; images straddle workgroups
    (130, 65, True, True, False),      # last layer: tap, no next QKV
    (257, 257, False, False, True),    # first layer: LN1 + QKV only
    (200, 50, True, False, False),     # MLP half alone (no tap, no next layer)
    (16, 16, True, True, True),        # a single wave's worth of rows
    (1370 * 2, 1370, True, True, True),  # two images of the north-star grid
])
def test_block_vs_oracle(block_fn, M, T, mlp, tap, qkv):
    rng = np.random.default_rng(M * 7 + T)
    w = make_weights(M + 1)
    x0 = (rng.standard_normal((M, D)) * 1.5 + rng.standard_normal((1, D)) * 0.5).astype(np.float32)
    att = _h(rng.standard_normal((M, D)).astype(np.float32))
    eps, q_scale = 1e-6, 0.125
    want = reference(w, x0, att, eps, mlp, tap, qkv, T, q_scale)

    d_mlp, d_qkv, v_mlp, v_qkv, v_tap = pack(w, block_fn)
    xd = dev(x0)
    attd = dev(att.astype(np.float16))
    cap = empty(M * D * 4)
    feat = empty(M * D * 2)
    q, k, v = (empty(M * D * 2) for _ in range(3))
    a = L.DinoBlockArgs()
    a.x, a.M, a.T, a.H, a.q_scale, a.eps = xd.ptr, M, T, H, q_scale, eps
    if mlp:
        a.att, a.w_mlp, a.vec_mlp, a.cap_x1 = attd.ptr, d_mlp.ptr, v_mlp.ptr, cap.ptr
    if tap:
        a.feat, a.vec_tap = feat.ptr, v_tap.ptr
    if qkv:
        a.q, a.k, a.v, a.w_qkv, a.vec_qkv = q.ptr, k.ptr, v.ptr, d_qkv.ptr, v_qkv.ptr
    L.vx_check(block_fn(C.byref(a), None))
    sync()

    got_x = xd.to_numpy(np.float32, (M, D))
    # residual stream: f32, the branch contributions went through f16 operands (2^-11 relative each)
    if mlp:
        assert rel_err(cap.to_numpy(np.float32, (M, D)), want["x1"]) < 1e-3
    assert rel_err(got_x, want["x"]) < 1e-3
    if not mlp:
        assert np.array_equal(got_x, x0), "the QKV-only instance must not touch the residual stream"
    if tap:
        assert rel_err(feat.to_numpy(np.float16, (M, D)).astype(np.float32), want["feat"]) < 2e-3
    if qkv:
        B = M // T
        for name, buf in zip("qkv", (q, k, v)):
            got = buf.to_numpy(np.float16, (B, H, T, 64)).astype(np.float32)
            assert rel_err(got, want[name]) < 4e-3, name


def _run_block(block_fn, w, x0, att, T, q_scale=0.125, eps=1e-6):
    M = x0.shape[0]
    d_mlp, d_qkv, v_mlp, v_qkv, v_tap = pack(w, block_fn)
    xd, attd = dev(x0), dev(att.astype(np.float16))
    cap, feat = empty(M * D * 4), empty(M * D * 2)
    q, k, v = (empty(M * D * 2) for _ in range(3))
    a = L.DinoBlockArgs()
    a.x, a.M, a.T, a.H, a.q_scale, a.eps = xd.ptr, M, T, H, q_scale, eps
    a.att, a.w_mlp, a.vec_mlp, a.cap_x1 = attd.ptr, d_mlp.ptr, v_mlp.ptr, cap.ptr
    a.feat, a.vec_tap = feat.ptr, v_tap.ptr
    a.q, a.k, a.v, a.w_qkv, a.vec_qkv = q.ptr, k.ptr, v.ptr, d_qkv.ptr, v_qkv.ptr
    L.vx_check(block_fn(C.byref(a), None))
    sync()
    B = M // T
    return dict(x=xd.to_numpy(np.float32, (M, D)), x1=cap.to_numpy(np.float32, (M, D)), feat=feat.to_numpy(np.float16, (M, D)).astype(np.float32),
                q=q.to_numpy(np.float16, (B, H, T, 64)).astype(np.float32), k=k.to_numpy(np.float16, (B, H, T, 64)).astype(np.float32),
                v=v.to_numpy(np.float16, (B, H, T, 64)).astype(np.float32))


def test_block_layerscale_spread(block_fn):
    """Trained DINOv2 LayerScale vectors span orders of magnitude (init 1e-5 .. 1; the synthetic files use 0.1 +- 0.01). The 16-token
    form folds lambda into f16-rounded weights (Wo' = f16(f16(Wo) lambda1), W2' likewise, csrc/depthany.cpp), so channels with a tiny
    lambda sit in f16's subnormal range and channels with a large one scale their rounding error up: lambda log-uniform in
    [1e-5, 4] per channel plus outliers at 1e-7, 8 and exactly 0, against the f32 oracle with the bounds of test_block_vs_oracle
    (errors relative to the largest magnitude of the tensor: a channel's folded rounding error is 2^-11 of ITS contribution)."""
    rng = np.random.default_rng(99)
    M, T = 384, 64
    w = make_weights(31)
    for name in ("lam1", "lam2"):
        lam = np.exp(rng.uniform(np.log(1e-5), np.log(4.0), D)).astype(np.float32)
        lam[[3, 77, 200]] = [1e-7, 8.0, 0.0]
        lam[rng.integers(0, D, 20)] *= -1.0  # (signs occur in trained vectors too)
        w[name] = lam
    x0 = (rng.standard_normal((M, D)) * 1.5).astype(np.float32)
    att = _h(rng.standard_normal((M, D)).astype(np.float32))
    want = reference(w, x0, att, 1e-6, True, True, True, T, 0.125)
    got = _run_block(block_fn, w, x0, att, T)
    for n in got:
        assert np.isfinite(got[n]).all(), n
    assert rel_err(got["x1"], want["x1"]) < 1e-3 and rel_err(got["x"], want["x"]) < 1e-3
    # per channel too: a channel with a small lambda must not inherit the error scale of the large ones. Its folded weights are f16
    # subnormals (W lambda ~ 5e-6 at lambda = 1e-4: about 1 % precision each, NOT flushed by the MFMA), so its branch contribution
    # (~3e-4) carries ~2 % error = 7e-6 absolute on a residual stream of magnitude ~5: floor 2e-5, two orders below the 2^-11
    # operand rounding of the ordinary channels' contributions
    err = np.abs(got["x"] - want["x"]).max(axis=0)
    scale = np.abs(want["x"] - x0).max(axis=0)
    bad = np.nonzero(~(err < 4e-3 * scale + 2e-5))[0]
    assert bad.size == 0, [(int(c), float(err[c]), float(scale[c]), float(w['lam1'][c]), float(w['lam2'][c])) for c in bad[:8]]
    assert rel_err(got["feat"], want["feat"]) < 2e-3
    for n in "qkv":
        assert rel_err(got[n], want[n]) < 4e-3, n


def test_block_large_hidden_preactivations(block_fn):
    """The 16-token form evaluates GELU on packed f16 (v_pk_*_f16, v_exp_f16, v_rcp_f16): x^2 overflows f16 beyond |x| = 255 and x
    itself beyond 65504. Hidden units whose pre-activation sits at +-300, at -7e4 (f16: -inf before the clamp; gelu = 0) and rows
    of moderately large values must come out as the f32 oracle's GELU rounded to f16. (+7e4 is +inf in any f16 hidden map, here as in
    the GEMM schedule's f16 store: not representable, not tested.)"""
    rng = np.random.default_rng(7)
    M, T = 256, 64
    w = make_weights(41)
    big = {5: 300.0, 6: -300.0, 700: -7.0e4, 701: 260.0, 1535: -1000.0, 1000: 250.0}
    for j, v in big.items():
        w["b1"][j] = v
    w["w1"][[5, 6, 700, 701, 1535]] *= 0.0  # these units' pre-activation is their bias exactly
    w["w1"][100:140] *= 40.0                # and forty units swing through +-100 with the input
    x0 = (rng.standard_normal((M, D)) * 1.5).astype(np.float32)
    att = _h(rng.standard_normal((M, D)).astype(np.float32))
    want = reference(w, x0, att, 1e-6, True, True, True, T, 0.125)
    got = _run_block(block_fn, w, x0, att, T)
    for n in got:
        assert np.isfinite(got[n]).all(), n
    assert rel_err(got["x"], want["x"]) < 1.5e-3
    assert rel_err(got["feat"], want["feat"]) < 3e-3
    for n in "qkv":
        assert rel_err(got[n], want[n]) < 5e-3, n


def test_block_rows_are_independent(block_fn):
    """A row's results do not depend on which workgroup / wave / lane processes it: the same rows placed at another
    offset of a larger problem give bit-identical outputs (the property the batch sharding relies on)."""
    M1, M2, T = 192, 448, 64
    rng = np.random.default_rng(5)
    w = make_weights(9)
    x = (rng.standard_normal((M2, D)) * 1.5).astype(np.float32)
    att = rng.standard_normal((M2, D)).astype(np.float16)
    d_mlp, d_qkv, v_mlp, v_qkv, v_tap = pack(w, block_fn)

    def run(xs, atts):
        M = xs.shape[0]
        xd, ad, feat = dev(xs), dev(atts), empty(M * D * 2)
        q, k, v = (empty(M * D * 2) for _ in range(3))
        a = L.DinoBlockArgs()
        a.x, a.M, a.T, a.H, a.q_scale, a.eps = xd.ptr, M, T, H, 0.125, 1e-6
        a.att, a.w_mlp, a.vec_mlp = ad.ptr, d_mlp.ptr, v_mlp.ptr
        a.feat, a.vec_tap = feat.ptr, v_tap.ptr
        a.q, a.k, a.v, a.w_qkv, a.vec_qkv = q.ptr, k.ptr, v.ptr, d_qkv.ptr, v_qkv.ptr
        L.vx_check(block_fn(C.byref(a), None))
        sync()
        return xd.to_numpy(np.float32, (M, D)), feat.to_numpy(np.uint16, (M, D)), q.to_numpy(np.uint16, (M // T, H, T, 64))

    xa, fa, qa = run(x, att)
    xb, fb, qb = run(x[256:256 + M1], att[256:256 + M1])
    assert np.array_equal(xa[256:256 + M1], xb) and np.array_equal(fa[256:256 + M1], fb)
    assert np.array_equal(qa[4:7], qb)


def test_block_argument_errors():
    a = L.DinoBlockArgs()
    assert api().vx_dino_block16_f16(C.byref(a), None) == 0  # empty problem
    x = empty(128 * D * 4)
    a.x, a.M = x.ptr, 128
    assert api().vx_dino_block16_f16(C.byref(a), None) == 0 and b"nothing to do" in api().vx_last_error()
    assert api().vx_dino_block_supported(384, 1536, 64) == 1 and api().vx_dino_block_supported(768, 3072, 64) == 0
