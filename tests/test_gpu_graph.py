"""-m gpu: the graph executor (csrc/graph.cpp behind `visp_graph_*`) on the device. Every node kind against a
plain PyTorch fp32 evaluation of the same op on the same f16-rounded operands (the reference pins its ops the same way:
tests/test_primitives.py:20-184), then Depth-Anything-V2 built through the layer the way the reference's arch code builds it
(vision.cpp_amd/graph.py::depthany_predict) against the CPU oracle (MAE < 1e-3, the north star's tolerance) and against the
hand-scheduled step of csrc/depthany.cpp."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle
from visioncpp_amd import _lib as L
from visioncpp_amd import graph as G
from visioncpp_amd import synth, vision

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def device():
    return vision.Device.init(vision.Backend.gpu)


def h(a):  # round to f16 like every activation / matrix weight of the executor
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


def rel(got, want):
    want = np.asarray(want, np.float64)
    return float(np.abs(np.asarray(got, np.float64) - want).max() / max(float(np.abs(want).max()), 1e-30))


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def run(g, feeds, outs):
    g.allocate()
    for k, v in feeds.items():
        g.set(k, v)
    g.compute()
    return [g.get(o) for o in outs]


def test_linear_layer_norm_gelu_mul_add(device):
    """One pre-LN MLP block (dino.cpp:48-56, 80-87): layer_norm -> linear + gelu (fused) -> linear -> layer scale -> residual."""
    rng = np.random.default_rng(1)
    C, Hd, T, B = 128, 512, 70, 3
    x = h(rng.standard_normal((B, T, C)))
    w = {"norm.weight": 1 + 0.1 * rng.standard_normal(C), "norm.bias": 0.1 * rng.standard_normal(C), "fc1.weight": h(0.05 * rng.standard_normal((Hd, C))),
         "fc1.bias": 0.05 * rng.standard_normal(Hd), "fc2.weight": h(0.05 * rng.standard_normal((C, Hd))), "fc2.bias": 0.05 * rng.standard_normal(C),
         "ls.lambda1": 0.3 + 0.1 * rng.standard_normal(C)}
    g = G.Graph(device)
    for k, v in w.items():
        g.add_weight(k, v, G.F16 if v.ndim == 2 else G.F32)
    m = G.ModelRef(g)
    xi = g.input((C, T, B), G.F16)
    ln = G.layer_norm(m["norm"], xi, 1e-6)
    hid = G.gelu(m, G.linear(m["fc1"], ln))
    y = G.add(m, xi, G.mul(m, G.linear(m["fc2"], hid), m["ls"].weights("lambda1")))
    g.output(ln, "ln"); g.output(y, "y")
    got_ln, got_y = run(g, {xi: x}, [ln, y])
    tw = {k: t(v) for k, v in w.items()}
    ref_ln = F.layer_norm(t(x), (C,), tw["norm.weight"], tw["norm.bias"], 1e-6)
    assert rel(got_ln, ref_ln.numpy()) < 2e-3
    ref_h = F.gelu(F.linear(t(h(ref_ln.numpy())), tw["fc1.weight"], tw["fc1.bias"]), approximate="tanh")
    ref_y = t(x) + F.linear(t(h(ref_h.numpy())), tw["fc2.weight"], tw["fc2.bias"]) * tw["ls.lambda1"]
    assert rel(got_y, ref_y.numpy()) < 3e-3
    lines = g.describe().splitlines()
    assert lines[1].startswith("gemm[gelu]") and lines[2].startswith("gemm[*scale][+res]")  # layer scale folded into fc2, residual in its epilogue


@pytest.mark.parametrize("T,B,heads", [(50, 2, 2), (257, 1, 6), (1370, 1, 6)])
def test_attention_node(device, T, B, heads):
    """split heads (reshape views) -> attention -> output linear (nn.cpp:210-244, dino.cpp:59-74) vs torch softmax(QK^T / 8) V."""
    rng = np.random.default_rng(T)
    C = heads * 64
    x = h(rng.standard_normal((B, T, C)))
    ws = {n: h(rng.standard_normal((C, C)) / math.sqrt(C)) for n in ("query", "key", "value", "dense")}
    bs = {n: 0.1 * rng.standard_normal(C) for n in ws}
    g = G.Graph(device)
    for n in ws:
        g.add_weight(f"{n}.weight", ws[n]); g.add_weight(f"{n}.bias", bs[n], G.F32)
    m = G.ModelRef(g)
    xi = g.input((C, T, B), G.F16)
    q, k, v = (G.reshape(m, G.linear(m[n], xi), 64, heads, T, B) for n in ("query", "key", "value"))
    y = g.output(G.attention(m, q, k, v, None, 1 / 8, m["dense"]), "y")
    (got,) = run(g, {xi: x}, [y])
    assert g.describe().splitlines()[0].startswith("gemm[qkv heads-major]")  # the three projections of one input: one launch
    # the same attention with q also read by an output: the projections stay three products + three head-major copies
    g2 = G.Graph(device)
    for n in ws:
        g2.add_weight(f"{n}.weight", ws[n]); g2.add_weight(f"{n}.bias", bs[n], G.F32)
    m2 = G.ModelRef(g2)
    xi2 = g2.input((C, T, B), G.F16)
    lq = G.linear(m2["query"], xi2)
    g2.output(lq, "q")
    q2, k2, v2 = G.reshape(m2, lq, 64, heads, T, B), *(G.reshape(m2, G.linear(m2[n], xi2), 64, heads, T, B) for n in ("key", "value"))
    y2 = g2.output(G.attention(m2, q2, k2, v2, None, 1 / 8, m2["dense"]), "y")
    (got2,) = run(g2, {xi2: x}, [y2])
    assert g2.describe().count("heads_major") == 3
    assert rel(got2, got) < 2e-3
    tq, tk, tv = (t(h(F.linear(t(x), t(ws[n]), t(bs[n])).numpy())).reshape(B, T, heads, 64).transpose(1, 2) for n in ("query", "key", "value"))
    att = torch.softmax(tq @ tk.transpose(-1, -2) / 8, -1) @ tv
    ref = F.linear(t(h(att.transpose(1, 2).reshape(B, T, C).numpy())), t(ws["dense"]), t(bs["dense"]))
    assert rel(got, ref.numpy()) < 4e-3


@pytest.mark.parametrize("cin,cout,k,stride,pad,H,W", [(64, 64, 3, 1, 1, 19, 23), (48, 64, 3, 1, 1, 30, 26), (96, 96, 3, 2, 1, 37, 37), (64, 32, 3, 1, 1, 100, 120),
                                                         (384, 48, 1, 1, 0, 12, 9), (32, 32, 3, 1, 1, 112, 98), (64, 8, 5, 1, 2, 21, 17)])
def test_conv_2d_node(device, cin, cout, k, stride, pad, H, W):
    """CWHN conv_2d (nn.cpp:72-100) in every form the lowering picks: 1x1 as a plain product, implicit GEMM (any kernel / stride / a Cin whose
    9-tap rows are not a multiple of the 64-wide k tile), the halo-in-LDS 3x3 kernel from 96 px wide -- vs torch conv2d."""
    rng = np.random.default_rng(cin + cout + k)
    B = 2
    x = h(rng.standard_normal((B, H, W, cin)))
    w = h(rng.standard_normal((cout, k, k, cin)) / math.sqrt(k * k * cin))  # OHWI
    b = 0.1 * rng.standard_normal(cout)
    g = G.Graph(device)
    g.add_weight("c.weight", w); g.add_weight("c.bias", b, G.F32)
    m = G.ModelRef(g)
    xi = g.input((cin, W, H, B), G.F16)
    y = g.output(G.conv_2d(m["c"], xi, stride, pad), "y")
    (got,) = run(g, {xi: x}, [y])
    ref = F.conv2d(t(x).permute(0, 3, 1, 2), t(w).permute(0, 3, 1, 2), t(b), stride=stride, padding=pad).permute(0, 2, 3, 1)
    assert got.shape == tuple(ref.shape)
    assert rel(got, ref.numpy()) < 2e-3


@pytest.mark.parametrize("hw", [(37, 37), (148, 40)])
def test_residual_conv_unit_fuses_and_matches(device, hw):
    """dpt::residual_conv (depth-anything.cpp:15-23) with feature_fusion's outer add (:28-31) and the 1x1 projection BEFORE the bilinear
    align_corners resize it commutes with (:36-40). On a map of at most 96 x 96: one launch per residual unit (kernels_rcu.hip; the second one
    applies the projection), three launches for the stage. On a larger one: two launches of the LDS-ring conv per unit ([relu-in][relu], then
    [+res], the outer add as a second residual of the same epilogue), six for the stage."""
    rng = np.random.default_rng(5)
    C, (H, W), B = 64, hw, 2
    x0, x1 = h(rng.standard_normal((B, H, W, C))), h(rng.standard_normal((B, H, W, C)))
    names = ["residual_layer1.convolution1", "residual_layer1.convolution2", "residual_layer2.convolution1", "residual_layer2.convolution2"]
    w = {n: h(rng.standard_normal((C, 3, 3, C)) / math.sqrt(9 * C)) for n in names}
    b = {n: 0.1 * rng.standard_normal(C) for n in names}
    wp, bp = h(rng.standard_normal((C, 1, 1, C)) / 8), 0.1 * rng.standard_normal(C)
    g = G.Graph(device)
    for n in names:
        g.add_weight(f"f.{n}.weight", w[n]); g.add_weight(f"f.{n}.bias", b[n], G.F32)
    g.add_weight("f.projection.weight", wp); g.add_weight("f.projection.bias", bp, G.F32)
    m = G.ModelRef(g)
    a, c = g.input((C, W, H, B), G.F16, "x0"), g.input((C, W, H, B), G.F16, "x1")
    y = g.output(G.dpt_feature_fusion(m["f"], a, c, (2 * W, 2 * H)), "y")
    (got,) = run(g, {a: x0, c: x1}, [y])
    d = g.describe()
    if max(hw) <= 96:
        assert d.count("residual_unit[relu, conv3x3, relu, conv3x3, + x, + x0]") == 1 and d.count("residual_unit[relu, conv3x3, relu, conv3x3, + x, conv1x1 before its resize]") == 1
        assert "gemm(conv1x1 before its resize)" not in d and f"bilinear_ac {W}x{H} -> {2 * W}x{2 * H} C=64" in d and "launches=3" in d
    else:
        assert d.count("dconv3x3[relu-in][relu]") == 2 and d.count("dconv3x3[+res][+res]") == 1 and d.count("dconv3x3[+res] M=") == 1
        assert "gemm(conv1x1 before its resize)" in d and f"bilinear_ac {W}x{H} -> {2 * W}x{2 * H} C=64" in d and "launches=6" in d

    def conv(v, n):
        return F.conv2d(v, t(w[n]).permute(0, 3, 1, 2), t(b[n]), padding=1)

    def rcu(v, p):
        o = t(h(torch.relu(conv(torch.relu(v), f"{p}.convolution1")).numpy()))
        return t(h((v + conv(o, f"{p}.convolution2")).numpy()))

    v0, v1 = t(x0).permute(0, 3, 1, 2), t(x1).permute(0, 3, 1, 2)
    r = rcu(t(h((v0 + rcu(v1, "residual_layer1")).numpy())), "residual_layer2")
    up = t(h(F.interpolate(r, size=(2 * H, 2 * W), mode="bilinear", align_corners=True).numpy()))
    ref = F.conv2d(up, t(wp).permute(0, 3, 1, 2), t(bp)).permute(0, 2, 3, 1)
    assert rel(got, ref.numpy()) < 4e-3


@pytest.mark.parametrize("cin,cout,s", [(48, 48, 4), (96, 96, 2), (64, 16, 2)])
def test_conv_transpose_2d_node(device, cin, cout, s):
    """kernel == stride transposed conv (nn.cpp:117-129) as a product + pixel shuffle; Cin = 48 / 96 take the zero-padded row copy."""
    rng = np.random.default_rng(cin + s)
    B, H, W = 2, 9, 11
    x = h(rng.standard_normal((B, H, W, cin)))
    w = h(rng.standard_normal((cin, cout, s, s)) / math.sqrt(cin))  # torch layout = ggml ne [kw, kh, Cout, Cin]
    b = 0.1 * rng.standard_normal(cout)
    g = G.Graph(device)
    g.add_weight("t.weight", w); g.add_weight("t.bias", b, G.F32)
    xi = g.input((cin, W, H, B), G.F16)
    y = g.output(G.conv_transpose_2d(G.ModelRef(g)["t"], xi, s), "y")
    (got,) = run(g, {xi: x}, [y])
    ref = F.conv_transpose2d(t(x).permute(0, 3, 1, 2), t(w), t(b), stride=s).permute(0, 2, 3, 1)
    assert rel(got, ref.numpy()) < 2e-3
    assert ("pad_rows" in g.describe()) == (cin % 64 != 0)


def test_slice_concat_repeat_patch_embed(device):
    """dino::prepare_tokens (dino.cpp:32-46): patch_embed on the f32 image tensor, cls token repeated over the batch and concatenated
    in front, position embeddings added; then the neck's cls-token slice (depth-anything.cpp:50-51)."""
    rng = np.random.default_rng(9)
    D, ps, pw, ph, B = 64, 14, 4, 4, 3
    img = rng.standard_normal((B, ph * ps, pw * ps, 3)).astype(np.float32)
    w = h(rng.standard_normal((D, ps, ps, 3)) / math.sqrt(ps * ps * 3))
    b = 0.1 * rng.standard_normal(D)
    cls = rng.standard_normal((1, 1, D)).astype(np.float32)
    pos = rng.standard_normal((1, 1 + pw * ph, D)).astype(np.float32)
    g = G.Graph(device)
    g.add_weight("e.patch_embeddings.projection.weight", w); g.add_weight("e.patch_embeddings.projection.bias", b, G.F32)
    g.add_weight("e.cls_token", cls, G.F32); g.add_weight("e.position_embeddings", pos, G.F32)
    m = G.ModelRef(g)
    xi = g.input((3, pw * ps, ph * ps, B), G.F32, "image")
    tok = G.dino_prepare_tokens(m["e"], xi, ps)
    sl = G.slice_(m, tok, G.SLICE_ALL, (1, tok.ne[1]))
    every_other = G.slice_(m, tok, (8, 40, 1), (0, tok.ne[1], 2))  # begin / end / step on two dimensions: the generic strided copy
    g.output(tok, "tok"); g.output(sl, "sl"); g.output(every_other, "eo")
    got_tok, got_sl, got_eo = run(g, {xi: img}, [tok, sl, every_other])
    pe = F.conv2d(t(h(img)).permute(0, 3, 1, 2), t(w).permute(0, 3, 1, 2), t(b), stride=ps).permute(0, 2, 3, 1).reshape(B, pw * ph, D)
    ref = torch.cat([t(h(cls)).expand(B, 1, D), t(h(pe.numpy()))], 1) + t(pos)
    assert got_tok.shape == (1, B, 1 + pw * ph, D)
    assert rel(got_tok[0], ref.numpy()) < 2e-3
    np.testing.assert_array_equal(got_sl[0], got_tok[0][:, 1:])
    np.testing.assert_array_equal(got_eo[0], got_tok[0][:, 0::2, 8:40])


def test_one_channel_head_relu_scale(device):
    """head.conv3 (1x1, 32 -> 1) + ReLU + max_depth scale (depth-anything.cpp:91-95): one f32 launch."""
    rng = np.random.default_rng(3)
    C, H, W, B = 32, 20, 30, 2
    x = h(rng.standard_normal((B, H, W, C)))
    w, b = h(rng.standard_normal((1, 1, 1, C)) / 4), np.array([0.2], np.float32)
    g = G.Graph(device)
    g.add_weight("conv3.weight", w); g.add_weight("conv3.bias", b, G.F32)
    m = G.ModelRef(g)
    xi = g.input((C, W, H, B), G.F16)
    y = g.output(G.scale(m, G.relu(m, G.conv_2d(m["conv3"], xi)), 20.0), "depth")
    (got,) = run(g, {xi: x}, [y])
    assert y.dtype == G.F32 and g.summary()["launches"] == 1
    ref = 20.0 * torch.relu((t(x) * t(w).reshape(C)).sum(-1) + 0.2)
    np.testing.assert_allclose(got[..., 0], ref.numpy(), rtol=1e-5, atol=1e-5)


def _oracle(cfg, seed):
    sd = synth.state_dict(cfg, seed)
    tensors, conv2d = synth.gguf_tensors(sd)
    om = oracle.Model(tensors, conv2d, "whcn")
    params = oracle.make_params(cfg.patch_size, cfg.embed_dim, cfg.n_layers, cfg.n_heads, cfg.image_size, 14, cfg.feature_layers, 1.0, oracle.GELU_GGML_F16_LUT)
    return om, params


def _pre(img):
    return oracle.image_u8_to_f32(img, oracle.RGB_U8, oracle.RGB_F32, (-0.485, -0.456, -0.406, 0), (1 / 0.229, 1 / 0.224, 1 / 0.225, 1))


def _norm(d):
    lo, hi = d.min(), d.max()
    return (d - lo) / (hi - lo)


@pytest.mark.parametrize("layout", ["whcn", "cwhn"])
def test_depth_anything_through_the_graph_layer(device, tmp_path, layout):
    """The whole model built node by node (graph.py::depthany_predict = the structure of dino.cpp + depth-anything.cpp) from a GGUF of
    either tensor layout, against the CPU oracle at every tapped boundary and at the output (MAE < 1e-3 on the normalised depth), and
    against the hand-scheduled step. 700 x 518: the position embeddings are bicubic-resized (folded on the host)."""
    cfg = synth.SMALL
    path = synth.write_gguf(tmp_path / f"small_{layout}.gguf", cfg, seed=0, layout=layout)
    W, H, B = 700, 518, 2
    imgs = synth.images(B, W, H, seed=3)
    g = G.Graph(device, G.Weights(path))
    m = G.ModelRef(g)
    xi = g.input((3, W, H, B), G.F32, "image")
    out = G.depthany_predict(m, xi, cfg.n_layers, cfg.n_heads)
    taps = [g.output(g.get_tensor(f"dino_layer_{i}"), f"dino_layer_{i}") for i in cfg.feature_layers]
    fused = g.output(g.get_tensor("neck.fusion_stage.layers.3"), "fusion_3")
    g.allocate()
    g.set(xi, np.stack([_pre(im) for im in imgs]))
    g.compute()
    depth = g.get(out)[..., 0]
    assert depth.shape == (B, H, W) and np.isfinite(depth).all()

    om, params = _oracle(cfg, 0)
    names = [f"dino_layer_{i}" for i in cfg.feature_layers] + ["fusion_3"]
    for b in range(B):
        caps = {n: 1 << 25 for n in names}
        want, cap = om.predict(params, _pre(imgs[b]), caps)
        for n, tn in zip(names, taps + [fused]):
            got = g.get(tn)
            got = got[0, b] if n.startswith("dino") else got[b]
            assert rel(got.reshape(-1), cap[n].reshape(-1)) < (2e-2 if n.startswith("dino") else 3e-2), n
        mae = float(np.abs(_norm(depth[b]) - _norm(want.reshape(H, W))).mean())
        print(f"{layout} image {b}: graph executor vs oracle MAE {mae:.2e}")
        assert mae < 1e-3

    model = vision.Model.load(path, device)
    static = model.compute_batch(imgs)
    for b in range(B):
        assert float(np.abs(_norm(depth[b]) - static[b]).mean()) < 5e-4


def test_model_kernel_groups_match_the_per_node_lowering(device, tmp_path):
    """The same node graph (the north star's widths, 4 layers, 112 x 112, u8 images in HBM through the two extension ops image_u8_to_f32 and
    image_normalize) lowered both ways: node groups on the model kernels (token-stationary block kernel on an f32 residual stream, LDS-ring convs,
    projection before resize, resizing halo loader, in-place cls-token slice, padded rows) and one launch per epilogue-fused node. Both meet the
    oracle bar (MAE < 1e-3 on the normalised depth) and agree with each other to f16 rounding of the intermediates; the device's min-max
    normalisation equals the host's."""
    cfg = synth.Config(embed_dim=384, n_layers=4, n_heads=6, image_size=112, feature_layers=(0, 1, 2, 3), name="wide4")
    path = synth.write_gguf(tmp_path / "wide4.gguf", cfg, seed=8)
    W = H = 112
    B = 3
    imgs = synth.images(B, W, H, seed=21)
    res = {}
    for fused in (True, False):
        g = G.Graph(device, G.Weights(path))
        g.set_fused_models(fused)
        u8 = g.input((3, W, H, B), G.U8, "image_u8")
        x = g.op(G.OP_IMAGE_U8_TO_F32, [u8], fparams=[0.485, 0.456, 0.406, 1 / 0.229, 1 / 0.224, 1 / 0.225])
        depth = G.depthany_predict(G.ModelRef(g), x, cfg.n_layers, cfg.n_heads, feature_layers=cfg.feature_layers)
        out = g.output(g.op(G.OP_IMAGE_NORMALIZE, [depth]), "normalized")
        g.allocate()
        d = g.describe()
        assert ("dino_block[" in d) == fused and ("preprocess_patches" in d) == fused and ("image_u8_to_f32" in d) == (not fused)
        g.set(u8, imgs)
        g.compute()
        res[fused] = (g.get(depth)[..., 0], g.get(out)[..., 0])
    om, params = _oracle(cfg, 8)
    for fused, (raw, norm) in res.items():
        for b in range(B):
            np.testing.assert_allclose(norm[b], _norm(raw[b]), atol=2e-6)  # image_normalize (image.cpp:537-582) on the device
            want, _ = om.predict(params, _pre(imgs[b]), {})
            assert float(np.abs(norm[b] - _norm(want.reshape(H, W))).mean()) < 1e-3, fused
    assert float(np.abs(res[True][1] - res[False][1]).mean()) < 5e-4


def test_hip_graph_replay_is_bit_identical(device, tmp_path):
    """compute() eagerly, then the same launch list replayed as one hipGraph: identical bits, and new input data is picked up."""
    cfg = synth.MINI
    path = synth.write_gguf(tmp_path / "mini.gguf", cfg, seed=4)
    g = G.Graph(device, G.Weights(path))
    xi = g.input((3, 112, 112, 2), G.F32, "image")
    out = G.depthany_predict(G.ModelRef(g), xi, cfg.n_layers, cfg.n_heads, feature_layers=cfg.feature_layers)
    g.allocate()
    g.use_hip_graph(True)
    a, b2 = (np.stack([_pre(im) for im in synth.images(2, 112, 112, seed=s)]) for s in (1, 2))
    g.set(xi, a); g.compute(); first = g.get(out)       # eager + capture
    g.compute(); again = g.get(out)                     # replay
    np.testing.assert_array_equal(first, again)
    g.set(xi, b2); g.compute(); other = g.get(out)
    assert np.abs(other - first).max() > 0
    om, params = _oracle(cfg, 4)
    want, _ = om.predict(params, a[0], {})
    assert float(np.abs(_norm(first[0, ..., 0]) - _norm(want.reshape(112, 112))).mean()) < 1e-3


def test_intermediates_are_not_readable_and_errors_surface(device):
    g = G.Graph(device)
    g.add_weight("fc.weight", np.eye(64, dtype=np.float32))
    m = G.ModelRef(g)
    x = g.input((64, 8), G.F16)
    mid = G.linear(m["fc"], x)
    y = g.output(G.linear(m["fc"], G.relu(m, mid)), "y")
    g.allocate()
    g.set(x, np.ones((8, 64)))
    g.compute()
    np.testing.assert_array_equal(g.get(y), np.ones((1, 1, 8, 64)))
    with pytest.raises(L.Error, match="recycled"):
        g.get(mid)  # not marked as an output: its buffer may have been reused (compute_graph_output is how the reference keeps one)
    with pytest.raises(ValueError):
        g.set(x, np.ones((4, 64)))
    g2 = G.Graph(device)
    g2.add_weight("q.weight", np.zeros((96, 96), np.float32))
    x2 = g2.input((96, 10, 1), G.F16)
    q = G.reshape(G.ModelRef(g2), G.linear(G.ModelRef(g2)["q"], x2), 32, 3, 10, 1)
    g2.output(g2.op(G.OP_ATTENTION, [q, q, q], fparams=[0.1]), "o")
    with pytest.raises(L.Error, match="head_dim 64"):
        g2.allocate()


def test_cpp_graph_layer_on_a_network_of_its_own():
    """tests/cpp/graph_check.cpp: include/visp/ml.h + nn.h from C++ (model_init, model_add_tensor, compute_graph_init, model_ref prefixes,
    compute_graph_input / _output, the nn.h builders, compute_graph_allocate, transfer_to_backend, compute, transfer_from_backend) on a token
    mixer + small conv decoder that the same program also evaluates in float loops. (Depth-Anything itself is built through this layer by the
    reference's own arch sources in tests/test_reference_sources_compile.py, and through the Python face below.)"""
    import subprocess
    from pathlib import Path

    exe = Path(__file__).resolve().parents[1] / "vision.cpp_amd" / "lib" / "graph_check"
    assert exe.exists(), "run __graft_entry__.build() first"
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0 and "graph_check ok" in r.stdout, (r.stdout, r.stderr)


def test_graphs_over_one_model_share_the_device_weights(device, tmp_path):
    """The reference rebuilds its graph when the input extent changes (vision.cpp:150-158) over the same model_weights. Here the second
    graph over a Weights object uploads nothing for the weights (only its own folded constants) and both graphs stay usable."""
    cfg = synth.MINI
    path = synth.write_gguf(tmp_path / "mini.gguf", cfg, seed=4)
    w = G.Weights(path)
    outs, graphs = [], []
    for (W, H) in [(112, 112), (168, 112)]:
        g = G.Graph(device, w)
        xi = g.input((3, W, H, 1), G.F32, "image")
        out = G.depthany_predict(G.ModelRef(g), xi, cfg.n_layers, cfg.n_heads, feature_layers=cfg.feature_layers)
        g.allocate()
        graphs.append((g, xi, out, W, H))
    first, second = graphs[0][0].summary()["constant_bytes"], graphs[1][0].summary()["constant_bytes"]
    assert first > 3_000_000 and second < first // 10, (first, second)  # the resized position embeddings are the second graph's own
    om, params = _oracle(cfg, 4)
    for g, xi, out, W, H in graphs:
        img = _pre(synth.images(1, W, H, seed=W)[0])
        g.set(xi, img[None])
        g.compute()
        want, _ = om.predict(params, img, {})
        assert float(np.abs(_norm(g.get(out)[0, ..., 0]) - _norm(want.reshape(H, W))).mean()) < 1e-3


# ---- ESRGAN through the graph layer (esrgan.cpp:13-79): planar maps, concat as planes of one buffer -----------------------------------------

def _esrgan_graph(device, path, cfg, shape):
    g = G.Graph(device, G.Weights(path))
    img = g.input(shape, G.F32, "image")
    out = G.esrgan_generate(G.ModelRef(g), img, cfg.scale, cfg.num_blocks)
    g.allocate()
    return g, img, out


@pytest.mark.parametrize("cfg_name,hw", [("ESRGAN_TINY", (56, 40)), ("ESRGAN_X4", (32, 48))])
def test_esrgan_generate_through_the_graph(device, tmp_path, cfg_name, hw):
    """The reference's esrgan_generate (esrgan.cpp:55-79) built node by node (conv_2d, leaky_relu, concat, scale, add, interpolate NEAREST) and lowered onto
    the LDS-ring conv in its planar layout: the f32 image as a value | residue plane, every conv_block into the plane behind its input (concat is no
    launch), conv5 * 0.2 + x with x from the halo, the rrdb's (.) * 0.2 + input as the second residual, the x2 resize in the up-conv's loader, the last
    conv straight to f32 RGB -- the launch list of this library's hand schedule (csrc/esrgan.cpp), derived from the graph. Checked against the CPU
    oracle (pinned to the reference's torch RRDBNet, tests/golden/make_golden_esrgan.py) and against the hand schedule on the same weights."""
    O = oracle
    cfg = getattr(synth, cfg_name)
    path = tmp_path / "esrgan.gguf"
    synth.write_esrgan_gguf(path, cfg, 7)
    Hh, Ww = hw
    B = 3
    imgs = synth.images(B, Ww, Hh, seed=11).astype(np.float32) / np.float32(255.0)  # [B, H, W, 3]
    g, img, out = _esrgan_graph(device, path, cfg, (3, Ww, Hh, B))
    lines = g.describe().strip().splitlines()
    n_up = int(math.log2(cfg.scale))
    assert lines[0].startswith("image_planes") and "[f32 image]" in lines[1] and "[rgb f32]" in lines[-2]
    assert len(lines) - 1 == 1 + 1 + 15 * cfg.num_blocks + 1 + n_up + 2  # image plane, first conv, 15 convs per rrdb, trunk, up-convs, hr conv + last conv
    text = "\n".join(lines)
    assert text.count("[*s + x]") == 3 * cfg.num_blocks and text.count("[*s + x][*s + res]") == cfg.num_blocks and text.count("[nearest x2 in the loader]") == n_up
    assert "concat" not in text and "planes_to_nhwc" not in text
    g.set(img, imgs)
    g.compute()
    got = g.get(out)
    assert got.shape == (B, Hh * cfg.scale, Ww * cfg.scale, 3) and got.dtype == np.float32
    sd = synth.esrgan_state_dict(cfg, 7)
    tensors, conv2d = synth.esrgan_gguf_tensors(sd)
    om = O.Model(tensors, conv2d, "whcn")
    for i in range(B):
        ref = O.esrgan_generate(om, cfg.scale, cfg.num_blocks, imgs[i])
        assert np.abs(got[i] - ref).mean() < 1e-3 and np.abs(got[i] - ref).max() < 8e-3
    m = vision.Model.load(path, device)
    hand = m.esrgan_generate(imgs)
    assert np.abs(got - hand).max() < 2e-3  # same kernels, same operands; the hand schedule's first conv is issued twice, nothing else differs


def test_esrgan_nodes_outside_the_patterns(device):
    """leaky_relu, interpolate NEAREST and a planar map read by something that is not the LDS-ring conv (here: a graph output and an add of two maps):
    the generic launches and the planes -> NHWC copy."""
    rng = np.random.default_rng(3)
    C, H, W, B = 32, 9, 7, 2
    x = h(rng.standard_normal((B, H, W, C)))
    g = G.Graph(device)
    m = G.ModelRef(g)
    xi = g.input((C, W, H, B), G.F16)
    a = G.leaky_relu(m, xi, 0.1)
    b = G.interpolate(m, a, (W * 3, H * 2), G.SCALE_MODE_NEAREST)
    ya, yb = g.output(a, "a"), g.output(b, "b")
    got_a, got_b = run(g, {xi: x}, [ya, yb])
    want_a = np.where(x > 0, x, 0.1 * x)
    np.testing.assert_allclose(got_a, want_a, rtol=2e-3, atol=1e-3)
    ref_b = F.interpolate(t(h(want_a)).permute(0, 3, 1, 2), size=(H * 2, W * 3), mode="nearest").permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(got_b, ref_b, rtol=2e-3, atol=1e-3)


def test_esrgan_interior_maps_read_back_from_their_planes(device, tmp_path):
    """Module boundaries of the generator kept as graph outputs (the first conv's map, the two rrdb outputs, the trunk sum): each lives as planes of a dense
    block / a planar buffer and is handed out as NHWC by one `planes_to_nhwc` copy; the schedule around them stays the planar one. Against the oracle's
    captures of the same boundaries (esrgan.cpp:58-67)."""
    O = oracle
    cfg = synth.ESRGAN_TINY
    path = tmp_path / "esrgan.gguf"
    synth.write_esrgan_gguf(path, cfg, 7)
    Hh, Ww, B = 24, 40, 2
    imgs = synth.images(B, Ww, Hh, seed=5).astype(np.float32) / np.float32(255.0)
    g = G.Graph(device, G.Weights(path))
    img = g.input((3, Ww, Hh, B), G.F32, "image")
    out = G.esrgan_generate(G.ModelRef(g), img, cfg.scale, cfg.num_blocks)
    keep = {"rrdb_0": g.output(g.get_tensor("model.1.sub.0"), "rrdb_0"), "rrdb_1": g.output(g.get_tensor("model.1.sub.1"), "rrdb_1")}
    g.allocate()
    text = g.describe()
    assert text.count("planes_to_nhwc") == 2 and text.count("dconv3x3(planes)") == 1 + 15 * cfg.num_blocks + 1 + 1 + 2 and "concat" not in text
    g.set(img, imgs)
    g.compute()
    got = {k: g.get(v) for k, v in keep.items()}
    res = g.get(out)
    sd = synth.esrgan_state_dict(cfg, 7)
    tensors, conv2d = synth.esrgan_gguf_tensors(sd)
    om = O.Model(tensors, conv2d, "whcn")
    for i in range(B):
        ref, caps = O.esrgan_generate(om, cfg.scale, cfg.num_blocks, imgs[i], {"rrdb_0": Hh * Ww * 64, "rrdb_1": Hh * Ww * 64})
        assert np.abs(res[i] - ref).mean() < 1e-3
        for k in keep:
            want = caps[k].reshape(Hh, Ww, 64)
            assert got[k].shape == (B, Hh, Ww, 64) and rel(got[k][i], want) < 6e-3, k


def test_dense_block_that_does_not_fit_the_pattern_falls_back_correctly(device):
    """A conv_block whose activation the conv epilogue cannot absorb (LeakyReLU 0.1) in a concat chain: the chain is NOT taken as planes of one buffer --
    the planar first map is copied out as NHWC once, the activation and the concat are launches of their own, the next conv runs on the NHWC map -- and
    the numbers are those of the same ops in torch."""
    rng = np.random.default_rng(9)
    H, W, B = 20, 24, 2
    img = rng.random((B, H, W, 3)).astype(np.float32)
    w0, b0 = h(rng.standard_normal((64, 3, 3, 3)) / math.sqrt(27)), 0.1 * rng.standard_normal(64)
    w1, b1 = h(rng.standard_normal((32, 3, 3, 64)) / math.sqrt(576)), 0.1 * rng.standard_normal(32)
    w2, b2 = h(rng.standard_normal((64, 3, 3, 96)) / math.sqrt(864)), 0.1 * rng.standard_normal(64)
    g = G.Graph(device)
    for k, (w, b) in {"c0": (w0, b0), "c1": (w1, b1), "c2": (w2, b2)}.items():
        g.add_weight(f"{k}.weight", w); g.add_weight(f"{k}.bias", b, G.F32)
    m = G.ModelRef(g)
    x = g.input((3, W, H, B), G.F32, "image")
    x0 = G.conv_2d(m["c0"], x, 1, 1)
    x1 = G.leaky_relu(m, G.conv_2d(m["c1"], x0, 1, 1), 0.1)
    c = G.concat(m, [x0, x1], 0)
    y = g.output(G.relu(m, G.conv_2d(m["c2"], c, 1, 1)), "y")
    (got,) = run(g, {x: img}, [y])
    d = g.describe()
    assert "planes_to_nhwc" in d and "concat part 0" in d and "leaky_relu n=" in d and d.count("dconv3x3(planes)") == 2

    def conv(v, w, b):
        return F.conv2d(v, t(w).permute(0, 3, 1, 2), t(b), padding=1)

    v0 = t(h(conv(t(img).permute(0, 3, 1, 2), w0, b0).numpy()))
    v1 = t(h(F.leaky_relu(t(h(conv(v0, w1, b1).numpy())), 0.1).numpy()))
    ref = torch.relu(conv(torch.cat([v0, v1], 1), w2, b2)).permute(0, 2, 3, 1).numpy()
    assert rel(got, ref) < 4e-3
