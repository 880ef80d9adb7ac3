"""Graph layer without a GPU (planning-only graphs, `Graph(None)`): shape inference with ggml's conventions, constant folding on
the host, the lowering's fusion decisions and the liveness arena -- the parts of the executor boundary (SURVEY section 8 rows a19 / b3;
reference include/visp/ml.h:154-256, src/visp/ml.cpp:531-642, 746-788) that are host logic."""
import numpy as np
import pytest
import torch

from oracle import oracle
from visioncpp_amd import _lib as L
from visioncpp_amd import graph as G
from visioncpp_amd import synth


def _g():
    return G.Graph(None)


def test_model_ref_prefixes_and_lookup():
    g = _g()
    w = g.add_weight("enc.layer.3.fc.weight", np.zeros((8, 64), np.float32))
    g.add_weight("enc.layer.3.fc.bias", np.zeros(8, np.float32), G.F32)
    m = G.ModelRef(g)
    assert m["enc"]["layer"][3]["fc"].weights("weight").index == w.index
    assert m["enc.layer"][3].find("fc.bias") is not None
    assert m["enc"].find("nothing") is None  # model_ref::find returns null (ml.h:221)
    with pytest.raises(KeyError, match="tensor not found: enc.nothing"):  # model_ref::weights asserts (ml.h:222)
        m["enc"].weights("nothing")
    assert m.with_prefix("enc.layer.3.fc").weights("weight").ne == (64, 8, 1, 1)  # torch [8, 64] = ggml ne [64, 8]
    with pytest.raises(L.Error, match="exists already"):
        g.add_weight("enc.layer.3.fc.bias", np.zeros(8, np.float32))


def test_shape_inference_follows_ggml_conventions():
    g = _g()
    m = G.ModelRef(g)
    g.add_weight("fc.weight", np.zeros((96, 64), np.float32))
    g.add_weight("c.weight", np.zeros((32, 3, 3, 64), np.float32))       # OHWI = ne [Cin, kw, kh, Cout]
    g.add_weight("t.weight", np.zeros((64, 48, 2, 2), np.float32))       # torch [Cin, Cout, kh, kw] = ne [kw, kh, Cout, Cin]
    x = g.input((64, 10, 7, 2), G.F16)
    assert G.linear(m["fc"], x).ne == (96, 10, 7, 2)
    assert G.conv_2d(m["c"], x, 1, 1).ne == (32, 10, 7, 2)
    assert G.conv_2d(m["c"], x, 2, 1).ne == (32, 5, 4, 2)
    assert G.conv_transpose_2d(m["t"], x, 2).ne == (48, 20, 14, 2)
    assert G.interpolate(m, x, (15, 9), G.BILINEAR_AC).ne == (64, 15, 9, 2)
    assert G.slice_(m, x, G.SLICE_ALL, (1, 10), G.SLICE_ALL, 1).ne == (64, 9, 7, 1)
    assert G.slice_(m, x, (0, 64, 2), (-3, 10)).ne == (32, 3, 7, 2)  # step, python-style negative begin
    assert G.concat(m, [x, x, x], 1).ne == (64, 30, 7, 2)
    assert G.reshape(m, x, 64, 70, 2).ne == (64, 70, 2, 1)
    q = G.reshape(m, x, 16, 4, 70, 2)
    assert g.op(G.OP_ATTENTION, [q, q, q], fparams=[0.25]).ne == (64, 70, 2, 1)
    for bad, msg in [(lambda: G.reshape(m, x, 63, 70, 2), "elements"), (lambda: G.conv_transpose_2d(m["t"], x, 4), "kernel == stride"),
                     (lambda: G.linear(m["fc"], G.reshape(m, x, 32, 20, 7, 2)), "does not match"),
                     (lambda: G.add(m, x, g.input((64, 1, 7, 2), G.F16)), "cannot broadcast"),
                     (lambda: G.slice_(m, x, G.SLICE_ALL, (10, 12)), "empty or out-of-range"),
                     (lambda: G.relu(m, g.input((3, 8, 8, 1), G.F32)), "is f32")]:
        with pytest.raises(L.Error, match=msg):
            bad()


def test_constants_fold_on_the_host_like_the_oracle_computes_them():
    """dino::interpolate_pos_encoding (dino.cpp:10-30) on weights alone never reaches the device: slice + reshape + bicubic + concat
    fold into one constant that equals the oracle's resize of the same embeddings (700 x 518 = 50 x 37 patches)."""
    rng = np.random.default_rng(0)
    D, side = 16, 5
    pos = rng.standard_normal((1, 1 + side * side, D)).astype(np.float32)
    g = _g()
    g.add_weight("position_embeddings", pos, G.F32)
    m = G.ModelRef(g)
    x = g.input((D, 1 + 7 * 4, 2), G.F16)
    out = G.dino_interpolate_pos_encoding(m, x, 7 * 14, 4 * 14, 14)
    assert out.is_constant and out.ne == (D, 1 + 7 * 4, 1, 1)
    got = g.read_constant(out).reshape(1 + 28, D)
    want_patch = oracle.interpolate_nhwc(pos[0, 1:].reshape(1, side, side, D), (4, 7), mode="bicubic", align_corners=False)
    np.testing.assert_array_equal(got[0], pos[0, 0])
    np.testing.assert_allclose(got[1:], want_patch.reshape(28, D), rtol=1e-5, atol=1e-6)
    # ... and torch agrees with both (tests/test_primitives.py:165-184 pins ggml's interpolate on torch.nn.functional)
    tp = torch.nn.functional.interpolate(torch.from_numpy(pos[0, 1:].reshape(1, side, side, D)).permute(0, 3, 1, 2), size=(4, 7), mode="bicubic", align_corners=False)
    np.testing.assert_allclose(got[1:], tp.permute(0, 2, 3, 1).reshape(28, D).numpy(), rtol=1e-4, atol=1e-5)
    # unchanged grid: the stored tensor itself
    same = G.dino_interpolate_pos_encoding(m, g.input((D, 26, 1), G.F16), 70, 70, 14)
    assert same.index == g.find("position_embeddings").index
    # a folded repeat + scale + add
    cls = g.add_weight("cls", rng.standard_normal((1, 1, D)).astype(np.float32), G.F32)
    r = G.add(m, G.scale(m, G.repeat(m, cls, D, 1, 3, 1), 2.0), cls)
    np.testing.assert_allclose(g.read_constant(r), np.broadcast_to(3.0 * g.read_constant(cls), (1, 3, 1, D)), rtol=1e-6)


def _planned(tmp_path_factory, fused_models):
    path = synth.write_gguf(tmp_path_factory.mktemp("g") / "small.gguf", synth.SMALL, seed=0)
    g = G.Graph(None, G.Weights(path))
    g.set_fused_models(fused_models)
    img = g.input((3, 518, 518, 2), G.F32, "image")
    out = G.depthany_predict(G.ModelRef(g), img, 12, 6)
    g.allocate()
    return g, img, out


@pytest.fixture(scope="module")
def planned(tmp_path_factory):  # one launch per epilogue-fused node
    return _planned(tmp_path_factory, False)


@pytest.fixture(scope="module")
def planned_fused(tmp_path_factory):  # node groups mapped onto the kernels written for them (the default)
    return _planned(tmp_path_factory, True)


def test_depth_anything_node_groups_lower_to_the_model_kernels(planned_fused):
    """The default lowering of the same 567 nodes: 54 launches -- the launch list of the hand-written step this library measured in rounds 1-3,
    now derived from the graph (62 launches), with one launch per residual unit on the 19^2 / 37^2 / 74^2 maps since round 4. dino.cpp:10-110: prepare_tokens = patches + cls rows + one GEMM whose epilogue writes f32 token rows; the
    first layer's LN1 + QKV, then per layer ONE attention and ONE token-stationary block launch (out-proj, both residuals, MLP, the taps'
    final LayerNorm, the next layer's LN1 + QKV). depth-anything.cpp:15-96: the cls-token slice is row addressing in the projection GEMM,
    the 48 / 96-channel projections are written with the padded rows their conv_transpose reads, every 3x3 / stride-1 conv is the LDS-ring
    kernel with ReLU-in / ReLU / two residual maps in its epilogue -- except dpt::residual_conv on maps of at most 96 x 96, which is ONE launch
    (relu, conv, relu, conv, + x [+ feature_fusion's other addend] [+ the 1x1 projection]) with the intermediate map in LDS --, the fusion
    projection runs BEFORE its resize, head.conv1 resizes in its halo loader, the head's tail is one kernel."""
    g, img, out = planned_fused
    lines = g.describe().strip().splitlines()
    assert g.summary()["launches"] == len(lines) - 1 == 54
    assert lines[0].startswith("im2col_patches 14x14 M=2738") and lines[1] == "cls_rows B=2" and lines[2].startswith("gemm[tokens f32: + bias + pos] M=2738 N=384 K=588")
    assert lines[3].startswith("dino_block[ln1 + qkv] M=2740")
    assert [l.split(" M=")[0] for l in lines[4:28]] == ["attention B=2 heads=6 T=1370", "dino_block[out-proj + mlp + next ln1 + qkv]"] * 2 + \
        (["attention B=2 heads=6 T=1370", "dino_block[out-proj + mlp + tap + next ln1 + qkv]"] + ["attention B=2 heads=6 T=1370", "dino_block[out-proj + mlp + next ln1 + qkv]"] * 2) * 3 + \
        ["attention B=2 heads=6 T=1370", "dino_block[out-proj + mlp + tap]"]
    text = "\n".join(lines)
    assert text.count("gemm(conv1x1)[rows 1.. of 1370] M=2738") == 4 and not any(l.startswith(("slice", "pad_rows")) for l in lines)
    assert text.count("gemm+pixel_shuffle") == 2 and "conv3x3s2 M=722 N=384 K=3456" in text
    assert text.count("dconv3x3 M=") == 4                                            # neck.convs (no bias, no activation)
    # fusion stages 0-2 (19^2, 37^2, 74^2): five residual units, three of them with the stage's projection; stage 3 (148^2) stays on the LDS-ring conv
    assert text.count("residual_unit[relu, conv3x3, relu, conv3x3, + x, conv1x1 before its resize]") == 3 and text.count("residual_unit[relu, conv3x3, relu, conv3x3, + x, + x0]") == 2
    assert text.count("dconv3x3[relu-in][relu]") == 2 and text.count("dconv3x3[+res][+res]") == 1 and text.count("dconv3x3[+res] M=") == 1
    assert text.count("gemm(conv1x1 before its resize)") == 1 and text.count("bilinear_ac") == 3
    assert lines[-3].startswith("dconv3x3[resize 148x148 in the loader] M=175232 N=32 K=576 <- head.conv1.weight")
    assert lines[-2].startswith("head_tail[resize 296x296 -> 518x518, conv3x3 32->32, relu, conv1x1 -> 1, relu] B=2 <- head.conv2.weight")
    assert g.get_tensor("dino_layer_11").ne == (384, 1370, 2, 1)
    # a group is only taken when nothing outside reads its interior: an output in the middle of a layer sends the encoder down the generic path
    path = synth.write_gguf(__import__("pathlib").Path(__import__("tempfile").mkdtemp()) / "small.gguf", synth.SMALL, seed=0)
    g2 = G.Graph(None, G.Weights(path))
    x = g2.input((3, 518, 518, 1), G.F32, "image")
    G.depthany_predict(G.ModelRef(g2), x, 12, 6)
    hidden = g2.get_tensor("backbone.encoder.layer.4")  # a layer output: allowed (copied out of the f32 stream) ...
    g2.output(hidden, "layer_4")
    g2.allocate()
    assert g2.describe().count("dino_block[") == 13 and "copy_f32 layer output" in g2.describe() and g2.get_tensor("layer_4").dtype == G.F32


def test_depth_anything_lowers_to_fused_launches(planned):
    """Without the model-kernel groups (set_fused_models(False)) 567 graph nodes (weights included) lower to 134 launches, 7 per encoder layer: the three q / k / v products of an attention are one
    GEMM with a head-major epilogue, LayerScale is folded into the packed weights so the residual rides in the product's epilogue,
    activations, ReLU-on-load and conv residuals are epilogues and loader flags, views cost nothing, and everything computed from weights
    alone was folded when the node was made."""
    g, img, out = planned
    lines = g.describe().strip().splitlines()
    s = g.summary()
    assert s["launches"] == len(lines) - 1 == 134
    text = "\n".join(lines)
    assert text.count("gemm[qkv heads-major] M=2740 N=1152 K=384") == 12  # query | key | value (dino.cpp:59-70): one product
    assert text.count("gemm[*scale][+res]") == 24                # out-proj and fc2 with layer_scale + residual (dino.cpp:48-50, 80-87)
    assert not any(l.startswith(("mul", "heads_major")) for l in lines)
    assert text.count("gemm[gelu]") == 12                       # fc1 + gelu (dino.cpp:52-56)
    assert text.count("attention B=2 heads=6 T=1370") == 12
    # residual_conv x 7 (depth-anything.cpp:15-23): two launches each; feature_fusion's x0 + ... rides in the same epilogue as a second residual
    assert text.count("[relu-in][relu]") == 7 and text.count("conv3x3[+res] M=") == 4 and text.count("conv3x3[+res][+res]") == 3
    assert text.count("gemm+pixel_shuffle") == 2                 # conv_transpose k == s (nn.cpp:117-129)
    assert "conv3x3s2 M=722 N=384 K=3456" in text                # reassemble 3: 3x3 stride 2 on 37 x 37
    # the head's tail -- resize, conv2, relu, conv3, relu -- is the one launch of the kernel made for it (kernels_headconv.hip)
    assert lines[-2].startswith("head_tail[resize 296x296 -> 518x518, conv3x3 32->32, relu, conv1x1 -> 1, relu] B=2 <- head.conv2.weight")
    assert not any(l.startswith(("relu", "gelu")) for l in lines)     # no stand-alone activation survives
    assert out.dtype == G.F32 and out.ne == (1, 518, 518, 2)
    assert g.get_tensor("dino_layer_11").ne == (384, 1370, 2, 1)      # ggml_format_name + ggml_get_tensor (dino.cpp:103-105)
    assert g.get_tensor("image").index == img.index


def test_arena_recycles_buffers_by_liveness(planned):
    """ggml_gallocr's job (ml.cpp:545-552): a buffer is reused after its last reader. Batch 2 at 518 x 518: 534 MB of node outputs
    live in < 100 MB; the four tapped feature maps (read by the neck long after their layer) survive the encoder."""
    g, _, _ = planned
    s = g.summary()
    assert s["unshared_bytes"] > 5 * s["arena_bytes"]
    largest = 2 * 296 * 296 * 64 * 2  # fusion 3's output at 296 x 296 x 64 (the 518 x 518 x 32 map never exists: the head's tail is one kernel)
    assert 2 * largest < s["arena_bytes"] < 4 * largest
    assert s["constant_bytes"] > 49_000_000  # every weight packed once (24.8 M parameters in f16 + padding)


def test_dead_nodes_and_missing_outputs():
    g = _g()
    g.add_weight("fc.weight", np.zeros((64, 64), np.float32))
    m = G.ModelRef(g)
    x = g.input((64, 8), G.F16)
    y = G.linear(m["fc"], x)
    G.gelu(m, G.linear(m["fc"], y))  # never reaches an output: not launched
    with pytest.raises(L.Error, match="no output"):
        g.allocate()
    g.output(G.relu(m, y), "y")
    g.allocate()
    assert g.describe().splitlines()[0].startswith("gemm[relu] M=8 N=64 K=64")
    assert g.summary()["launches"] == 1
    with pytest.raises(L.Error, match="already allocated"):
        G.relu(m, y)
    with pytest.raises(L.Error, match="without a device"):
        g.compute()


def test_fusion_respects_other_readers():
    """An activation with a second reader, or a producer that is itself an output, must stay a launch of its own."""
    g = _g()
    g.add_weight("fc.weight", np.zeros((64, 64), np.float32))
    g.add_weight("c.weight", np.zeros((64, 3, 3, 64), np.float32))
    m = G.ModelRef(g)
    x = g.input((64, 8, 8, 1), G.F16)
    y = G.linear(m["fc"], x)
    a = G.gelu(m, y)
    g.output(G.add(m, a, y), "sum")  # y has two readers: gelu is its own launch; the add cannot fold into the gemm either
    r = G.relu(m, x)
    c1 = G.conv_2d(m["c"], r, 1, 1)
    g.output(G.add(m, c1, r), "c")  # relu(x) has two readers: materialised; conv + add still fuse (r is computed first)
    g.allocate()
    lines = g.describe().splitlines()
    assert lines[0].startswith("gemm M=64 N=64") and lines[1].startswith("gelu n=4096") and lines[2].startswith("add n=4096")
    assert lines[3].startswith("relu n=4096") and lines[4].startswith("conv3x3[+res] M=64")


def test_esrgan_lowers_to_the_planar_dense_block_schedule(tmp_path):
    """esrgan_generate (esrgan.cpp:55-79) as a graph, planned without a device: no concat, no activation, no scale / add launch -- every one of them is a
    plane address or an epilogue of the LDS-ring conv -- and the x2 nearest resize is the up-conv's loader."""
    cfg = synth.ESRGAN_TINY
    synth.write_esrgan_gguf(tmp_path / "e.gguf", cfg, 7)
    g = G.Graph(None, G.Weights(tmp_path / "e.gguf"))
    img = g.input((3, 40, 56, 3), G.F32, "image")
    out = G.esrgan_generate(G.ModelRef(g), img, cfg.scale, cfg.num_blocks)
    assert out.ne == (3, 80, 112, 3)
    g.allocate()
    lines = g.describe().strip().splitlines()
    assert g.summary()["launches"] == len(lines) - 1 == 36
    assert lines[0] == "image_planes [3, 40, 56, 3]" and lines[1].startswith("dconv3x3(planes)[f32 image] M=6720 N=64 K=27 <- model.0.weight")
    rdb = [l.split(" M=")[0] for l in lines[2:17]]
    assert rdb == (["dconv3x3(planes)[leaky_relu]"] * 4 + ["dconv3x3(planes)[*s + x]"]) * 2 + ["dconv3x3(planes)[leaky_relu]"] * 4 + ["dconv3x3(planes)[*s + x][*s + res]"]
    assert [l.split(" <- ")[0].split(" K=")[1] for l in lines[2:7]] == ["576", "864", "1152", "1440", "1728"]  # a conv_block reads every plane written before it
    assert lines[32].startswith("dconv3x3(planes)[*s + res] M=6720 N=64 K=576 <- model.1.sub.2.weight")          # trunk conv + the first conv's output
    assert lines[33].startswith("dconv3x3(planes)[nearest x2 in the loader][leaky_relu] M=26880 N=64 K=576 <- model.3.weight")
    assert lines[34].startswith("dconv3x3(planes)[leaky_relu] M=26880") and lines[35].startswith("dconv3x3(planes)[rgb f32] M=26880 N=3 K=576 <- model.7.weight")
    # without the model-kernel groups the f32 image conv has no lowering: said so, not guessed
    g2 = G.Graph(None, G.Weights(tmp_path / "e.gguf"))
    g2.set_fused_models(False)
    G.esrgan_generate(G.ModelRef(g2), g2.input((3, 40, 56, 1), G.F32, "image"), cfg.scale, cfg.num_blocks)
    with pytest.raises(L.Error, match="Cin = 3 must be a multiple of 8|read before it is computed|f32"):
        g2.allocate()
