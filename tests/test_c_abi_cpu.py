"""CPU-side checks of the drop-in boundary (no GPU, no compute calls): the library loads,
exports every symbol the headers declare, parses GGUF files written with the reference's
on-disk contract, and reports errors the way the reference C API does (return 0 + message,
src/visp/c-api.cpp:6-21)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

from visioncpp_amd import _lib as L
from visioncpp_amd import gguf, synth

ROOT = Path(__file__).resolve().parents[1]


def test_library_exports_every_declared_symbol():
    api = L.get_lib()
    declared = set()
    for hdr in ("visp_c_api.h", "visp_hip_kernels.h"):
        text = (ROOT / "include" / hdr).read_text()
        declared |= set(re.findall(r"\b(visp_[a-z0-9_]+|vx_[a-z0-9_]+)\s*\(", text))
    declared -= {"vx_gemm_args", "visp_image_view"}
    assert len(declared) >= 55
    for sym in sorted(declared):
        assert hasattr(api, sym), f"{sym} declared in include/ but not exported"
    assert declared <= set(L.C_API_SYMBOLS + L.KERNEL_SYMBOLS), declared - set(L.C_API_SYMBOLS + L.KERNEL_SYMBOLS)


def test_reference_symbol_set_is_complete():
    ref = ["visp_get_last_error", "visp_image_destroy", "visp_backend_load_all", "visp_device_init", "visp_device_destroy",
           "visp_device_type", "visp_device_name", "visp_device_description", "visp_model_detect_family", "visp_model_load",
           "visp_model_destroy", "visp_model_compute"]  # reference c-api.cpp:145-253
    api = L.get_lib()
    for s in ref:
        getattr(api, s)


def test_gguf_roundtrip_and_family_detection(tmp_path):
    path = synth.write_gguf(tmp_path / "tiny.gguf", synth.TINY, seed=3)
    f = gguf.GGUFFile(path)
    assert f.kv["general.architecture"] == "depthanything"
    assert f.kv["dino.embed_dim"] == 32 and f.kv_types["dino.embed_dim"] == gguf.T_I32
    assert f.kv["depthanything.feature_layers"] == [0, 1, 2, 3]
    sd = synth.state_dict(synth.TINY, 3)
    tensors, conv2d = synth.gguf_tensors(sd)
    assert f.kv["depthanything.conv2d_weights"] == conv2d
    assert f.tensor_names == list(tensors.keys())
    for k, v in tensors.items():
        np.testing.assert_array_equal(f.tensors[k], v)
    fam = C.c_int32(-1)
    L.check(L.get_lib().visp_model_detect_family(str(path).encode(), C.byref(fam)))
    assert fam.value == 2  # model_family::depth_anything


def test_other_architectures_are_detected(tmp_path):
    for arch, fam in [("mobile-sam", 0), ("birefnet", 1), ("migan", 3), ("esrgan", 4), ("something-else", 5)]:
        w = gguf.GGUFWriter(tmp_path / f"{arch}.gguf", arch)
        w.add_tensor("t", np.zeros(4, np.float32))
        w.write()
        out = C.c_int32(-1)
        L.check(L.get_lib().visp_model_detect_family(str(tmp_path / f"{arch}.gguf").encode(), C.byref(out)))
        assert out.value == fam


def test_errors_follow_the_reference_convention(tmp_path):
    api = L.get_lib()
    fam = C.c_int32()
    assert api.visp_model_detect_family(b"/nonexistent/model.gguf", C.byref(fam)) == 0
    assert b"Failed to load GGUF model" in api.visp_get_last_error()
    bad = tmp_path / "bad.gguf"
    bad.write_bytes(b"NOPE" + bytes(64))
    assert api.visp_model_detect_family(str(bad).encode(), C.byref(fam)) == 0
    assert b"bad magic" in api.visp_get_last_error()
    with pytest.raises(L.Error):
        L.check(api.visp_model_detect_family(str(bad).encode(), C.byref(fam)))
    dev = C.c_void_p()
    # there is no CPU backend behind this ABI: asking for one fails loudly instead of falling back
    assert api.visp_device_init(1, C.byref(dev)) == 0
    assert b"no suitable device" in api.visp_get_last_error()


def test_image_view_abi_layout():
    # reference image_view {i32x2 extent; int stride; image_format format; void const* data} (image.h:37-41)
    assert C.sizeof(L.ImageView) == 24
    assert [f[0] for f in L.ImageView._fields_] == ["width", "height", "stride", "format", "data"]
    assert L.ImageView.data.offset == 16


def test_converter_matches_reference_contract(tmp_path):
    """tools/convert_depth_anything.py (scripts/convert.py:428-475 restated): safetensors in, GGUF out, identical
    to the file synth.write_gguf produces for the same tensors, and readable by the C++ loader."""
    from safetensors.numpy import save_file

    from visioncpp_amd import convert

    cfg = synth.Config(embed_dim=384, n_layers=12, n_heads=6, image_size=70, name="s70")  # real dims, 5x5 pos grid
    sd = synth.state_dict(cfg, seed=2)
    save_file(sd, str(tmp_path / "m.safetensors"))
    loaded = convert.load_safetensors(tmp_path / "m.safetensors")  # file order (what the reference's converter iterates too)
    assert set(loaded) == set(sd)
    out = convert.convert_depth_anything(loaded, tmp_path / "m.gguf", image_size=70)
    ref = synth.write_gguf(tmp_path / "ref.gguf", cfg, sd=loaded)
    a, b = gguf.GGUFFile(out), gguf.GGUFFile(ref)
    assert a.kv == b.kv and a.tensor_names == b.tensor_names
    for k in a.tensor_names:
        np.testing.assert_array_equal(a.tensors[k], b.tensors[k])
    # on-disk rules: NHWC patch embed, untouched ConvTranspose, f32 cls/pos, conv2d index list
    assert a.tensors["backbone.embeddings.patch_embeddings.projection.weight"].shape == (384, 14, 14, 3)
    assert a.tensors["neck.reassemble_stage.layers.0.resize.weight"].shape == (48, 48, 4, 4)
    assert a.tensors["backbone.embeddings.position_embeddings"].dtype == np.float32
    assert a.tensors["head.conv1.weight"].dtype == np.float16
    idx = a.kv["depthanything.conv2d_weights"]
    assert a.tensor_names.index("head.conv1.weight") in idx and a.tensor_names.index("neck.convs.0.weight") in idx
    assert a.tensor_names.index("neck.reassemble_stage.layers.0.projection.weight") not in idx
    fam = C.c_int32(-1)
    L.check(L.get_lib().visp_model_detect_family(str(out).encode(), C.byref(fam)))
    assert fam.value == 2


def test_esrgan_gguf_is_detected_and_tile_layout_matches_oracle(tmp_path):
    """Host-only pieces of the ESRGAN row: family detection of the synthetic checkpoint (vision.cpp:7-21) and
    tile_layout / tile_scale (image.cpp:612-629) of the product against the oracle's restatement."""
    from oracle import oracle as O
    from visioncpp_amd.vision import esrgan_tile_layout

    lib = L.get_lib()
    p = synth.write_esrgan_gguf(tmp_path / "e.gguf", synth.ESRGAN_TINY, 3)
    fam = C.c_int32(-1)
    assert lib.visp_model_detect_family(str(p).encode(), C.byref(fam)) == 1 and fam.value == 4
    f = gguf.GGUFFile(p)
    assert f.kv["esrgan.scale"] == 2 and f.kv["esrgan.block_count"] == 2 and f.kv["esrgan.tensor_data_layout"] == "whcn"
    for (w, h, s) in [(256, 256, 4), (1000, 700, 2), (224, 224, 4), (225, 17, 1), (640, 481, 8)]:
        got = esrgan_tile_layout(w, h, s)
        t = O.tile_scale(O.tile_layout(w, h, 224, 16, 16), s)
        assert got == {k: getattr(t, k) for k in got}, (w, h, s)


def test_esrgan_converter_accepts_both_key_layouts(tmp_path):
    """convert_esrgan (reference scripts/convert.py:504-527): an old-arch state dict converts to the file the synthetic
    writer produces; the BasicSR / Real-ESRGAN 'new arch' names give the same file (the renaming spandrel performs)."""
    from visioncpp_amd import convert

    cfg = synth.EsrganConfig(num_blocks=2, scale=4, name="c")
    sd = synth.esrgan_state_dict(cfg, 5)
    ref = gguf.GGUFFile(synth.write_esrgan_gguf(tmp_path / "ref.gguf", cfg, sd=sd))
    a = gguf.GGUFFile(convert.convert_esrgan(sd, tmp_path / "a.gguf"))
    # the same weights under the new-arch names, in shuffled order
    top = {"model.0": "conv_first", "model.1.sub.2": "conv_body", "model.3": "conv_up1", "model.6": "conv_up2", "model.8": "conv_hr", "model.10": "conv_last"}
    new = {}
    for k, v in reversed(list(sd.items())):
        base, _, leaf = k.rpartition(".")
        if base in top:
            new[f"{top[base]}.{leaf}"] = v
        else:
            p = base.split(".")  # model.1.sub.N.RDBk.convj.0
            new[f"body.{p[3]}.rdb{p[4][3:]}.{p[5]}.{leaf}"] = v
    b = gguf.GGUFFile(convert.convert_esrgan(new, tmp_path / "b.gguf"))
    for f in (a, b):
        assert f.kv["general.architecture"] == "esrgan" and f.kv["esrgan.scale"] == 4 and f.kv["esrgan.block_count"] == 2
        assert f.kv["esrgan.filter_count"] == 64 and list(f.kv["esrgan.conv2d_weights"]) == list(ref.kv["esrgan.conv2d_weights"])
        assert list(f.tensors) == list(ref.tensors)   # same names in the same order (the conv2d index list depends on it)
        for name, t in f.tensors.items():
            assert t.dtype == ref.tensors[name].dtype and np.array_equal(t, ref.tensors[name])
    with pytest.raises(ValueError, match="not an RRDBNet"):
        convert.convert_esrgan({"foo.weight": np.zeros(3, np.float32)}, tmp_path / "x.gguf")


def test_ctypes_structs_match_the_c_headers(tmp_path):
    """The Python mirrors of the argument blocks must have the C layout: compile a probe against include/*.h with gcc
    and compare sizeof / offsetof of the last field (catches a field added on one side only)."""
    import subprocess

    root = Path(__file__).resolve().parents[1]
    probe = tmp_path / "probe.c"
    probe.write_text('''
#include <stddef.h>
#include <stdio.h>
#include "visp_hip_kernels.h"
#include "visp_c_api.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(vx_gemm_args), offsetof(vx_gemm_args, debug_stamps), sizeof(vx_dconv_args),
           offsetof(vx_dconv_args, stamps), sizeof(vx_tile_layout), sizeof(visp_image_view), sizeof(visp_timing), sizeof(visp_esrgan_info));
    return 0;
}
''')
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-I", str(root / "include"), str(probe), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = [C.sizeof(L.GemmArgs), L.GemmArgs.debug_stamps.offset, C.sizeof(L.DconvArgs), L.DconvArgs.stamps.offset, C.sizeof(L.TileLayout),
            C.sizeof(L.ImageView), C.sizeof(L.Timing), C.sizeof(L.EsrganInfo)]
    assert got == want, (got, want)


def test_mobile_sam_gguf_is_detected_and_bias_packing_is_host_code(tmp_path):
    """write_tinyvit_gguf follows convert_sam's header contract (family 0 by architecture string), and the attention-bias
    packer (host code in the HIP library) lays the [heads][N][N] table out in MFMA accumulator order with -inf pads."""
    from visioncpp_amd import synth
    lib = L.get_lib()
    p = synth.write_tinyvit_gguf(tmp_path / "sam.gguf", seed=5)
    fam = C.c_int32(-1)
    L.check(lib.visp_model_detect_family(L.path_to_char_p(p), C.byref(fam)))
    assert fam.value == 0
    N, heads, QB = 49, 3, 2
    assert lib.vx_window_attention_bias_bytes(N, heads) == heads * QB * QB * 64 * 16 * 2
    assert lib.vx_window_attention_bias_bytes(196, 5) == 5 * 7 * 7 * 64 * 16 * 2
    bias = np.random.default_rng(0).standard_normal((heads, N, N)).astype(np.float16).astype(np.float32)
    packed = np.zeros(heads * QB * QB * 64 * 16, np.float16)
    L.vx_check(lib.vx_window_attention_pack_bias(bias.ctypes.data, N, heads, packed.ctypes.data))
    pk = packed.reshape(heads, QB, QB, 64, 16).astype(np.float32)
    lane, e = np.meshgrid(np.arange(64), np.arange(16), indexing="ij")
    for qb in range(QB):
        for kb in range(QB):
            q = qb * 32 + (lane & 31)
            key = kb * 32 + (e >> 2) * 8 + 4 * (lane >> 5) + (e & 3)
            want = np.where(key >= N, -np.inf, np.where(q < N, bias[:, np.minimum(q, N - 1), np.minimum(key, N - 1)], 0.0))
            assert np.array_equal(pk[:, qb, kb], want)
    assert lib.vx_window_attention_pack_bias(bias.ctypes.data, 300, heads, packed.ctypes.data) == 0  # more than 256 tokens


def test_mobile_sam_converter_writes_the_reference_contract(tmp_path):
    """convert_sam on a checkpoint-shaped state dict (image_encoder. / prompt_encoder. / mask_decoder. keys, BatchNorm
    bookkeeping tensors included) gives the file the synthetic writer gives, and the loader's family detection accepts it."""
    from visioncpp_amd import convert, gguf, synth
    cfg = synth.TINYVIT_5M
    enc, dec = synth.tinyvit_state_dict(cfg, 6), synth.sam_decoder_state_dict(7)
    ckpt = {"image_encoder." + k: v for k, v in enc.items()}
    ckpt["image_encoder.patch_embed.seq.0.bn.num_batches_tracked"] = np.array(3, np.int64)
    ckpt.update(dec)
    p = convert.convert_sam(ckpt, tmp_path / "conv.gguf")
    ref = synth.write_mobile_sam_gguf(tmp_path / "ref.gguf", cfg, enc_sd=enc, dec_sd=dec)
    a, b = gguf.GGUFFile(p), gguf.GGUFFile(ref)
    assert list(a.tensors) == list(b.tensors)
    for name, t in a.tensors.items():
        assert t.dtype == b.tensors[name].dtype and np.array_equal(t, b.tensors[name]), name
    assert a.tensors["dec.iou_token.weight"].dtype == np.float32 and a.tensors["enc.neck.2.weight"].dtype == np.float16
    assert max(len(n) for n in a.tensors) < 64
    fam = C.c_int32(-1)
    L.check(L.get_lib().visp_model_detect_family(L.path_to_char_p(p), C.byref(fam)))
    assert fam.value == 0
    with pytest.raises(ValueError, match="not a MobileSAM"):
        convert.convert_sam({"foo": np.zeros(3, np.float32)}, tmp_path / "x.gguf")


def test_image_scale_matches_oracle_and_reference_vector():
    """visp_image_scale (host code of the product, csrc/image_resize.cpp) against the oracle's restatement of
    stb_image_resize (exact for u8, 1e-6 for f32) and against the reference's vector tests/test-image.cpp:186-203."""
    import numpy as np

    from oracle import oracle
    from visioncpp_amd import vision

    img = np.zeros((8, 8, 4), np.uint8)
    for i in range(64):
        img[i // 8, i % 8] = [255, 4 * (i // 8), 4 * (i % 8), 255]
    res = vision.image_scale(img, 4, 4, vision.ImageFormat.rgba_u8)
    for i in range(16):
        assert list(res[i // 4, i % 4]) == [255, 2 + 8 * (i // 4), 2 + 8 * (i % 4), 255]
    rng = np.random.default_rng(0)
    F = vision.ImageFormat
    for fmt, ofmt, ch in ((F.rgb_u8, oracle.RGB_U8, 3), (F.rgba_u8, oracle.RGBA_U8, 4), (F.bgra_u8, oracle.BGRA_U8, 4), (F.argb_u8, oracle.ARGB_U8, 4),
                          (F.alpha_u8, oracle.ALPHA_U8, 1), (F.rgba_f32, oracle.RGBA_F32, 4), (F.rgb_f32, oracle.RGB_F32, 3), (F.alpha_f32, oracle.ALPHA_F32, 1)):
        for (h, w, oh, ow) in ((33, 47, 70, 90), (64, 48, 17, 23), (20, 30, 20, 30), (100, 37, 37, 100), (5, 7, 70, 56)):
            is_f = fmt.value >= F.rgba_f32.value
            a = (rng.random((h, w, ch)).astype(np.float32) * 2 - 0.5) if is_f else rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
            a = a if ch > 1 else a[..., 0]
            got, want = vision.image_scale(a, ow, oh, fmt), oracle.image_scale(a, ofmt, ow, oh)
            if is_f:
                assert np.abs(got - want).max() < 1e-6, (fmt, h, w)
            else:
                np.testing.assert_array_equal(got, want)
    with pytest.raises(Exception, match="resize"):
        vision.image_scale(np.zeros((4, 4, 3), np.uint8), 0, 4)


def test_gguf_tensor_infos_are_validated(tmp_path):
    """Crafted headers (64-bit offsets / dims are untrusted): offsets that wrap, dims of 0 or with an overflowing
    product, unaligned offsets and truncated data are refused with an error, never turned into a pointer."""
    import struct

    api = L.get_lib()
    good = tmp_path / "good.gguf"
    w = gguf.GGUFWriter(good, "depthanything")
    w.add_tensor("a", np.arange(16, dtype=np.float32))
    w.add_tensor("b", np.arange(8, dtype=np.float32))
    w.write()
    n = C.c_int32()
    L.check(api.visp_gguf_validate(str(good).encode(), C.byref(n)))
    assert n.value == 2
    raw = bytearray(good.read_bytes())
    # tensor info of "a": name (u64 len + bytes), u32 n_dims, u64 dims[n], u32 type, u64 offset
    pos = raw.index(b"\x01\x00\x00\x00\x00\x00\x00\x00a") + 9
    assert struct.unpack_from("<I", raw, pos)[0] == 1 and struct.unpack_from("<Q", raw, pos + 4)[0] == 16
    dim_at, off_at = pos + 4, pos + 4 + 8 + 4

    def variant(name, at, value):
        b = bytearray(raw)
        struct.pack_into("<Q", b, at, value)
        p = tmp_path / name
        p.write_bytes(b)
        return api.visp_gguf_validate(str(p).encode(), None), api.visp_get_last_error()

    for name, at, value, msg in [
        ("wrap.gguf", off_at, 2 ** 64 - 32, b"out of bounds"),       # base + offset + size wraps
        ("far.gguf", off_at, 1 << 40, b"out of bounds"),
        ("unaligned.gguf", off_at, 4, b"out of bounds"),
        ("zero_dim.gguf", dim_at, 0, b"bad dimension"),
        ("huge_dim.gguf", dim_at, 2 ** 63, b"bad dimension"),
        ("big.gguf", dim_at, 1 << 40, b"out of bounds"),              # 4 TiB of f32
    ]:
        ok, err = variant(name, at, value)
        assert ok == 0 and msg in err, (name, err)
    trunc = tmp_path / "trunc.gguf"
    trunc.write_bytes(bytes(raw[:-8]))
    assert api.visp_gguf_validate(str(trunc).encode(), None) == 0 and b"out of bounds" in api.visp_get_last_error()


def test_image_scale_upsample_against_an_independent_f64_evaluation():
    """The enlarging branch of image_scale -- what every input that is not at the model extent takes (640 x 480 -> 700 x 518) -- has no
    vector in the reference's tests (tests/test-image.cpp:186-203 is a 2:1 Mitchell reduction). stb_image_resize v0.9x, which
    the reference fetches, enlarges with Catmull-Rom: written out here from the library's published definition in float64 and
    numpy, independently of csrc/image_resize.cpp and of oracle.image_scale:
      * output pixel i has its centre at (i + 0.5) / scale in source coordinates; source pixels n with centre n + 0.5 inside the
        +-2 support contribute k(distance), first = floor(lower + 0.5), last = floor(upper - 0.5), coefficients scaled to sum 1;
      * k(x) = 1 - x^2 (2.5 - 1.5 x) for x < 1, 2 - x (4 + x (0.5 x - 2.5)) for x < 2;  * reads beyond the image clamp to the edge;
      * u8 colour is filtered in linear light (sRGB decode / encode), alpha and float images as they are; alpha-weighted when the
        format has an alpha channel.
    u8 results may differ by one level where the float pipeline's rounding and stb's table-based sRGB encoder land on the other
    side of a boundary; f32 images must agree to float rounding."""
    import numpy as np

    from visioncpp_amd import vision

    def kern(x):
        x = np.abs(x)
        return np.where(x < 1, 1 - x * x * (2.5 - 1.5 * x), np.where(x < 2, 2 - x * (4 + x * (0.5 * x - 2.5)), 0.0))

    def weights(n_in, n_out):
        scale = n_out / n_in
        Wm = np.zeros((n_out, n_in))
        for i in range(n_out):
            c = (i + 0.5) / scale
            lo, hi = (i + 0.5 - 2 * scale) / scale, (i + 0.5 + 2 * scale) / scale
            first, last = int(np.floor(lo + 0.5)), int(np.floor(hi - 0.5))
            ns = np.arange(first, last + 1)
            k = kern(c - (ns + 0.5))
            k = k / k.sum()
            for n, kv in zip(ns, k):
                Wm[i, min(max(n, 0), n_in - 1)] += kv  # edge clamp
        return Wm

    def resize(a, oh, ow):  # a [h, w, c] float64
        return np.einsum("yh,hwc->ywc", weights(a.shape[0], oh), np.einsum("xw,hwc->hxc", weights(a.shape[1], ow), a))

    def to_linear(u):
        c = u / 255.0
        return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)

    def to_srgb(l):
        l = np.clip(l, 0.0, 1.0)
        return np.where(l <= 0.0031308, l * 12.92, 1.055 * l ** (1 / 2.4) - 0.055) * 255.0

    rng = np.random.default_rng(42)
    F = vision.ImageFormat
    for (h, w, oh, ow) in ((48, 64, 52, 70), (9, 7, 31, 40), (30, 20, 33, 23)):  # (an axis at scale exactly 1 takes stb's reducing filter)
        # smooth content (a resize of noise mostly measures the encoder's boundaries), full range
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([127 + 120 * np.sin(xx / 3.1 + c) * np.cos(yy / 4.3 - c) for c in range(3)], -1)
        rgb = np.clip(base + rng.normal(0, 6, base.shape), 0, 255).astype(np.uint8)
        got = vision.image_scale(rgb, ow, oh, F.rgb_u8).astype(np.int32)
        want = to_srgb(resize(to_linear(rgb.astype(np.float64)), oh, ow))
        d = np.abs(got - np.rint(want))
        assert d.max() <= 1 and d.mean() < 0.03, (h, w, d.max(), d.mean())
        # float images: no colour space, plain filter
        f = rng.random((h, w, 3)).astype(np.float32)
        gotf = vision.image_scale(f, ow, oh, F.rgb_f32)
        assert np.abs(gotf - resize(f.astype(np.float64), oh, ow)).max() < 2e-5
        # rgba: colour premultiplied by alpha in linear light, filtered, divided back; alpha itself linear
        al = np.clip(128 + 100 * np.sin(xx / 5.0) + rng.normal(0, 5, (h, w)), 1, 255).astype(np.uint8)
        rgba = np.concatenate([rgb, al[..., None]], -1)
        gota = vision.image_scale(rgba, ow, oh, F.rgba_u8).astype(np.int32)
        a_lin = al.astype(np.float64)[..., None] / 255.0
        pm = resize(to_linear(rgb.astype(np.float64)) * a_lin, oh, ow)
        a_out = resize(a_lin, oh, ow)
        col = to_srgb(pm / np.maximum(a_out, 1e-12))
        da = np.abs(gota[..., 3] - np.rint(np.clip(a_out[..., 0], 0, 1) * 255.0))
        dc = np.abs(gota[..., :3] - np.rint(col))[a_out[..., 0] > 0.05]
        assert da.max() <= 1 and dc.max() <= 2 and dc.mean() < 0.06, (da.max(), dc.max(), dc.mean())


def test_model_file_surface_and_null_pointers(tmp_path):
    """visp_file_* = the reference's model_file (ml.h:85-103, ml.cpp:206-254): key/values by name, i32 only for get_int, exact-length i32 arrays,
    a missing key named in the error; and the graph entries report a null out pointer instead of writing through it."""
    import ctypes as C

    api = L.get_lib()
    path = synth.write_gguf(tmp_path / "m.gguf", synth.MINI, seed=1)
    f = C.c_void_p()
    L.check(api.visp_file_load(L.path_to_char_p(path), C.byref(f)))
    n, v = C.c_int64(), C.c_int32()
    L.check(api.visp_file_n_tensors(f, C.byref(n)))
    assert n.value > 100
    L.check(api.visp_file_get_int(f, b"dino.embed_dim", C.byref(v)))
    assert v.value == synth.MINI.embed_dim
    arr = (C.c_int32 * 4)()
    L.check(api.visp_file_get_int_array(f, b"depthanything.feature_layers", arr, 4))
    assert list(arr) == list(synth.MINI.feature_layers)
    need = C.c_int64()
    L.check(api.visp_file_get_string(f, b"general.architecture", None, 0, C.byref(need)))
    buf = C.create_string_buffer(need.value)
    L.check(api.visp_file_get_string(f, b"general.architecture", buf, need.value, None))
    assert buf.value == b"depthanything"
    assert api.visp_file_get_int(f, b"dino.no_such_key", C.byref(v)) == 0 and b"dino.no_such_key" in api.visp_get_last_error()
    assert api.visp_file_get_int_array(f, b"depthanything.feature_layers", arr, 3) == 0  # wrong length
    assert api.visp_file_get_int(None, b"x", C.byref(v)) == 0 and b"null" in api.visp_get_last_error()
    w = C.c_void_p()
    L.check(api.visp_weights_from_file(f, C.byref(w)))
    g = C.c_void_p()
    L.check(api.visp_graph_create(w, C.byref(g)))
    for call in (lambda: api.visp_graph_find_weight(g, b"head.conv1.weight", None), lambda: api.visp_graph_get_tensor(g, b"x", None),
                 lambda: api.visp_graph_input(g, 0, (C.c_int64 * 4)(3, 14, 14, 1), b"in", None),
                 lambda: api.visp_graph_op(g, 5, (C.c_int32 * 1)(0), 1, None, 0, None, 0, None)):
        assert call() == 0 and b"null" in api.visp_get_last_error()
    idx = C.c_int32()
    L.check(api.visp_graph_find_weight(g, b"head.conv1.weight", C.byref(idx)))
    assert api.visp_graph_read_constant(g, idx.value, None, 1 << 30) == 0 and b"null" in api.visp_get_last_error()
    api.visp_graph_destroy(g)
    api.visp_weights_destroy(w)
    api.visp_file_destroy(f)
