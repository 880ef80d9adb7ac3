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
