"""Row b3 of SURVEY section 8 (the executor boundary the reference's arch code is written against), checked instead of claimed: the
reference's own src/visp/arch/dino.cpp, src/visp/arch/depth-anything.cpp and (round 4) src/visp/arch/esrgan.cpp are compiled WHERE THEY LIE, unmodified, against this
repository's include tree (include/visp/{ml,nn,vision,builders}.h, include/visp/arch/*.h forwarders, include/util/*.h), linked with
lib/libvisioncpp.so, and run: Depth-Anything built by the reference's depthany_predict lowers to the same launch list as the graph the
Python face builds for the same file, and the reference's process_input / process_output / image_extent agree with this backend's.

Build-container only: /root/reference does not exist on the GPU box, where this module skips. Objects go to a temporary directory;
no reference text is stored in the repository. This is a drop-in check of the API, not an oracle: what runs underneath is this
backend's own graph layer, and nothing here is used as a parity reference."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle
from visioncpp_amd import graph as G
from visioncpp_amd import synth, vision

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference/src/visp/arch")
SOURCES = [REF / "dino.cpp", REF / "depth-anything.cpp"]
ESRGAN_SOURCE = REF / "esrgan.cpp"
FLAGS = ["-std=c++20", "-O1", "-Wall", "-DVISP_GGML_NAMES", "-DVISP_ARCH_FROM_SOURCE", "-I", str(ROOT / "include")]

pytestmark = pytest.mark.skipif(not all(s.exists() for s in SOURCES), reason="the reference tree is only present in the build container")


def test_reference_arch_sources_pass_the_front_end_unmodified():
    for src in SOURCES + [ESRGAN_SOURCE]:
        r = subprocess.run(["g++", *FLAGS, "-fsyntax-only", str(src)], capture_output=True, text=True)
        assert r.returncode == 0 and "warning" not in r.stderr, f"{src.name}:\n{r.stderr}"


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    """dino.o + depth-anything.o from the reference sources + tests/cpp/arch_source_driver.cpp + libvisioncpp.so"""
    out = tmp_path_factory.mktemp("arch_from_source")
    objs = []
    for src in SOURCES:
        obj = out / (src.stem + ".o")
        subprocess.run(["g++", *FLAGS, "-c", str(src), "-o", str(obj)], check=True)
        objs.append(str(obj))
    exe = out / "arch_source_driver"
    lib_dir = ROOT / "vision.cpp_amd" / "lib"
    subprocess.run(["g++", *FLAGS, str(ROOT / "tests" / "cpp" / "arch_source_driver.cpp"), *objs, "-o", str(exe), "-L", str(lib_dir), "-lvisioncpp",
                    f"-Wl,-rpath,{lib_dir}"], check=True)
    return exe


def _python_launch_list(path, cfg, w, h):
    g = G.Graph(None, G.Weights(path))
    m = G.ModelRef(g)
    image = g.input((3, w, h, 1), G.F32)
    G.depthany_predict(m, image, cfg.n_layers, cfg.n_heads, cfg.patch_size, cfg.feature_layers)
    g.allocate()
    return g.describe()


# the north star's widths (384 / 1536, 6 heads of 64) with four layers and four distinct taps (the reference asserts on 4 features)
WIDE = synth.Config(embed_dim=384, n_layers=4, n_heads=6, image_size=112, feature_layers=(0, 1, 2, 3), name="wide4")


@pytest.mark.parametrize("cfg_name,extent", [("MINI", (70, 56)), ("WIDE", (112, 112)), ("MINI", (98, 42))])
def test_depth_anything_built_by_the_reference_sources(driver, tmp_path, cfg_name, extent):
    cfg = WIDE if cfg_name == "WIDE" else getattr(synth, cfg_name)
    path = synth.write_gguf(tmp_path / "m.gguf", cfg, seed=3)
    r = subprocess.run([str(driver), str(path), str(extent[0]), str(extent[1])], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert lines[0].startswith("arch=depthanything")
    # depthany_detect_params + depthany_image_extent as the reference defines them vs this backend's own (csrc/depthany.cpp)
    want = oracle.depthany_image_extent(extent[0], extent[1], cfg.image_size, 14)
    taps = ",".join(str(v) for v in cfg.feature_layers)
    assert lines[1] == (f"params patch={cfg.patch_size} dim={cfg.embed_dim} layers={cfg.n_layers} heads={cfg.n_heads} size={cfg.image_size} "
                        f"taps={taps} extent={want[0]}x{want[1]}")
    assert lines[2] == f"output ne=1,{want[0]},{want[1]},1"
    n_tok = (want[0] // cfg.patch_size) * (want[1] // cfg.patch_size) + 1
    for i, layer in enumerate(cfg.feature_layers):
        assert lines[3 + i] == f"tap dino_layer_{layer} ne={cfg.embed_dim},{n_tok},1"
    # the graph the reference's builders made lowers to the launch list of the graph the Python face makes for the same file
    # (the summary line differs in what it may: the reference marks the four taps as graph outputs, so their buffers are persistent)
    got, ref = [ln for ln in lines[9:] if ln.strip()], _python_launch_list(path, cfg, *want).strip().splitlines()
    assert got[:-1] == ref[:-1] and len(got) > 30
    assert got[-1].split()[0] == ref[-1].split()[0] and got[-1].startswith("launches=")


def test_host_steps_of_the_reference_sources_agree_with_the_library(driver, tmp_path):
    """process_input / process_output compiled from the reference's file call this backend's image_scale, image_u8_to_f32 and
    image_normalize; the sums must equal what the library's own pipeline pieces give for the same bytes."""
    cfg = synth.MINI
    path = synth.write_gguf(tmp_path / "m.gguf", cfg, seed=3)
    w, h = 90, 64
    r = subprocess.run([str(driver), str(path), str(w), str(h)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = {ln.split()[0]: ln for ln in r.stdout.splitlines() if ln.startswith("process_")}
    ext = oracle.depthany_image_extent(w, h, cfg.image_size, 14)
    i = np.arange(w * h * 3, dtype=np.int64)
    image = ((i * 37 + (i >> 5) * 11) & 255).astype(np.uint8).reshape(h, w, 3)
    scaled = vision.image_scale(image, ext[0], ext[1])
    mean, std = np.float32([0.485, 0.456, 0.406]), np.float32([0.229, 0.224, 0.225])
    want_in = ((scaled.astype(np.float32) / np.float32(255.0) - mean) * (np.float32(1.0) / std)).astype(np.float64).sum()
    s_in = float(got["process_input"].split("sum=")[1])
    assert got["process_input"].startswith(f"process_input {ext[0]}x{ext[1]} ") and abs(s_in - want_in) < 1e-3 * max(1.0, abs(want_in))
    k = np.arange(ext[0] * ext[1], dtype=np.int64)
    raw = ((k * 13) % 1009).astype(np.float32) * np.float32(0.25) + np.float32(3.0)
    norm = (raw - raw.min()) / (raw.max() - raw.min())
    back = vision.image_scale(norm.reshape(ext[1], ext[0]), w, h, vision.ImageFormat.alpha_f32)
    s_out = float(got["process_output"].split("sum=")[1])
    assert got["process_output"].startswith(f"process_output {w}x{h} ") and abs(s_out - float(back.astype(np.float64).sum())) < 1e-3 * back.size


# ---- the ESRGAN generator (round 4: leaky_relu, NEAREST interpolate, channel concat, scale / add chains in the graph layer) --------------------------

@pytest.fixture(scope="module")
def esrgan_driver(tmp_path_factory):
    """esrgan.o from the reference source + tests/cpp/esrgan_source_driver.cpp + libvisioncpp.so"""
    out = tmp_path_factory.mktemp("esrgan_from_source")
    obj = out / "esrgan.o"
    subprocess.run(["g++", *FLAGS, "-c", str(ESRGAN_SOURCE), "-o", str(obj)], check=True)
    exe = out / "esrgan_source_driver"
    lib_dir = ROOT / "vision.cpp_amd" / "lib"
    subprocess.run(["g++", *FLAGS, str(ROOT / "tests" / "cpp" / "esrgan_source_driver.cpp"), str(obj), "-o", str(exe), "-L", str(lib_dir), "-lvisioncpp",
                    f"-Wl,-rpath,{lib_dir}"], check=True)
    return exe


@pytest.mark.parametrize("cfg_name,tile", [("ESRGAN_TINY", (40, 56, 3)), ("ESRGAN_X4", (48, 32, 2))])
def test_esrgan_built_by_the_reference_source(esrgan_driver, tmp_path, cfg_name, tile):
    """esrgan_detect_params + esrgan_generate as src/visp/arch/esrgan.cpp defines them: the graph they build lowers to the launch list of the graph the
    Python face builds for the same file -- the planar dense-block schedule (tests/test_graph_cpu.py says what that list is)."""
    cfg = getattr(synth, cfg_name)
    path = tmp_path / "e.gguf"
    synth.write_esrgan_gguf(path, cfg, 7)
    w, h, n = tile
    r = subprocess.run([str(esrgan_driver), str(path), str(w), str(h), str(n)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert lines[0] == f"arch=esrgan scale={cfg.scale} blocks={cfg.num_blocks} graph_size={512 + cfg.num_blocks * 192}"
    assert lines[1] == f"result ne=3,{w * cfg.scale},{h * cfg.scale},{n}"
    g = G.Graph(None, G.Weights(path))
    G.esrgan_generate(G.ModelRef(g), g.input((3, w, h, n), G.F32), cfg.scale, cfg.num_blocks)
    g.allocate()
    got, ref = [ln for ln in lines[2:] if ln.strip()], g.describe().strip().splitlines()
    assert got == ref and len(got) == 1 + 1 + 15 * cfg.num_blocks + 1 + int(np.log2(cfg.scale)) + 2 + 1


def test_esrgan_detect_params_refuses_another_architecture(esrgan_driver, tmp_path):
    path = synth.write_gguf(tmp_path / "m.gguf", synth.MINI, seed=3)
    r = subprocess.run([str(esrgan_driver), str(path), "16", "16", "1"], capture_output=True, text=True)
    assert r.returncode == 1 and "Architecture expected to be 'esrgan', but was 'depthanything'" in r.stderr
