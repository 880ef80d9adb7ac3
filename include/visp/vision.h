// visp/vision.h -- the source-level C++ API of the MI355X backend, shaped like the reference's public header
// (reference include/visp/vision.h:124-347, include/visp/image.h:17-110, include/visp/ml.h:32-64): same namespace, type and
// function names, argument meaning and error behaviour (exceptions carrying the library's message), so that a C++ caller of
// the reference's high-level API -- scripts/pkg-check/main.cpp:22-44 is the model -- builds against this backend unchanged.
// Header-only over the binary-stable C ABI (include/visp_c_api.h, lib/libvisioncpp.so): link with -lvisioncpp.
//
// Covered: backend_init / backend_device, image_view / image_data / image_alloc / image_clear / image_scale / image_u8_to_f32 /
// image_normalize, per family *_load_model + *_compute for depth_anything, esrgan, birefnet and sam (sam_encode + sam_compute with a
// point or a box), and the Depth-Anything pipeline pieces depthany_params / depthany_detect_params / depthany_image_extent /
// depthany_process_input / depthany_process_output (vision.h:236-252).
// The ml.h layer -- model_file, model_weights, compute_graph, model_ref, tensor, transfer_* -- is visp/ml.h, the nn.h builders visp/nn.h.
// The graph builders of the reference's arch sources (dino::*, dpt::*, depthany_predict) are DECLARED in visp/builders.h and defined by
// those sources themselves: src/visp/arch/dino.cpp and depth-anything.cpp compile unmodified against these headers with
// -DVISP_GGML_NAMES -DVISP_ARCH_FROM_SOURCE (tests/test_reference_sources_compile.py). With VISP_ARCH_FROM_SOURCE the pipeline pieces
// the reference defines in depth-anything.cpp (image_extent / process_input / process_output) are declared here, not defined.
// Not covered: the *_predict builders of the other families (hand schedules only), migan_* (family not built).
#pragma once

#include <array>
#include <cstdint>
#include <cstring>
#include <exception>
#include <memory>
#include <span>
#include <string>
#include <utility>

#include "../visp_c_api.h"

namespace visp {
using std::byte;
using std::span;
using f32x4 = std::array<float, 4>; // per-channel offset / scale of image_u8_to_f32 (arithmetic: util/math.h)

struct exception : std::exception { // reference src/util/... visp::exception: what() = the library's message
    std::string message;
    explicit exception(std::string m) : message(std::move(m)) {}
    char const* what() const noexcept override { return message.c_str(); }
};
namespace detail {
inline void check(int32_t ok) {
    if (!ok) throw exception(visp_get_last_error());
}
} // namespace detail

//
// images (include/visp/image.h)

struct i32x2 {
    int32_t v[2] = {0, 0};
    constexpr i32x2() = default;
    constexpr i32x2(int32_t x, int32_t y) : v{x, y} {}
    constexpr int32_t& operator[](int i) { return v[i]; }
    constexpr int32_t operator[](int i) const { return v[i]; }
    friend constexpr bool operator==(i32x2 a, i32x2 b) { return a.v[0] == b.v[0] && a.v[1] == b.v[1]; }
    friend constexpr bool operator!=(i32x2 a, i32x2 b) { return !(a == b); }
};
struct box_2d { i32x2 top_left, bottom_right; }; // sam_compute box prompt (vision.h:139-160)

enum class image_format : int32_t { rgba_u8, bgra_u8, argb_u8, rgb_u8, alpha_u8, rgba_f32, rgb_f32, alpha_f32 };
constexpr int n_channels(image_format f) {
    switch (f) {
        case image_format::rgb_u8: case image_format::rgb_f32: return 3;
        case image_format::alpha_u8: case image_format::alpha_f32: return 1;
        default: return 4;
    }
}
constexpr bool is_float(image_format f) { return int(f) >= int(image_format::rgba_f32); }
constexpr int n_bytes(image_format f) { return n_channels(f) * (is_float(f) ? 4 : 1); }

struct image_data;
struct image_view { // include/visp/image.h:37-41: non-owning
    i32x2 extent;
    int32_t stride = 0;
    image_format format = image_format::rgba_u8;
    void const* data = nullptr;

    image_view() = default;
    image_view(i32x2 e, image_format f, void const* d) : extent(e), stride(e[0] * n_bytes(f)), format(f), data(d) {}
    image_view(i32x2 e, int32_t s, image_format f, void const* d) : extent(e), stride(s), format(f), data(d) {}
    image_view(image_data const& img);
    image_view(i32x2 e, std::span<float const> d) : image_view(e, image_format::alpha_f32, d.data()) {} // one float per pixel
    std::span<uint8_t const> as_bytes() const { return {static_cast<uint8_t const*>(data), size_t(stride) * size_t(extent[1])}; }
    std::span<float const> as_floats() const { return {static_cast<float const*>(data), size_t(extent[0]) * size_t(extent[1]) * size_t(n_channels(format))}; }
};
struct image_data { // owning; include/visp/image.h:60-64
    i32x2 extent;
    image_format format = image_format::rgba_u8;
    std::unique_ptr<uint8_t[]> data;
};
inline image_view::image_view(image_data const& img) : image_view(img.extent, img.format, img.data.get()) {}

inline image_data image_alloc(i32x2 extent, image_format format) {
    size_t n = size_t(extent[0]) * size_t(extent[1]) * size_t(n_bytes(format));
    return image_data{extent, format, std::unique_ptr<uint8_t[]>(new uint8_t[n ? n : 1])};
}
inline void image_clear(image_data& img) { std::memset(img.data.get(), 0, size_t(img.extent[0]) * size_t(img.extent[1]) * size_t(n_bytes(img.format))); }

namespace detail {
inline visp_image_view c_view(image_view const& v) { return {v.extent[0], v.extent[1], v.stride, int32_t(v.format), const_cast<void*>(v.data)}; }
inline image_data take(visp_image_view const& v, visp_image_data* owner) { // copies the library-owned result into an image_data
    image_data out = image_alloc({v.width, v.height}, image_format(v.format));
    const size_t row = size_t(v.width) * size_t(n_bytes(out.format));
    for (int y = 0; y < v.height; ++y) std::memcpy(out.data.get() + y * row, static_cast<uint8_t const*>(v.data) + size_t(y) * size_t(v.stride), row);
    visp_image_destroy(owner);
    return out;
}
} // namespace detail

inline image_data image_scale(image_view const& img, i32x2 target) { // src/visp/image.cpp:352-356 (stb_image_resize semantics)
    visp_image_view in = detail::c_view(img), out{};
    visp_image_data* owner = nullptr;
    detail::check(visp_image_scale(&in, target[0], target[1], &out, &owner));
    return detail::take(out, owner);
}

// image.cpp:215-255: dst = (src / 255 + offset) * scale per channel, any u8 format -> the float format `format`
inline image_data image_u8_to_f32(image_view const& img, image_format format, std::array<float, 4> offset = {0, 0, 0, 0}, std::array<float, 4> scale = {1, 1, 1, 1}) {
    visp_image_view in = detail::c_view(img), out{};
    visp_image_data* owner = nullptr;
    detail::check(visp_image_u8_to_f32(&in, int32_t(format), offset.data(), scale.data(), &out, &owner));
    return detail::take(out, owner);
}
// image.cpp:537-582: min-max of an alpha_f32 image mapped to [min, max]
inline image_data image_normalize(image_view const& img, float min = 0, float max = 1) {
    visp_image_view in = detail::c_view(img), out{};
    visp_image_data* owner = nullptr;
    detail::check(visp_image_normalize(&in, min, max, &out, &owner));
    return detail::take(out, owner);
}

//
// backend (include/visp/ml.h:32-64)

enum class backend_type : int32_t { cpu = 1, gpu = 2, vulkan = gpu | 1 << 8 };

struct backend_device {
    visp_device* handle = nullptr;
    backend_device() = default;
    explicit backend_device(visp_device* h) : handle(h) {}
    backend_device(backend_device&& o) noexcept : handle(std::exchange(o.handle, nullptr)) {}
    backend_device& operator=(backend_device&& o) noexcept { std::swap(handle, o.handle); return *this; }
    ~backend_device() { if (handle) visp_device_destroy(handle); }
    backend_type type() const { return backend_type(visp_device_type(handle)); }
    char const* name() const { return visp_device_name(handle); }
    char const* description() const { return visp_device_description(handle); }
};
inline backend_device backend_init() { // first available device (here: the first gfx950 GPU)
    visp_device* d = nullptr;
    detail::check(visp_device_init(VISP_BACKEND_AUTO, &d));
    return backend_device(d);
}
inline backend_device backend_init(backend_type t) { // backend_type::cpu throws: this build has no CPU backend
    visp_device* d = nullptr;
    detail::check(visp_device_init(int32_t(t), &d));
    return backend_device(d);
}
inline bool backend_is_available(backend_type t) { return t == backend_type::gpu; }

//
// models (include/visp/vision.h)

namespace detail {
template <int Family>
struct model_handle {
    visp_model* handle = nullptr;
    model_handle() = default;
    explicit model_handle(visp_model* h) : handle(h) {}
    model_handle(model_handle&& o) noexcept : handle(std::exchange(o.handle, nullptr)) {}
    model_handle& operator=(model_handle&& o) noexcept { std::swap(handle, o.handle); return *this; }
    ~model_handle() { if (handle) visp_model_destroy(handle, Family); }
};
template <int Family>
inline visp_model* load(char const* filepath, backend_device const& dev) {
    visp_model* m = nullptr;
    check(visp_model_load(filepath, dev.handle, Family, &m));
    return m;
}
template <int Family>
inline image_data compute(visp_model* m, image_view const& image, int32_t* args = nullptr, int32_t n_args = 0) {
    visp_image_view in = c_view(image), out{};
    visp_image_data* owner = nullptr;
    check(visp_model_compute(m, Family, &in, 1, args, n_args, &out, &owner));
    return take(out, owner);
}
} // namespace detail

// Depth-Anything (vision.h:224-252, 339-347). The device must outlive the model.
using depthany_model = detail::model_handle<VISP_DEPTH_ANYTHING>;
inline depthany_model depthany_load_model(char const* filepath, backend_device const& dev) { return depthany_model(detail::load<VISP_DEPTH_ANYTHING>(filepath, dev)); }
inline image_data depthany_compute(depthany_model& model, image_view image) { // -> alpha_f32 in [0, 1] at the input extent (vision.cpp:147-167)
    visp_image_view in = detail::c_view(image), out{};
    visp_image_data* owner = nullptr;
    detail::check(visp_depthany_compute_f32(model.handle, &in, &out, &owner));
    return detail::take(out, owner);
}

// --- Depth Anything pipeline (vision.h:236-252, arch/depth-anything.cpp:112-149). The reference reads the parameters from a
// model_file; this backend has no GGUF surface in its public header, so they come from the loaded model.
struct dino_params { int patch_size = 14, embed_dim = 384, n_layers = 12, n_heads = 6; }; // vision.h dino_params
struct depthany_params {
    int image_size = 518;
    int image_multiple = 14;
    i32x2 image_extent = {518, 518};
    float max_depth = 1;
    std::array<int, 4> feature_layers = {2, 5, 8, 11};
    dino_params dino;
};
// --- ESRGAN pipeline (vision.h:294-304): scale factor and number of RRDB blocks, from the GGUF keys esrgan.scale / esrgan.block_count
struct esrgan_params {
    int scale = 4;
    int n_blocks = 23;
};
#ifdef VISP_ARCH_FROM_SOURCE
// the reference's own definitions (src/visp/arch/depth-anything.cpp:112-149) are part of the build
i32x2 depthany_image_extent(i32x2 input_extent, depthany_params const&);
image_data depthany_process_input(image_view image, depthany_params const&);
image_data depthany_process_output(std::span<float const> output_data, i32x2 target_extent, depthany_params const&);
#else
// round the short side up to a multiple of image_multiple (at least image_size), keep the aspect ratio, round both up (:112-117)
inline i32x2 depthany_image_extent(i32x2 extent, depthany_params const& p) {
    auto next_multiple = [](int x, int m) { return (x + m - 1) / m * m; };
    const int min_side = extent[0] < extent[1] ? extent[0] : extent[1];
    const int tgt = next_multiple(min_side, p.image_multiple) > p.image_size ? next_multiple(min_side, p.image_multiple) : p.image_size;
    return i32x2(next_multiple(extent[0] * tgt / min_side, p.image_multiple), next_multiple(extent[1] * tgt / min_side, p.image_multiple));
}
#endif
inline depthany_params depthany_detect_params(depthany_model const& model, i32x2 input_extent = {}) { // (:119-128)
    visp_depthany_info info{};
    detail::check(visp_depthany_get_info(model.handle, &info));
    depthany_params p;
    p.image_size = info.image_size;
    p.image_multiple = info.image_multiple;
    p.max_depth = info.max_depth;
    for (int i = 0; i < 4; ++i) p.feature_layers[size_t(i)] = info.feature_layers[i];
    p.dino = dino_params{info.patch_size, info.embed_dim, info.n_layers, info.n_heads};
    if (input_extent[0] > 0 && input_extent[1] > 0) p.image_extent = depthany_image_extent(input_extent, p);
    return p;
}
#ifndef VISP_ARCH_FROM_SOURCE
// image_scale to the model extent where it differs, then (u8 / 255 - mean) / std -> rgb_f32 (:130-140)
inline image_data depthany_process_input(image_view image, depthany_params const& p) {
    image_data resized;
    if (image.extent[0] != p.image_extent[0] || image.extent[1] != p.image_extent[1]) {
        resized = image_scale(image, p.image_extent);
        image = image_view(resized);
    }
    return image_u8_to_f32(image, image_format::rgb_f32, {-0.485f, -0.456f, -0.406f, 0.f}, {1.f / 0.229f, 1.f / 0.224f, 1.f / 0.225f, 1.f});
}
// min-max normalise the raw depth at the model extent, scale to the caller's extent where it differs (:142-149)
inline image_data depthany_process_output(std::span<float const> output_data, i32x2 target_extent, depthany_params const& p) {
    image_data normalized = image_normalize(image_view(p.image_extent, image_format::alpha_f32, output_data.data()));
    if (normalized.extent[0] != target_extent[0] || normalized.extent[1] != target_extent[1]) return image_scale(normalized, target_extent);
    return normalized;
}
#endif

// ESRGAN (vision.h:284-304): any size, tiled, -> rgba_u8 at scale x the input extent (vision.cpp:220-253)
using esrgan_model = detail::model_handle<VISP_ESRGAN>;
inline esrgan_model esrgan_load_model(char const* filepath, backend_device const& dev) { return esrgan_model(detail::load<VISP_ESRGAN>(filepath, dev)); }
inline image_data esrgan_compute(esrgan_model& model, image_view image) { return detail::compute<VISP_ESRGAN>(model.handle, image); }

// BiRefNet (vision.h birefnet_*; vision.cpp:98-132): any 8-bit colour image -> alpha_u8 foreground mask at the input extent
using birefnet_model = detail::model_handle<VISP_BIREFNET>;
inline birefnet_model birefnet_load_model(char const* filepath, backend_device const& dev) { return birefnet_model(detail::load<VISP_BIREFNET>(filepath, dev)); }
inline image_data birefnet_compute(birefnet_model& model, image_view image) { return detail::compute<VISP_BIREFNET>(model.handle, image); }

// MobileSAM (vision.h:139-160): sam_encode once per image, then any number of prompts (vision.cpp:26-92) -> alpha_u8 mask
using sam_model = detail::model_handle<VISP_SAM>;
inline sam_model sam_load_model(char const* filepath, backend_device const& dev) { return sam_model(detail::load<VISP_SAM>(filepath, dev)); }
inline void sam_encode(sam_model& model, image_view image) {
    visp_image_view in = detail::c_view(image);
    detail::check(visp_sam_encode(model.handle, &in));
}
namespace detail {
inline image_data sam_prompt(sam_model& model, int32_t const* prompt, int32_t n) {
    visp_image_view out{};
    visp_image_data* owner = nullptr;
    check(visp_sam_compute(model.handle, prompt, n, &out, &owner));
    return take(out, owner);
}
} // namespace detail
inline image_data sam_compute(sam_model& model, i32x2 point) {
    int32_t p[2] = {point[0], point[1]};
    return detail::sam_prompt(model, p, 2);
}
inline image_data sam_compute(sam_model& model, box_2d box) {
    int32_t p[4] = {box.top_left[0], box.top_left[1], box.bottom_right[0], box.bottom_right[1]};
    return detail::sam_prompt(model, p, 4);
}

} // namespace visp
