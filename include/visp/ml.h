// visp/ml.h -- the graph layer of the MI355X backend with the reference's vocabulary (reference include/visp/ml.h:85-300): model
// weights by name, compute_graph, model_ref (a graph + a name prefix: `m["encoder.layer"][3]["norm1"]`), compute_graph_input / _output,
// transfer_to_backend / transfer_from_backend, slice / concat / interpolate. Header-only over the C ABI (`visp_weights_*`,
// `visp_graph_*` in include/visp_c_api.h); the builders of src/visp/nn.h are in visp/nn.h.
//
// What the reference implements on ggml (graph container, gallocr, one backend kernel per node) is this library's own executor
// (csrc/graph.cpp): nn-level nodes lowered once into fused launches, constants folded on the host, a liveness arena in HBM.
// Differences a caller sees, all in how a graph comes to be, none in how it is built or run:
//   * model_load / model_init / model_transfer exist with the reference's roles (ml.cpp:206-217, 285-301, 449-516); model_load_weights(path)
//     is the three in one call. Tensors are uploaded when the first graph that uses them is allocated, packed for the kernel that reads
//     them, and shared by later graphs;
//   * compute_graph_init takes the weights its model_refs look names up in (ggml: a second context passed to model_ref);
//   * `tensor` is a value handle {graph, index, shape}: `x->ne[1]` and `nelements(x)` read as in the reference, there is no ggml_tensor;
//   * 2D maps are CWHN ([C, W, H, N], model_build_flag::cwhn): the layout helpers of nn.h are identities, interpolate takes CWHN.
// Source compatibility is checked, not claimed: tests/test_reference_sources_compile.py compiles the reference's src/visp/arch/dino.cpp and
// depth-anything.cpp where they lie, unmodified, against this header set and builds Depth-Anything through them.
#pragma once

#include <array>
#include <cstddef>
#include <initializer_list>
#include <limits>
#include <string>
#include <vector>

#include "vision.h"

namespace visp {

using byte = std::byte;
using tensor_name = std::string;                 // reference: fixed_string<128> (util/string.h)
using i64x2 = std::array<int64_t, 2>;
using i64x4 = std::array<int64_t, 4>;

// the two ggml names the reference's callers spell out (compute_graph_input(m, GGML_TYPE_F32, ...)) and its interpolate modes
constexpr int32_t GGML_TYPE_F32 = 0, GGML_TYPE_F16 = 1;
constexpr int32_t GGML_SCALE_MODE_NEAREST = 0, GGML_SCALE_MODE_BILINEAR = 1, GGML_SCALE_MODE_BICUBIC = 2, GGML_SCALE_FLAG_ALIGN_CORNERS = 1 << 8;

enum class model_build_flag : uint32_t { cwhn = 1 << 0, conv_2d_direct_cwhn = 1 << 1, concat_n = 1 << 2, f16_conv_transpose = 1 << 3, window_partition = 1 << 4, flash_attention = 1 << 5 };
// a set of model_build_flag bits: `flags |= model_build_flag::cwhn`, `flags & model_build_flag::cwhn ? a : b` (ml.h:66-80)
struct model_build_flags {
    uint32_t bits = 0;
    constexpr model_build_flags() = default;
    constexpr model_build_flags(model_build_flag f) : bits(uint32_t(f)) {}
    constexpr explicit model_build_flags(uint32_t b) : bits(b) {}
    constexpr explicit operator bool() const { return bits != 0; }
    constexpr bool has(model_build_flag f) const { return (bits & uint32_t(f)) != 0; }
    constexpr model_build_flags& operator|=(model_build_flags o) { bits |= o.bits; return *this; }
    constexpr model_build_flags& operator&=(model_build_flags o) { bits &= o.bits; return *this; }
    friend constexpr model_build_flags operator|(model_build_flags a, model_build_flags b) { return model_build_flags(a.bits | b.bits); }
    friend constexpr model_build_flags operator&(model_build_flags a, model_build_flags b) { return model_build_flags(a.bits & b.bits); }
    friend constexpr model_build_flags operator~(model_build_flags a) { return model_build_flags(~a.bits); }
    friend constexpr bool operator==(model_build_flags a, model_build_flags b) { return a.bits == b.bits; }
};
constexpr model_build_flags operator|(model_build_flag a, model_build_flag b) { return model_build_flags(a) | model_build_flags(b); }
constexpr model_build_flags operator~(model_build_flag a) { return ~model_build_flags(a); }
// what this backend builds with: NHWC maps, fused attention (ml.cpp:166-186 backend_default_flags)
inline model_build_flags backend_default_flags(backend_type) { return model_build_flag::cwhn | model_build_flag::flash_attention; }

//
// tensors

struct tensor_shape {
    int64_t ne[4] = {1, 1, 1, 1};
    int32_t type = GGML_TYPE_F16;
};
struct tensor {
    visp_graph* graph = nullptr;
    int32_t id = -1;
    tensor_shape shape;
    tensor() = default;
    tensor(std::nullptr_t) {}
    tensor(visp_graph* g, int32_t i) : graph(g), id(i) {
        if (g && i >= 0) detail::check(visp_graph_tensor_info(g, i, &shape.type, shape.ne, nullptr));
    }
    tensor_shape const* operator->() const { return &shape; } // x->ne[1], x->type
    explicit operator bool() const { return id >= 0; }
    friend bool operator==(tensor const& a, tensor const& b) { return a.graph == b.graph && a.id == b.id; }
};
inline std::array<int64_t, 4> nelements(tensor t) { return {t->ne[0], t->ne[1], t->ne[2], t->ne[3]}; } // auto [c, w, h, n] = nelements(x)
inline int64_t n_elements(tensor t) { return t->ne[0] * t->ne[1] * t->ne[2] * t->ne[3]; }

//
// model file (ml.h:85-103): a GGUF file in memory, key/values by name. get_int wants an i32 value, get_array an i32 array of exactly
// out.size() entries; a missing key throws with its name (ml.cpp:223-254).

struct model_file {
    visp_file* handle = nullptr;
    std::string path;
    model_file() = default;
    model_file(visp_file* h, std::string p) : handle(h), path(std::move(p)) {}
    model_file(model_file&& o) noexcept : handle(std::exchange(o.handle, nullptr)), path(std::move(o.path)) {}
    model_file& operator=(model_file&& o) noexcept { std::swap(handle, o.handle); std::swap(path, o.path); return *this; }
    ~model_file() { if (handle) visp_file_destroy(handle); }

    int64_t n_tensors() const { int64_t n = 0; detail::check(visp_file_n_tensors(handle, &n)); return n; }
    int get_int(char const* name) const { int32_t v = 0; detail::check(visp_file_get_int(handle, name, &v)); return v; }
    void get_array(char const* name, std::span<int> out_values) const { detail::check(visp_file_get_int_array(handle, name, out_values.data(), int64_t(out_values.size()))); }
    std::string get_string(char const* name) const {
        int64_t need = 0;
        detail::check(visp_file_get_string(handle, name, nullptr, 0, &need));
        std::string v(size_t(need), '\0');
        detail::check(visp_file_get_string(handle, name, v.data(), need, nullptr));
        v.resize(size_t(need) - 1);
        return v;
    }
    std::string arch() const { return get_string("general.architecture"); }
};
inline model_file model_load(char const* filepath) {
    visp_file* f = nullptr;
    detail::check(visp_file_load(filepath, &f));
    return model_file(f, filepath);
}

//
// model weights (ml.h:105-149)

struct model_weights {
    visp_weights* handle = nullptr;
    model_weights() = default;
    explicit model_weights(visp_weights* h) : handle(h) {}
    model_weights(model_weights&& o) noexcept : handle(std::exchange(o.handle, nullptr)) {}
    model_weights& operator=(model_weights&& o) noexcept { std::swap(handle, o.handle); return *this; }
    ~model_weights() { if (handle) visp_weights_destroy(handle); }
};
inline model_weights model_load_weights(char const* filepath) {
    visp_weights* w = nullptr;
    detail::check(visp_weights_load(filepath, &w));
    return model_weights(w);
}
inline model_weights model_init(size_t /*n_tensors*/ = 0) { // an empty store; model_transfer / model_add_tensor fill it (ml.cpp:285-301)
    visp_weights* w = nullptr;
    detail::check(visp_weights_create(&w));
    return model_weights(w);
}
// the file's f16 / f32 tensors become the model's weights (ml.cpp:449-516). Nothing is uploaded here: device images are made per consumer
// role when a graph over the weights is allocated, so the device argument of the reference's signature is optional.
inline void model_transfer(model_file const& file, model_weights& weights) {
    visp_weights* w = nullptr;
    detail::check(visp_weights_from_file(file.handle, &w));
    weights = model_weights(w);
}
inline void model_transfer(model_file const& file, model_weights& weights, backend_device const&) { model_transfer(file, weights); }
inline void model_add_tensor(model_weights& w, char const* name, int32_t type, i64x4 ne, std::span<float const> data) {
    detail::check(visp_weights_add(w.handle, name, type, ne.data(), data.data()));
}

//
// compute graph (ml.h:167-182)

struct compute_graph {
    visp_graph* handle = nullptr;
    compute_graph() = default;
    explicit compute_graph(visp_graph* h) : handle(h) {}
    compute_graph(compute_graph&& o) noexcept : handle(std::exchange(o.handle, nullptr)) {}
    compute_graph& operator=(compute_graph&& o) noexcept { std::swap(handle, o.handle); return *this; }
    ~compute_graph() { if (handle) visp_graph_destroy(handle); }
    explicit operator bool() const { return handle != nullptr; }
};
inline compute_graph compute_graph_init(model_weights& weights) {
    visp_graph* g = nullptr;
    detail::check(visp_graph_create(weights.handle, &g));
    return compute_graph(g);
}
// lowers the graph to launches, uploads the weights it uses, plans and allocates its arena (ml.cpp:545-552). Throws on failure.
inline bool compute_graph_allocate(compute_graph& g, backend_device const& dev) {
    detail::check(visp_graph_allocate(g.handle, dev.handle));
    return true;
}
// lower and plan only: the launch list and the arena size without a device (compute_graph_describe shows them)
inline void compute_graph_plan(compute_graph& g) { detail::check(visp_graph_allocate(g.handle, nullptr)); }
inline void compute(compute_graph const& g, backend_device const&) { detail::check(visp_graph_compute(g.handle)); } // blocks until done (ml.cpp:559-562)
inline std::string compute_graph_describe(compute_graph const& g) { // one line per launch + the arena summary
    int64_t need = 0;
    detail::check(visp_graph_describe(g.handle, nullptr, 0, &need));
    std::string s(size_t(need), '\0');
    detail::check(visp_graph_describe(g.handle, s.data(), need, nullptr));
    s.resize(size_t(need) - 1);
    return s;
}

//
// model_ref (ml.h:199-245)

struct model_ref {
    compute_graph* graph = nullptr; // `*m.graph` is what the reference hands to ggml_build_forward_expand
    model_build_flags flags = model_build_flag::cwhn | model_build_flag::flash_attention;
    tensor_name prefix;

    model_ref() = default;
    model_ref(compute_graph& g) : graph(&g) {}
    explicit model_ref(compute_graph* g, model_build_flags f = model_build_flag::cwhn | model_build_flag::flash_attention, tensor_name p = {}) : graph(g), flags(f), prefix(std::move(p)) {}

    visp_graph* handle() const { return graph ? graph->handle : nullptr; }
    tensor_name full(char const* name) const { return prefix.empty() ? tensor_name(name) : prefix + "." + name; }
    tensor find(char const* name) const { // null tensor if not found
        int32_t id = -1;
        detail::check(visp_graph_find_weight(handle(), full(name).c_str(), &id));
        return id < 0 ? tensor() : tensor(handle(), id);
    }
    tensor weights(char const* name) const { // throws if not found (the reference asserts)
        tensor t = find(name);
        if (!t) throw exception("tensor not found: " + full(name));
        return t;
    }
    model_ref with_prefix(tensor_name new_prefix) const { return model_ref(graph, flags, std::move(new_prefix)); }
    model_ref operator[](char const* sub_module) const { return with_prefix(full(sub_module)); }
    model_ref operator[](tensor_name const& sub_module) const { return with_prefix(full(sub_module.c_str())); }
    model_ref operator[](int sub_module) const { return with_prefix(full(std::to_string(sub_module).c_str())); }
};

inline tensor named(model_ref const& m, tensor t) { // ml.cpp:640-643: the tensor takes the current prefix as its name
    detail::check(visp_graph_set_name(m.handle(), t.id, m.prefix.c_str()));
    return t;
}
inline tensor set_name(model_ref const& m, tensor t, char const* name) { // ggml_set_name / ggml_format_name
    detail::check(visp_graph_set_name(m.handle(), t.id, name));
    return t;
}
inline tensor get_tensor(model_ref const& m, char const* name) { // ggml_get_tensor
    int32_t id = -1;
    detail::check(visp_graph_get_tensor(m.handle(), name, &id));
    return id < 0 ? tensor() : tensor(m.handle(), id);
}
inline tensor compute_graph_input(model_ref const& m, int32_t type, i64x4 ne, tensor_name name = "input") {
    int32_t id = -1;
    detail::check(visp_graph_input(m.handle(), type, ne.data(), name.c_str(), &id));
    return tensor(m.handle(), id);
}
inline tensor compute_graph_output(model_ref const& m, tensor t, tensor_name name = "output") {
    detail::check(visp_graph_output(m.handle(), t.id, name.c_str()));
    return t;
}

namespace detail {
inline uint16_t f32_to_f16(float f) { // IEEE binary16, round to nearest even
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u, ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) return uint16_t(sign | 0x7c00u | (ax > 0x7f800000u ? 0x200u : 0u));
    if (ax >= 0x477ff000u) return uint16_t(sign | 0x7c00u);
    if (ax < 0x33000001u) return uint16_t(sign);
    const int e = int(ax >> 23) - 127;
    const uint32_t man = (ax & 0x7fffffu) | 0x800000u;
    const int shift = e < -14 ? 13 + (-14 - e) : 13;
    uint32_t q = man >> shift;
    const uint32_t rem = man & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) ++q;
    return uint16_t(sign | (e < -14 ? q : (uint32_t(e + 14) << 10) + q));
}
inline tensor op(model_ref const& m, int32_t kind, std::initializer_list<tensor> src, std::initializer_list<int64_t> ip = {}, std::initializer_list<float> fp = {}) {
    int32_t ids[4] = {-1, -1, -1, -1}, n = 0, id = -1;
    for (tensor const& t : src) ids[n++] = t.id;
    check(visp_graph_op(m.handle(), kind, ids, n, ip.begin(), int32_t(ip.size()), fp.begin(), int32_t(fp.size()), &id));
    return tensor(m.handle(), id);
}
} // namespace detail

//
// tensor data and transfer (ml.h:154-193)

struct tensor_data {
    tensor x;
    std::unique_ptr<byte[]> data;
    size_t n_bytes = 0;
    std::span<float> as_f32() { return {reinterpret_cast<float*>(data.get()), n_bytes / 4}; }
    std::span<float const> as_f32() const { return {reinterpret_cast<float const*>(data.get()), n_bytes / 4}; }
    std::span<byte> as_bytes() { return {data.get(), n_bytes}; }
};
inline size_t n_bytes(tensor x) { return size_t(n_elements(x)) * (x->type == GGML_TYPE_F32 ? 4 : 2); }
inline tensor_data tensor_alloc(tensor x) { // host memory for a tensor's f32 image
    tensor_data d{x, std::unique_ptr<byte[]>(new byte[size_t(n_elements(x)) * 4]), size_t(n_elements(x)) * 4};
    return d;
}
inline void transfer_to_backend(tensor x, std::span<byte const> data) { detail::check(visp_graph_tensor_set(x.graph, x.id, data.data(), data.size())); }
inline void transfer_to_backend(tensor x, std::span<float const> data) { // f32 host data; an f16 tensor converts
    if (x->type == GGML_TYPE_F32) { detail::check(visp_graph_tensor_set(x.graph, x.id, data.data(), data.size_bytes())); return; }
    std::vector<uint16_t> h(data.size());
    for (size_t i = 0; i < h.size(); ++i) h[i] = detail::f32_to_f16(data[i]);
    detail::check(visp_graph_tensor_set(x.graph, x.id, h.data(), h.size() * 2));
}
inline void transfer_to_backend(tensor x, image_view const& img) { // a float image into an f32 tensor [C, W, H, 1] (ml.cpp:708-716)
    if (!is_float(img.format)) throw exception("transfer_to_backend: the image must be a float format");
    transfer_to_backend(x, img.as_floats());
}
inline void transfer_to_backend(tensor_data const& d) { transfer_to_backend(d.x, d.as_f32()); }
inline void transfer_from_backend(tensor x, std::span<float> dst) { detail::check(visp_graph_tensor_get(x.graph, x.id, dst.data(), dst.size_bytes(), 1)); }
inline tensor_data transfer_from_backend(tensor x) { // the tensor's values as f32
    tensor_data d = tensor_alloc(x);
    transfer_from_backend(x, d.as_f32());
    return d;
}

//
// tensor operations (ml.h:262-293, ml.cpp:746-788)

struct slice_t {
    int64_t begin, end, step;
    static constexpr int64_t max = std::numeric_limits<int64_t>::max() / 4;
    constexpr slice_t() : begin(0), end(max), step(1) {}
    constexpr slice_t(int64_t index) : begin(index), end(index + 1), step(1) {}
    constexpr slice_t(int64_t b, int64_t e, int64_t s = 1) : begin(b), end(e), step(s) {}
};
// `x[0, 0:64, 16:32, :]` (numpy order) is slice(m, x, {}, {16, 32}, {0, 64}, 0). Copies (the reference returns a view).
inline tensor slice(model_ref const& m, tensor x, slice_t s0, slice_t s1 = {}, slice_t s2 = {}, slice_t s3 = {}) {
    return detail::op(m, VISP_OP_SLICE, {x}, {s0.begin, s0.end, s0.step, s1.begin, s1.end, s1.step, s2.begin, s2.end, s2.step, s3.begin, s3.end, s3.step});
}
inline tensor concat(model_ref const& m, std::initializer_list<tensor> src, int dim) { // n-ary: folded left to right
    tensor out;
    for (tensor const& t : src) {
        if (!t) continue;
        out = out ? detail::op(m, VISP_OP_CONCAT, {out, t}, {dim}) : t;
    }
    return out;
}
// up- or downsample a CWHN map to target = {w, h}
inline tensor interpolate(model_ref const& m, tensor x, i64x2 target, int32_t mode) { return detail::op(m, VISP_OP_INTERPOLATE, {x}, {target[0], target[1], mode}); }

} // namespace visp
