// visp/nn.h -- the neural-network building blocks of the reference's src/visp/nn.h with the same names and argument meaning,
// over this backend's graph layer (visp/ml.h). A module's parameters are looked up under the model_ref's prefix ("weight",
// "bias"), as in the reference. 2D maps are CWHN throughout, so the permute / contiguous helpers are identities.
// Below them: the element-wise and shape ops the reference's arch code calls on ggml directly, under plain names (add, mul, gelu,
// relu, scale, reshape_3d / _4d, repeat_4d, cont), and -- opt-in, #define VISP_GGML_NAMES before including -- under the ggml_* names
// with a model_ref where ggml takes its context, which is how that code passes it.
#pragma once

#include <cstdarg>
#include <cstdio>

#include "ml.h"

namespace visp {

inline tensor linear(model_ref m, tensor x) { // nn.cpp:6-12
    tensor b = m.find("bias");
    return b ? detail::op(m, VISP_OP_LINEAR, {x, m.weights("weight"), b}) : detail::op(m, VISP_OP_LINEAR, {x, m.weights("weight")});
}
inline tensor layer_norm(model_ref m, tensor x, float eps = 1e-5f) { // nn.cpp:14-19
    return detail::op(m, VISP_OP_LAYER_NORM, {x, m.weights("weight"), m.weights("bias")}, {}, {eps});
}

inline bool is_whcn(model_ref m) { return !(m.flags & model_build_flag::cwhn); }
inline bool is_cwhn(model_ref m) { return !!(m.flags & model_build_flag::cwhn); }
inline tensor permute_cwhn_to_whcn(model_ref, tensor x) { return x; }
inline tensor permute_whcn_to_cwhn(model_ref, tensor x) { return x; }
inline tensor cwhn_to_contiguous_2d(model_ref, tensor x) { return x; }
inline tensor whcn_to_contiguous_2d(model_ref, tensor x) { return x; }
inline tensor contiguous_2d_to_cwhn(model_ref, tensor x) { return x; }
inline tensor contiguous_2d_to_whcn(model_ref, tensor x) { return x; }
inline std::array<int64_t, 4> nelements_whcn(model_ref const&, tensor t) { return {t->ne[1], t->ne[2], t->ne[0], t->ne[3]}; }

inline tensor conv_2d(model_ref m, tensor x, int stride = 1, int pad = 0) { // nn.cpp:72-100; weight [Cin, kw, kh, Cout]
    tensor b = m.find("bias");
    return b ? detail::op(m, VISP_OP_CONV_2D, {x, m.weights("weight"), b}, {stride, pad}) : detail::op(m, VISP_OP_CONV_2D, {x, m.weights("weight")}, {stride, pad});
}
inline tensor conv_transpose_2d(model_ref m, tensor x, int stride) { // nn.cpp:117-129; weight [kw, kh, Cout, Cin], kernel == stride
    tensor b = m.find("bias");
    return b ? detail::op(m, VISP_OP_CONV_TRANSPOSE_2D, {x, m.weights("weight"), b}, {stride}) : detail::op(m, VISP_OP_CONV_TRANSPOSE_2D, {x, m.weights("weight")}, {stride});
}
// image [C, W, H, N] f32 -> [D, W / p, H / p, N]: the projection conv with kernel == stride (nn.cpp:166-180; DINOv2 has no norm)
inline tensor patch_embed(model_ref m, tensor x, int patch_size) {
    model_ref p = m["projection"];
    tensor b = p.find("bias");
    tensor out = b ? detail::op(m, VISP_OP_PATCH_EMBED, {x, p.weights("weight"), b}, {patch_size}) : detail::op(m, VISP_OP_PATCH_EMBED, {x, p.weights("weight")}, {patch_size});
    if (m.find("norm.weight")) out = layer_norm(m["norm"], out);
    return out;
}

// element-wise and shape ops
inline tensor add(model_ref const& m, tensor a, tensor b) { return detail::op(m, VISP_OP_ADD, {a, b}); }
inline tensor mul(model_ref const& m, tensor a, tensor b) { return detail::op(m, VISP_OP_MUL, {a, b}); }
inline tensor gelu(model_ref const& m, tensor x) { return detail::op(m, VISP_OP_GELU, {x}); }
inline tensor relu(model_ref const& m, tensor x) { return detail::op(m, VISP_OP_RELU, {x}); }
inline tensor scale(model_ref const& m, tensor x, float s) { return detail::op(m, VISP_OP_SCALE, {x}, {}, {s}); }
inline tensor leaky_relu(model_ref const& m, tensor x, float negative_slope) { return detail::op(m, VISP_OP_LEAKY_RELU, {x}, {}, {negative_slope}); }
inline tensor cont(model_ref const& m, tensor x) { return detail::op(m, VISP_OP_CONT, {x}); }
inline tensor reshape_4d(model_ref const& m, tensor x, int64_t n0, int64_t n1, int64_t n2, int64_t n3) { return detail::op(m, VISP_OP_RESHAPE, {x}, {n0, n1, n2, n3}); }
inline tensor reshape_3d(model_ref const& m, tensor x, int64_t n0, int64_t n1, int64_t n2) { return reshape_4d(m, x, n0, n1, n2, 1); }
inline tensor repeat_4d(model_ref const& m, tensor x, int64_t n0, int64_t n1, int64_t n2, int64_t n3) { return detail::op(m, VISP_OP_REPEAT, {x}, {n0, n1, n2, n3}); }

struct attention_qkv { tensor q, k, v; };
// attention with an optional output linear layer; q, k, v [head_dim, n_heads, n_tokens, batch] (nn.cpp:210-244). Masks belong to the
// window-attention families, which this layer does not express.
inline tensor attention(model_ref m, tensor q, tensor k, tensor v, tensor mask, float scale_, model_ref m_out) {
    if (mask) throw exception("attention: masks are not built in the graph layer");
    tensor x = detail::op(m, VISP_OP_ATTENTION, {q, k, v}, {}, {scale_});
    return linear(m_out, x);
}

#ifdef VISP_GGML_NAMES
// the two calls the reference's arch code makes without a context (dino.cpp:104-105): the tensor knows its graph
inline tensor ggml_format_name(tensor t, char const* fmt, ...) __attribute__((format(printf, 2, 3)));
inline tensor ggml_format_name(tensor t, char const* fmt, ...) {
    char buf[128];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    detail::check(visp_graph_set_name(t.graph, t.id, buf));
    return t;
}
// ggml adds the tensor to the forward graph so that it is computed and kept; here: it becomes an output under its current name
inline void ggml_build_forward_expand(compute_graph& g, tensor t) { detail::check(visp_graph_output(g.handle, t.id, nullptr)); }
inline tensor ggml_add(model_ref const& m, tensor a, tensor b) { return add(m, a, b); }
inline tensor ggml_add_inplace(model_ref const& m, tensor a, tensor b) { return add(m, a, b); }
inline tensor ggml_mul(model_ref const& m, tensor a, tensor b) { return mul(m, a, b); }
inline tensor ggml_gelu(model_ref const& m, tensor x) { return gelu(m, x); }
inline tensor ggml_relu(model_ref const& m, tensor x) { return relu(m, x); }
inline tensor ggml_relu_inplace(model_ref const& m, tensor x) { return relu(m, x); }
inline tensor ggml_scale(model_ref const& m, tensor x, float s) { return scale(m, x, s); }
inline tensor ggml_scale_inplace(model_ref const& m, tensor x, float s) { return scale(m, x, s); }
inline tensor ggml_leaky_relu(model_ref const& m, tensor x, float negative_slope, bool /*inplace*/) { return leaky_relu(m, x, negative_slope); }
inline tensor ggml_cont(model_ref const& m, tensor x) { return cont(m, x); }
inline tensor ggml_reshape_3d(model_ref const& m, tensor x, int64_t n0, int64_t n1, int64_t n2) { return reshape_3d(m, x, n0, n1, n2); }
inline tensor ggml_reshape_4d(model_ref const& m, tensor x, int64_t n0, int64_t n1, int64_t n2, int64_t n3) { return reshape_4d(m, x, n0, n1, n2, n3); }
inline tensor ggml_repeat_4d(model_ref const& m, tensor x, int64_t n0, int64_t n1, int64_t n2, int64_t n3) { return repeat_4d(m, x, n0, n1, n2, n3); }
#endif

} // namespace visp
