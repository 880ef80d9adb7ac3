// DINOv2 backbone on the graph layer (visp/ml.h, visp/nn.h): what the reference builds in src/visp/arch/dino.cpp (declared in
// src/visp/arch/dino.h), module by module under the same weight names. Header-only.
#pragma once

#include <cmath>
#include <span>
#include <vector>

#include "../nn.h"

namespace visp {
namespace dino {

// position embeddings for a w x h image: the stored tensor when the patch grid is the stored (square) one, else cls embedding +
// bicubic resize of the patch embeddings (dino.cpp:10-30). Computed from weights alone: folded on the host, no launch.
inline tensor interpolate_pos_encoding(model_ref m, tensor x, int64_t w, int64_t h, int patch_size) {
    tensor pos_embed = m.weights("position_embeddings");
    const int64_t n_patch = x->ne[1] - 1, n = pos_embed->ne[1] - 1;
    if (n_patch == n && w == h) return pos_embed;
    tensor class_embed = slice(m, pos_embed, {}, {0}, {}, {});
    tensor patches = slice(m, pos_embed, {}, {1, n + 1}, {}, {});
    const int64_t dim = x->ne[0], side = int64_t(std::sqrt(float(n)) + 0.01f);
    const i64x2 target = {w / patch_size, h / patch_size};
    patches = reshape_4d(m, patches, dim, side, side, 1);
    patches = interpolate(m, patches, target, GGML_SCALE_MODE_BICUBIC);
    patches = reshape_3d(m, patches, dim, target[0] * target[1], 1);
    return concat(m, {class_embed, patches}, 1);
}

inline tensor prepare_tokens(model_ref m, tensor x, int patch_size) { // dino.cpp:32-46
    auto [c, w, h, n] = nelements(x);
    x = patch_embed(m["patch_embeddings"], x, patch_size);
    x = reshape_3d(m, x, x->ne[0], x->ne[1] * x->ne[2], x->ne[3]);
    tensor cls_token = m.weights("cls_token");
    if (cls_token->ne[2] != n) cls_token = repeat_4d(m, cls_token, cls_token->ne[0], 1, n, 1);
    x = concat(m, {cls_token, x}, 1);
    return add(m, x, interpolate_pos_encoding(m, x, w, h, patch_size));
}

inline tensor layer_scale(model_ref m, tensor x) { return mul(m, x, m.weights("lambda1")); } // dino.cpp:48-50

inline tensor mlp(model_ref m, tensor x) { // dino.cpp:52-57
    x = linear(m["fc1"], x);
    x = gelu(m, x);
    return linear(m["fc2"], x);
}

inline tensor self_attention(model_ref m, tensor x, int n_heads) { // dino.cpp:59-74
    auto [c, n, b, _] = nelements(x);
    auto project = [&, c = c, n = n, b = b](model_ref mp, tensor t) { return reshape_4d(mp, linear(mp, t), c / n_heads, n_heads, n, b); };
    tensor q = project(m["attention.query"], x), k = project(m["attention.key"], x), v = project(m["attention.value"], x);
    const float scale_ = 1.0f / std::sqrt(float(c) / float(n_heads));
    return attention(m, q, k, v, nullptr, scale_, m["output.dense"]);
}

inline tensor layer(model_ref m, tensor x, dino_params const& p) { // dino.cpp:76-90
    tensor attn = layer_norm(m["norm1"], x, 1e-6f);
    attn = self_attention(m["attention"], attn, p.n_heads);
    x = add(m, x, layer_scale(m["layer_scale1"], attn));
    tensor ffn = layer_norm(m["norm2"], x, 1e-6f);
    ffn = mlp(m["mlp"], ffn);
    x = add(m, x, layer_scale(m["layer_scale2"], ffn));
    return named(m, x);
}

} // namespace dino

// the outputs of the listed layers, each through the shared final layernorm, named "dino_layer_<i>" (dino.cpp:92-110)
inline std::vector<tensor> dino_get_intermediate_layers(model_ref m, tensor x, std::span<const int> layers, dino_params const& p) {
    x = dino::prepare_tokens(m["embeddings"], x, p.patch_size);
    std::vector<tensor> outputs;
    model_ref encoder = m["encoder.layer"];
    for (int i = 0; i < p.n_layers; ++i) {
        x = dino::layer(encoder[i], x, p);
        bool wanted = false;
        for (int l : layers) wanted = wanted || l == i;
        if (wanted) outputs.push_back(set_name(m, layer_norm(m["layernorm"], x, 1e-6f), ("dino_layer_" + std::to_string(i)).c_str()));
    }
    return outputs;
}

} // namespace visp
