// Depth-Anything-V2 (DPT neck + head on DINOv2 features) on the graph layer: what the reference builds in
// src/visp/arch/depth-anything.cpp:15-110, module by module under the same weight names; depthany_predict is declared in the
// reference's include/visp/vision.h:246-248. Header-only. The hand-scheduled step behind depthany_compute (csrc/depthany.cpp) is
// the fast path of this model; this is the same model through the generic executor.
#pragma once

#include "dino.h"

namespace visp {
namespace dpt {

constexpr int32_t bilinear_align_corners = GGML_SCALE_MODE_BILINEAR | GGML_SCALE_FLAG_ALIGN_CORNERS;

inline tensor residual_conv(model_ref m, tensor x) { // depth-anything.cpp:15-23; lowers to conv[relu-in][relu] + conv[+res]
    tensor out = relu(m, x);
    out = conv_2d(m["convolution1"], out, 1, 1);
    out = relu(m, out);
    out = conv_2d(m["convolution2"], out, 1, 1);
    return named(m, add(m, x, out));
}

inline tensor feature_fusion(model_ref m, tensor x0, tensor x1, int64_t const* size) { // depth-anything.cpp:25-42; size = the next level's ne
    tensor x = x0;
    if (x1) x = add(m, x, residual_conv(m["residual_layer1"], x1));
    x = residual_conv(m["residual_layer2"], x);
    const int64_t w = size ? size[1] : x->ne[1] * 2, h = size ? size[2] : x->ne[2] * 2;
    x = interpolate(m, x, {w, h}, bilinear_align_corners);
    return named(m, conv_2d(m["projection"], x));
}

inline tensor neck(model_ref m, std::span<tensor> features, int64_t patch_w, int64_t patch_h) { // depth-anything.cpp:44-79
    if (features.size() != 4) throw exception("dpt::neck: expected 4 feature maps");
    std::array<tensor, 4> layer;
    model_ref reassemble = m["reassemble_stage.layers"];
    for (int i = 0; i < 4; ++i) {
        tensor x = features[size_t(i)];
        x = slice(m, x, {}, {1, x->ne[1]}, {}, {}); // drop the cls token
        x = reshape_4d(m, x, x->ne[0], patch_w, patch_h, x->ne[2]);
        x = conv_2d(reassemble[i]["projection"], x); // 1x1: a plain product on the token rows
        switch (i) {
            case 0: x = conv_transpose_2d(reassemble[i]["resize"], x, 4); break;
            case 1: x = conv_transpose_2d(reassemble[i]["resize"], x, 2); break;
            case 3: x = conv_2d(reassemble[i]["resize"], x, 2, 1); break;
        }
        layer[size_t(i)] = x;
    }
    model_ref convs = m["convs"];
    for (int i = 0; i < 4; ++i) layer[size_t(i)] = conv_2d(convs[i], layer[size_t(i)], 1, 1);
    model_ref fusion = m["fusion_stage.layers"];
    tensor fused = feature_fusion(fusion[0], layer[3], nullptr, layer[2]->ne);
    fused = feature_fusion(fusion[1], fused, layer[2], layer[1]->ne);
    fused = feature_fusion(fusion[2], fused, layer[1], layer[0]->ne);
    return feature_fusion(fusion[3], fused, layer[0], nullptr);
}

inline tensor head(model_ref m, tensor x, int64_t w, int64_t h, float max_depth) { // depth-anything.cpp:81-96
    tensor out = conv_2d(m["conv1"], x, 1, 1);
    out = interpolate(m, out, {w, h}, bilinear_align_corners);
    out = relu(m, conv_2d(m["conv2"], out, 1, 1));
    out = relu(m, conv_2d(m["conv3"], out)); // one channel: an f32 map
    if (max_depth != 1) out = scale(m, out, max_depth);
    return out;
}

} // namespace dpt

// image [3, w, h, n] f32 (normalised: depthany_process_input) -> depth [1, w, h, n] f32, marked as the graph's output
inline tensor depthany_predict(model_ref m, tensor image, depthany_params const& p) { // depth-anything.cpp:100-110
    auto [c, w, h, n] = nelements(image);
    const int64_t w_patch = w / p.dino.patch_size, h_patch = h / p.dino.patch_size;
    std::vector<tensor> features = dino_get_intermediate_layers(m["backbone"], image, p.feature_layers, p.dino);
    tensor fused = dpt::neck(m["neck"], features, w_patch, h_patch);
    tensor depth = dpt::head(m["head"], fused, w, h, p.max_depth);
    return compute_graph_output(m, depth);
}

} // namespace visp
