// visp/builders.h -- declarations of the graph builders that the reference DEFINES in its arch sources (src/visp/arch/dino.cpp,
// src/visp/arch/depth-anything.cpp, src/visp/arch/esrgan.cpp) on top of visp/ml.h + visp/nn.h. This backend ships no copy of those definitions: an integrator
// compiles the reference's two files against this include tree (-DVISP_GGML_NAMES -DVISP_ARCH_FROM_SOURCE) and links libvisioncpp.so;
// tests/test_reference_sources_compile.py does exactly that and checks the launch list the resulting graph lowers to. The library's own
// Depth-Anything path (visp_model_load / visp_model_compute) builds the same node graph internally (csrc/depthany_graph.cpp).
#pragma once

#include <span>
#include <vector>

#include "ml.h"

namespace visp {

// DINOv2 backbone: parameters from the GGUF keys dino.*; the token features after the listed layers, each through the final LayerNorm
dino_params dino_detect_params(model_file const&);
std::vector<tensor> dino_get_intermediate_layers(model_ref, tensor image, std::span<int const> layer_ids, dino_params const&);

// Depth-Anything: parameters from depthanything.* (+ the extent rule when an input extent is given); image [3, W, H, N] f32 -> depth [1, W, H, N]
depthany_params depthany_detect_params(model_file const&, i32x2 input_extent = {});
tensor depthany_predict(model_ref, tensor image, depthany_params const&);

// ESRGAN / Real-ESRGAN generator (src/visp/arch/esrgan.cpp): parameters from esrgan.*; image [3, W, H, N] f32 -> [3, W * scale, H * scale, N] f32
esrgan_params esrgan_detect_params(model_file const&);
int esrgan_estimate_graph_size(esrgan_params const&);
tensor esrgan_generate(model_ref, tensor image, esrgan_params const&);

namespace esrgan { // module level (src/visp/arch/esrgan.h)
tensor upsample(model_ref m, tensor x);
tensor conv_block(model_ref m, tensor x);
tensor risidual_dense_block(model_ref m, tensor x);
tensor rrdb(model_ref m, tensor x);
} // namespace esrgan

namespace dino { // module level, prefixes as in the HF state dict
tensor interpolate_pos_encoding(model_ref m, tensor x, int64_t w, int64_t h, int patch_size);
tensor prepare_tokens(model_ref m, tensor x, int patch_size);
tensor layer_scale(model_ref m, tensor x);
tensor mlp(model_ref m, tensor x);
tensor self_attention(model_ref m, tensor x, int n_heads);
tensor layer(model_ref m, tensor x, dino_params const& p);
std::vector<tensor> get_intermediate_layers(model_ref m, tensor x, std::span<int const> layers, dino_params const& p);
} // namespace dino

namespace dpt { // neck + head; `size` = {.., w, h, ..} in the ne order of a CWHN map, null: twice the input
tensor residual_conv(model_ref m, tensor x);
tensor feature_fusion(model_ref m, tensor x0, tensor x1, int64_t const* size = nullptr);
tensor neck(model_ref m, std::span<tensor> features, int64_t patch_w, int64_t patch_h);
tensor head(model_ref m, tensor fused, int64_t w, int64_t h, float max_depth);
} // namespace dpt

} // namespace visp
