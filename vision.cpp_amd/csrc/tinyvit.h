// TinyViT image encoder of MobileSAM on the MI355X backend (BASELINE.json configs[4]): model
// load (mobile-sam GGUF -> packed f16 weights), static schedule, batched executor. Mirrors sam_load_model / sam_encode
// (reference src/visp/vision.cpp:26-52; graph src/visp/arch/mobile-sam.cpp:20-215, sam_process_input :533-547), and the
// prompt encoder + mask decoder + mask post-processing behind sam_compute (vision.cpp:54-92, mobile-sam.cpp:207-583).
#pragma once
#include <vector>

#include "depthany.h"

namespace visp {

struct tiny_vit_layer { int resolution, embed_dim, depth, num_heads, window_size; bool downsample; };
struct tiny_vit_params { // mobile-sam.h:16-37
    int img_size = 1024;
    tiny_vit_layer layers[4] = {{256, 64, 2, 2, 7, true}, {128, 128, 2, 4, 7, true}, {64, 160, 6, 5, 14, true}, {64, 320, 2, 10, 7, false}};
};

struct packed_dw { size_t w = 0, b = 0; int C = 0; };   // depthwise 3x3: f16 [9][C] + f32 bias [C]
struct tv_mbconv_weights { packed_gemm conv1, conv3; packed_dw conv2; size_t conv3_frag = SIZE_MAX; }; // conv3_frag: vx_mbconv_pack_w3 image
struct tv_merge_weights { packed_gemm conv1, conv3; packed_dw conv2; int stride = 2; };
// opt-in e4m3 image of a linear weight (kernels_gemm_fp8.hip): rows [N pad 128][K pad 128] + one f32 scale per row, offsets into sam_model::fp8_arena
struct fp8_linear { size_t w = SIZE_MAX, s = SIZE_MAX; int N = 0, Kp = 0; };
struct tv_block_weights {
    packed_vec attn_ln_w, attn_ln_b, bias; // bias: attention_biases_indexed packed f16 (vx_window_attention_pack_bias), n = f16 count
    packed_gemm qkv, proj, fc1, fc2;
    fp8_linear fc1_e4m3, fc2_e4m3;
    packed_dw local_conv;
    packed_vec mlp_ln_w, mlp_ln_b;
};
struct tinyvit_weights {
    packed_gemm pe0, pe2; // patch_embed.seq.0 (3 -> 32, input channels 3..5 repeat 0..2), seq.2
    std::vector<tv_mbconv_weights> mbconv;
    tv_merge_weights merge[3];
    std::vector<tv_block_weights> blocks[4];
    packed_gemm neck0, neck2;
    packed_vec neck1_w, neck1_b, neck3_w, neck3_b;
};

// prompt encoder + mask decoder (mobile-sam.cpp:207-483); present when the file holds the dec.* / prompt_encoder.* tensors
struct sam_attn_weights { packed_gemm q, k, v, o; };
struct sam_twoway_weights {
    sam_attn_weights self_attn, t2i, i2t;
    packed_gemm lin1, lin2;
    packed_vec norm_w[4], norm_b[4];
};
struct samdec_weights {
    bool present = false;
    int dim = 256, heads = 8, res = 64;
    std::vector<float> gaussian;                       // host: positional_encoding_gaussian_matrix [2][dim/2]
    std::vector<float> point_embed[4], not_a_point;    // host: label embeddings [dim]
    std::vector<float> output_tokens;                  // host: [iou_token | mask_tokens] = [5][dim]
    packed_vec no_mask, dense_pe;                      // arena f16: [dim], [res*res][dim] (n = f16 count)
    std::vector<sam_twoway_weights> layers;
    sam_attn_weights final_attn;
    packed_vec final_norm_w, final_norm_b;
    packed_gemm up0, up3;                              // conv_transpose k2 s2 as GEMM + pixel shuffle
    packed_vec up_norm_w, up_norm_b;
    int up_c1 = 0, up_c2 = 0;
    packed_gemm hyper[4][3], iou_head[3];
    // The prompt encoder's tables above and the iou head are host arithmetic: their f32 values ALSO live in the weight arena
    // (tables_off: gaussian | point_embed[4] | not_a_point | output_tokens = 11 dim floats; the iou head as its packed GEMMs),
    // so that a rank whose arena arrived by RCCL broadcast (load_no_upload) rebuilds them in sam_weights_ready.
    size_t tables_off = SIZE_MAX;
    std::vector<float> iou_w[3], iou_b[3]; // host f32 copies [n][k] of the f16-rounded iou head weights, biases
};

struct sam_model : model_base { // vision.h sam_model counterpart (encoder part)
    sam_model() : model_base(family_sam) {}
    backend_device const* backend = nullptr;
    tiny_vit_params params;
    tinyvit_weights weights;
    device_buffer weight_arena;
    bool weights_uploaded = false;
    device_buffer ws;
    samdec_weights dec;
    device_buffer dec_ws;         // decoder scratch
    std::vector<float> last_masks; // [4][mask_size^2] logits of the last sam_compute (f16-rounded), and their iou predictions
    float last_iou[4] = {0, 0, 0, 0};
    device_buffer embed;          // image embedding of the last sam_encode: f32 [64, 64, 256] (NHWC)
    i32x2 image_extent = {{0, 0}}; // extent of the image passed to sam_encode (vision.h sam_model::image_extent)
    bool timing = false, captures = false;
    // BASELINE.json configs[4] ("fp8 GGUF weights on CDNA4 fp8 MFMA"), opt-in and never the default: the MLPs of the transformer stages
    // (mobile-sam.cpp:150-160: fc1 + gelu, fc2 + residual -- 39 % of the encoder's time) on the block-scaled e4m3 matrix instruction, activations
    // quantised per token on the way in. The decision on accuracy stands (tests/test_fp8_decision.py: mask IoU below the bar); this is the measured form of it.
    bool fp8_mlp = false;
    device_buffer fp8_arena, fp8_ws;
    std::vector<timing_entry> last_timing;
    std::map<std::string, capture_entry> capture_bufs;
    ~sam_model();
};
void sam_set_fp8_mlp(sam_model&, bool enable); // builds the e4m3 weight images on first use (from the f16 weights on the device)

sam_model* sam_load_model(char const* filepath, backend_device const& dev, int flags = load_default);
void sam_weights_ready(sam_model&);
// B images rgb_u8 [B, 1024, 1024, 3] already at the model extent, on the device -> image embeddings f32 [B, 64, 64, 256] (NHWC)
void sam_encode_batch_device(sam_model&, void const* rgb_dev, int batch, void* out_dev, void* stream);
void sam_encode_batch_host(sam_model&, uint8_t const* rgb, int batch, float* out);
// reference API (vision.cpp:36-52): any extent, any u8 colour format; longest side scaled to 1024, edge-replicated to the
// square (sam_process_input); the embedding stays on the device in model.embed
void sam_encode(sam_model&, image_view image);
// reference API (vision.cpp:54-92): point (n = 2) or box (n = 4) in pixels of the image given to sam_encode -> alpha_u8 mask
// at that image's extent (best of the first three masks by predicted iou)
image_data sam_compute(sam_model&, int const* prompt, int n);

} // namespace visp
