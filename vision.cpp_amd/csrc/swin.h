// SWIN transformer encoder -- the backbone of BiRefNet -- on the MI355X backend: model load (the
// "bb.*" tensors of a birefnet GGUF -> packed f16 weights), static schedule, batched executor. Mirrors swin_detect_params /
// swin_precompute / swin_encode (reference src/visp/arch/swin.cpp:264-319, include/visp/vision.h swin_params). The BiRefNet
// decoder is not part of this backend yet, so the family's visp_model_compute stays refused; the encoder has its own entry
// points (include/visp_c_api.h visp_swin_*).
#pragma once
#include <vector>

#include "depthany.h"

namespace visp {

// per-kernel-group timing of one pass (HIP events between launches), shared by the swin and birefnet executors
struct timing_marks {
    std::vector<std::pair<std::string, void*>> marks;
    std::vector<timing_entry> acc;
    void mark(const char* name, double flops, double bytes, void* stream);
    void finish(void* stream, std::vector<timing_entry>& out, bool append = false);
};

struct swin_layer_t { int depth, n_heads, n_features; };
struct swin_params { // vision.h swin_params; swin_t_params / swin_l_params of swin.cpp:264-291
    int embed_dim = 96, window_size = 7;
    swin_layer_t layers[4] = {{2, 3, 96}, {2, 6, 192}, {6, 12, 384}, {2, 24, 768}};
};
swin_params swin_detect_params(model_file const&); // swin.cpp:293-302 (+ the layer table keys of this repo's test files)

struct swin_block_weights {
    packed_vec norm1_w, norm1_b, norm2_w, norm2_b;
    packed_vec bias;             // 4 packed f16 images of vx_swin_attention_pack_bias (n = f16 count)
    packed_gemm qkv, proj, fc1, fc2; // qkv rows re-ordered per head as q | k | v (the window attention kernel's layout)
};
struct swin_weights {
    packed_gemm patch_embed;     // 4x4 stride-4 conv on the value + residue input pixels (8 channels)
    packed_vec pe_norm_w, pe_norm_b;
    bool pe_norm = false;
    std::vector<swin_block_weights> blocks[4];
    packed_vec merge_norm_w[3], merge_norm_b[3];
    packed_gemm merge_reduction[3];
    packed_vec out_norm_w[4], out_norm_b[4];
};

struct swin_model : model_base { // the encoder half of vision.h birefnet_model
    swin_model() : model_base(family_birefnet) {}
    bool full = false; // true: this object is a birefnet_model (birefnet.h) -- encoder + decoder
    backend_device const* backend = nullptr;
    swin_params params;
    swin_weights weights;
    device_buffer weight_arena;
    device_buffer ws;
    bool timing = false, captures = false;
    // Shift-mask semantics. The reference's swin::layer hands the layer's attn_mask to EVERY block (swin.cpp:226-237) and
    // swin::block forwards it to window_attention unconditionally (:128-139), so the -inf edge mask of compute_attention_mask
    // (:165-210) also acts in the UNSHIFTED blocks (the windows of the last row / column). That is what the reference computes,
    // and the default here. Its torch twin (tests/test_birefnet.py:249-255, the original Swin) masks shifted blocks only:
    // mask_shifted_only = true (visp_swin_set_mask_mode(m, 1) / VISP_SWIN_MASK_SHIFTED_ONLY=1 at load) selects that.
    bool mask_shifted_only = false;
    std::vector<timing_entry> last_timing;
    std::map<std::string, capture_entry> capture_bufs;
    virtual ~swin_model();
};

swin_model* swin_load_model(char const* filepath, backend_device const& dev, char const* prefix = "bb");
void swin_load_into(swin_model& model, model_file const& file, backend_device const& dev, char const* prefix); // weights of an open file

// one output of the encoder: row stride ld (elements; 0 = C), f32 or f16 -- lets a caller write a stage straight into a wider
// concatenation buffer (birefnet::encode_concat)
struct swin_out { void* ptr = nullptr; int ld = 0; bool f32 = true; };
// the encoder on already pre-processed pixels: in8 = f16 [B, h, w, 8] (value + residue, vx_tv_preprocess layout)
void swin_encode_pixels(swin_model& m, void const* in8, int B, int w, int h, swin_out const outs[4], void* stream, void const* rgb_dev = nullptr);
// dims[i] = {w_i, h_i, C_i} of the four outputs for an image of extent (w, h)
void swin_output_dims(swin_model const& m, int w, int h, int dims[4][3]);
// rgb_u8 [B, h, w, 3] in device memory -> outs[i] f32 [B, h_i, w_i, C_i] (NHWC = the reference's CWHN tensors); w, h multiples of
// 32 (patch 4, three even merges). stream NULL: the device's stream, synchronised before returning.
void swin_encode_batch_device(swin_model& m, void const* rgb_dev, int B, int w, int h, void* const outs[4], void* stream);
void swin_encode_batch_host(swin_model& m, uint8_t const* rgb, int B, int w, int h, float* const outs[4]);

} // namespace visp
