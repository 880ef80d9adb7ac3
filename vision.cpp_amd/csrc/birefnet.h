// BiRefNet (dichotomous segmentation; SURVEY section 8f ranks 3-4, BASELINE.json configs[3]) on the MI355X backend: the SWIN
// encoder of swin.h run at two scales, the squeeze block and the decoder with deformable-convolution ASPP blocks. Mirrors
// birefnet_load_model / birefnet_compute (reference src/visp/vision.cpp:98-132), birefnet_detect_params / image_extent /
// process_input / process_output / predict (src/visp/arch/birefnet.cpp:252-323).
#pragma once
#include "swin.h"

namespace visp {

struct birefnet_params { // vision.h birefnet_params
    int image_size = 1024; // -1: dynamic (input extent rounded up to image_multiple)
    int image_multiple = 128;
    i32x2 image_extent = {{1024, 1024}};
};

struct bf_deform_weights { packed_gemm offmod, conv; int k = 1; }; // offsets | modulator logits from one conv; kernel weights x bn scale
struct bf_block_weights {  // basic_decoder_block (birefnet.cpp:144-150)
    packed_gemm conv_in, conv_out, gap, conv1;
    bf_deform_weights aspp[4]; // aspp1, aspp_deforms.0..2: kernels 1, 1, 3, 7
    int cin = 0, inter = 0, planes = 0, cout = 0;
};
struct bf_ipt_weights { packed_gemm conv1, conv_out; int cout = 0; };
struct birefnet_weights {
    bf_block_weights squeeze, block[4];      // block[i] = decoder.block(4 - i)
    bf_ipt_weights ipt[5];                   // ipt[i] = decoder.ipt_blk(5 - i): grids 32, 16, 8, 4, 1
    packed_gemm lateral[3], gdt[3], gdt_attn[3]; // [i] = level 4 - i
    packed_gemm conv_out1;
};

struct birefnet_model : swin_model { // vision.h birefnet_model counterpart; the swin_model part is the encoder
    birefnet_model() { full = true; }
    birefnet_params bparams;
    birefnet_weights dec;
    device_buffer dec_arena;
    device_buffer dws; // decoder / feature workspace
    ~birefnet_model() override;
};

birefnet_model* birefnet_load_model(char const* filepath, backend_device const& dev);
i32x2 birefnet_image_extent(i32x2 input_extent, birefnet_params const& p); // birefnet.cpp:283-301 (no allocation cap: 288 GB)
// rgb_u8 [B, h, w, 3] (device) -> sigmoid mask f32 [B, h, w] (device); w, h multiples of 32. stream NULL: own stream + sync.
void birefnet_compute_batch_device(birefnet_model& m, void const* rgb_dev, int B, int w, int h, void* mask_dev, void* stream);
void birefnet_compute_batch_host(birefnet_model& m, uint8_t const* rgb, int B, int w, int h, float* mask);
image_data birefnet_compute(birefnet_model& m, image_view image); // vision.cpp:108-132: any 8-bit colour image -> alpha_u8 mask at its extent

} // namespace visp
