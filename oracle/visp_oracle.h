/*
 * visp_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C f32 restatement of what the reference's ggml CPU backend computes on the
 * Depth-Anything-V2 hot path: f16 weights widened to f32 at load
 * (reference src/visp/ml.cpp:115-135, 449-516), CWHN (=NHWC) layout, f32 arithmetic,
 * tanh-GELU through an fp16 look-up table (docs/model-implementation-guide.md:284-288).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library. The product (vision.cpp_amd/csrc) never links, includes or calls it.
 *
 * PINNING: the reference's own implementation (C++ on the un-vendored ggml submodule)
 * cannot be compiled in this environment, and it ships no DINO/DPT module tests.
 * The oracle is pinned (tests/test_oracle_*.py) against
 *   - every literal vector the reference's tests hold for this path
 *     (tests/test-image.cpp:62-184,283-301; tests/test-ml.cpp:18-103),
 *   - the torch functionals the reference's tests/test_primitives.py pins ggml ops to
 *     (linear, layer_norm, interpolate, conv_transpose2d), fixtures in tests/golden/,
 *   - HuggingFace transformers' DepthAnythingForDepthEstimation (the model whose
 *     state-dict names the reference's GGUF uses verbatim, scripts/convert.py:428-475).
 * Whole-path parity against the reference *binary* stays unpinned (no ggml, no GGUF).
 *
 * ESRGAN part (vo_esrgan_*, vo_tile_*): pinned (tests/test_oracle_esrgan.py) against the literal tile_merge
 * vectors of tests/test-image.cpp:303-360 and against outputs of the reference's own torch modules
 * (tests/test_esrgan.py:70-212 ResidualDenseBlock_5C / RRDB / RRDBNet, imported in the CPU container by
 * tests/golden/make_golden_esrgan.py; fixtures tests/golden/esrgan_*.npz).
 *
 * TinyViT part (vo_tinyvit_*, vo_attention_rel_bias, vo_conv2d_depthwise_nhwc): pinned (tests/test_oracle_tinyvit.py)
 * against outputs of the reference's own torch TinyViT (tests/test_mobile_sam.py:18-765, imported in the CPU container by
 * tests/golden/make_golden_tinyvit.py; fixture tests/golden/tinyvit_5m.npz: the 5M configuration at 1024x1024, samples
 * of every stage boundary). Exact (2e-4) with the torch twin's GELU forms, within the reference's module tolerance in
 * the ggml form (tanh GELU everywhere).
 */
#ifndef VISP_ORACLE_H
#define VISP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ggml type ids used by the reference's files (tests/workbench.py:14-19) */
enum { VO_F32 = 0, VO_F16 = 1, VO_I32 = 26 };
/* visp::image_format (include/visp/image.h:17-29) */
enum {
    VO_RGBA_U8 = 0, VO_BGRA_U8, VO_ARGB_U8, VO_RGB_U8, VO_ALPHA_U8,
    VO_RGBA_F32, VO_RGB_F32, VO_ALPHA_F32
};
enum { VO_LAYOUT_UNKNOWN = 0, VO_LAYOUT_WHCN = 1, VO_LAYOUT_CWHN = 2 };
enum { VO_GELU_GGML_F16_LUT = 0, VO_GELU_TANH_F32 = 1, VO_GELU_ERF_F32 = 2 };

/* A tensor as it sits in a GGUF file: ne[0] is the contiguous axis (ggml order). */
typedef struct {
    const char* name;
    const void* data;
    int32_t type;
    int64_t ne[4];
} vo_tensor;

typedef struct {
    int patch_size, embed_dim, n_layers, n_heads; /* dino_params, vision.h:124-129 */
    int image_size, image_multiple;               /* depthany_params, vision.h:236-243 */
    int feature_layers[4];
    float max_depth;
    int gelu_mode;
} vo_depthany_params;

typedef struct vo_model vo_model;

/* Named intermediate capture (the reference workbench's capture idea, tests/workbench.cpp:754-760). */
typedef struct {
    const char* name;
    float* dst;
    int64_t capacity; /* in floats */
    int64_t written;  /* out: number of floats produced (may exceed capacity: nothing copied then) */
} vo_capture;

int vo_num_threads(void);
void vo_set_num_threads(int n);

/* ---- scalar conversions ------------------------------------------------------------ */
void vo_f16_to_f32(const uint16_t* src, float* dst, int64_t n);
void vo_f32_to_f16(const float* src, uint16_t* dst, int64_t n);

/* ---- image ops (src/visp/image.cpp) -------------------------------------------------- */
/* image.cpp:215-255 + image-impl.h:17-34: dst = (src/255 + offset) * scale, clamped tile reads */
int vo_image_u8_to_f32(const uint8_t* src, int sw, int sh, int sstride, int sformat,
                       float* dst, int dw, int dh, int dformat,
                       const float offset[4], const float scale[4], int tile_x, int tile_y);
/* image.cpp:257-288 + image-impl.h:36-43: uint8(clamp(v*scale+offset,0,1)*255) */
int vo_image_f32_to_u8(const float* src, int w, int h, int sformat, uint8_t* dst, int dformat,
                       float scale, float offset);
/* image.cpp:537-576 */
void vo_image_normalize(const float* src, float* dst, int w, int h, int channels, float mn, float mx);
/* image.cpp:328-356 (stb_image_resize v0.97 semantics, see the .c); src / dst tightly packed in `format`; returns 1 on success */
int vo_image_scale(const void* src, int w, int h, int format, void* dst, int ow, int oh);
/* depth-anything.cpp:112-117 */
void vo_depthany_image_extent(int w, int h, int image_size, int image_multiple, int* ow, int* oh);

/* ---- weight transfer (src/visp/ml.cpp:331-340, 449-516) ------------------------------- */
/* Converts one tensor to f32 and, if whcn_to_cwhn, permutes a conv2d kernel
 * [kw,kh,Cin,Cout] -> [Cin,kw,kh,Cout] (depthwise [kw,kh,1,C] -> [C,1,kw,kh]).
 * out_ne receives the permuted shape. dst must hold nelements floats (int32 copied raw). */
int vo_transfer_tensor(const vo_tensor* src, int whcn_to_cwhn, void* dst, int64_t out_ne[4]);

/* ---- primitives (src/visp/nn.cpp; semantics in SURVEY Appendix A) ---------------------- */
/* fp8 what-if (tests/test_fp8_decision.py): 1 = linear inputs rounded to e4m3 per row; not reference behaviour */
void vo_set_linear_act_quant(int mode);
float vo_round_e4m3(float v);
void vo_linear(const float* x, int64_t M, int64_t K, const float* w /*[N][K]*/, const float* b,
               int64_t N, float* y);
void vo_layer_norm(const float* x, int64_t M, int64_t C, const float* w, const float* b, float eps,
                   float* y);
void vo_gelu(const float* x, float* y, int64_t n, int mode);
/* q,k,v: [N][H*hd] (head h at column h*hd), out [N][H*hd]; softmax(q k^T * scale) v */
void vo_attention(const float* q, const float* k, const float* v, int64_t N, int H, int hd,
                  float scale, float* out);
/* NHWC conv, weight [Cout][kh][kw][Cin] (cwhn after transfer), cross-correlation */
void vo_conv2d_nhwc(const float* x, int B, int H, int W, int Cin, const float* w, const float* bias,
                    int Cout, int kh, int kw, int stride, int pad, float* y);
/* torch conv_transpose2d(padding=0); weight torch [Cin][Cout][kh][kw] == ggml ne [kw,kh,Cout,Cin] */
void vo_conv_transpose2d_nhwc(const float* x, int B, int H, int W, int Cin, const float* w,
                              const float* bias, int Cout, int kh, int kw, int stride, float* y);
/* ggml_interpolate BILINEAR (|ALIGN_CORNERS) on NHWC data */
void vo_interpolate_bilinear_nhwc(const float* x, int B, int H, int W, int C, int OH, int OW,
                                  int align_corners, float* y);
/* ggml_interpolate BICUBIC (a = -0.75, torch semantics) on NHWC data */
void vo_interpolate_bicubic_nhwc(const float* x, int B, int H, int W, int C, int OH, int OW,
                                 int align_corners, float* y);

/* ---- model --------------------------------------------------------------------------- */
/* Mirrors model_transfer(file, weights, cpu-device, F32, cwhn): every float tensor -> f32,
 * tensors whose index is listed in conv2d_idx are permuted when src_layout is whcn. */
vo_model* vo_model_create(const vo_tensor* tensors, int n_tensors, const int32_t* conv2d_idx,
                          int n_conv2d, int src_layout);
void vo_model_destroy(vo_model*);
int vo_model_n_tensors(const vo_model*);
/* returns f32 data and shape (post-transfer) or NULL */
const float* vo_model_tensor(const vo_model*, const char* name, int64_t ne[4]);
const char* vo_last_error(void);

/* depthany_predict (depth-anything.cpp:100-110): image = normalised rgb_f32 [h][w][3],
 * w,h multiples of patch_size; out = raw depth [h][w] (before image_normalize). */
int vo_depthany_predict(const vo_model*, const vo_depthany_params*, const float* image, int w, int h,
                        float* out, vo_capture* captures, int n_captures);

/* depthany_compute (vision.cpp:147-167) for an rgb_u8 image whose extent already equals
 * depthany_image_extent(extent): process_input, predict, process_output (min-max to [0,1]). */
int vo_depthany_compute(const vo_model*, const vo_depthany_params*, const uint8_t* rgb, int w, int h,
                        float* out_normalized, float* out_raw /*nullable*/);

/* ---- ESRGAN / Real-ESRGAN (reference src/visp/arch/esrgan.cpp, vision.cpp:208-253) ---- */
typedef struct { int scale, n_blocks; } vo_esrgan_params;
/* esrgan_generate (esrgan.cpp:55-79): x = rgb_f32 [h][w][3] in [0,1] -> out [h*scale][w*scale][3]; tensors are
 * looked up as "model.0.weight", "model.1.sub.<i>.RDB<k>.conv<j>.0.weight", ... exactly as the reference does */
int vo_esrgan_generate(const vo_model*, const vo_esrgan_params*, const float* x, int w, int h, float* out,
                       vo_capture* captures, int n_captures);
/* one residual dense block (esrgan.cpp:27-41), prefix e.g. "model.1.sub.0.RDB1"; x [h][w][nf] in/out */
int vo_esrgan_rdb(const vo_model*, const char* prefix, float* x, int w, int h, int nf);

/* tile_layout (src/visp/image.cpp:612-651) */
typedef struct { int image_w, image_h, overlap_x, overlap_y, n_x, n_y, tile_w, tile_h; } vo_tile_layout;
void vo_tile_layout_init(vo_tile_layout*, int w, int h, int max_tile_size, int overlap, int align);
void vo_tile_scale(const vo_tile_layout* in, int scale, vo_tile_layout* out);
/* tile_merge (image.cpp:653-693): blends an rgb_f32 tile into dst (rgb_f32 image, zero-initialised) */
void vo_tile_merge(const float* tile, float* dst, int tile_x, int tile_y, const vo_tile_layout*);
/* esrgan_compute (vision.cpp:220-253): any u8 colour image -> rgba_u8 [h*scale][w*scale][4] */
int vo_esrgan_compute(const vo_model*, const vo_esrgan_params*, const uint8_t* img, int w, int h, int format,
                      uint8_t* out_rgba);

/* ---- TinyViT image encoder of MobileSAM (reference src/visp/arch/mobile-sam.cpp:20-215) ----
 * GGUF names as scripts/convert.py:204-247 writes them ("enc." prefix, BatchNorm fused into "<conv>.c.weight/.c.bias",
 * "attn.attention_biases_indexed" [heads][N][N]). Layer table as mobile-sam.h:16-37 (fixed in the reference; a
 * parameter here so that small instances can be pinned against the reference's torch modules). */
typedef struct { int resolution, embed_dim, depth, num_heads, window_size, downsample; } vo_tinyvit_layer;
typedef struct { int img_size; vo_tinyvit_layer layers[4]; } vo_tinyvit_params;
/* tiny_vit (mobile-sam.cpp:188-215): image = normalised rgb_f32 [img][img][3]; out = [res3][res3][256] (NHWC) */
int vo_tinyvit_encode(const vo_model*, const char* prefix /* "enc" */, const vo_tinyvit_params*, const float* image, float* out,
                      vo_capture* captures, int n_captures);
/* test knob: GELU form inside MBConv / everywhere else (default: ggml's f16-table tanh form for both) */
void vo_tinyvit_set_gelu_modes(int mbconv_mode, int other_mode);
/* tiny_vit_block (mobile-sam.cpp:133-160) on tokens x [res*res][dim], in place */
int vo_tinyvit_block(const vo_model*, const char* prefix, float* x, int res, int dim, int heads, int window);
/* attention_rel_bias on windows x [n_win][N][dim] -> y same shape (mobile-sam.cpp:122-131, nn.cpp:182-244) */
int vo_attention_rel_bias(const vo_model*, const char* prefix, const float* x, int n_win, int N, int dim, int heads, float* y);
/* conv_2d_depthwise + bias on NHWC (nn.cpp:102-115); weight [C][1][kw][kh] after transfer */
void vo_conv2d_depthwise_nhwc(const float* x, int H, int W, int C, const float* w, const float* bias, int k, int stride, int pad, float* y);

/* ---- MobileSAM prompt encoder + mask decoder (sam_compute: vision.cpp:54-84, mobile-sam.cpp:207-531, 556-583) ---- */
/* pixel prompt of the original image (n = 2 point, 4 box) -> [-1, 1] coordinates (preprocess_point / preprocess_box) */
void vo_sam_process_prompt(const int* prompt, int n, int image_w, int image_h, int image_size, float out[4]);
/* embed_points (point + sentinel) / embed_box -> sparse prompt [2][dim] */
int vo_sam_embed_prompt(const vo_model*, const float coords[4], int is_box, float* out, int* dim);
/* predict_masks with the no-mask dense prompt: embed NHWC [res][res][dim], sparse [n][dim] -> masks [4][(4 res)^2], iou [4] */
int vo_sam_predict_masks(const vo_model*, const float* embed, int res, int dim, const float* sparse, int n_sparse, float* masks, float* iou);
void vo_sam_process_mask(const float* mask, int mask_size, int image_size, int target_w, int target_h, uint8_t* out);
/* whole sam_compute after sam_encode; out_mask alpha_u8 [image_h][image_w]; iou_out [4] and masks_out [4][(4 res)^2] optional */
int vo_sam_compute(const vo_model*, const float* embed, int res, int dim, int image_w, int image_h, const int* prompt, int n_prompt,
                   uint8_t* out_mask, float* iou_out, float* masks_out);

/* ---- SWIN transformer encoder, the BiRefNet backbone (reference src/visp/arch/swin.cpp) ---- */
typedef struct { int embed_dim, window_size, depths[4], n_heads[4]; } vo_swin_params; /* swin_t: 96, 7, {2,2,6,2}, {3,6,12,24} (swin.cpp:266-275) */
void vo_swin_rel_pos_index(int window, int32_t* dst /*[ws^4]*/);                      /* swin.cpp:26-38 */
void vo_swin_attention_mask(int w, int h, int window, float* out /*[nw_y*nw_x][ws^2][ws^2]*/); /* swin.cpp:165-213 */
/* block (swin.cpp:117-163) on tokens x [h*w][C] in place; the mask (vo_swin_attention_mask(w, h)) is applied iff non-NULL, as swin::block does */
int vo_swin_block(const vo_model*, const char* prefix, float* x, int w, int h, int C, int heads, int window, int shift, const float* mask);
/* patch_merging (swin.cpp:140-161): *out malloc'd [(h/2)*(w/2)][*cout], free with vo_free */
int vo_swin_patch_merging(const vo_model*, const char* prefix, const float* x, int w, int h, int C, float** out, int* cout);
/* swin_encode (swin.cpp:237-262): normalised rgb_f32 image [H][W][3] -> four normed stage outputs (NHWC, malloc'd: vo_free) */
/* 0 (default) = the reference as written: swin::layer passes the shift mask to every block (swin.cpp:226-237); 1 = shifted
 * blocks only (the reference's torch twin, HuggingFace Swin). Process-wide; used by vo_swin_encode and vo_birefnet_predict. */
void vo_swin_set_mask_mode(int shifted_only);
int vo_swin_get_mask_mode(void);
int vo_swin_encode(const vo_model*, const char* prefix, const vo_swin_params*, const float* image, int W, int H, float* outs[4],
                   int dims[4][3], vo_capture* captures, int n_captures);
void vo_free(void* p);

/* ---- BiRefNet (reference src/visp/arch/birefnet.cpp): two-scale SWIN encode, squeeze block, decoder -------------------------
 * The deformable convolution is PARITY UNPINNED (no torchvision / ggml here): torchvision's published algorithm restated. */
void vo_deform_conv2d_nhwc(const float* x, int H, int W, int Cin, const float* w /*[Cout][kh][kw][Cin]*/, int Cout, int kh, int kw,
                           const float* offset /*[OH][OW][2 kh kw]: (dy, dx) per tap*/, const float* mask /*[OH][OW][kh kw] or NULL*/, int stride,
                           int pad, float* y /*[OH][OW][Cout]*/);
/* image_to_patches (birefnet.cpp:158-167): [IH][IW][C] -> [h][w][gw*gh*C], channel = gx + gw (gy + gh c) */
void vo_image_to_patches(const float* image, int IW, int IH, int C, int w, int h, float* patches);
/* birefnet::encode (birefnet.cpp:43-73): feats[i] malloc'd NHWC (vo_free), dims[i] = {w, h, C} */
int vo_birefnet_encode(const vo_model*, const vo_swin_params*, const float* image, int W, int H, float* feats[4], int dims[4][3]);
/* birefnet_predict (birefnet.cpp:252-260): normalised rgb_f32 [H][W][3] -> sigmoid mask [H][W]; captures: feature_0..3, squeeze, p4..p1 */
int vo_birefnet_predict(const vo_model*, const vo_swin_params*, const float* image, int W, int H, float* out, vo_capture* captures, int n_captures);

/* dino building blocks, exposed for module-level parity tests */
int vo_dino_layer(const vo_model*, const char* prefix, int n_heads, int gelu_mode, float* x /*[N][C] in/out*/,
                  int64_t N, int64_t C);

#ifdef __cplusplus
}
#endif
#endif
