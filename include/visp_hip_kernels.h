/*
 * visp_hip_kernels.h -- thin C ABI between the host C++ of the backend and its HIP
 * translation units (gfx950 / MI355X only). Host code never includes HIP headers; every
 * pointer below is a device pointer unless it says "host", `stream` is a hipStream_t.
 *
 * Each launcher replaces the ggml ops the reference's graph emits for the Depth-Anything
 * path; the reference call sites are cited per entry.
 * All functions return 1 on success and 0 on error (message via vx_last_error()), the
 * convention of the reference's C API (src/visp/c-api.cpp:6-21).
 */
#ifndef VISP_HIP_KERNELS_H
#define VISP_HIP_KERNELS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VX_API __attribute__((visibility("default")))

/* ---- runtime ------------------------------------------------------------------------- */
VX_API const char* vx_last_error(void);
VX_API int vx_device_count(void);
VX_API int vx_set_device(int index);
/* name/arch strings are copied into caller buffers; total/free memory in bytes */
VX_API int vx_device_info(int index, char* name, int name_cap, char* arch, int arch_cap,
                          size_t* total_mem, size_t* free_mem, int* n_cu);
VX_API int vx_malloc(void** ptr, size_t bytes);
VX_API int vx_free(void* ptr);
VX_API int vx_memset(void* ptr, int value, size_t bytes, void* stream);
VX_API int vx_memcpy_h2d(void* dst, const void* host_src, size_t bytes, void* stream);
VX_API int vx_memcpy_d2h(void* host_dst, const void* src, size_t bytes, void* stream);
VX_API int vx_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream);
/* pinned host memory + copies that do NOT synchronise (the overlapped host pipeline: H2D / compute / D2H on three streams) */
VX_API int vx_malloc_host(void** ptr, size_t bytes);
VX_API int vx_free_host(void* ptr);
VX_API int vx_memcpy_h2d_async(void* dst, const void* pinned_src, size_t bytes, void* stream);
VX_API int vx_memcpy_d2h_async(void* pinned_dst, const void* src, size_t bytes, void* stream);
VX_API int vx_event_sync(void* ev);
VX_API int vx_stream_create(void** stream);
VX_API int vx_stream_destroy(void* stream);
VX_API int vx_stream_sync(void* stream);
VX_API int vx_event_create(void** ev);
VX_API int vx_event_destroy(void* ev);
VX_API int vx_event_record(void* ev, void* stream);
VX_API int vx_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on stop */
VX_API int vx_stream_wait_event(void* stream, void* ev);            /* fork/join between streams (also under capture) */
/* hipGraph capture of a launch sequence */
VX_API int vx_graph_begin_capture(void* stream);
VX_API int vx_graph_end_capture(void* stream, void** graph_exec);
VX_API int vx_graph_launch(void* graph_exec, void* stream);
VX_API int vx_graph_destroy(void* graph_exec);

/* ---- GEMM family: C[M,N] = A[M,K] * W[N,K]^T (f16 in, f32 accumulate on MFMA) ----------
 * replaces ggml_mul_mat (+ggml_add bias, +ggml_gelu, +ggml_mul lambda, +ggml_add residual)
 * emitted by linear() src/visp/nn.cpp:6-12, conv_2d 1x1 nn.cpp:76-81, dino.cpp:48-90,
 * conv_transpose_2d nn.cpp:117-129 (k == stride => a GEMM + pixel shuffle),
 * and ggml_conv_2d 3x3 (nn.cpp:83-97) as an implicit GEMM over NHWC input. */
enum vx_epilogue {
    VX_EPI_F16 = 0,       /* out f16 [M, ldo] = acc + bias                                      */
    VX_EPI_F16_GELU = 1,  /* ... then tanh-GELU (ggml_gelu, dino.cpp:54)                        */
    VX_EPI_F16_RELU = 2,  /* ... then ReLU                                                      */
    VX_EPI_RESID_F32 = 3, /* x f32 [M, ldo] += lambda[n] * (acc + bias)   (dino.cpp:48-50,80-87) */
    VX_EPI_TOKENS = 4,    /* patch embed: x f32 [(m/P)*(P+1) + 1 + m%P, n] = acc + bias + pos[1 + m%P, n]
                             (dino.cpp:32-46, pos-embed add fused)                              */
    VX_EPI_QKV = 5,       /* scatter into head-major q, k, v f16 [B,H,T,64]; q scaled             */
    VX_EPI_PIXSHUF = 6,   /* conv-transpose k==s: n = (dy*s+dx)*Cout + co ->
                             out f16 [b, y*s+dy, x*s+dx, co]  (nn.cpp:117-129)                  */
    VX_EPI_F16_ADD = 7,   /* out f16 = [relu](acc + bias) + res1 + res2 (nullable), conv residual units
                             (depth-anything.cpp:15-30)                                          */
    VX_EPI_HEAD_OUT = 8,  /* N == 32: out f32 [M] = head_scale * relu(sum_n relu(acc+bias)[n] * lambda[n] + head_bias):
                             head.conv2 + ReLU + head.conv3 (1x1 -> 1) + ReLU fused (depth-anything.cpp:87-94) */
};

typedef struct {
    /* A operand: plain rows, or implicit im2col over an NHWC image when conv_kh > 0 */
    const void* A;      /* f16 */
    int64_t lda;        /* elements between consecutive A rows (plain mode)                      */
    int a_group;        /* plain mode row remap: arow = (m / a_group) * a_group_stride + a_row_off + m % a_group */
    int a_group_stride; /* (a_group == 0: identity). Used to skip the cls token (depth-anything.cpp:50) */
    int a_row_off;
    /* implicit-GEMM conv (conv_kh > 0): A is [B, H, W, Cin] f16, m = (b, oy, ox), k = (ky, kx, c) */
    int conv_kh, conv_kw, conv_stride, conv_pad;
    int conv_H, conv_W, conv_Cin, conv_OH, conv_OW;
    int a_relu;         /* apply ReLU to A elements while loading (depth-anything.cpp:17,19)    */

    const void* W;      /* f16 [N, K], K padded to a multiple of 64 with zeros                   */
    const float* bias;  /* f32 [N] or NULL                                                      */
    int M, N, K;

    int epi;            /* enum vx_epilogue */
    void* out;
    int64_t ldo;
    int relu;           /* VX_EPI_F16_ADD: apply ReLU before adding residuals                   */
    const float* lambda; /* VX_EPI_RESID_F32: f32 [N]                                           */
    const float* pos;   /* VX_EPI_TOKENS: f32 [(P+1), N]                                        */
    int tokens_P;       /* VX_EPI_TOKENS: patches per image                                     */
    /* VX_EPI_QKV */
    void* q; void* k; void* vt;
    int qkv_T, qkv_Tp, qkv_H; /* tokens per image, (unused), heads; vt receives v as [B,H,T,64]    */
    float q_scale;
    /* VX_EPI_PIXSHUF */
    int ps_s, ps_Cout, ps_H, ps_W; /* stride, real Cout, input H, W; ldo = output channel stride */
    /* VX_EPI_F16_ADD */
    const void* res1; const void* res2; /* f16 [M, ldo] or NULL */
    int n_valid;        /* columns >= n_valid are not stored (N padded for tiling); 0 = N        */
    int stages;         /* LDS ring depth of the k-loop: 0 = kernel default (tuning knob for benches) */
    float head_bias, head_scale; /* VX_EPI_HEAD_OUT: conv3 bias, max_depth; lambda = conv3 weights f32 [N] */
    /* VX_EPI_F16_ADD extras (TinyViT): post_gelu: out = gelu(acc + bias + res1 [+ res2]) (mb_conv tail, mobile-sam.cpp:88-90).
     * win_ws > 0: the M rows are in window order of a win_res x win_res map (window_partition with zero padding,
     * mobile-sam.cpp:25-46); out and res1/res2 are addressed at the PIXEL row of each window row, padded rows are dropped:
     * window_reverse + residual add fused (mobile-sam.cpp:48-64, 146-149) */
    int post_gelu, win_ws, win_res;
    /* SWIN (swin.cpp:141-156): win_res_h > 0 makes the map win_res (width) x win_res_h (height); win_shift > 0 undoes the cyclic
     * shift as well (window row -> padded position -> + shift modulo the padded extent -> pixel, dropped if outside the map) */
    int win_res_h, win_shift;
    /* split-K for deep reductions on few output tiles (3x3 convs with thousands of input channels on small maps): k_splits > 1
     * cuts the k-loop into that many ranges, one workgroup each, partial sums f32 in k_partial [k_splits][M][N] (caller's scratch),
     * summed in a fixed order by a second launch that applies bias / ReLU / res1 -- deterministic. Epilogues F16, F16_RELU, F16_ADD. */
    int k_splits; float* k_partial;
    void* debug_stamps; /* diagnostics only: u64 [blocks][8] s_memtime stamps per phase, NULL in product */
} vx_gemm_args;

VX_API int vx_gemm_f16(const vx_gemm_args* args, void* stream);
/* heuristic for vx_gemm_args.k_splits: > 1 only when the tile grid fills less than half the chip and the k-loop has >= 16 tiles */
VX_API int vx_gemm_pick_k_splits(int M, int N, int K);

/* 3x3 / stride 1 / pad 1 NHWC convolution with the input halo staged once in LDS (Cin, Cout in {32, 64};
 * epilogues F16, F16_RELU, F16_ADD, HEAD_OUT). Same argument block as the implicit-GEMM form (conv_* fields,
 * W = [Cout][Kp] with k = (ky,kx,c)); used for the DPT fusion / head convs at >= 96 px (depth-anything.cpp:15-23, 81-94). */
VX_API int vx_conv3x3_supported(const vx_gemm_args* args);
VX_API int vx_conv3x3_f16(const vx_gemm_args* args, void* stream);

/* ---- ESRGAN dense-block 3x3 convolution (esrgan.cpp:13-79) ---------------------------------
 * Activations are f16, PLANAR in groups of 32 channels: a C-channel map is C/32 planes of [B, H, W, 32], consecutive
 * planes `*_plane` elements apart. A residual dense block keeps [x|x1|x2|x3|x4] as six planes of one buffer: a conv
 * reads the first cin/32 planes and writes its 32 (or 64) outputs as the next plane(s), so concat
 * (esrgan.cpp:29-36) is never materialised.
 *   v = conv(x) + bias;  act: v = max(v, 0.2 v);  res1: v = v*s1 + res1;  res2: v = v*s2 + res2.
 *   x_residual (cout = 64): v = v*s1 + x[planes 0,1] with x taken from the halo already in LDS (no second read);
 *   mutually exclusive with res1, res2 still applies afterwards.
 * up2: the source is [B, H/2, W/2, 32] per plane and is nearest-upsampled on the fly (esrgan.cpp:13-19).
 * w: packed by the host as [cin/32][9 taps][cout][32] f16 with the four 16-byte groups of every (tap, n) row
 *    stored at position g ^ ((n >> 2) & 3)  (see packer::conv in csrc/esrgan.cpp).
 * VX_DC_RGB_F32: cout = 32 (3 real), out = f32 [B, H, W, 3], no activation / residuals.
 * VX_DC_HEAD_F32: cout = 32, out = f32 [B, H, W] = head_scale * relu(sum_n relu(v)[n] * head_w[n] + head_bias)
 *   (head.conv2 + ReLU + head.conv3 1x1 + ReLU of the DPT head, depth-anything.cpp:87-94).
 * The same kernel serves NHWC maps: *_pix = elements between pixels (0 = 32 = planar), *_plane = elements between
 *   consecutive 32-channel groups (32 for NHWC). act: 0 none, 1 LeakyReLU 0.2, 2 ReLU. a_relu: convolve relu(x). */
enum { VX_DC_F16 = 0, VX_DC_RGB_F32 = 1, VX_DC_HEAD_F32 = 2 };
typedef struct {
    const void* x; int64_t x_plane; int cin; int up2;
    int B, H, W;                 /* output (= conv-space) extent */
    const void* w; const float* bias; int cout;
    int epi, act;
    float s1; const void* res1; int64_t res1_plane;
    float s2; const void* res2; int64_t res2_plane;
    void* out; int64_t out_plane;
    int x_residual;
    int64_t x_pix, out_pix, res1_pix, res2_pix;
    int a_relu;
    const float* head_w; float head_bias, head_scale;
    void* stamps; /* diagnostics only: u64 [blocks][8] = cycles in {dma wait, barrier, dma issue + halo cursor, mfma loop, epilogue, tile setup}, steps, end clock; NULL in product */
    /* round 3 -- ggml_interpolate(BILINEAR | ALIGN_CORNERS) fused into the consumer (depth-anything.cpp:36-38, 84-85; ml.cpp:782-788):
     * bil_hs > 0: x is the LOW-resolution map [B, bil_hs, bil_ws, .] and the conv runs on its bilinear resize to H x W, interpolated by
     *   the halo loader (source patch by LDS-DMA, packed-f16 weights); cout = 32, no up2 / a_relu / x_residual, H + W <= 2048 and a
     *   scale that keeps an 18 x 34 halo inside a 12 x 21 source patch (about <= 0.575; vx_dconv_bilinear_supported).
     * res2_hs > 0: res2 is [B, res2_hs, res2_ws, .] and is resized to H x W where the epilogue adds it (same arithmetic). */
    int bil_hs, bil_ws;
    int res2_hs, res2_ws;
} vx_dconv_args;
VX_API int vx_dconv_bilinear_supported(int cout, int H, int W, int hs, int ws);
VX_API int vx_dconv3x3_f16(const vx_dconv_args* args, void* stream);
/* sets the kernels' dynamic-LDS attribute (call once per process before capturing launches into a hipGraph) */
VX_API int vx_dconv_prepare(void);

/* ---- one launch per DPT residual unit on small maps (depth-anything.cpp:15-33; kernels_rcu.hip) -------------
 *   out = P( conv2( relu( conv1( relu(x) ) + b1 ) ) + b2 + x [+ res2] ),  P = identity or the 1x1 projection wp (+ bp)
 * x, res2, out: f16 NHWC [B, H, W, 64]; biases f32 [64] or NULL. w1, w2: f16 [9 taps][64 n][64 c], wp: f16 [64 n][64 c], every (tap, n) row of
 * 128 bytes with its eight 16-byte groups stored at position g ^ ((n >> 1) & 7) (the linear LDS image of a tap's slab).
 * The intermediate map stays in LDS. H, W <= 96 (vx_rcu_supported); larger maps are the LDS-ring conv's. */
typedef struct {
    const void* x; const void* w1; const float* b1; const void* w2; const float* b2;
    const void* res2; const void* wp; const float* bp;
    void* out;
    int B, H, W;
} vx_rcu_args;
VX_API int vx_rcu_supported(int H, int W);
VX_API int vx_rcu_fused_f16(const vx_rcu_args* args, void* stream);

/* tile_layout (include/visp/image.h:163-181, src/visp/image.cpp:612-651) */
typedef struct { int image_w, image_h, overlap_x, overlap_y, n_x, n_y, tile_w, tile_h; } vx_tile_layout;
/* image_u8_to_f32 with tile offset for every tile (vision.cpp:236-241) -> one plane f16 [B*n_tiles, tile_h, tile_w, 32]
 * (channels 0..2 = value, 3..5 = f16 rounding residue, rest 0). format = visp::image_format (u8 colour). */
VX_API int vx_esrgan_tiles_in(const uint8_t* img, int B, int w, int h, int format, const vx_tile_layout* t, void* out, void* stream);
/* the same tiles as f32 rgb [B*n_tiles, tile_h, tile_w, 3]: the input tensor of the generator's graph (which splits it into value | residue itself) */
VX_API int vx_esrgan_tiles_in_f32(const uint8_t* img, int B, int w, int h, int format, const vx_tile_layout* t, float* out, void* stream);
/* tile_merge of all tiles in the reference's order + image_f32_to_u8 rgba (image.cpp:653-693, vision.cpp:246-252).
 * tiles: f32 [B, n_y, n_x, tile_h, tile_w, 3] (t = the SCALED layout); out_f32 [B,h,w,3] and/or out_rgba [B,h,w,4]. */
VX_API int vx_esrgan_tiles_out(const float* tiles, int B, const vx_tile_layout* t, float* out_f32, uint8_t* out_rgba, void* stream);

/* ---- TinyViT (MobileSAM image encoder) helpers, kernels_tinyvit.hip -------------------------------------------------
 * u8 rgb -> f16 [pixels][8]: (v/255 - mean)/std (sam_process_input, mobile-sam.cpp:533-547) as value + f16 residue */
VX_API int vx_tv_preprocess(const uint8_t* rgb, void* out, int64_t n_pixels, void* stream);
/* depthwise 3x3 pad 1 (+ bias, optional GELU) on NHWC f16: conv_2d_depthwise, nn.cpp:102-115; w f16 [9][C], bias f32 [C] */
VX_API int vx_dwconv3x3_f16(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int C, int stride, int gelu, void* stream);
/* second half of mb_conv in one launch (mobile-sam.cpp:82-90; kernels_mbconv.hip): y = gelu(x + conv3(gelu(dw3x3(h) + b2)) + b3)
 * h f16 [B,H,W,C] (= gelu(conv1(x))), w2 f16 [9][C], w3 f16 = vx_mbconv_pack_w3 (host code) of the row-major [Cout][C] weight
 * (MFMA fragment order, same byte count), x / y f16 [B,H,W,Cout]. Built for C = 256, Cout = 64, W % 64 == 0
 * (vx_mbconv_dw_pw_supported); the depthwise output only exists as an LDS tile. */
VX_API int vx_mbconv_dw_pw_supported(int C, int Cout, int W);
VX_API int vx_mbconv_pack_w3(const void* w3_rows_host, void* packed_host);
VX_API int vx_mbconv_dw_pw_f16(const void* h, const void* w2, const float* b2, const void* w3, const float* b3, const void* x, void* y, int B, int H,
                               int W, int C, int Cout, void* stream);
/* LayerNorm of f16 rows (nn.cpp:14-19). ws > 0: output rows in window order of a res x res map (window_partition,
 * mobile-sam.cpp:25-46; padded positions = norm of zero = bias). out_f32: f32 output. C <= 512. */
VX_API int vx_layernorm_f16(const void* x, const float* w, const float* b, void* y, int64_t rows_out, int C, float eps, int res, int ws,
                            int out_f32, void* stream);
/* window attention with relative position bias, head_dim 32, on MFMA (mobile-sam.cpp:112-131; kernels_winattn.hip): qkv f16
 * [n_windows*N][heads*96] (per head q|k|v), out f16 [n_windows*N][heads*32]; N <= 256. bias_packed: the f16 image that
 * vx_window_attention_pack_bias (host code) makes of attention_biases_indexed f32 [heads][N][N] -- accumulator order,
 * padded keys = -inf -- vx_window_attention_bias_bytes(N, heads) bytes, uploaded by the caller. */
VX_API size_t vx_window_attention_bias_bytes(int N, int heads);
VX_API int vx_window_attention_pack_bias(const float* bias_host, int N, int heads, void* packed_host);
VX_API int vx_window_attention_f16(const void* qkv, const void* bias_packed, void* out, int n_windows, int N, int heads, void* stream);
/* ---- SWIN encoder (BiRefNet backbone, swin.cpp), kernels_swin.hip + the masked form of the window attention --------------
 * shifted windows (swin.cpp:165-213): bias_packed holds FOUR images (vx_swin_attention_pack_bias: bias, + last-column mask,
 * + last-row mask, + corner mask); window (wy, wx) of an nwy x nwx image picks its class. nwx = nwy = 0: one image. */
VX_API int vx_swin_attention_pack_bias(const float* table /*[(2ws-1)^2][heads]*/, int ws, int heads, void* packed_host /*4 x bias_bytes(ws*ws, heads)*/);
VX_API int vx_window_attention_masked_f16(const void* qkv, const void* bias_packed, void* out, int n_windows, int N, int heads, int nwx, int nwy,
                                          void* stream);
/* LayerNorm of f16 rows [.., C] (C % 8 == 0, <= 2048). ws > 0: norm1 + pad + roll(-shift) + window_partition of swin::block
 * (swin.cpp:124-139): x is [B, H, W, C], output rows are window tokens (zeros where the padded map has no pixel). ws = 0: rows in
 * place; out_f32 writes f32 (the per-stage output norms, swin.cpp:255-258). */
VX_API int vx_swin_layernorm_f16(const void* x, const float* w, const float* b, void* y, int64_t rows_out, int C, float eps, int H, int W, int ws,
                                 int shift, int out_f32, void* stream);
VX_API int vx_swin_layernorm_strided_f16(const void* x, const float* w, const float* b, void* y, int64_t rows, int C, float eps, int ldy, int out_f32,
                                         void* stream); /* plain rows, output row stride ldy elements (0 = C) */
/* patch_merging up to the reduction linear (swin.cpp:140-158): 2x2 gather in the reference's concat order + LayerNorm(4C);
 * x [B, H, W, C] -> y [B * H/2 * W/2, 4C] */
VX_API int vx_swin_merge_layernorm_f16(const void* x, const float* w, const float* b, void* y, int B, int H, int W, int C, float eps, void* stream);
/* window_reverse + roll(+shift) + crop + shortcut (swin.cpp:141-156): y [B,H,W,C] = x + a[window row] */
VX_API int vx_swin_window_reverse_add_f16(const void* a, const void* x, void* y, int B, int H, int W, int C, int ws, int shift, void* stream);
/* window_reverse + residual: y[b,py,px,:] = x[b,py,px,:] + a[window row of (py,px),:] (mobile-sam.cpp:48-64, 146-149) */
VX_API int vx_window_reverse_add_f16(const void* a, const void* x, void* y, int B, int res, int ws, int C, void* stream);
/* ---- BiRefNet glue (two-scale encode + decoder, birefnet.cpp), kernels_birefnet.hip ---------------------------------------
 * normalised (birefnet_process_input) and downscaled by 2 (bilinear, align_corners): rgb u8 [B,H,W,3] -> f16 [B,H/2,W/2][8] value + residue */
VX_API int vx_bf_preprocess_half(const uint8_t* rgb, void* out8, int B, int H, int W, void* stream);
/* image_to_patches of the normalised image (birefnet.cpp:158-167): -> f16 [B, h, w, (IW/w)*(IH/h)*3], channel = gx + gw (gy + gh c) */
VX_API int vx_bf_patches(const uint8_t* rgb, void* out, int B, int IH, int IW, int h, int w, void* stream);
/* bilinear align-corners resize of f16 NHWC maps; lds / ldd = row strides in elements (channel slices of wider buffers); C % 8 == 0 */
VX_API int vx_bf_resize_f16(const void* src, int lds, void* dst, int ldd, int B, int h, int w, int C, int oh, int ow, void* stream);
/* sampling half of deformable_conv_2d (birefnet.cpp:83-92 = torchvision deform_conv2d, stride 1, pad k/2): x f16 [B,h,w,C]; offmod
 * [B*h*w, ldom] = per pixel 2 k^2 offsets (dy, dx per tap) then k^2 modulator logits; cols [B*h*w, k*k*C] = 2 sigmoid(mod) * sample */
VX_API int vx_bf_deform_cols_f16(const void* x, const void* offmod, int ldom, void* cols, int B, int h, int w, int C, int k, void* stream);
/* per-image pixel mean [B, n, C] (row stride ld) -> [B, C]; broadcast of [B, C] rows to every pixel of a slice; y *= sigmoid(a); final mask */
VX_API int vx_bf_mean_f16(const void* x, int ld, void* y, float* acc_scratch /*f32 [B*C]*/, int B, int64_t n, int C, void* stream);
VX_API int vx_bf_broadcast_f16(const void* g, int ldg, void* dst, int ldd, int B, int64_t n, int C, void* stream);
VX_API int vx_bf_mul_sigmoid_f16(void* y, int ldy, const void* a, int lda, int64_t rows, int C, void* stream);
VX_API int vx_bf_sigmoid_out_f32(const void* a, int lda, float* out, int64_t n, void* stream);
/* y f16 = a + b[i mod b_period]; a f16 or f32 (SAM decoder: queries + query_pe, keys + key_pe, embedding + no_mask_embed) */
VX_API int vx_add_rows_f16(const void* a, int a_is_f32, const void* b, int64_t b_period, void* y, int64_t n, void* stream);
/* attention with few queries or few keys (SAM mask decoder, mobile-sam.cpp:306-320): q [Nq][heads*hd], k, v [Nk][heads*hd]
 * -> out [Nq][heads*hd]; hd in {8, 16, 32}, Nk <= 4096; scale = 1/sqrt(hd) */
VX_API int vx_small_attention_f16(const void* q, const void* k, const void* v, void* out, int Nq, int Nk, int heads, int hd, void* stream);
/* sam::interpolate_bilinear (mobile-sam.cpp:485-516) for sam_process_mask (:556-583): source element (x, y) at
 * src[(y * sstride + x) * step], f16 or f32; dst f32 [dh][dw], or u8 thresholded at 0 (0 / 255) when out_u8 */
VX_API int vx_sam_interpolate(const void* src, int src_f16, int sw, int sh, int sstride, int step, void* dst, int dw, int dh, int out_u8, void* stream);
/* y = gelu(a + b), b nullable (mb_conv tail, mobile-sam.cpp:88-90) */
VX_API int vx_add_gelu_f16(const void* a, const void* b, void* y, int64_t n, void* stream);

/* ---- fused multi-head attention, head_dim 64 (nn.cpp:210-244, dino.cpp:59-74) ------------
 * q,k,v: f16 [B,H,T,64], q pre-scaled by VX_ATTN_Q_SCALE = log2(e) / sqrt(64): the kernel works in the exp2 domain and the
 * producer of q (QKV epilogue / block kernel, their q_scale argument) folds the factor in; out: f16 [B*T, H*64].
 * softmax in f32, S never leaves registers; V is transposed on the fly by ds_read_b64_tr_b16. */
#define VX_ATTN_Q_SCALE 0.18033688011112042f
VX_API int vx_attention_f16(const void* q, const void* k, const void* v, void* out, int B, int H, int T,
                            void* stream);
/* test hook: a lane's partial row sum above `limit` sends a key tile from the lagging-reference fast path to the full path
 * (0 = always the full path; negative = restore the default 1024) */
VX_API void vx_attention_set_fast_limit(float limit);
/* diagnostics (tools/attn_stamps.py): u64 [blocks][waves][4] = cycles in {wait + barrier, scores + softmax, PV, lifetime}; NULL = off */
VX_API void vx_attention_set_stamps(void* stamps);

/* ---- token-stationary DINOv2 block (kernels_block16.hip; embed dim 384, mlp 1536, head dim 64) ----------------------------
 * One launch per layer replaces everything between two attentions of dino::layer (src/visp/arch/dino.cpp:48-90):
 *   att != NULL:  x += lambda1 * (att Wo^T + bo);  x += lambda2 * (gelu(LN2(x) W1^T + b1) W2^T + b2)
 *   feat != NULL: feat = LN_final(x) as f16 rows (get_intermediate_layers, dino.cpp:100-107)
 *   q != NULL:    q, k, v = LN1(x) Wqkv^T + b of the NEXT layer, head-major f16 [B, H, T, 64], q scaled by q_scale
 * (att == NULL, q != NULL: the first layer's LN1 + QKV on x as it is.)
 * att f16 [M, 384]; x f32 [M, 384] in place; weights as the slab streams of vx_dino_block16_pack_mlp / _qkv. LayerScale arrives FOLDED
 * in by the caller: wo / w2 rows and bo / b2 scaled by lambda1 / lambda2 -- the residual stream then stays in the MFMA accumulators from
 * the attention output to the next layer's q, k, v (x read once, written once per launch).
 * vec_mlp f32 = bo' | (unused) | ln2.w | ln2.b | b1[1536] | b2' | (unused) (3840 floats), vec_qkv f32 = ln1.w | ln1.b | bqkv[1152]
 * (1920 floats), vec_tap f32 = lnf.w | lnf.b (768 floats). M % T == 0 when q != NULL. 16 tokens per wave on v_mfma_f32_16x16x32_f16,
 * two waves per SIMD. */
typedef struct {
    const void* att; float* x;
    const void* w_mlp; const float* vec_mlp;
    const void* w_qkv; const float* vec_qkv;
    const float* vec_tap; void* feat;
    void *q, *k, *v;
    int M, T, H;
    float q_scale, eps;
    float* cap_x1; /* tests only: f32 [M, 384] copy of x after the attention half; NULL in the product path */
    void* stamps;  /* diagnostics only: u64 [workgroups][16] s_memtime at the phase boundaries; NULL in the product path */
} vx_dino_block_args;
VX_API int vx_dino_block_supported(int embed_dim, int hidden, int head_dim);
VX_API size_t vx_dino_block_mlp_bytes(void);
VX_API size_t vx_dino_block_qkv_bytes(void);
/* host code: f16 row-major wo' [384][384], w1 [1536][384], w2' [384][1536] -> out (vx_dino_block_mlp_bytes());
 * wqkv [1152][384] (q rows, then k, then v) -> out (vx_dino_block_qkv_bytes()) */
VX_API int vx_dino_block16_pack_mlp(const void* wo, const void* w1, const void* w2, void* out);
VX_API int vx_dino_block16_pack_qkv(const void* wqkv, void* out);
VX_API int vx_dino_block16_f16(const vx_dino_block_args* args, void* stream);

/* ---- e4m3 GEMM on the block-scaled matrix instruction (kernels_gemm_fp8.hip), opt-in: BASELINE.json configs[4] "fp8 GGUF weights on CDNA4 fp8 MFMA".
 * The reference has no fp8 tensor type; the format is this backend's own: OCP e4m3 values + ONE f32 scale per row (per output channel of a weight,
 * per token of an activation), K padded to a multiple of 128 with zeros:
 *   out f16 [M, ldo] = act( a_scale[m] * w_scale[n] * sum_k A8[m, k] * W8[n, k] + bias[n] ) [+ res]   (linear + bias [+ GELU] [+ residual], nn.cpp:6-12, mobile-sam.cpp MLP);
 * n_valid and ldo multiples of 8 */
typedef struct {
    const void* A; const float* a_scale; /* e4m3 [M][Kp], f32 [M] (vx_quantize_rows_e4m3) */
    const void* W; const float* w_scale; /* e4m3 [N][Kp], f32 [N] (vx_quantize_rows_e4m3_host), N % 128 == 0 */
    const float* bias;                   /* f32 [N] or NULL */
    int M, N, Kp, n_valid;               /* columns >= n_valid are not stored */
    void* out; int64_t ldo;
    int act;                             /* 0 none, 1 tanh-GELU */
    const void* res;                     /* f16 [M, ldo] added to the result, or NULL (the MLP's residual, mobile-sam.cpp:150-160) */
} vx_gemm_fp8_args;
VX_API int vx_gemm_fp8_supported(int N, int K);
VX_API int vx_gemm_fp8(const vx_gemm_fp8_args* args, void* stream);
/* f16 rows [M][ldx] (first K columns) -> e4m3 rows [M][Kp] + scale[m] = absmax / 448 (device) */
VX_API int vx_quantize_rows_e4m3(const void* x_f16, int64_t ldx, void* q, float* scale, int M, int K, int Kp, void* stream);
/* host code: f32 rows [N][K] -> e4m3 [N][Kp] (round to nearest even, saturating at 448) + scale[n] */
VX_API int vx_quantize_rows_e4m3_host(const float* w, int N, int K, int Kp, void* q_out, float* scale_out);

/* ---- LayerNorm (nn.cpp:14-19): x f32 [M,C] -> y f16 [M,C]; biased variance, eps in sqrt --- */
VX_API int vx_layernorm_f32_f16(const float* x, const float* w, const float* b, void* y, int M, int C,
                                float eps, void* stream);

/* ---- pre-processing (depth-anything.cpp:130-140, image.cpp:215-255) + im2col of the 14x14
 * stride-14 patch embedding (nn.cpp:166-180): rgb_u8 [B,H,W,3] -> f16 [B*P, Kp],
 * k = (ky*ps + kx)*3 + c, zero padded to Kp; value = (u8/255 - mean[c]) / std[c] ----------- */
VX_API int vx_preprocess_patches(const uint8_t* rgb, void* patches, int B, int H, int W, int ps, int Kp,
                                 const float mean[3], const float inv_std[3], void* stream);
/* same normalisation to plain NHWC f32 [B,H,W,3] (the tensor the reference uploads) */
VX_API int vx_preprocess_f32(const uint8_t* rgb, float* out, int B, int H, int W, const float mean[3],
                             const float inv_std[3], void* stream);
/* cls token rows: x[b*(P+1), :] = cls + pos[0, :]  (dino.cpp:37-44) */
VX_API int vx_write_cls_rows(float* x, const float* cls, const float* pos, int B, int T, int C, void* stream);

/* ---- bilinear resize, align_corners (ml.cpp:782-788, depth-anything.cpp:36-38,83-85),
 * NHWC f16 [B,H,W,C] -> [B,OH,OW,C], C % 8 == 0 ------------------------------------------- */
VX_API int vx_bilinear_ac_f16(const void* x, void* y, int B, int H, int W, int C, int OH, int OW, void* stream);

/* ---- head tail: depth[b,y,x] = max_depth * relu(sum_c x[b,y,x,c]*w[c] + bias), x f16 NHWC
 * (already ReLU-ed conv2 output), C <= 64 (depth-anything.cpp:89-94) ----------------------- */
VX_API int vx_head_out_f32(const void* x, const float* w, float bias, float max_depth, float* depth,
                           int64_t n_pixels, int C, void* stream);

/* ---- the DPT head's tail as one kernel (depth-anything.cpp:84-94; kernels_headconv.hip): x f16 [B, hs, ws, 32] --bilinear, align_corners-->
 * [B, H, W, 32] -> conv 3x3 (32 -> 32, pad 1) + bias + ReLU -> conv 1x1 (32 -> 1) + b3 + ReLU -> * scale -> out f32 [B, H, W]. The 3x3 kernel
 * is held in registers: `wfrag` = vx_headconv_pack (host code) of the GEMM-family rows [32][Kp], k = (ky, kx, c); vx_headconv_frag_bytes()
 * bytes. Supported: cin = cout = 32 and a resize whose 18 x 34 halo fits a 12 x 21 source patch (scale up to about 0.58). */
VX_API size_t vx_headconv_frag_bytes(void);
VX_API int vx_headconv_pack(const void* w_rows, int Kp, void* out_frag);
VX_API int vx_headconv_supported(int cin, int cout, int H, int W, int hs, int ws);
/* diagnostics (tools/headconv_stamps.py): u64 [blocks][4][8] per-phase cycles of the next launches; NULL = off */
VX_API void vx_headconv_set_stamps(void* stamps);
VX_API int vx_headconv_bil_f16(const void* x, const void* wfrag, const float* bias, const float* w3, float b3, float scale, float* out, int B, int H, int W,
                               int hs, int ws, void* stream);

/* ---- post-processing: per-image min/max then (v-min)/(max-min) (image.cpp:537-576);
 * minmax: f32 [B,2] scratch ------------------------------------------------------------------ */
VX_API int vx_minmax_normalize(const float* depth, float* out, float* minmax, int B, int64_t pixels_per_image,
                               void* stream);
/* alpha_f32 -> alpha_u8: uint8(clamp(v,0,1)*255) (image-impl.h:36-38) */
VX_API int vx_f32_to_u8(const float* src, uint8_t* dst, int64_t n, void* stream);

/* ---- graph executor glue (csrc/graph.cpp, kernels_graph.hip): the stand-alone forms of the ops the reference's arch code emits between
 * its matrix products (ml.cpp:746-788 slice / concat, ggml_add / ggml_mul broadcasts, ggml_gelu / relu / scale, ggml_cont of a permute) */
/* dst[i . dst_stride] = scale * src[i . src_stride], i over ne (element strides, 0 = broadcast a source dimension) */
VX_API int vx_copy_strided_f16(const void* src, void* dst, const int64_t ne[4], const int64_t src_stride[4], const int64_t dst_stride[4], float scale,
                               void* stream);
/* y[i] = a[i] (op) b[i mod b_period]; op 0 add, 1 mul; each operand f16 or f32 */
VX_API int vx_binary_rows(int op, const void* a, int a_f32, const void* b, int b_f32, int64_t b_period, void* y, int y_f32, int64_t n, void* stream);
/* op 0 tanh-GELU (ggml_gelu), 1 ReLU, 2 x * s; f16 */
VX_API int vx_unary_f16(int op, const void* x, void* y, int64_t n, float s, void* stream);
/* op: 0 gelu, 1 relu, 2 x * s, 3 leaky relu max(x, s x) */
VX_API int vx_convert(const void* x, int x_f32, void* y, int y_f32, int64_t n, void* stream);
/* ggml_interpolate(NEAREST) of an NHWC f16 map (ml.cpp:782-788; esrgan.cpp:15) */
VX_API int vx_nearest_f16(const void* x, void* y, int B, int H, int W, int C, int OH, int OW, void* stream);
/* f32 image [pixels][C <= 16] -> one 32-channel f16 plane: values, then what the f16 rounding dropped, then zeros (input of an image's first conv) */
VX_API int vx_image_planes_f32(const float* x, void* y, int64_t n_pix, int C, void* stream);
/* patch_embed's im2col (nn.cpp:166-180) on the f32 image tensor [B,H,W,C]: rows (b,py,px), k = (ky,kx,c), zero padded to Kp, f16 */
/* 1x1 convolution to one channel: out f32 [M] = scale * act(sum_c x[m,c] w[c] + bias), x f16 [M][C] (depth-anything.cpp:91-95) */
VX_API int vx_conv1x1_to1_f32(const void* x, const float* w, float bias, int relu, float scale, float* out, int64_t M, int C, void* stream);
VX_API int vx_im2col_patches_f32(const float* x, void* patches, int B, int H, int W, int C, int ps, int Kp, void* stream);

#ifdef __cplusplus
}
#endif
#endif
