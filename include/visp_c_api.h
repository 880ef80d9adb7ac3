/*
 * visp_c_api.h -- the drop-in C ABI of the MI355X backend (lib/libvisioncpp.so).
 *
 * Part 1 re-exports, with identical names, signatures and semantics, the 11 symbols of the
 * reference's C API (reference src/visp/c-api.cpp:145-253), i.e. exactly what the reference's
 * ctypes binding (bindings/python/visioncpp/_lib.py:118-171) binds. Return value 1 = ok,
 * 0 = error with the message available from visp_get_last_error() (thread-local,
 * c-api.cpp:6-21). Built behind it: depth_anything, esrgan, sam (MobileSAM: image encoder + prompt encoder / mask decoder) and
 * birefnet (swin-T / BiRefNet-lite); migan returns an error ("not built in this backend").
 *
 * Part 2 is the batched, device-resident extension the reference does not have (its
 * depthany_compute is batch 1, src/visp/vision.cpp:155): it is what bench.py and a
 * data-parallel caller use.
 */
#ifndef VISP_C_API_H
#define VISP_C_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VISP_API __attribute__((visibility("default")))

/* visp::image_format, include/visp/image.h:17-29 */
enum visp_image_format {
    VISP_RGBA_U8 = 0, VISP_BGRA_U8, VISP_ARGB_U8, VISP_RGB_U8, VISP_ALPHA_U8,
    VISP_RGBA_F32, VISP_RGB_F32, VISP_ALPHA_F32
};
/* visp::backend_type, include/visp/ml.h:32-36 */
enum visp_backend_type { VISP_BACKEND_AUTO = 0, VISP_BACKEND_CPU = 1, VISP_BACKEND_GPU = 2, VISP_BACKEND_VULKAN = 258 };
/* visp::model_family, include/visp/vision.h:86-94 */
enum visp_model_family { VISP_SAM = 0, VISP_BIREFNET, VISP_DEPTH_ANYTHING, VISP_MIGAN, VISP_ESRGAN, VISP_FAMILY_COUNT };

/* == visp::image_view {i32x2 extent; int stride; image_format format; void const* data}
 * (include/visp/image.h:37-41) == c-api.cpp:121-127 visp_image_view == _lib.py:36-43 ImageView */
typedef struct visp_image_view {
    int32_t width;
    int32_t height;
    int32_t stride; /* bytes per row */
    int32_t format;
    void* data;
} visp_image_view;

typedef struct visp_image_data visp_image_data; /* visp::image_data, owned by the library */
typedef struct visp_device visp_device;         /* visp::backend_device */
typedef struct visp_model visp_model;           /* any_model (c-api.cpp:193) */

/* ---- Part 1: the reference's symbols ------------------------------------------------- */
VISP_API char const* visp_get_last_error(void);                                   /* c-api.cpp:147 */
VISP_API void visp_image_destroy(visp_image_data* img);                           /* :153 */
VISP_API int32_t visp_backend_load_all(char const* dir);                          /* :159 (no-op here: returns 1 backend) */
VISP_API int32_t visp_device_init(int32_t type, visp_device** out_device);        /* :164 */
VISP_API void visp_device_destroy(visp_device* d);                                /* :174 */
VISP_API int32_t visp_device_type(visp_device const* d);                          /* :178 */
VISP_API char const* visp_device_name(visp_device const* d);                      /* :182 */
VISP_API char const* visp_device_description(visp_device const* d);               /* :188 */
VISP_API int32_t visp_model_detect_family(char const* filepath, int32_t* out_family); /* :198 */
VISP_API int32_t visp_model_load(char const* filepath, visp_device const* dev, int32_t arch, visp_model** out); /* :206 */
VISP_API void visp_model_destroy(visp_model* model, int32_t arch);                /* :222 */
/* Depth-Anything: inputs[0] any u8 format; returns alpha_u8 after image_normalize (c-api.cpp:72-77) */
VISP_API int32_t visp_model_compute(visp_model* model, int32_t family, visp_image_view* inputs, int32_t n_inputs,
                                    int32_t* args, int32_t n_args, visp_image_view* out_image,
                                    visp_image_data** out_data);                  /* :230 */

/* ---- Part 2: MI355X extension ---------------------------------------------------------- */
/* like visp_device_init(GPU) but on an explicit HIP device index (one process per GPU) */
VISP_API int32_t visp_hip_device_init(int32_t device_index, visp_device** out_device);

enum { VISP_LOAD_DEFAULT = 0, VISP_LOAD_NO_UPLOAD = 1 /* parse + allocate only; weights arrive by broadcast */ };
VISP_API int32_t visp_model_load_ex(char const* filepath, visp_device const* dev, int32_t arch, int32_t flags,
                                    visp_model** out);
/* packed device weight arena (one allocation) -- the unit RCCL broadcasts at load */
VISP_API int32_t visp_depthany_weights_arena(visp_model* m, void** device_ptr, size_t* n_bytes);
/* call after a VISP_LOAD_NO_UPLOAD model's arena has been filled by the broadcast */
VISP_API int32_t visp_depthany_weights_ready(visp_model* m);

typedef struct visp_depthany_info {
    int32_t patch_size, embed_dim, n_layers, n_heads;
    int32_t image_size, image_multiple;
    int32_t feature_layers[4];
    float max_depth;
} visp_depthany_info;
VISP_API int32_t visp_depthany_get_info(visp_model const* m, visp_depthany_info* out);
/* depthany_image_extent (depth-anything.cpp:112-117) */
VISP_API int32_t visp_depthany_image_extent(visp_model const* m, int32_t w, int32_t h, int32_t* out_w, int32_t* out_h);

/* Pre-allocates the activation workspace for `batch` images of w x h (w,h multiples of the
 * patch size). Also called implicitly on the first compute of a new shape. */
VISP_API int32_t visp_depthany_reserve(visp_model* m, int32_t batch, int32_t w, int32_t h);

/* Batched hot path, everything on the device: rgb_u8 [batch,h,w,3] -> out [batch,h,w] f32 in [0,1]
 * (depthany_compute semantics per image: process_input, predict, image_normalize).
 * raw_out (nullable) receives the un-normalised depth. Asynchronous on `stream` (hipStream_t,
 * NULL = the model's own stream, in which case the call returns after synchronising). */
VISP_API int32_t visp_depthany_compute_batch_device(visp_model* m, void const* rgb_u8_dev, int32_t batch, int32_t w,
                                                    int32_t h, void* out_dev, void* raw_out_dev, void* stream);
/* the reference's C++ depthany_compute (vision.cpp:147-167): one image of any extent / u8 colour format -> alpha_f32 in [0, 1] at
 * the caller's extent (visp_model_compute returns the same map as alpha_u8, c-api.cpp:72-77); result owned by *out_data */
VISP_API int32_t visp_depthany_compute_f32(visp_model* m, visp_image_view const* image, visp_image_view* out_image, visp_image_data** out_data);
/* same with host buffers (H2D + compute + D2H on the model's stream, blocking) */
VISP_API int32_t visp_depthany_compute_batch_host(visp_model* m, uint8_t const* rgb_u8, int32_t batch, int32_t w,
                                                  int32_t h, float* out, float* raw_out);
/* Multi-GPU from C: models[i] is a depth_anything model loaded on its own device (visp_hip_device_init(i) + visp_model_load);
 * the batch is cut into n_models contiguous shards (sizes differ by at most one), each shard runs on its model's device from
 * its own host thread, outputs land at the matching offsets of `out` [batch, h, w] f32. No collective is involved: images are
 * independent. (Per-device kernel attributes are set per (kernel, device), so several devices in one process are fine. Models that share
 * one visp_device share its compute stream: their shards take turns instead of running concurrently.) */
VISP_API int32_t visp_depthany_compute_sharded(visp_model* const* models, int32_t n_models, uint8_t const* rgb_u8, int32_t batch,
                                               int32_t w, int32_t h, float* out);
/* Overlapped host pipeline (the reference's benchmark loop -- upload, compute, download per call, tests/benchmark.cpp:55-91 --
 * with the transfers hidden): n_slots (2..8) slots of pinned staging + device buffers; submit() chains H2D (copy stream) ->
 * forward (the model's stream) -> D2H (second copy stream) by events and returns at once with a ticket; wait() blocks until
 * that batch's normalised depth maps [batch, h, w] f32 are in pinned host memory and returns them (valid until the slot is
 * reused, i.e. for n_slots - 1 further submits). input() = the next slot's pinned rgb buffer [batch, h, w, 3] to fill in
 * place (then submit(NULL)); submit(ptr) copies from pageable memory. One thread per pipeline. */
typedef struct visp_depthany_pipeline visp_depthany_pipeline;
VISP_API int32_t visp_depthany_pipeline_create(visp_model* m, int32_t batch, int32_t w, int32_t h, int32_t n_slots, visp_depthany_pipeline** out);
VISP_API void visp_depthany_pipeline_destroy(visp_depthany_pipeline* p);
VISP_API int32_t visp_depthany_pipeline_input(visp_depthany_pipeline* p, uint8_t** out_pinned);
VISP_API int32_t visp_depthany_pipeline_submit(visp_depthany_pipeline* p, uint8_t const* rgb_u8_or_null, int32_t* out_ticket);
VISP_API int32_t visp_depthany_pipeline_wait(visp_depthany_pipeline* p, int32_t ticket, float const** out_pinned);
/* capture the launch sequence of the current reserved shape into a hipGraph and replay it on
 * every later compute of that shape (enable = 0 turns it off) */
VISP_API int32_t visp_depthany_use_graph(visp_model* m, int32_t enable);
/* encoder schedule: -1 = auto (default: 1 where the model has the kernel's shape), 0 = one launch per op group, 1 = one attention
 * + one token-stationary block launch per layer (csrc/kernels_block16.hip; embed dim 384 / mlp 1536 / head dim 64 only).
 * Results agree to f16 rounding. */
VISP_API int32_t visp_depthany_set_schedule(visp_model* m, int32_t schedule);
/* sub-batches of one step on parallel HIP streams (parallel branches of the step's hipGraph): 0 = automatic (3 from batch 24, 2
 * from batch 8), 1 = none, up to 4. Results are bit-identical for every value (images are independent). */
VISP_API int32_t visp_depthany_set_split(visp_model* m, int32_t n);

/* Named intermediate tensors of the last compute, converted to f32 on the host (parity tests;
 * the reference's counterpart is the workbench capture, tests/workbench.cpp:754-760).
 * Names: tokens, layer_<i>, dino_layer_<i>, reassemble_<i>, neck_conv_<i>, fusion_<i>, head_conv1, depth.
 * Only available when captures were enabled before the compute. */
VISP_API int32_t visp_depthany_enable_captures(visp_model* m, int32_t enable);
VISP_API int32_t visp_depthany_read_capture(visp_model* m, char const* name, float* host_out, int64_t capacity,
                                            int64_t* n_written, int64_t shape[4]);

/* per-kernel-group timing of the last compute (HIP events on the compute stream); writes up to
 * cap entries, returns count via n. Only when enabled. */
typedef struct visp_timing { char name[32]; float ms; int32_t launches; double flops; double bytes; } visp_timing;
VISP_API int32_t visp_depthany_enable_timing(visp_model* m, int32_t enable);
VISP_API int32_t visp_depthany_read_timing(visp_model* m, visp_timing* out, int32_t cap, int32_t* n);

/* ---- ESRGAN extension (family 4; reference vision.h:284-304, vision.cpp:208-253) ---------------------------
 * visp_model_load / visp_model_compute work as in the reference (one image in, rgba_u8 at extent*scale out).
 * The batched entry points take B images of one extent and push all their tiles through the network together. */
typedef struct visp_esrgan_info { int32_t scale, n_blocks, n_filters, growth, tile_group; } visp_esrgan_info;
VISP_API int32_t visp_esrgan_get_info(visp_model const* m, visp_esrgan_info* out);
/* tiles per network pass (workspace bound / cache-locality knob); default 64 or $VISP_ESRGAN_TILE_GROUP */
VISP_API int32_t visp_esrgan_set_tile_group(visp_model* m, int32_t tiles);
VISP_API int32_t visp_esrgan_weights_arena(visp_model* m, void** device_ptr, size_t* n_bytes);
VISP_API int32_t visp_esrgan_weights_ready(visp_model* m);
/* tile_scale(tile_layout(extent, 224, 16), scale): out8 = image w,h, overlap x,y, n_tiles x,y, tile w,h (image.cpp:612-629) */
/* host-only: read a GGUF file completely (header, key/values, tensor infos, bounds of every tensor's data) as visp_model_load
 * does before it packs weights; 0 + error message if the file is malformed */
VISP_API int32_t visp_gguf_validate(char const* filepath, int32_t* out_n_tensors);
/* host-only image_scale of the reference (src/visp/image.cpp:328-356: stb_image_resize semantics, csrc/image_resize.cpp):
 * any supported format; the result is owned by *out_data (visp_image_destroy) */
VISP_API int32_t visp_image_scale(visp_image_view const* src, int32_t width, int32_t height, visp_image_view* out_image,
                                  visp_image_data** out_data);
/* host-only image_u8_to_f32 of the reference (src/visp/image.cpp:215-255, image-impl.h:17-34): dst = (src / 255 + offset) * scale per
 * channel, any u8 format -> the float format `format` (what *_process_input applies); result owned by *out_data */
VISP_API int32_t visp_image_u8_to_f32(visp_image_view const* src, int32_t format, float const offset[4], float const scale[4],
                                      visp_image_view* out_image, visp_image_data** out_data);
/* host-only image_normalize of the reference (src/visp/image.cpp:537-582): min-max of an alpha_f32 image mapped to [min, max]
 * (depthany_process_output); result owned by *out_data */
VISP_API int32_t visp_image_normalize(visp_image_view const* src, float min, float max, visp_image_view* out_image, visp_image_data** out_data);
VISP_API int32_t visp_esrgan_tile_layout(int32_t w, int32_t h, int32_t scale, int32_t out8[8]);
/* img: u8 [B,h,w,channels(format)] (rgba/bgra/argb/rgb), out: rgba_u8 [B, h*scale, w*scale, 4]; device pointers;
 * stream = hipStream_t or NULL (NULL: the device's stream, synchronised before returning) */
VISP_API int32_t visp_esrgan_compute_batch_device(visp_model* m, void const* img, int32_t batch, int32_t w, int32_t h, int32_t format,
                                                  void* out_rgba, void* stream);
VISP_API int32_t visp_esrgan_compute_batch_host(visp_model* m, uint8_t const* img, int32_t batch, int32_t w, int32_t h, int32_t format,
                                                uint8_t* out_rgba);
/* esrgan_generate (esrgan.cpp:55-79) on n rgb_f32 tiles [n,h,w,3] -> [n,h*scale,w*scale,3] (host buffers; for tests) */
VISP_API int32_t visp_esrgan_generate_host(visp_model* m, float const* rgb, int32_t n, int32_t w, int32_t h, float* out);
VISP_API int32_t visp_esrgan_enable_timing(visp_model* m, int32_t enable);
VISP_API int32_t visp_esrgan_read_timing(visp_model* m, visp_timing* out, int32_t cap, int32_t* n);

/* ---- MobileSAM extension (family 0; reference vision.h:186-222, vision.cpp:26-92, mobile-sam.cpp) ----
 * visp_model_load / visp_model_compute work as in the reference: one image + 2 (point) or 4 (box) integer arguments in
 * pixels of that image -> alpha_u8 mask at the image's extent (sam_encode + sam_compute, c-api.cpp:34-52). A file
 * with only the enc.* tensors loads too; sam_compute then fails. */
/* sam_encode: any extent / u8 colour format; the embedding is kept on the device with the model */
VISP_API int32_t visp_sam_encode(visp_model* m, visp_image_view const* image);
/* embedding of the last visp_sam_encode: f32 [64, 64, 256] (rows, columns, channels), shape returned via shape[3] */
VISP_API int32_t visp_sam_read_embedding(visp_model* m, float* host_out, int64_t capacity, int64_t shape[3]);
/* sam_compute on the embedding of the last visp_sam_encode (several prompts per image, vision.cpp:54-92) */
VISP_API int32_t visp_sam_compute(visp_model* m, int32_t const* prompt, int32_t n_prompt, visp_image_view* out_image, visp_image_data** out_data);
/* test hook: mask logits [4][256][256] (f16-rounded; kept only while captures are enabled) and iou predictions [4] of the last sam_compute */
VISP_API int32_t visp_sam_read_masks(visp_model* m, float* masks, int64_t capacity, float iou[4]);
/* rgb: u8 [B, 1024, 1024, 3] already at the model extent -> out f32 [B, 64, 64, 256]; device pointers;
 * stream = hipStream_t or NULL (NULL: the device's stream, synchronised before returning) */
VISP_API int32_t visp_sam_encode_batch_device(visp_model* m, void const* rgb, int32_t batch, void* out, void* stream);
VISP_API int32_t visp_sam_encode_batch_host(visp_model* m, uint8_t const* rgb, int32_t batch, float* out);
/* opt-in, never the default (BASELINE.json configs[4]: "fp8 GGUF weights on CDNA4 fp8 MFMA"; the reference has no fp8 type): the MLPs of the transformer
 * stages run on the block-scaled e4m3 matrix instruction -- weights quantised per output channel (on first use, from the f16 weights), activations per
 * token. Changes the embedding by several percent of its scale (tests/test_fp8_decision.py, tests/test_gpu_tinyvit.py): a measurement, not a recommendation. */
VISP_API int32_t visp_sam_set_fp8_mlp(visp_model* m, int32_t enable);
VISP_API int32_t visp_sam_weights_arena(visp_model* m, void** device_ptr, size_t* n_bytes);
VISP_API int32_t visp_sam_weights_ready(visp_model* m);
/* test hook as for depth_anything: "patch_embed", "layer_0" .. "layer_3" = the stage outputs of the last batch, f16 -> f32 */
VISP_API int32_t visp_sam_enable_captures(visp_model* m, int32_t enable);
VISP_API int32_t visp_sam_read_capture(visp_model* m, char const* name, float* host_out, int64_t capacity, int64_t* n_written, int64_t shape[4]);
VISP_API int32_t visp_sam_enable_timing(visp_model* m, int32_t enable);
VISP_API int32_t visp_sam_read_timing(visp_model* m, visp_timing* out, int32_t cap, int32_t* n);

/* ---- BiRefNet (family 1; reference src/visp/arch/birefnet.cpp, vision.cpp:98-132): visp_model_load + visp_model_compute run
 * birefnet_compute (any 8-bit colour image -> alpha_u8 mask at its extent). Batched extension: rgb_u8 [batch, h, w, 3] at the
 * model extent (visp_birefnet_image_extent; multiples of 64) -> sigmoid mask f32 [batch, h, w]. */
VISP_API int32_t visp_birefnet_image_extent(visp_model* m, int32_t w, int32_t h, int32_t* out_w, int32_t* out_h);
VISP_API int32_t visp_birefnet_compute_batch_device(visp_model* m, void const* rgb_u8, int32_t batch, int32_t w, int32_t h, void* mask_f32, void* stream);
VISP_API int32_t visp_birefnet_compute_batch_host(visp_model* m, uint8_t const* rgb_u8, int32_t batch, int32_t w, int32_t h, float* mask);

/* ---- SWIN encoder, the backbone of BiRefNet (reference src/visp/arch/swin.cpp, swin_encode) ---------
 * The encoder of a birefnet GGUF (tensors bb.*, KV swin.embed_dim) on its own: visp_swin_load reads only the backbone; the
 * visp_swin_* calls also accept a full birefnet model handle. Destroyed with visp_model_destroy(m, VISP_BIREFNET). */
VISP_API int32_t visp_swin_load(char const* filepath, visp_device const* dev, visp_model** out);
/* dims[3 i .. 3 i + 2] = {w_i, h_i, C_i} of output i for an image of extent (w, h) (swin.cpp:229-262) */
VISP_API int32_t visp_swin_output_dims(visp_model* m, int32_t w, int32_t h, int32_t dims[12]);
/* rgb_u8 [batch, h, w, 3] (device) -> outs[i] f32 [batch, h_i, w_i, C_i] (device; NHWC = the reference's CWHN result tensors):
 * birefnet_process_input's normalisation (birefnet.cpp:259-270) + swin_encode. w, h multiples of 32. stream = hipStream_t or
 * NULL (NULL: the device's stream, synchronised before returning). */
VISP_API int32_t visp_swin_encode_batch_device(visp_model* m, void const* rgb_u8, int32_t batch, int32_t w, int32_t h, void* const outs[4],
                                               void* stream);
VISP_API int32_t visp_swin_encode_batch_host(visp_model* m, uint8_t const* rgb_u8, int32_t batch, int32_t w, int32_t h, float* const outs[4]);
/* test hooks as for the other families: "patch_embed", "block_<layer>_<i>" (f16 -> f32), per-kernel-group timing */
/* Shift-mask semantics of the window attention. 0 (default) = the reference as written: swin::layer passes the layer's attn_mask to
 * every block and swin::block forwards it unconditionally (swin.cpp:128-139, 226-237), so the edge windows of UNSHIFTED blocks are
 * masked too. 1 = shifted blocks only, as the reference's torch twin / the original Swin (tests/test_birefnet.py:249-255). */
VISP_API int32_t visp_swin_set_mask_mode(visp_model* m, int32_t shifted_only);
VISP_API int32_t visp_swin_enable_captures(visp_model* m, int32_t enable);
VISP_API int32_t visp_swin_read_capture(visp_model* m, char const* name, float* host_out, int64_t capacity, int64_t* n_written, int64_t shape[4]);
VISP_API int32_t visp_swin_enable_timing(visp_model* m, int32_t enable);
VISP_API int32_t visp_swin_read_timing(visp_model* m, visp_timing* out, int32_t cap, int32_t* n);

/* ---- graph layer: the executor boundary the reference's arch code is written against -----------------------------------------------------------
 * replaces include/visp/ml.h:154-256 (compute_graph, compute_graph_init / _allocate, compute, model_ref::weights / find, compute_graph_input /
 * _output, transfer_to_backend / transfer_from_backend) and the builders of src/visp/nn.h + ml.cpp:746-788, which the reference implements on
 * ggml (ggml_new_tensor / ggml_mul_mat / ggml_add / ... / ggml_gallocr / ggml_backend_graph_compute). A tensor handle is an index into its
 * graph (>= 0); shapes are ggml's ne order (ne[0] contiguous). include/visp/ml.h is the C++ face of these entries (model_ref, tensor, the
 * nn.h builder names). One generic entry adds a node: `op` is a visp_graph_op, `src` the operand handles, `iparams` / `fparams` its integer
 * and float arguments (listed per op below). */
typedef struct visp_graph visp_graph;
typedef struct visp_weights visp_weights; /* model_weights (ml.h:126-149): tensors by name + their packed device images, shared by the graphs over them */
enum visp_graph_op {
    VISP_OP_LINEAR = 2,            /* src x, w [K,N], (b)                                  nn.cpp:6-12 */
    VISP_OP_LAYER_NORM = 3,        /* src x, w, b; f0 eps                                  nn.cpp:14-19 */
    VISP_OP_GELU = 4, VISP_OP_RELU = 5,
    VISP_OP_SCALE = 6,             /* f0 factor */
    VISP_OP_ADD = 7, VISP_OP_MUL = 8, /* second operand broadcast over trailing dimensions */
    VISP_OP_CONV_2D = 9,           /* src x [C,W,H,N], w [Cin,kw,kh,Cout], (b); i0 stride, i1 pad        nn.cpp:72-100 */
    VISP_OP_CONV_TRANSPOSE_2D = 10,/* src x, w [kw,kh,Cout,Cin], (b); i0 stride (== kernel)               nn.cpp:117-129 */
    VISP_OP_INTERPOLATE = 11,      /* i0 w, i1 h, i2 mode: 1 bilinear, 2 bicubic, | 256 align_corners     ml.cpp:782-788 */
    VISP_OP_ATTENTION = 12,        /* src q, k, v [head_dim, heads, tokens, batch]; f0 scale               nn.cpp:210-244 */
    VISP_OP_CONCAT = 13,           /* src a, b; i0 dim                                                      ml.cpp:770-780 */
    VISP_OP_SLICE = 14,            /* i[3d], i[3d+1], i[3d+2] = begin, end, step of dimension d            ml.cpp:746-768 */
    VISP_OP_RESHAPE = 15,          /* i0..i3 ne */
    VISP_OP_REPEAT = 16,           /* i0..i3 ne */
    VISP_OP_PATCH_EMBED = 17,      /* src x f32 [C,W,H,N], w, (b); i0 patch size                           nn.cpp:166-180 */
    VISP_OP_CONT = 18,
    /* extensions for callers that keep images in HBM (the reference does both steps on the host): */
    VISP_OP_IMAGE_U8_TO_F32 = 19,  /* src x u8 input [3,W,H,N]; f0..f2 mean, f3..f5 1/std: (x / 255 - mean) / std        image.cpp:215-255 */
    VISP_OP_IMAGE_NORMALIZE = 20,  /* src x f32 [1,W,H,N]: per-image min-max to [0, 1]                                   image.cpp:537-582 */
    VISP_OP_LEAKY_RELU = 21        /* f0 negative slope                                                                    ggml_leaky_relu, esrgan.cpp:17, 24 */
};
/* every f16 / f32 tensor of the file becomes a weight (model_load + model_transfer, ml.cpp:206-217, 449-516); conv kernels listed in
 * <arch>.conv2d_weights of a whcn file are presented as [Cin,kw,kh,Cout]. Device images are made when a graph that uses a tensor is
 * allocated, once per (tensor, role), and are reused by later graphs over the same weights. */
VISP_API int32_t visp_weights_load(char const* gguf_path, visp_weights** out);
/* model_file (ml.h:85-103; ml.cpp:206-281): a GGUF file read into memory, its key/values by name. get_int needs an i32 value, the array
 * getter an i32 array of exactly n entries (the reference's rules); a missing key is an error naming it. */
typedef struct visp_file visp_file;
VISP_API int32_t visp_file_load(char const* gguf_path, visp_file** out);
VISP_API void visp_file_destroy(visp_file* f);
VISP_API int32_t visp_file_n_tensors(visp_file const* f, int64_t* out);
VISP_API int32_t visp_file_get_int(visp_file const* f, char const* key, int32_t* out);
VISP_API int32_t visp_file_get_int_array(visp_file const* f, char const* key, int32_t* out, int64_t n);
VISP_API int32_t visp_file_get_string(visp_file const* f, char const* key, char* out, int64_t capacity, int64_t* needed);
VISP_API int32_t visp_weights_from_file(visp_file const* f, visp_weights** out); /* model_transfer: the file's tensors as model weights */
VISP_API int32_t visp_weights_create(visp_weights** out);
VISP_API int32_t visp_weights_add(visp_weights* w, char const* name, int32_t dtype, int64_t const ne[4], float const* data);
VISP_API void visp_weights_destroy(visp_weights* w); /* graphs over the weights keep them alive */
/* compute_graph_init (ml.cpp:531-543). weights == NULL: the graph starts with an empty store of its own (visp_graph_add_weight) */
VISP_API int32_t visp_graph_create(visp_weights* weights, visp_graph** out);
VISP_API void visp_graph_destroy(visp_graph* g);
VISP_API int32_t visp_graph_add_weight(visp_graph* g, char const* name, int32_t dtype, int64_t const ne[4], float const* data, int32_t* out);
VISP_API int32_t visp_graph_find_weight(visp_graph* g, char const* name, int32_t* out); /* *out = -1 when absent (model_ref::find) */
VISP_API int32_t visp_graph_input(visp_graph* g, int32_t dtype /* 0 f32, 1 f16, 24 u8 image bytes */, int64_t const ne[4], char const* name, int32_t* out);
VISP_API int32_t visp_graph_op(visp_graph* g, int32_t op, int32_t const* src, int32_t n_src, int64_t const* iparams, int32_t n_iparams,
                               float const* fparams, int32_t n_fparams, int32_t* out);
VISP_API int32_t visp_graph_set_name(visp_graph* g, int32_t tensor, char const* name);
VISP_API int32_t visp_graph_get_tensor(visp_graph* g, char const* name, int32_t* out); /* ggml_get_tensor; -1 when absent */
VISP_API int32_t visp_graph_output(visp_graph* g, int32_t tensor, char const* name);
VISP_API int32_t visp_graph_tensor_info(visp_graph const* g, int32_t tensor, int32_t* dtype, int64_t ne[4], int32_t* is_constant);
/* constants (weights and everything folded from them): their f32 values */
VISP_API int32_t visp_graph_read_constant(visp_graph const* g, int32_t tensor, float* out, int64_t capacity);
/* compute_graph_allocate (ml.cpp:545-552): lower to launches, pack the weights the graph uses, plan + allocate the arena on `dev`.
 * dev == NULL: lower and plan only -- no device work (visp_graph_describe shows the launch list and the arena size) */
VISP_API int32_t visp_graph_allocate(visp_graph* g, visp_device const* dev);
/* before visp_graph_allocate. 1 (default): the lowering may replace whole node groups by the kernels written for them -- the dino::layer group
 * (embed 384 / mlp 1536 / heads of 64) by attention + one token-stationary block launch on an f32 residual stream, 3x3 convs by the LDS-ring conv,
 * interpolate -> conv by the resizing halo loader, interpolate -> conv 1x1 by projection-then-resize, a sliced GEMM operand by row addressing.
 * 0: one launch per epilogue-fused node on f16 activations. */
VISP_API int32_t visp_graph_set_fused_models(visp_graph* g, int32_t enable);
VISP_API int32_t visp_graph_use_hip_graph(visp_graph* g, int32_t enable); /* replay the launch list as one hipGraph from the second compute on */
VISP_API int32_t visp_graph_compute(visp_graph* g);   /* blocking (ml.cpp:559-562) */
VISP_API int32_t visp_graph_tensor_set(visp_graph* g, int32_t tensor, void const* data, size_t n_bytes);
VISP_API int32_t visp_graph_tensor_get(visp_graph* g, int32_t tensor, void* data, size_t n_bytes, int32_t as_f32);
/* one line per launch, then "launches=.. arena_bytes=.. unshared_bytes=.. constant_bytes=.."; returns the length needed */
VISP_API int32_t visp_graph_describe(visp_graph const* g, char* out, int64_t capacity, int64_t* needed);

#ifdef __cplusplus
}
#endif
#endif
