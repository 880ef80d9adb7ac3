// Depth-Anything-V2 on the MI355X backend: model load (GGUF -> packed f16 weight arena in HBM),
// static per-shape schedule and the batched executor. Host C++ only; all device work goes
// through the vx_* C ABI (include/visp_hip_kernels.h).
//
// Mirrors the reference's high-level API for this family (include/visp/vision.h:224-252,
// 339-347; src/visp/vision.cpp:137-167; src/visp/arch/depth-anything.cpp; src/visp/arch/dino.cpp).
#pragma once
#include <array>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "gguf.h"
#include "image.h"

namespace visp {

enum class backend_type : int32_t { cpu = 1, gpu = 2, vulkan = 2 | 1 << 8 }; // include/visp/ml.h:32-36

struct backend_device { // include/visp/ml.h:44-55, here: one HIP device + its compute stream
    int index = 0;
    std::string name, description;
    void* stream = nullptr;
    size_t total_mem = 0;
    int n_cu = 0;
    backend_type type() const { return backend_type::gpu; }
    // Every model and graph made on this device shares `stream`. Whatever enqueues on it, captures a hipGraph on it or synchronises it
    // holds the device's turn (device_turn below), so that host threads driving different models of one device cannot interleave a
    // capture with another thread's launches ("capturing stream has unjoined work"). Recursive: entry points nest.
    mutable std::recursive_mutex turn;
    ~backend_device();
};
// makes the device current for the calling thread and holds its turn for the scope
struct device_turn {
    std::unique_lock<std::recursive_mutex> lock;
    explicit device_turn(backend_device const& dev);
};
backend_device* backend_init(int device_index); // throws visp::exception if no gfx950 device

struct dino_params { // vision.h:124-129
    int patch_size = 16, embed_dim = 768, n_layers = 12, n_heads = 12;
};
struct depthany_params { // vision.h:236-243
    int image_size = 518, image_multiple = 14;
    i32x2 image_extent = {{518, 518}};
    float max_depth = 1;
    std::array<int, 4> feature_layers = {2, 5, 8, 11};
    dino_params dino;
};

dino_params dino_detect_params(model_file const&);                        // dino.cpp:119-126
depthany_params depthany_detect_params(model_file const&);                // depth-anything.cpp:119-128
i32x2 depthany_image_extent(i32x2 extent, depthany_params const&);        // depth-anything.cpp:112-117

// One GEMM-shaped weight in a packed arena: f16 [N][K] (both padded for the kernel's tiles, pads are zero) + optional f32 bias [N]
// (the model loaders that pack for hand schedules: tinyvit.cpp, swin.cpp, birefnet.cpp)
struct packed_gemm {
    size_t w = 0, b = SIZE_MAX; // byte offsets into the arena
    size_t dw = SIZE_MAX;       // 3x3 convs with Cin % 32 == 0, Cout in {32, 64}: the same kernel as vx_dconv3x3_f16 slabs
    int d_cin = 0;              // [cin/32][9][cout][32] f16 (16-byte groups at g ^ ((n >> 2) & 3)), see kernels_dconv.hip
    int N = 0, K = 0;           // padded
    int n_real = 0, k_real = 0; // for FLOP accounting and n_valid
};
struct packed_vec { size_t off = SIZE_MAX; int n = 0; }; // f32 vector

struct device_buffer { void* ptr = nullptr; size_t bytes = 0; };

struct capture_entry { void* dev = nullptr; int64_t shape[4] = {1, 1, 1, 1}; bool f16 = true; };

struct timing_entry { std::string name; float ms = 0; int launches = 0; double flops = 0, bytes = 0; };

// every model handle starts with its family (the C ABI's handles are untyped: c-api.cpp:193 any_model)
enum model_family_id : int32_t { family_sam = 0, family_birefnet, family_depth_anything, family_migan, family_esrgan, family_count };
struct model_base {
    int32_t family;
    explicit model_base(int32_t f) : family(f) {}
};

struct depthany_pipeline;
struct depthany_step; // the lowered graphs of one (batch, extent, schedule, split): csrc/depthany.cpp
struct weight_store;  // csrc/graph.h
struct depthany_model : model_base { // vision.h:339-347 counterpart: {backend, weights, params, graph, input, output}
    depthany_model();
    backend_device const* backend = nullptr;
    depthany_params params;
    std::shared_ptr<weight_store> store; // the model's tensors by name + their device images per consumer role, all in one arena (shared by cloned executors)
    device_buffer weight_arena;          // that arena: what rank 0 broadcasts and the other ranks receive (not owned here: the store frees it)
    bool weights_uploaded = false;
    bool block_shape = false; // embed dim 384 / mlp 1536 / head dim 64: the encoder groups lower to the token-stationary block kernel
    std::vector<std::unique_ptr<depthany_step>> steps; // a few lowered shapes, least recently used first out
    device_buffer host_io;    // device staging of the blocking host entry
    bool use_graph = false, captures = false, timing = false;
    int split = 0; // sub-batches of a step on parallel streams: 0 = automatic (3 from batch 24, 2 from batch 8; $VISP_SPLIT overrides), 1 = none, 2..4
    bool timing_split = false; // with timing: keep the step's sub-batch split (launches timed per stream while the other streams run)
    depthany_pipeline* shard_pipeline = nullptr; // visp_depthany_compute_sharded's overlapped host pipeline for this model (owned, lazily made)
    int schedule = -1; // lowering: -1 auto (node groups on the kernels written for them), 0 one launch per epilogue-fused node (GEMM launches), 1 = auto, refused where the model has not the block kernel's shape
    std::map<std::string, capture_entry> capture_bufs;
    std::vector<timing_entry> last_timing;
    // side streams of the sub-batches (fork / join by events, captured into the hipGraph as parallel branches)
    void* aux_stream[3] = {nullptr, nullptr, nullptr};
    void* fork_event = nullptr;
    void* join_event[3] = {nullptr, nullptr, nullptr};
    ~depthany_model();
};
// drops the captured hipGraphs (a setting that changes what a step launches was changed)
void depthany_drop_captured_steps(depthany_model&);

enum load_flags { load_default = 0, load_no_upload = 1 };
depthany_model* depthany_load_model(char const* filepath, backend_device const& dev, int flags = load_default);
// after a load_no_upload model's arena has been filled (RCCL broadcast from the rank that read the file)
void depthany_weights_ready(depthany_model&);

// a second executor over the SAME device weights: its own workspace, streams, events and hipGraph, so that two forwards can be in
// flight on the device at once (the source model must outlive it)
depthany_model* depthany_clone_executor(depthany_model const&);

void depthany_reserve(depthany_model&, int batch, int w, int h);
// rgb_u8 [B,h,w,3] on the device -> out f32 [B,h,w] normalised (+ raw depth if raw_out != null)
void depthany_compute_batch_device(depthany_model&, void const* rgb_dev, int batch, int w, int h, void* out_dev,
                                   void* raw_out_dev, void* stream);
void depthany_compute_batch_host(depthany_model&, uint8_t const* rgb, int batch, int w, int h, float* out, float* raw_out);

// Overlapped host pipeline: batches of `batch` images of one extent flow through n_slots slots; for slot k the upload of its
// input (pinned staging -> HBM on a copy stream), the forward (the model's compute stream) and the download of its output
// (HBM -> pinned, second copy stream) are chained by events, so the upload of batch k+1 and the download of batch k-1 run
// under the compute of batch k. The reference times upload + compute + download per call (tests/benchmark.cpp:55-91); this
// is that loop with the transfers hidden. VISP_PIPELINE_EXECUTORS=2 lets consecutive batches overlap ON the device as well
// (alternating executors: own workspace, graph and compute stream over the same weights). Measured at batch 32: 7.0-7.2 ms per
// batch against 6.85 with one executor -- three sub-batch streams already fill the chip, six contend -- so the default is one.
struct depthany_pipeline {
    depthany_model* model = nullptr;
    std::vector<depthany_model*> exec;      // exec[0] = model, the others are clones owned by the pipeline
    std::vector<void*> compute_stream;      // one per executor; [0] = the backend's stream
    int batch = 0, w = 0, h = 0, n_slots = 0;
    size_t in_bytes = 0, out_bytes = 0;
    struct slot {
        void *pin_in = nullptr, *pin_out = nullptr, *dev_in = nullptr, *dev_out = nullptr;
        void *uploaded = nullptr, *computed = nullptr, *downloaded = nullptr; // events
        bool busy = false;
    };
    std::vector<slot> slots;
    void *h2d_stream = nullptr, *d2h_stream = nullptr;
    int next = 0;
    long n_submitted = 0;
    ~depthany_pipeline();
};
depthany_pipeline* depthany_pipeline_create(depthany_model&, int batch, int w, int h, int n_slots);
// next slot's pinned input buffer (fill it, then submit(nullptr)) -- or pass pageable memory to submit and it is copied in
uint8_t* depthany_pipeline_input(depthany_pipeline&);
int depthany_pipeline_submit(depthany_pipeline&, uint8_t const* rgb_or_null); // returns the ticket (= slot index)
// blocks until that batch is back in pinned host memory; returns it (valid until the slot is submitted again)
float const* depthany_pipeline_wait(depthany_pipeline&, int ticket);

// one shard of the multi-device entry: pageable host buffers through the overlapped pipeline in chunks of 32 (bit-identical to
// depthany_compute_batch_host)
void depthany_compute_shard_host(depthany_model&, uint8_t const* rgb, int count, int w, int h, float* out);

// reference API: any extent, any u8 colour format, batch 1 (vision.cpp:147-167) -> alpha_f32 at the input extent
image_data depthany_compute(depthany_model&, image_view image);

} // namespace visp
