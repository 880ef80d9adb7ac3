// Tensor graph + executor: the layer the reference's arch code is written against (include/visp/ml.h:154-256 compute_graph,
// model_ref, compute_graph_input/output, transfer_*; src/visp/ml.cpp:531-741, 746-788; src/visp/nn.h), rebuilt for the MI355X
// backend. The reference delegates this layer to ggml (graph container, ggml_gallocr liveness allocator, one backend kernel per
// node). Here a graph is a list of nn-level nodes (linear, layer_norm, conv_2d, conv_transpose_2d, attention, interpolate, the
// element-wise glue, slice / concat / repeat / reshape) that `graph_allocate` lowers ONCE into a launch list on the fused kernel
// families of this library:
//   * activation / residual / ReLU-on-load fusion by use count (linear -> gelu, conv -> relu, relu -> conv, conv -> add become
//     GEMM epilogues and loader flags: what ggml runs as 2-4 nodes and HBM round trips is one launch),
//   * constant folding on the host of everything computed from weights alone (the bicubic position-embedding resize, the
//     repeated cls token: dino.cpp:10-44),
//   * weights packed per consumer role (f16 [N][K pad 64] GEMM operand, pixel-shuffle row order for conv_transpose, f32 vectors),
//   * a liveness-based arena in HBM (a buffer is recycled after its last reader), the whole launch list optionally captured
//     into one hipGraph.
// Tensors follow ggml's conventions at the API: ne[0] is the contiguous dimension, activations of 2D ops are CWHN
// ([C, W, H, N] = NHWC in memory, model_build_flag::cwhn), linear weights are [in, out], conv weights [Cin, kw, kh, Cout],
// conv_transpose weights [kw, kh, Cout, Cin]. Activations are f16 in HBM (f32 accumulation inside the kernels), vectors f32.
#pragma once
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "depthany.h"
#include "gguf.h"

namespace visp {

enum graph_op : int32_t {
    gop_input = 0,
    gop_weight,
    gop_linear,           // x, w [K, N], (b)                         nn.cpp:6-12
    gop_layer_norm,       // x, w, b; f0 = eps                         nn.cpp:14-19
    gop_gelu,             // ggml_gelu                                 dino.cpp:54
    gop_relu,             // ggml_relu
    gop_scale,            // f0 = factor                               depth-anything.cpp:93-95
    gop_add,              // a + b, b broadcast over trailing dims     ggml_add
    gop_mul,              // a * b, same broadcast                     dino.cpp:48-50
    gop_conv_2d,          // x CWHN, w [Cin, kw, kh, Cout], (b); i0 = stride, i1 = pad       nn.cpp:72-100
    gop_conv_transpose_2d,// x CWHN, w [kw, kh, Cout, Cin], (b); i0 = stride (== kernel)      nn.cpp:117-129
    gop_interpolate,      // x CWHN; i0 = w, i1 = h, i2 = mode (ggml: 1 bilinear, 2 bicubic, | 256 align_corners)  ml.cpp:782-788
    gop_attention,        // q, k, v [hd, heads, T, B]; f0 = scale -> [hd * heads, T, B]         nn.cpp:210-244 (without the output linear)
    gop_concat,           // a, b; i0 = dim                            ml.cpp:770-780
    gop_slice,            // x; i[3d..3d+2] = begin, end, step         ml.cpp:746-768
    gop_reshape,          // x; i0..i3 = ne (a view)
    gop_repeat,           // x; i0..i3 = ne (source dims 1 or equal)   dino.cpp:38-40
    gop_patch_embed,      // x f32 [C, W, H, N], w conv weight, (b); i0 = patch size            nn.cpp:166-180
    gop_cont,             // a view: every tensor of this executor is contiguous
    // extensions for callers that keep images in HBM (the batched entries of this library); the reference does both steps on the host
    gop_image_u8_to_f32,  // x u8 [3, W, H, N] -> f32: (x / 255 - f0..f2) * f3..f5     image.cpp:215-255 as depthany_process_input calls it
    gop_image_normalize,  // x f32 [1, W, H, N] -> f32: per-image min-max to [0, 1]     image.cpp:537-582 (depthany_process_output)
    gop_leaky_relu,       // f0 = negative slope                         ggml_leaky_relu (esrgan.cpp:17, 24)
    gop_count
};
const char* graph_op_name(int32_t op);

constexpr int32_t gdt_f32 = 0, gdt_f16 = 1, gdt_u8 = 24; // ggml type ids (24 = I8: one byte per element; image inputs)

struct graph_node {
    int32_t op = gop_input;
    int32_t dtype = gdt_f16;
    int64_t ne[4] = {1, 1, 1, 1};
    int src[4] = {-1, -1, -1, -1};
    int n_src = 0;
    int64_t ip[12] = {0};
    float fp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::string name;
    bool is_output = false;
    bool constant = false;        // a weight, or computed from weights alone: evaluated on the host when the node is made
    std::vector<float> host;      // folded constants: the values
    const float* cdata = nullptr; // weights: the values live in the weight store
    const float* values() const { return cdata ? cdata : host.data(); }
    // filled by graph_allocate
    int alias_of = -1;            // shares the buffer of that node (views, fused activations)
    int buffer = -1;              // index into graph::buffers (materialised nodes)
    int64_t n_elements() const { return ne[0] * ne[1] * ne[2] * ne[3]; }
    size_t n_bytes() const { return (size_t)n_elements() * (dtype == gdt_f16 ? 2 : (dtype == gdt_u8 ? 1 : 4)); }
};

struct graph_buffer {
    size_t bytes = 0, offset = 0;
    int first = 0, last = 0; // launch indices of the first writer and the last reader
    bool persistent = false; // inputs and outputs
    void* external = nullptr; // graph_bind_external: an input / output that lives in the caller's memory instead of the arena
};

struct graph_launch {
    std::string desc; // e.g. "gemm[gelu] M=43840 N=1536 K=384 <- layer0.mlp.fc1": what tests and `describe` show
    std::function<void(void* stream)> run;
    std::string group; // timing group (per-group HIP-event times of the model entries: "block", "attention", "fusion_rcu", ...)
    double flops = 0, bytes = 0; // algorithmic work of the launch (2 x MACs; operand bytes), for the roofline lines
};

// model_weights (ml.h:126-149): the tensors of a model by name, f32 on the host, and -- once a graph over them has been allocated on a
// device -- their packed device images per consumer role (GEMM operand, pixel-shuffle rows, f32 vector, ...). Shared by every graph built
// over the model: rebuilding the graph for another input extent (the reference's lazy graph build, vision.cpp:150-158) uploads nothing.
struct weight_store {
    struct entry {
        int32_t dtype = gdt_f16;
        int64_t ne[4] = {1, 1, 1, 1};
        std::vector<float> data;
    };
    std::map<std::string, entry, std::less<>> tensors;
    backend_device const* dev = nullptr;                    // bound by the first graph_allocate with a device
    std::map<std::pair<std::string, int>, void*> packs;     // (name, role) -> device image
    std::vector<void*> allocs;
    size_t device_bytes = 0;
    // One device arena for every image (optional; depthany's loader sizes it from a planning pass): images are then bump-allocated in
    // lowering order, which is the same on every rank -- the arena IS the model on the device, one RCCL broadcast moves it. no_data: the
    // tensors came from a header-only read (zeros on the host): images are laid out but not uploaded (the broadcast fills them).
    device_buffer arena;
    size_t arena_used = 0;
    bool no_data = false;
    std::mutex mutex;                                       // guards dev / packs / allocs / device_bytes
    ~weight_store();
};
std::shared_ptr<weight_store> weights_create();
// all f16 / f32 tensors of a GGUF file (conv kernels listed in <arch>.conv2d_weights of a whcn file are presented as CWHN)
std::shared_ptr<weight_store> weights_load(char const* gguf_path);
std::shared_ptr<weight_store> weights_from_file(model_file const& file); // model_transfer (ml.cpp:449-516) from a file already read
void weights_add(weight_store&, char const* name, int32_t dtype, const int64_t ne[4], const float* data);

struct graph {
    backend_device const* dev = nullptr; // null: build / fold / plan only (no device work; the CPU tests)
    std::shared_ptr<weight_store> store;
    std::vector<graph_node> nodes;
    std::map<std::string, int, std::less<>> weights; // name -> node (made when first looked up)
    std::map<std::string, int, std::less<>> named;   // graph_set_name, inputs, outputs

    bool allocated = false;
    std::vector<graph_buffer> buffers;
    std::vector<graph_launch> launches;
    size_t arena_bytes = 0, sum_bytes = 0; // with recycling / if every buffer had its own storage
    size_t plan_store_bytes = 0;           // planning only (no device): bytes of weight images this graph would add to the shared store
    device_buffer arena;
    std::vector<void*> const_allocs;       // packed weights and constants (device)
    size_t const_bytes = 0;
    bool use_hip_graph = false;
    void* graph_exec = nullptr;
    // fused_models: the lowering may replace whole node groups by the kernels written for them (the dino::layer group -> attention + one
    // token-stationary block launch with an f32 residual stream, interpolate -> conv 3x3 -> the resizing halo loader, ...). false: one
    // launch per (epilogue-fused) node on f16 activations -- what visp_depthany_set_schedule(model, 0) selects.
    bool fused_models = true;
    ~graph();
};

graph* graph_create(std::shared_ptr<weight_store> weights); // null: the graph gets a store of its own (graph_add_weight)
int graph_add_weight(graph&, char const* name, int32_t dtype, const int64_t ne[4], const float* data);
int graph_find_weight(graph&, char const* name); // -1 if absent
int graph_input(graph&, int32_t dtype, const int64_t ne[4], char const* name);
int graph_add(graph&, int32_t op, const int* src, int n_src, const int64_t* ip, int n_ip, const float* fp, int n_fp);
void graph_set_name(graph&, int t, char const* name);
int graph_get_tensor(graph&, char const* name); // -1 if absent
void graph_output(graph&, int t, char const* name);
void graph_allocate(graph&, backend_device const* dev); // dev == null: lower and plan only
void graph_compute(graph&);
void graph_tensor_set(graph&, int t, const void* data, size_t bytes);            // the tensor's own dtype
void graph_tensor_get(graph&, int t, void* data, size_t bytes, bool as_f32);     // as_f32: converted on the host
// an input or output tensor lives at `ptr` (device memory of the caller, at least the tensor's size) instead of in the graph's arena;
// may be called again with another pointer between computes (not while a captured hipGraph of this graph is in use)
void graph_bind_external(graph&, int t, void* ptr);
void* graph_tensor_device_ptr(graph&, int t);                                    // where an input / output tensor lives
std::string graph_describe(graph const&); // one line per launch, then the arena summary

} // namespace visp
