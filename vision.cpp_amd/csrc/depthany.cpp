#include "depthany.h"
#include "graph.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "../../include/visp_hip_kernels.h"
#include "visp_util.h"

namespace visp {

#define VX(call)                                        \
    do {                                                \
        if (!(call)) throw except("%s", vx_last_error()); \
    } while (0)

namespace {
template <typename T>
T round_up(T x, T m) { return (x + m - 1) / m * m; }
} // namespace

//
// backend (reference src/visp/ml.cpp:59-95)

device_turn::device_turn(backend_device const& dev) : lock(dev.turn) { VX(vx_set_device(dev.index)); }

backend_device* backend_init(int device_index) {
    int n = vx_device_count();
    if (n <= 0) throw except("Failed to initialize backend, no suitable device available");
    if (device_index < 0 || device_index >= n) throw except("Failed to initialize backend, device index %d out of range (%d devices)", device_index, n);
    VX(vx_set_device(device_index));
    char name[256] = {0}, arch[128] = {0};
    size_t total = 0;
    int n_cu = 0;
    VX(vx_device_info(device_index, name, sizeof name, arch, sizeof arch, &total, nullptr, &n_cu));
    if (strncmp(arch, "gfx950", 6) != 0)
        throw except("Failed to initialize backend: device %d is %s, this backend is built for gfx950 (MI355X) only", device_index, arch);
    auto* d = new backend_device;
    d->index = device_index;
    d->name = std::string("HIP") + std::to_string(device_index);
    d->description = std::string(name) + " (" + arch + ")";
    d->total_mem = total;
    d->n_cu = n_cu;
    VX(vx_stream_create(&d->stream));
    return d;
}
backend_device::~backend_device() {
    if (stream) vx_stream_destroy(stream);
}

//
// params (reference dino.cpp:119-126, depth-anything.cpp:112-128)

dino_params dino_detect_params(model_file const& file) {
    dino_params p{};
    p.patch_size = file.get_int("dino.patch_size");
    p.embed_dim = file.get_int("dino.embed_dim");
    p.n_heads = file.get_int("dino.n_heads");
    p.n_layers = file.get_int("dino.n_layers");
    return p;
}
depthany_params depthany_detect_params(model_file const& file) {
    depthany_params p;
    p.dino = dino_detect_params(file);
    p.image_size = file.get_int("depthanything.image_size");
    file.get_array("depthanything.feature_layers", p.feature_layers.data(), 4);
    return p;
}
i32x2 depthany_image_extent(i32x2 extent, depthany_params const& p) {
    int min_side = std::min(extent[0], extent[1]);
    int tgt_side = std::max(p.image_size, next_multiple(min_side, p.image_multiple));
    i32x2 target = {{extent[0] * tgt_side / min_side, extent[1] * tgt_side / min_side}};
    return i32x2{{next_multiple(target[0], p.image_multiple), next_multiple(target[1], p.image_multiple)}};
}


//
// The model on the graph layer. What the reference does with ggml -- depthany_predict builds the graph through model_ref (vision.cpp:147-158,
// depth-anything.cpp:100-110), compute() runs it (ml.cpp:559-562) -- is what this backend does with csrc/graph.h: the nodes below are
// lowered by graph_allocate onto the kernels written for their groups (token-stationary block kernel, LDS-ring conv, resizing loaders, head
// kernel). The weights are a weight_store (tensors by name; device images per consumer role in ONE arena, which is what RCCL broadcasts).

namespace {

// Depth-Anything-V2 as graph nodes. Module paths are the GGUF tensor-name prefixes (= the HF state dict, convert.py:437-442); the arithmetic is
// the reference's (dino.cpp:10-110, depth-anything.cpp:15-110), written against graph_add instead of ggml.
struct net_builder {
    graph& g;
    depthany_params const& P;
    bool keep; // name the module boundaries AND keep them readable (graph outputs): the parity tests' captures

    int weight(std::string const& name) const {
        const int t = graph_find_weight(g, name.c_str());
        if (t < 0) throw except("tensor not found: %s", name.c_str());
        return t;
    }
    int find(std::string const& name) const { return graph_find_weight(g, name.c_str()); }
    int node(int32_t op, std::vector<int> const& src, std::vector<int64_t> const& ip = {}, std::vector<float> const& fp = {}) const {
        return graph_add(g, op, src.data(), (int)src.size(), ip.data(), (int)ip.size(), fp.data(), (int)fp.size());
    }
    int boundary(int t, std::string const& name) const {
        if (keep) graph_output(g, t, name.c_str());
        else graph_set_name(g, t, name.c_str());
        return t;
    }
    std::array<int64_t, 4> shape(int t) const { return {g.nodes[t].ne[0], g.nodes[t].ne[1], g.nodes[t].ne[2], g.nodes[t].ne[3]}; }

    int with_params(int32_t op, std::string const& mod, int x, std::vector<int64_t> const& ip = {}) const { // weight [+ bias] of a module
        const int b = find(mod + ".bias");
        return b >= 0 ? node(op, {x, weight(mod + ".weight"), b}, ip) : node(op, {x, weight(mod + ".weight")}, ip);
    }
    int dense(std::string const& mod, int x) const { return with_params(gop_linear, mod, x); }
    int conv(std::string const& mod, int x, int stride = 1, int pad = 0) const { return with_params(gop_conv_2d, mod, x, {stride, pad}); }
    int norm(std::string const& mod, int x) const { return node(gop_layer_norm, {x, weight(mod + ".weight"), weight(mod + ".bias")}, {}, {1e-6f}); }
    int resize(int x, int64_t w, int64_t h) const { return node(gop_interpolate, {x}, {w, h, 1 | 256}); } // bilinear, align_corners
    int relu(int x) const { return node(gop_relu, {x}); }
    int add(int a, int b) const { return node(gop_add, {a, b}); }

    // position embeddings for a pw x ph grid: as stored, or the patch part bicubic-resized (all constants: folded on the host)
    int position_embedding(int64_t D, int64_t pw, int64_t ph, bool square) const {
        const int pos = weight("backbone.embeddings.position_embeddings");
        const int64_t stored = g.nodes[pos].ne[1] - 1;
        if (stored == pw * ph && square) return pos;
        const int64_t side = (int64_t)(std::sqrt((float)stored) + 0.01f);
        const int64_t all[3] = {0, (int64_t)1 << 60, 1};
        auto rows = [&](int64_t b, int64_t e) {
            return node(gop_slice, {pos}, {all[0], all[1], all[2], b, e, 1, all[0], all[1], all[2], all[0], all[1], all[2]});
        };
        int grid = node(gop_reshape, {rows(1, stored + 1)}, {D, side, side, 1});
        grid = node(gop_interpolate, {grid}, {pw, ph, 2}); // bicubic
        grid = node(gop_reshape, {grid}, {D, pw * ph, 1, 1});
        return node(gop_concat, {rows(0, 1), grid}, {1});
    }

    // image f32 [3, W, H, B] -> depth f32 [1, W, H, B]
    int build(int image) const {
        const auto [c, W, H, B] = shape(image);
        (void)c;
        const int ps = P.dino.patch_size, D = P.dino.embed_dim, NH = P.dino.n_heads;
        const int64_t pw = W / ps, ph = H / ps, T = pw * ph + 1;
        const std::string emb = "backbone.embeddings.";

        // ---- tokens: patch projection, cls token in front, position embeddings on top
        int x = with_params(gop_patch_embed, emb + "patch_embeddings.projection", image, {ps});
        x = node(gop_reshape, {x}, {D, pw * ph, B, 1});
        int cls = weight(emb + "cls_token");
        if (B > 1) cls = node(gop_repeat, {cls}, {D, 1, B, 1});
        x = node(gop_concat, {cls, x}, {1});
        x = boundary(add(x, position_embedding(D, pw, ph, W == H)), "tokens");

        // ---- encoder: pre-LN blocks with LayerScale; the taps go through the shared final LayerNorm
        std::vector<int> taps;
        const float scale = 1.0f / std::sqrt((float)D / (float)NH);
        for (int i = 0; i < P.dino.n_layers; ++i) {
            const std::string L = "backbone.encoder.layer." + std::to_string(i) + ".";
            const int ln1 = norm(L + "norm1", x);
            int qkv[3];
            const char* part[3] = {"query", "key", "value"};
            for (int k = 0; k < 3; ++k) qkv[k] = node(gop_reshape, {dense(L + "attention.attention." + part[k], ln1)}, {D / NH, NH, T, B});
            int att = node(gop_attention, {qkv[0], qkv[1], qkv[2]}, {}, {scale});
            att = dense(L + "attention.output.dense", att);
            x = add(x, node(gop_mul, {att, weight(L + "layer_scale1.lambda1")}));
            int mlp = dense(L + "mlp.fc1", norm(L + "norm2", x));
            mlp = dense(L + "mlp.fc2", node(gop_gelu, {mlp}));
            x = boundary(add(x, node(gop_mul, {mlp, weight(L + "layer_scale2.lambda1")})), "layer_" + std::to_string(i));
            for (int f = 0; f < 4; ++f)
                if (P.feature_layers[f] == i && taps.size() < 4) {
                    const int t = norm("backbone.layernorm", x);
                    graph_output(g, t, ("dino_layer_" + std::to_string(i)).c_str()); // (kept in every build: the neck reads them long after their layer)
                    taps.push_back(t);
                }
        }
        if (taps.size() != 4) throw except("depthany: expected 4 feature layers, found %d", (int)taps.size());

        // ---- neck: reassemble (cls token off, 1x1 projection, resample), 3x3 to the fusion width
        int level[4];
        for (int j = 0; j < 4; ++j) {
            const std::string R = "neck.reassemble_stage.layers." + std::to_string(j) + ".";
            const int64_t all[3] = {0, (int64_t)1 << 60, 1};
            int y = node(gop_slice, {taps[(size_t)j]}, {all[0], all[1], all[2], 1, T, 1, all[0], all[1], all[2], all[0], all[1], all[2]});
            y = node(gop_reshape, {y}, {D, pw, ph, B});
            y = conv(R + "projection", y);
            if (j == 0) y = with_params(gop_conv_transpose_2d, R + "resize", y, {4});
            if (j == 1) y = with_params(gop_conv_transpose_2d, R + "resize", y, {2});
            if (j == 3) y = conv(R + "resize", y, 2, 1);
            boundary(y, "reassemble_" + std::to_string(j));
            level[j] = boundary(conv("neck.convs." + std::to_string(j), y, 1, 1), "neck_conv_" + std::to_string(j));
        }
        // ---- fusion, coarse to fine: [x + unit1(skip)] -> unit2 -> resize to the next level (x2 at the end) -> 1x1 projection
        auto unit = [&](std::string const& mod, int v) { // relu -> conv -> relu -> conv, + input
            const int a = conv(mod + ".convolution1", relu(v), 1, 1);
            return add(v, conv(mod + ".convolution2", relu(a), 1, 1));
        };
        int fused = -1;
        for (int i = 0; i < 4; ++i) {
            const std::string F = "neck.fusion_stage.layers." + std::to_string(i) + ".";
            const int skip = level[3 - i];
            int v = i == 0 ? skip : add(fused, unit(F + "residual_layer1", skip));
            v = unit(F + "residual_layer2", v);
            const auto s = i < 3 ? shape(level[2 - i]) : std::array<int64_t, 4>{0, 2 * g.nodes[v].ne[1], 2 * g.nodes[v].ne[2], 0};
            fused = boundary(conv(F + "projection", resize(v, s[1], s[2])), "fusion_" + std::to_string(i));
        }
        // ---- head
        int d = boundary(conv("head.conv1", fused, 1, 1), "head_conv1");
        d = relu(conv("head.conv2", resize(d, W, H), 1, 1));
        d = relu(conv("head.conv3", d));
        if (P.max_depth != 1) d = node(gop_scale, {d}, {}, {P.max_depth});
        return d;
    }
};

} // namespace

// The lowered graphs of one (batch, extent, schedule, split, captures): one graph per sub-batch -- sub-batches run on parallel streams (captured
// as parallel branches of the step's hipGraph), so that kernels with different bottlenecks of different sub-batches overlap
struct depthany_step {
    int B = 0, W = 0, H = 0;
    bool fused = true, captures = false;
    struct part {
        std::unique_ptr<graph> g;
        int b0 = 0, nb = 0;
        int in = -1, raw = -1, out = -1;
    };
    std::vector<part> parts;
    device_buffer staging;     // u8 input | normalised out | raw depth of the whole batch: what a captured hipGraph reads and writes
    void* graph_exec = nullptr;
    long last_use = 0;
    ~depthany_step() {
        if (graph_exec) vx_graph_destroy(graph_exec);
        vx_free(staging.ptr);
    }
};

namespace {

int pick_split(depthany_model const& m, int B, bool captures) {
    static const int split_env = getenv("VISP_SPLIT") ? atoi(getenv("VISP_SPLIT")) : 0;
    // measured at batch 32 (profiles/r02_split_streams.txt, r03_split_ab.txt): 3 sub-batches; 2 / 4 within 2 %
    const int want = m.split > 0 ? m.split : (split_env > 0 ? split_env : (B >= 24 ? 3 : (B >= 8 ? 2 : 1)));
    return ((!m.timing || m.timing_split) && !captures && want > 1 && want <= 4 && B >= 2 * want) ? want : 1;
}

std::unique_ptr<graph> build_part(depthany_model& m, int nb, int W, int H, bool fused, bool captures, depthany_step::part& part) {
    std::unique_ptr<graph> g(graph_create(m.store));
    g->fused_models = fused;
    const int64_t ne[4] = {3, W, H, nb};
    part.in = graph_input(*g, gdt_u8, ne, "image_u8");
    const float norm[6] = {0.485f, 0.456f, 0.406f, 1.f / 0.229f, 1.f / 0.224f, 1.f / 0.225f}; // depthany_process_input (depth-anything.cpp:130-140)
    const int image = graph_add(*g, gop_image_u8_to_f32, &part.in, 1, nullptr, 0, norm, 6);
    net_builder net{*g, m.params, captures};
    part.raw = net.build(image);
    graph_output(*g, part.raw, "depth");
    part.out = graph_add(*g, gop_image_normalize, &part.raw, 1, nullptr, 0, nullptr, 0); // depthany_process_output (depth-anything.cpp:142-149)
    graph_output(*g, part.out, "normalized");
    return g;
}

depthany_step& ensure_step(depthany_model& m, int B, int W, int H) {
    depthany_params const& P = m.params;
    if (B <= 0 || W <= 0 || H <= 0) throw except("depthany: invalid batch/extent %d x %dx%d", B, W, H);
    if (!m.weights_uploaded) throw except("depthany: weights were not uploaded (VISP_LOAD_NO_UPLOAD): fill the arena and call visp_depthany_weights_ready first");
    if (W % P.dino.patch_size || H % P.dino.patch_size) throw except("depthany: extent %dx%d is not a multiple of the patch size %d", W, H, P.dino.patch_size);
    static const bool block_off = getenv("VISP_NO_BLOCK") != nullptr;
    const bool fused = m.schedule != 0 && !(block_off && m.schedule < 0);
    const int n_split = pick_split(m, B, m.captures);
    static long clock = 0;
    for (auto& s : m.steps)
        if (s->B == B && s->W == W && s->H == H && s->fused == fused && s->captures == m.captures && (int)s->parts.size() == n_split) {
            s->last_use = ++clock;
            return *s;
        }
    while (m.steps.size() >= 3) { // a few shapes stay lowered (the timing pass, a second extent); the least recently used one goes
        auto lru = std::min_element(m.steps.begin(), m.steps.end(), [](auto const& a, auto const& b) { return a->last_use < b->last_use; });
        m.steps.erase(lru);
    }
    auto s = std::make_unique<depthany_step>();
    s->B = B; s->W = W; s->H = H; s->fused = fused; s->captures = m.captures;
    s->parts.resize((size_t)n_split);
    for (int j = 0; j < n_split; ++j) {
        depthany_step::part& p = s->parts[(size_t)j];
        p.b0 = (int)((long)B * j / n_split);
        p.nb = (int)((long)B * (j + 1) / n_split) - p.b0;
        p.g = build_part(m, p.nb, W, H, fused, m.captures, p);
        graph_allocate(*p.g, m.backend);
    }
    s->last_use = ++clock;
    m.steps.push_back(std::move(s));
    return *m.steps.back();
}

// input / output slices of the parts live at (rgb, out, raw) + the part's image offset
void bind_parts(depthany_step& s, const void* rgb, void* out, void* raw) {
    const size_t px = (size_t)s.W * s.H;
    for (auto& p : s.parts) {
        graph_bind_external(*p.g, p.in, const_cast<uint8_t*>(static_cast<const uint8_t*>(rgb)) + (size_t)p.b0 * px * 3);
        graph_bind_external(*p.g, p.out, static_cast<float*>(out) + (size_t)p.b0 * px);
        graph_bind_external(*p.g, p.raw, raw ? static_cast<float*>(raw) + (size_t)p.b0 * px : nullptr);
    }
}

// the parts' launch lists on parallel streams (part 0 on `stream`, the others on the model's side streams, forked and joined by events);
// with timing on, HIP events around every launch on its own stream, summed per group
#ifdef VISP_TIMING_ABLATIONS
// diagnostic builds only (make ABLATE=1): VISP_ABLATE_LAUNCHES="a-b,c,d-e" skips those indices of every part's launch list (results invalid): what
// a group of launches costs END TO END in the overlapped step
bool ablated_launch(int l) {
    static const std::vector<std::pair<int, int>> ranges = [] {
        std::vector<std::pair<int, int>> r;
        const char* e = getenv("VISP_ABLATE_LAUNCHES");
        while (e && *e) {
            char* end = nullptr;
            const int a = (int)strtol(e, &end, 10);
            int b = a;
            if (*end == '-') b = (int)strtol(end + 1, &end, 10);
            r.emplace_back(a, b);
            e = *end ? end + 1 : end;
        }
        return r;
    }();
    for (auto const& ab : ranges)
        if (l >= ab.first && l <= ab.second) return true;
    return false;
}
#endif

void run_parts(depthany_model& m, depthany_step& s, void* stream) {
    const bool timed = m.timing;
    struct stamp { void* ev; int part, launch; };
    std::vector<stamp> stamps;
    auto mark = [&](int part, int launch, void* strm) {
        void* ev = nullptr;
        VX(vx_event_create(&ev));
        VX(vx_event_record(ev, strm));
        stamps.push_back({ev, part, launch});
    };
    const int n = (int)s.parts.size();
    if (n > 1) VX(vx_event_record(m.fork_event, stream));
    for (int j = 0; j < n; ++j) {
        void* strm = j == 0 ? stream : m.aux_stream[j - 1];
        if (j > 0) VX(vx_stream_wait_event(strm, m.fork_event));
        graph& g = *s.parts[(size_t)j].g;
        for (int l = 0; l < (int)g.launches.size(); ++l) {
            if (timed) mark(j, l, strm);
#ifdef VISP_TIMING_ABLATIONS
            if (ablated_launch(l)) continue;
#endif
            g.launches[(size_t)l].run(strm);
        }
        if (timed) mark(j, -1, strm);
        if (j > 0) {
            VX(vx_event_record(m.join_event[j - 1], strm));
            VX(vx_stream_wait_event(stream, m.join_event[j - 1]));
        }
    }
    if (!timed) return;
    std::map<std::string, timing_entry> by;
    std::vector<std::string> order;
    for (size_t i = 0; i + 1 < stamps.size(); ++i) {
        if (stamps[i].launch < 0) continue; // the end mark of a part
        graph_launch const& l = s.parts[(size_t)stamps[i].part].g->launches[(size_t)stamps[i].launch];
        float ms = 0;
        VX(vx_event_elapsed_ms(stamps[i].ev, stamps[i + 1].ev, &ms));
        static const bool detail = getenv("VISP_TIMING_DETAIL") != nullptr; // one row per launch (tools/dpt_launches.py)
        std::string name = l.group.empty() ? "other" : l.group;
        if (detail) name += "#" + std::to_string(i) + " " + l.desc.substr(0, 40);
        auto it = by.find(name);
        if (it == by.end()) { order.push_back(name); it = by.emplace(name, timing_entry{name, 0, 0, 0, 0}).first; }
        it->second.ms += ms;
        it->second.launches += 1;
        it->second.flops += l.flops;
        it->second.bytes += l.bytes;
    }
    m.last_timing.clear();
    for (auto const& nm : order) m.last_timing.push_back(by[nm]);
    for (auto& st : stamps) vx_event_destroy(st.ev);
}

// copies of the module boundaries a captured run kept (graph outputs of the single part) for visp_depthany_read_capture
void collect_captures(depthany_model& m, depthany_step& s, void* stream) {
    graph& g = *s.parts[0].g;
    for (auto const& kv : g.named) {
        graph_node const& n = g.nodes[kv.second];
        if (!n.is_output || n.op == gop_input || kv.first == "normalized") continue;
        int r = kv.second;
        while (g.nodes[r].alias_of >= 0) r = g.nodes[r].alias_of;
        if (g.nodes[r].buffer < 0) continue;
        capture_entry& c = m.capture_bufs[kv.first];
        if (c.dev) vx_free(c.dev);
        c.f16 = g.nodes[r].dtype == gdt_f16;
        // ggml ne order -> the slowest dimension first, batch in front: tokens [D, T, B, 1] -> {B, T, D, 1}; maps [C, W, H, B] -> {B, H, W, C}
        const int64_t* ne = n.ne;
        const bool token_rows = kv.first == "tokens" || kv.first.rfind("layer_", 0) == 0 || kv.first.rfind("dino_layer_", 0) == 0;
        if (token_rows) { c.shape[0] = ne[2]; c.shape[1] = ne[1]; c.shape[2] = ne[0]; c.shape[3] = 1; }
        else { c.shape[0] = ne[3]; c.shape[1] = ne[2]; c.shape[2] = ne[1]; c.shape[3] = ne[0]; }
        const size_t bytes = (size_t)n.n_elements() * (c.f16 ? 2 : 4);
        VX(vx_malloc(&c.dev, bytes));
        VX(vx_memcpy_d2d(c.dev, graph_tensor_device_ptr(g, kv.second), bytes, stream));
    }
}

} // namespace

depthany_model::depthany_model() : model_base(family_depth_anything) {}

depthany_model* depthany_load_model(char const* filepath, backend_device const& dev, int flags) {
    const bool with_data = !(flags & load_no_upload);
    model_file file = model_load(filepath, /*header_only=*/!with_data);
    if (file.arch() != "depthanything")
        throw except("Model %s has architecture '%.*s', expected 'depthanything'", filepath, (int)file.arch().size(), file.arch().data());

    auto model = std::make_unique<depthany_model>();
    model->backend = &dev;
    model->params = depthany_detect_params(file);
    depthany_params const& P = model->params;
    const int D = P.dino.embed_dim;
    if (D % P.dino.n_heads != 0 || D / P.dino.n_heads != 64)
        throw except("Unsupported DINO head dim %d (this backend implements head_dim 64)", P.dino.n_heads ? D / P.dino.n_heads : 0);
    if (D % 128 != 0) throw except("Unsupported embed dim %d (must be a multiple of 128)", D);
    model->store = weights_from_file(file); // a header-only read gives zero-filled tensors: shapes, and an arena layout to receive the broadcast into
    {
        auto fc1 = model->store->tensors.find("backbone.encoder.layer.0.mlp.fc1.weight");
        model->block_shape = P.dino.n_layers > 0 && fc1 != model->store->tensors.end() && vx_dino_block_supported(D, (int)fc1->second.ne[1], 64) != 0;
    }

    device_turn turn(dev);
    VX(vx_dconv_prepare());
    for (void*& s : model->aux_stream) VX(vx_stream_create(&s));
    VX(vx_event_create(&model->fork_event));
    for (void*& e : model->join_event) VX(vx_event_create(&e));

    // The weight arena: one planning pass of the model's graph at its native extent says how many bytes of images the lowering makes of the
    // weights (per consumer role); the second pass packs them into one allocation in lowering order -- the same order on every rank.
    const int side = P.image_size - P.image_size % P.dino.patch_size;
    size_t need = 0;
    {
        depthany_step::part scratch;
        std::unique_ptr<graph> plan = build_part(*model, 1, side, side, true, false, scratch);
        graph_allocate(*plan, nullptr);
        need = plan->plan_store_bytes;
    }
    weight_store& ws = *model->store;
    ws.arena.bytes = need + (1u << 20);
    VX(vx_malloc(&ws.arena.ptr, ws.arena.bytes));
    {
        depthany_step::part scratch;
        std::unique_ptr<graph> warm = build_part(*model, 1, side, side, true, false, scratch);
        graph_allocate(*warm, &dev);
    }
    model->weight_arena.ptr = ws.arena.ptr;
    model->weight_arena.bytes = round_up<size_t>(ws.arena_used, 256);
    model->weights_uploaded = with_data;
    return model.release();
}

void depthany_weights_ready(depthany_model& m) {
    device_turn turn(*m.backend);
    // the few tensors the HOST needs when it builds a graph -- position embeddings and cls token (folded for other grids), the scalar bias of the
    // last conv (a kernel argument) -- come back out of the arena the broadcast filled
    weight_store& ws = *m.store;
    for (const char* name : {"backbone.embeddings.position_embeddings", "backbone.embeddings.cls_token", "head.conv3.bias"}) {
        auto t = ws.tensors.find(name);
        if (t == ws.tensors.end()) throw except("depthany: tensor %s is missing", name);
        std::vector<float>& host = t->second.data;
        if (auto p = ws.packs.find({name, 0}); p != ws.packs.end()) { // its f32 image
            VX(vx_memcpy_d2h(host.data(), p->second, host.size() * 4, m.backend->stream));
        } else if (auto q = ws.packs.find({name, 1}); q != ws.packs.end()) { // its f16 image (what the per-node lowering concatenates into the tokens)
            std::vector<uint16_t> h(host.size());
            VX(vx_memcpy_d2h(h.data(), q->second, h.size() * 2, m.backend->stream));
            for (size_t i = 0; i < h.size(); ++i) host[i] = f16_to_f32(h[i]);
        } else throw except("depthany: the arena holds no image of %s", name);
    }
    ws.no_data = true; // (stays: every other image this rank has is the one in the arena)
    m.weights_uploaded = true;
}

depthany_model::~depthany_model() {
    delete shard_pipeline; // (its executor 0 is this model: only the slots, streams and clones go)
    shard_pipeline = nullptr;
    if (backend) vx_set_device(backend->index);
    steps.clear();
    for (void* s : aux_stream) vx_stream_destroy(s);
    vx_event_destroy(fork_event);
    for (void* e : join_event) vx_event_destroy(e);
    for (auto& c : capture_bufs) vx_free(c.second.dev);
}

depthany_model* depthany_clone_executor(depthany_model const& src) {
    if (!src.weights_uploaded) throw except("depthany: cannot clone an executor before the weights are on the device");
    device_turn turn(*src.backend);
    auto m = std::make_unique<depthany_model>();
    m->backend = src.backend;
    m->params = src.params;
    m->store = src.store; // the same tensors and device images
    m->weight_arena = src.weight_arena;
    m->block_shape = src.block_shape;
    m->weights_uploaded = true;
    m->use_graph = src.use_graph;
    m->schedule = src.schedule;
    for (void*& s : m->aux_stream) VX(vx_stream_create(&s));
    VX(vx_event_create(&m->fork_event));
    for (void*& e : m->join_event) VX(vx_event_create(&e));
    return m.release();
}

void depthany_drop_captured_steps(depthany_model& m) {
    if (m.backend) vx_set_device(m.backend->index);
    for (auto& s : m.steps)
        if (s->graph_exec) { vx_graph_destroy(s->graph_exec); s->graph_exec = nullptr; }
}

void depthany_reserve(depthany_model& m, int B, int W, int H) {
    device_turn turn(*m.backend);
    ensure_step(m, B, W, H);
}

void depthany_compute_batch_device(depthany_model& m, void const* rgb_dev, int batch, int w, int h, void* out_dev, void* raw_out_dev, void* stream) {
    if (!m.weights_uploaded) throw except("depthany: weights were not uploaded (load_no_upload) and no arena broadcast was marked complete");
    device_turn turn(*m.backend);
    depthany_step& s = ensure_step(m, batch, w, h);
    const bool own_stream = stream == nullptr;
    void* strm = own_stream ? m.backend->stream : stream;
    const size_t in_bytes = (size_t)batch * h * w * 3, out_bytes = (size_t)batch * h * w * 4;
    if (m.use_graph && !m.captures && !m.timing) {
        // the captured launch sequence bakes pointers in: it reads and writes the step's own staging buffers
        if (!s.staging.ptr) {
            s.staging.bytes = round_up<size_t>(in_bytes, 256) + 2 * round_up<size_t>(out_bytes, 256);
            VX(vx_malloc(&s.staging.ptr, s.staging.bytes));
        }
        uint8_t* in = static_cast<uint8_t*>(s.staging.ptr);
        float* out = reinterpret_cast<float*>(in + round_up<size_t>(in_bytes, 256));
        float* raw = reinterpret_cast<float*>(in + round_up<size_t>(in_bytes, 256) + round_up<size_t>(out_bytes, 256));
        VX(vx_memcpy_d2d(in, rgb_dev, in_bytes, strm));
        if (!s.graph_exec) {
            bind_parts(s, in, out, raw);
            run_parts(m, s, strm); // eager once: every kernel attribute is set before the capture
            VX(vx_graph_begin_capture(strm));
            try {
                run_parts(m, s, strm);
            } catch (...) {
                void* dead = nullptr;
                vx_graph_end_capture(strm, &dead);
                if (dead) vx_graph_destroy(dead);
                throw;
            }
            VX(vx_graph_end_capture(strm, &s.graph_exec));
        }
        VX(vx_graph_launch(s.graph_exec, strm));
        VX(vx_memcpy_d2d(out_dev, out, out_bytes, strm));
        if (raw_out_dev) VX(vx_memcpy_d2d(raw_out_dev, raw, out_bytes, strm));
    } else {
        bind_parts(s, rgb_dev, out_dev, raw_out_dev);
        run_parts(m, s, strm);
        if (m.captures) collect_captures(m, s, strm);
    }
    if (own_stream) VX(vx_stream_sync(strm));
}

void depthany_compute_batch_host(depthany_model& m, uint8_t const* rgb, int batch, int w, int h, float* out, float* raw_out) {
    device_turn turn(*m.backend);
    void* s = m.backend->stream;
    const size_t in_bytes = (size_t)batch * h * w * 3, out_bytes = (size_t)batch * h * w * 4;
    if (m.host_io.bytes < in_bytes + 2 * out_bytes + 512) {
        VX(vx_free(m.host_io.ptr));
        m.host_io = {};
        VX(vx_malloc(&m.host_io.ptr, in_bytes + 2 * out_bytes + 512));
        m.host_io.bytes = in_bytes + 2 * out_bytes + 512;
    }
    uint8_t* in = static_cast<uint8_t*>(m.host_io.ptr);
    uint8_t* o = in + round_up<size_t>(in_bytes, 256);
    uint8_t* r = o + round_up<size_t>(out_bytes, 256);
    VX(vx_memcpy_h2d(in, rgb, in_bytes, s));
    const bool g = m.use_graph;
    m.use_graph = false; // one blocking call: nothing to replay
    try {
        depthany_compute_batch_device(m, in, batch, w, h, o, r, s);
    } catch (...) {
        m.use_graph = g;
        throw;
    }
    m.use_graph = g;
    VX(vx_memcpy_d2h(out, o, out_bytes, s));
    if (raw_out) VX(vx_memcpy_d2h(raw_out, r, out_bytes, s));
}

//
// overlapped host pipeline

depthany_pipeline* depthany_pipeline_create(depthany_model& m, int batch, int w, int h, int n_slots) {
    if (n_slots < 2 || n_slots > 8) throw except("depthany pipeline: %d slots (2..8)", n_slots);
    device_turn turn(*m.backend);
    depthany_reserve(m, batch, w, h);
    auto p = std::make_unique<depthany_pipeline>();
    p->model = &m; p->batch = batch; p->w = w; p->h = h; p->n_slots = n_slots;
    p->in_bytes = (size_t)batch * h * w * 3;
    p->out_bytes = (size_t)batch * h * w * 4;
    VX(vx_stream_create(&p->h2d_stream));
    VX(vx_stream_create(&p->d2h_stream));
    const char* ne = getenv("VISP_PIPELINE_EXECUTORS");
    const int n_exec = ne ? std::max(1, std::min(2, atoi(ne))) : 1; // two forwards in flight measured SLOWER (7.0-7.2 vs 6.85 ms per batch of 32)
    p->exec.push_back(&m);
    p->compute_stream.push_back(m.backend->stream);
    for (int e = 1; e < n_exec; ++e) {
        p->exec.push_back(depthany_clone_executor(m));
        void* cs = nullptr;
        VX(vx_stream_create(&cs));
        p->compute_stream.push_back(cs);
        depthany_reserve(*p->exec.back(), batch, w, h);
    }
    p->slots.resize((size_t)n_slots);
    for (auto& s : p->slots) {
        VX(vx_malloc_host(&s.pin_in, p->in_bytes));
        VX(vx_malloc_host(&s.pin_out, p->out_bytes));
        VX(vx_malloc(&s.dev_in, p->in_bytes));
        VX(vx_malloc(&s.dev_out, p->out_bytes));
        VX(vx_event_create(&s.uploaded));
        VX(vx_event_create(&s.computed));
        VX(vx_event_create(&s.downloaded));
    }
    return p.release();
}

depthany_pipeline::~depthany_pipeline() {
    if (model) vx_set_device(model->backend->index);
    for (auto& s : slots) {
        if (s.busy) vx_event_sync(s.downloaded);
        vx_free_host(s.pin_in); vx_free_host(s.pin_out);
        vx_free(s.dev_in); vx_free(s.dev_out);
        vx_event_destroy(s.uploaded); vx_event_destroy(s.computed); vx_event_destroy(s.downloaded);
    }
    if (h2d_stream) vx_stream_destroy(h2d_stream);
    if (d2h_stream) vx_stream_destroy(d2h_stream);
    for (size_t e = 1; e < exec.size(); ++e) {
        vx_stream_sync(compute_stream[e]);
        delete exec[e];
        vx_stream_destroy(compute_stream[e]);
    }
}

uint8_t* depthany_pipeline_input(depthany_pipeline& p) {
    auto& s = p.slots[(size_t)p.next];
    if (s.busy) throw except("depthany pipeline: slot %d still holds an unread result (wait for its ticket first)", p.next);
    return static_cast<uint8_t*>(s.pin_in);
}

int depthany_pipeline_submit(depthany_pipeline& p, uint8_t const* rgb) {
    depthany_model& m = *p.model;
    device_turn turn(*m.backend);
    const int ticket = p.next;
    auto& s = p.slots[(size_t)ticket];
    if (s.busy) throw except("depthany pipeline: slot %d still holds an unread result (wait for its ticket first)", ticket);
    const size_t e = (size_t)(p.n_submitted++ % (long)p.exec.size()); // consecutive batches alternate executors
    depthany_model& em = *p.exec[e];
    em.use_graph = m.use_graph;
    if (em.schedule != m.schedule) { // visp_depthany_set_schedule on the model after the pipeline was made: the executors follow it,
        em.schedule = m.schedule;    // and a launch sequence captured for the other schedule is dropped
        depthany_drop_captured_steps(em);
    }
    if (rgb && rgb != s.pin_in) memcpy(s.pin_in, rgb, p.in_bytes);
    void* cs = p.compute_stream[e];
    VX(vx_memcpy_h2d_async(s.dev_in, s.pin_in, p.in_bytes, p.h2d_stream));
    VX(vx_event_record(s.uploaded, p.h2d_stream));
    VX(vx_stream_wait_event(cs, s.uploaded));
    depthany_compute_batch_device(em, s.dev_in, p.batch, p.w, p.h, s.dev_out, nullptr, cs);
    VX(vx_event_record(s.computed, cs));
    VX(vx_stream_wait_event(p.d2h_stream, s.computed));
    VX(vx_memcpy_d2h_async(s.pin_out, s.dev_out, p.out_bytes, p.d2h_stream));
    VX(vx_event_record(s.downloaded, p.d2h_stream));
    s.busy = true;
    p.next = (p.next + 1) % p.n_slots;
    return ticket;
}

float const* depthany_pipeline_wait(depthany_pipeline& p, int ticket) {
    if (ticket < 0 || ticket >= p.n_slots) throw except("depthany pipeline: bad ticket %d", ticket);
    auto& s = p.slots[(size_t)ticket];
    if (!s.busy) throw except("depthany pipeline: ticket %d has nothing in flight", ticket);
    device_turn turn(*p.model->backend);
    VX(vx_event_sync(s.downloaded));
    s.busy = false;
    return static_cast<float const*>(s.pin_out);
}

// One shard of visp_depthany_compute_sharded on this model: pageable host images in, pageable results out, through the overlapped
// pipeline in chunks of the tuned step (32 images; pinned staging, H2D / forward / D2H of consecutive chunks on three streams,
// hipGraph replay) instead of one blocking pageable round trip. A last partial chunk runs as a full one (the pinned slot's other
// images are whatever it held: images are independent) -- no second workspace shape, no graph re-capture. Bit-identical to
// depthany_compute_batch_host.
void depthany_compute_shard_host(depthany_model& m, uint8_t const* rgb, int count, int w, int h, float* out) {
    constexpr int chunk = 32;
    if (count < chunk / 2) { // small shards: nothing to overlap, and a 32-image step for a few images wastes the device
        depthany_compute_batch_host(m, rgb, count, w, h, out, nullptr);
        return;
    }
    if (!m.shard_pipeline || m.shard_pipeline->w != w || m.shard_pipeline->h != h) {
        delete m.shard_pipeline;
        m.shard_pipeline = nullptr;
        m.shard_pipeline = depthany_pipeline_create(m, chunk, w, h, 3);
    }
    depthany_pipeline& p = *m.shard_pipeline;
    const size_t px = (size_t)w * h;
    const int n_chunks = (count + chunk - 1) / chunk;
    const bool graph = m.use_graph;
    m.use_graph = true;
    std::vector<std::pair<int, int>> flight; // (ticket, chunk index)
    auto retire = [&]() {
        const auto [ticket, k] = flight.front();
        flight.erase(flight.begin());
        float const* res = depthany_pipeline_wait(p, ticket);
        const int n = std::min(chunk, count - k * chunk);
        memcpy(out + (size_t)k * chunk * px, res, (size_t)n * px * 4);
    };
    try {
        for (int k = 0; k < n_chunks; ++k) {
            if ((int)flight.size() == p.n_slots) retire();
            const int n = std::min(chunk, count - k * chunk);
            uint8_t* pin = depthany_pipeline_input(p);
            memcpy(pin, rgb + (size_t)k * chunk * px * 3, (size_t)n * px * 3);
            flight.push_back({depthany_pipeline_submit(p, nullptr), k});
        }
        while (!flight.empty()) retire();
    } catch (...) {
        m.use_graph = graph;
        for (auto& f : flight) { // leave no slot busy behind an error
            try { depthany_pipeline_wait(p, f.first); } catch (...) {}
        }
        throw;
    }
    m.use_graph = graph;
}

// reference src/visp/vision.cpp:147-167
image_data depthany_compute(depthany_model& m, image_view image) {
    if (is_float(image.format) || n_channels(image.format) < 3)
        throw except("depthany: unsupported input image format [%d], expected an 8-bit colour image", int(image.format));
    i32x2 res = depthany_image_extent(image.extent, m.params);
    m.params.image_extent = res;
    // depthany_process_input: image_scale to the model extent happens in the caller's format -- for rgba / bgra / argb stb resizes alpha-weighted (premultiplied),
    // so colours next to transparent pixels differ from a resize of the opaque rgb -- and the alpha channel is dropped afterwards
    // (image_u8_to_f32 to rgb_f32 in the reference)
    const i32x2 caller_extent = image.extent;
    image_data resized;
    if (image.extent != res) {
        resized = image_scale(image, res);
        image = view_of(resized);
    }
    image_data rgb = image_to_rgb_u8(image);
    image_view rgb_view = view_of(rgb);
    image_data out = image_alloc(res, image_format::alpha_f32);
    depthany_compute_batch_host(m, static_cast<const uint8_t*>(rgb_view.data), 1, res[0], res[1],
                                reinterpret_cast<float*>(out.data.get()), nullptr);
    if (res != caller_extent) return image_scale(view_of(out), caller_extent); // depthany_process_output
    return out;
}

} // namespace visp
