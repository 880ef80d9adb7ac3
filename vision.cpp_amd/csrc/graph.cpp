// Tensor graph + executor (see graph.h). Host C++ only: device work goes through the vx_* C ABI.
#include "graph.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <optional>

#include "../../include/visp_hip_kernels.h"
#include "visp_util.h"

namespace visp {

#define VX(call)                                          \
    do {                                                  \
        if (!(call)) throw except("%s", vx_last_error()); \
    } while (0)

namespace {

template <typename T>
T round_up(T x, T m) { return (x + m - 1) / m * m; }

const char* const op_names[gop_count] = {"input", "weight", "linear", "layer_norm", "gelu", "relu", "scale", "add", "mul", "conv_2d",
                                         "conv_transpose_2d", "interpolate", "attention", "concat", "slice", "reshape", "repeat",
                                         "patch_embed", "cont", "image_u8_to_f32", "image_normalize", "leaky_relu"};

std::string shape_str(const int64_t ne[4]) {
    char b[96];
    snprintf(b, sizeof b, "[%lld, %lld, %lld, %lld]", (long long)ne[0], (long long)ne[1], (long long)ne[2], (long long)ne[3]);
    return b;
}

// ---- host evaluation of constants (f32, ggml index order: i0 fastest) ------------------------------------------------------

struct host_view {
    const float* p;
    const int64_t* ne;
    float at(int64_t i0, int64_t i1, int64_t i2, int64_t i3) const { return p[((i3 * ne[2] + i2) * ne[1] + i1) * ne[0] + i0]; }
};

// ggml's bicubic (a = -0.75, half-pixel centres, taps clamped to the map) on CWHN: dino.cpp:22-27 runs it on the stored position
// embeddings; the same arithmetic as depthany.cpp's interpolate_pos, for any [C, W, H, N]
void host_bicubic(const float* src, const int64_t sne[4], float* dst, int64_t tw, int64_t th) {
    const int64_t C = sne[0], W = sne[1], H = sne[2], N = sne[3];
    auto coeffs = [](float t, float c[4]) {
        const float a = -0.75f;
        float x;
        x = t + 1.0f; c[0] = ((a * x - 5.0f * a) * x + 8.0f * a) * x - 4.0f * a;
        x = t;        c[1] = ((a + 2.0f) * x - (a + 3.0f)) * x * x + 1.0f;
        x = 1.0f - t; c[2] = ((a + 2.0f) * x - (a + 3.0f)) * x * x + 1.0f;
        x = 2.0f - t; c[3] = ((a * x - 5.0f * a) * x + 8.0f * a) * x - 4.0f * a;
    };
    const float sfy = (float)th / (float)H, sfx = (float)tw / (float)W;
    for (int64_t n = 0; n < N; ++n)
        for (int64_t oy = 0; oy < th; ++oy) {
            const float sy = ((float)oy + 0.5f) / sfy - 0.5f;
            const int64_t iy = (int64_t)std::floor(sy);
            float cy[4];
            coeffs(sy - (float)iy, cy);
            for (int64_t ox = 0; ox < tw; ++ox) {
                const float sx = ((float)ox + 0.5f) / sfx - 0.5f;
                const int64_t ix = (int64_t)std::floor(sx);
                float cx[4];
                coeffs(sx - (float)ix, cx);
                float* o = dst + ((n * th + oy) * tw + ox) * C;
                for (int64_t c = 0; c < C; ++c) o[c] = 0.0f;
                for (int j = 0; j < 4; ++j) {
                    const int64_t yy = std::clamp<int64_t>(iy - 1 + j, 0, H - 1);
                    for (int i = 0; i < 4; ++i) {
                        const int64_t xx = std::clamp<int64_t>(ix - 1 + i, 0, W - 1);
                        const float wgt = cy[j] * cx[i];
                        const float* s = src + ((n * H + yy) * W + xx) * C;
                        for (int64_t c = 0; c < C; ++c) o[c] += wgt * s[c];
                    }
                }
            }
        }
}

// bilinear with align_corners as vx_bilinear_ac_f16 / ggml do it: sf = (out - 1) / (in - 1), src = i / sf
void host_bilinear_ac(const float* src, const int64_t sne[4], float* dst, int64_t tw, int64_t th) {
    const int64_t C = sne[0], W = sne[1], H = sne[2], N = sne[3];
    const float sfx = tw > 1 && W > 1 ? (float)(tw - 1) / (float)(W - 1) : (float)tw / (float)W;
    const float sfy = th > 1 && H > 1 ? (float)(th - 1) / (float)(H - 1) : (float)th / (float)H;
    for (int64_t n = 0; n < N; ++n)
        for (int64_t oy = 0; oy < th; ++oy) {
            const float sy = (float)oy / sfy;
            const int64_t y0 = std::min<int64_t>((int64_t)sy, H - 1), y1 = std::min<int64_t>(y0 + 1, H - 1);
            const float fy = sy - (float)y0;
            for (int64_t ox = 0; ox < tw; ++ox) {
                const float sx = (float)ox / sfx;
                const int64_t x0 = std::min<int64_t>((int64_t)sx, W - 1), x1 = std::min<int64_t>(x0 + 1, W - 1);
                const float fx = sx - (float)x0;
                float* o = dst + ((n * th + oy) * tw + ox) * C;
                const float* a = src + ((n * H + y0) * W + x0) * C;
                const float* b = src + ((n * H + y0) * W + x1) * C;
                const float* c = src + ((n * H + y1) * W + x0) * C;
                const float* d = src + ((n * H + y1) * W + x1) * C;
                for (int64_t k = 0; k < C; ++k) {
                    const float top = a[k] + (b[k] - a[k]) * fx, bot = c[k] + (d[k] - c[k]) * fx;
                    o[k] = top + (bot - top) * fy;
                }
            }
        }
}

// b broadcast over the trailing dimensions of a: b.ne[i] == a.ne[i] up to some dimension, 1 from there on. Returns b's period.
int64_t broadcast_period(const int64_t a[4], const int64_t b[4], const char* what) {
    bool ones = false;
    int64_t period = 1;
    for (int i = 0; i < 4; ++i) {
        if (!ones && b[i] == a[i]) { period *= b[i]; continue; }
        if (b[i] == 1) { ones = ones || a[i] != 1; continue; }
        throw except("%s: cannot broadcast %s onto %s (the second operand must equal the first in its leading dimensions and be 1 in the rest)", what,
                     shape_str(b).c_str(), shape_str(a).c_str());
    }
    return period;
}

void fold_constant(graph& g, graph_node& n) {
    auto src = [&](int i) -> graph_node const& { return g.nodes[n.src[i]]; };
    const int64_t total = n.n_elements();
    n.host.resize((size_t)total);
    float* out = n.host.data();
    switch (n.op) {
        case gop_reshape:
        case gop_cont: n.host.assign(src(0).values(), src(0).values() + total); break;
        case gop_scale:
            for (int64_t i = 0; i < total; ++i) out[i] = src(0).values()[i] * n.fp[0];
            break;
        case gop_add:
        case gop_mul: {
            const int64_t period = src(1).n_elements();
            const float *a = src(0).values(), *b = src(1).values();
            for (int64_t i = 0; i < total; ++i) out[i] = n.op == gop_add ? a[i] + b[i % period] : a[i] * b[i % period];
        } break;
        case gop_slice: {
            host_view v{src(0).values(), src(0).ne};
            int64_t o = 0;
            for (int64_t i3 = 0; i3 < n.ne[3]; ++i3)
                for (int64_t i2 = 0; i2 < n.ne[2]; ++i2)
                    for (int64_t i1 = 0; i1 < n.ne[1]; ++i1)
                        for (int64_t i0 = 0; i0 < n.ne[0]; ++i0)
                            out[o++] = v.at(n.ip[0] + i0 * n.ip[2], n.ip[3] + i1 * n.ip[5], n.ip[6] + i2 * n.ip[8], n.ip[9] + i3 * n.ip[11]);
        } break;
        case gop_repeat: {
            host_view v{src(0).values(), src(0).ne};
            int64_t o = 0;
            for (int64_t i3 = 0; i3 < n.ne[3]; ++i3)
                for (int64_t i2 = 0; i2 < n.ne[2]; ++i2)
                    for (int64_t i1 = 0; i1 < n.ne[1]; ++i1)
                        for (int64_t i0 = 0; i0 < n.ne[0]; ++i0) out[o++] = v.at(i0 % v.ne[0], i1 % v.ne[1], i2 % v.ne[2], i3 % v.ne[3]);
        } break;
        case gop_concat: {
            const int dim = (int)n.ip[0];
            host_view a{src(0).values(), src(0).ne}, b{src(1).values(), src(1).ne};
            int64_t o = 0;
            for (int64_t i3 = 0; i3 < n.ne[3]; ++i3)
                for (int64_t i2 = 0; i2 < n.ne[2]; ++i2)
                    for (int64_t i1 = 0; i1 < n.ne[1]; ++i1)
                        for (int64_t i0 = 0; i0 < n.ne[0]; ++i0) {
                            int64_t idx[4] = {i0, i1, i2, i3};
                            if (idx[dim] < a.ne[dim]) out[o++] = a.at(idx[0], idx[1], idx[2], idx[3]);
                            else { idx[dim] -= a.ne[dim]; out[o++] = b.at(idx[0], idx[1], idx[2], idx[3]); }
                        }
        } break;
        case gop_interpolate: {
            const int mode = (int)n.ip[2];
            if ((mode & 255) == 2 && !(mode & 256)) host_bicubic(src(0).values(), src(0).ne, out, n.ip[0], n.ip[1]);
            else if ((mode & 255) == 1 && (mode & 256)) host_bilinear_ac(src(0).values(), src(0).ne, out, n.ip[0], n.ip[1]);
            else throw except("interpolate: mode %d is not built (bicubic, and bilinear | align_corners, are)", mode);
        } break;
        default: throw except("graph: %s on constants is not folded", graph_op_name(n.op));
    }
}

bool foldable(int32_t op) {
    return op == gop_reshape || op == gop_cont || op == gop_scale || op == gop_add || op == gop_mul || op == gop_slice || op == gop_repeat ||
           op == gop_concat || op == gop_interpolate;
}

void check_tensor(graph const& g, int t, const char* what) {
    if (t < 0 || t >= (int)g.nodes.size()) throw except("%s: tensor handle %d is not part of this graph", what, t);
}

} // namespace

const char* graph_op_name(int32_t op) { return op >= 0 && op < gop_count ? op_names[op] : "?"; }

graph::~graph() {
    if (dev) vx_set_device(dev->index); // frees go to the device the graph lives on, whatever the calling thread's current device is
    if (graph_exec) vx_graph_destroy(graph_exec);
    if (arena.ptr) vx_free(arena.ptr);
    for (void* p : const_allocs) vx_free(p);
}
weight_store::~weight_store() {
    if (dev) vx_set_device(dev->index);
    if (arena.ptr) vx_free(arena.ptr);
    for (void* p : allocs) vx_free(p);
}

std::shared_ptr<weight_store> weights_create() { return std::make_shared<weight_store>(); }

void weights_add(weight_store& ws, char const* name, int32_t dtype, const int64_t ne[4], const float* data) {
    if (!name || !*name) throw except("weights: a tensor needs a name");
    if (ws.tensors.count(name)) throw except("weights: '%s' exists already", name);
    if (ws.dev) throw except("weights: '%s' added after a graph over these weights was allocated", name);
    weight_store::entry e;
    e.dtype = dtype == gdt_f32 ? gdt_f32 : gdt_f16;
    int64_t n = 1;
    for (int i = 0; i < 4; ++i) {
        if (ne[i] <= 0) throw except("weights: '%s' has a non-positive extent", name);
        e.ne[i] = ne[i];
        n *= ne[i];
    }
    if (data) e.data.assign(data, data + n);
    else e.data.assign((size_t)n, 0.0f);
    ws.tensors.emplace(name, std::move(e));
}

std::shared_ptr<weight_store> weights_load(char const* path) { return weights_from_file(model_load(path)); }

std::shared_ptr<weight_store> weights_from_file(model_file const& f) {
    auto ws = weights_create();
    const bool file_whcn = f.tensor_layout() == layout_whcn;
    std::vector<int32_t> conv2d = f.conv2d_weights();
    std::vector<float> tmp, perm;
    for (int idx = 0; idx < (int)f.tensors.size(); ++idx) {
        gguf_tensor const& t = f.tensors[idx];
        if (t.type != GGML_F32 && t.type != GGML_F16) continue; // index tables etc. are not weights of this executor
        const int64_t n = t.n_elements();
        tmp.assign((size_t)n, 0.0f);
        if (!t.data) ws->no_data = true; // header-only read (a rank that receives the packed arena by broadcast): shapes only
        else if (t.type == GGML_F32) memcpy(tmp.data(), t.data, (size_t)n * 4);
        else
            for (int64_t i = 0; i < n; ++i) tmp[i] = f16_to_f32(reinterpret_cast<const uint16_t*>(t.data)[i]);
        int64_t ne[4] = {t.ne[0], t.ne[1], t.ne[2], t.ne[3]};
        const float* data = tmp.data();
        if (file_whcn && std::binary_search(conv2d.begin(), conv2d.end(), idx)) {
            // torch OIHW (ne [kw, kh, Cin, Cout]) -> OHWI (ne [Cin, kw, kh, Cout]): what model_transfer does for a cwhn backend (ml.cpp:449-498)
            const int64_t kw = ne[0], kh = ne[1], ci = ne[2], co = ne[3];
            perm.resize((size_t)n);
            for (int64_t o = 0; o < co; ++o)
                for (int64_t c = 0; c < ci; ++c)
                    for (int64_t y = 0; y < kh; ++y)
                        for (int64_t x = 0; x < kw; ++x) perm[((o * kh + y) * kw + x) * ci + c] = tmp[((o * ci + c) * kh + y) * kw + x];
            ne[0] = ci; ne[1] = kw; ne[2] = kh; ne[3] = co;
            data = perm.data();
        }
        weights_add(*ws, t.name.c_str(), t.type, ne, data);
    }
    return ws;
}

graph* graph_create(std::shared_ptr<weight_store> weights) {
    graph* g = new graph;
    g->store = weights ? std::move(weights) : weights_create();
    return g;
}

static void require_building(graph const& g, const char* what) {
    if (g.allocated) throw except("%s: the graph is already allocated (build a new one)", what);
}

int graph_find_weight(graph& g, char const* name) {
    std::string_view key(name ? name : "");
    auto it = g.weights.find(key);
    if (it != g.weights.end()) return it->second;
    auto st = g.store->tensors.find(key);
    if (st == g.store->tensors.end()) return -1;
    require_building(g, "model_ref::find");
    graph_node n;
    n.op = gop_weight;
    n.dtype = st->second.dtype;
    for (int i = 0; i < 4; ++i) n.ne[i] = st->second.ne[i];
    n.name = st->first;
    n.constant = true;
    n.cdata = st->second.data.data(); // std::map nodes do not move
    g.nodes.push_back(std::move(n));
    const int id = (int)g.nodes.size() - 1;
    g.weights.emplace(st->first, id);
    return id;
}

int graph_add_weight(graph& g, char const* name, int32_t dtype, const int64_t ne[4], const float* data) {
    require_building(g, "graph_add_weight");
    weights_add(*g.store, name, dtype, ne, data);
    return graph_find_weight(g, name);
}

int graph_input(graph& g, int32_t dtype, const int64_t ne[4], char const* name) {
    require_building(g, "graph_input");
    if (dtype != gdt_f32 && dtype != gdt_f16 && dtype != gdt_u8) throw except("graph_input: dtype %d (0 = f32, 1 = f16, 24 = u8 image bytes)", dtype);
    graph_node n;
    n.op = gop_input;
    n.dtype = dtype;
    for (int i = 0; i < 4; ++i) {
        if (ne[i] <= 0) throw except("graph_input: non-positive extent");
        n.ne[i] = ne[i];
    }
    n.name = name && *name ? name : "input";
    g.nodes.push_back(n);
    const int id = (int)g.nodes.size() - 1;
    g.named[g.nodes[id].name] = id;
    return id;
}

void graph_set_name(graph& g, int t, char const* name) {
    check_tensor(g, t, "graph_set_name");
    g.nodes[t].name = name ? name : "";
    if (name && *name) g.named[name] = t;
}

int graph_get_tensor(graph& g, char const* name) {
    auto it = g.named.find(std::string_view(name ? name : ""));
    if (it != g.named.end()) return it->second;
    return graph_find_weight(g, name);
}

void graph_output(graph& g, int t, char const* name) {
    require_building(g, "graph_output");
    check_tensor(g, t, "graph_output");
    g.nodes[t].is_output = true;
    if (name && *name) graph_set_name(g, t, name);
}

int graph_add(graph& g, int32_t op, const int* src, int n_src, const int64_t* ip, int n_ip, const float* fp, int n_fp) {
    require_building(g, "graph_add");
    if (op <= gop_weight || op >= gop_count) throw except("graph_add: op %d is not a node op", op);
    if (n_src < 1 || n_src > 4 || n_ip < 0 || n_ip > 12 || n_fp < 0 || n_fp > 8) throw except("graph_add(%s): bad argument counts", graph_op_name(op));
    graph_node n;
    n.op = op;
    n.n_src = n_src;
    for (int i = 0; i < n_src; ++i) {
        check_tensor(g, src[i], graph_op_name(op));
        n.src[i] = src[i];
    }
    for (int i = 0; i < n_ip; ++i) n.ip[i] = ip[i];
    for (int i = 0; i < n_fp; ++i) n.fp[i] = fp[i];
    auto S = [&](int i) -> graph_node const& { return g.nodes[n.src[i]]; };
    const char* nm = graph_op_name(op);
    auto need_src = [&](int lo, int hi) {
        if (n_src < lo || n_src > hi) throw except("%s: takes %d..%d tensors, got %d", nm, lo, hi, n_src);
    };
    auto same_shape = [&]() { for (int i = 0; i < 4; ++i) n.ne[i] = S(0).ne[i]; };
    n.dtype = gdt_f16;

    switch (op) {
        case gop_linear: {
            need_src(2, 3);
            graph_node const &x = S(0), &w = S(1);
            if (!w.constant) throw except("linear: the weight must be a model weight");
            if (w.ne[2] != 1 || w.ne[3] != 1 || w.ne[0] != x.ne[0])
                throw except("linear: weight %s does not match input %s", shape_str(w.ne).c_str(), shape_str(x.ne).c_str());
            if (n_src == 3 && S(2).n_elements() != w.ne[1]) throw except("linear: bias has %lld elements, expected %lld", (long long)S(2).n_elements(), (long long)w.ne[1]);
            same_shape();
            n.ne[0] = w.ne[1];
        } break;
        case gop_layer_norm:
            need_src(3, 3);
            if (S(1).n_elements() != S(0).ne[0] || S(2).n_elements() != S(0).ne[0]) throw except("layer_norm: weight / bias must have %lld elements", (long long)S(0).ne[0]);
            same_shape();
            break;
        case gop_gelu:
        case gop_relu:
        case gop_leaky_relu:
        case gop_scale:
        case gop_cont:
            need_src(1, 1);
            same_shape();
            if ((op == gop_relu || op == gop_scale) && !S(0).constant) n.dtype = S(0).dtype; // the f32 tail of a one-channel head
            break;
        case gop_add:
        case gop_mul: {
            need_src(2, 2);
            if (S(0).n_elements() < S(1).n_elements()) { std::swap(n.src[0], n.src[1]); } // commutative: the larger operand first
            broadcast_period(S(0).ne, S(1).ne, nm);
            same_shape();
        } break;
        case gop_conv_2d: {
            need_src(2, 3);
            graph_node const &x = S(0), &w = S(1);
            const int64_t stride = n.ip[0] > 0 ? n.ip[0] : 1, pad = n.ip[1];
            n.ip[0] = stride;
            if (!w.constant) throw except("conv_2d: the kernel must be a model weight");
            if (w.ne[0] != x.ne[0]) throw except("conv_2d: kernel %s (ne = [Cin, kw, kh, Cout]) does not match input %s (CWHN)", shape_str(w.ne).c_str(), shape_str(x.ne).c_str());
            if (n_src == 3 && S(2).n_elements() != w.ne[3]) throw except("conv_2d: bias has %lld elements, expected %lld", (long long)S(2).n_elements(), (long long)w.ne[3]);
            n.ne[0] = w.ne[3];
            n.ne[1] = (x.ne[1] + 2 * pad - w.ne[1]) / stride + 1;
            n.ne[2] = (x.ne[2] + 2 * pad - w.ne[2]) / stride + 1;
            n.ne[3] = x.ne[3];
            if (n.ne[1] <= 0 || n.ne[2] <= 0) throw except("conv_2d: the kernel does not fit the input");
            // a 1x1 convolution to ONE channel is a model's final map (depth-anything.cpp:91-94): kept in f32, like the static schedule's output
            if (w.ne[3] == 1 && w.ne[1] == 1 && w.ne[2] == 1 && stride == 1 && pad == 0) n.dtype = gdt_f32;
            // ... and so is a 3x3 convolution to an RGB image (esrgan.cpp:75)
            if (w.ne[3] == 3 && w.ne[1] == 3 && w.ne[2] == 3 && stride == 1 && pad == 1) n.dtype = gdt_f32;
        } break;
        case gop_conv_transpose_2d: {
            need_src(2, 3);
            graph_node const &x = S(0), &w = S(1);
            const int64_t stride = n.ip[0];
            if (!w.constant) throw except("conv_transpose_2d: the kernel must be a model weight");
            if (w.ne[3] != x.ne[0]) throw except("conv_transpose_2d: kernel %s (ne = [kw, kh, Cout, Cin]) does not match input %s", shape_str(w.ne).c_str(), shape_str(x.ne).c_str());
            if (w.ne[0] != stride || w.ne[1] != stride) throw except("conv_transpose_2d: kernel %lldx%lld with stride %lld is not built (kernel == stride is)", (long long)w.ne[0], (long long)w.ne[1], (long long)stride);
            if (n_src == 3 && S(2).n_elements() != w.ne[2]) throw except("conv_transpose_2d: bias has %lld elements, expected %lld", (long long)S(2).n_elements(), (long long)w.ne[2]);
            n.ne[0] = w.ne[2];
            n.ne[1] = x.ne[1] * stride;
            n.ne[2] = x.ne[2] * stride;
            n.ne[3] = x.ne[3];
        } break;
        case gop_interpolate:
            need_src(1, 1);
            if (n.ip[0] <= 0 || n.ip[1] <= 0) throw except("interpolate: non-positive target extent");
            same_shape();
            n.ne[1] = n.ip[0];
            n.ne[2] = n.ip[1];
            break;
        case gop_attention: {
            need_src(3, 3);
            graph_node const &q = S(0), &k = S(1), &v = S(2);
            for (int i = 0; i < 4; ++i)
                if (k.ne[i] != v.ne[i] || (i != 2 && q.ne[i] != k.ne[i])) throw except("attention: q %s, k %s, v %s do not match ([head_dim, heads, tokens, batch])", shape_str(q.ne).c_str(), shape_str(k.ne).c_str(), shape_str(v.ne).c_str());
            n.ne[0] = q.ne[0] * q.ne[1];
            n.ne[1] = q.ne[2];
            n.ne[2] = q.ne[3];
            n.ne[3] = 1;
        } break;
        case gop_concat: {
            need_src(2, 2);
            const int64_t dim = n.ip[0];
            if (dim < 0 || dim > 3) throw except("concat: dimension %lld", (long long)dim);
            same_shape();
            for (int i = 0; i < 4; ++i)
                if (i != dim && S(0).ne[i] != S(1).ne[i]) throw except("concat: %s and %s differ outside dimension %lld", shape_str(S(0).ne).c_str(), shape_str(S(1).ne).c_str(), (long long)dim);
            n.ne[dim] = S(0).ne[dim] + S(1).ne[dim];
        } break;
        case gop_slice:
            need_src(1, 1);
            for (int d = 0; d < 4; ++d) {
                int64_t &b = n.ip[3 * d], &e = n.ip[3 * d + 1], &s = n.ip[3 * d + 2];
                const int64_t ext = S(0).ne[d];
                if (s <= 0) s = 1;
                if (b < 0) b += ext; // python-style negative indices (ml.cpp:752-756)
                if (e < 0) e += ext;
                e = std::min(e, ext);
                if (b < 0 || b >= e) throw except("slice: empty or out-of-range selection in dimension %d of %s", d, shape_str(S(0).ne).c_str());
                n.ne[d] = (e - b + s - 1) / s;
            }
            break;
        case gop_reshape: {
            need_src(1, 1);
            int64_t total = 1;
            for (int i = 0; i < 4; ++i) {
                if (n.ip[i] <= 0) throw except("reshape: non-positive extent");
                n.ne[i] = n.ip[i];
                total *= n.ip[i];
            }
            if (total != S(0).n_elements()) throw except("reshape: %s has %lld elements, the target %s has %lld", shape_str(S(0).ne).c_str(), (long long)S(0).n_elements(), shape_str(n.ne).c_str(), (long long)total);
        } break;
        case gop_repeat:
            need_src(1, 1);
            for (int i = 0; i < 4; ++i) {
                n.ne[i] = n.ip[i];
                if (n.ip[i] <= 0 || (S(0).ne[i] != 1 && S(0).ne[i] != n.ip[i])) throw except("repeat: %s to %s (source dimensions must be 1 or already equal)", shape_str(S(0).ne).c_str(), shape_str(n.ip).c_str());
            }
            break;
        case gop_patch_embed: {
            need_src(2, 3);
            graph_node const &x = S(0), &w = S(1);
            const int64_t ps = n.ip[0];
            if (!w.constant) throw except("patch_embed: the kernel must be a model weight");
            if (x.dtype != gdt_f32) throw except("patch_embed: the input image tensor is f32 (the tensor the reference uploads, or image_u8_to_f32 of a u8 input)");
            if (ps <= 0 || w.ne[1] != ps || w.ne[2] != ps || w.ne[0] != x.ne[0]) throw except("patch_embed: kernel %s does not match patch size %lld and input %s", shape_str(w.ne).c_str(), (long long)ps, shape_str(x.ne).c_str());
            if (x.ne[1] % ps || x.ne[2] % ps) throw except("patch_embed: extent %lldx%lld is not a multiple of the patch size %lld", (long long)x.ne[1], (long long)x.ne[2], (long long)ps);
            n.ne[0] = w.ne[3];
            n.ne[1] = x.ne[1] / ps;
            n.ne[2] = x.ne[2] / ps;
            n.ne[3] = x.ne[3];
        } break;
        case gop_image_u8_to_f32: {
            need_src(1, 1);
            graph_node const& x = S(0);
            if (x.op != gop_input || x.dtype != gdt_u8 || x.ne[0] != 3) throw except("image_u8_to_f32: the operand is a u8 input tensor [3, W, H, N]");
            if (n_fp != 6) throw except("image_u8_to_f32: six float parameters (mean r g b, 1 / std r g b)");
            same_shape();
            n.dtype = gdt_f32;
        } break;
        case gop_image_normalize: {
            need_src(1, 1);
            if (S(0).constant || S(0).dtype != gdt_f32 || S(0).ne[0] != 1) throw except("image_normalize: the operand is a one-channel f32 map [1, W, H, N]");
            same_shape();
            n.dtype = gdt_f32;
        } break;
        default: throw except("graph_add: op %d", op);
    }
    // f32 tensors exist as inputs only; an f32 operand anywhere else would need kernels this executor does not have
    if (op != gop_patch_embed && op != gop_image_u8_to_f32 && op != gop_image_normalize)
        for (int i = 0; i < n_src; ++i) {
            const bool head_tail = (op == gop_relu || op == gop_scale) && S(i).op != gop_input;
            // (a 3x3 conv on the f32 image of <= 16 channels: the first conv of an image-to-image network, esrgan.cpp:58)
            const bool image_conv = op == gop_conv_2d && i == 0 && S(0).op == gop_input && S(0).ne[0] <= 16 && S(1).ne[1] == 3 && S(1).ne[2] == 3 && n.ip[0] == 1 && n.ip[1] == 1;
            if (!S(i).constant && S(i).dtype != gdt_f16 && !head_tail && !image_conv)
                throw except("%s: operand %d is f32; only patch_embed and an image's first 3x3 conv read the f32 input tensor", nm, i);
        }

    bool all_const = true;
    for (int i = 0; i < n_src; ++i) all_const = all_const && S(i).constant;
    if (all_const && foldable(op)) {
        n.constant = true;
        fold_constant(g, n);
    } else if (all_const) {
        throw except("%s: every operand is a constant; the executor computes this op on activations only", nm);
    }
    g.nodes.push_back(std::move(n));
    return (int)g.nodes.size() - 1;
}

// ---- lowering ----------------------------------------------------------------------------------------------------------------

namespace {

struct packed_operand {
    void* w = nullptr;
    float* bias = nullptr;
    int N = 0, K = 0, n_real = 0, k_real = 0;
};

struct lowering {
    graph& g;
    std::vector<char> needed, skip;
    std::vector<int> uses;
    std::vector<std::vector<int>> consumers;
    std::vector<char> relu_on_load;
    std::vector<std::pair<std::vector<int>, std::vector<int>>> io; // per launch: buffers read, buffers written
    std::map<std::pair<std::pair<int, int>, std::string>, void*> const_cache; // (node, role, derived-from) -> device copy made for this graph

    explicit lowering(graph& gr) : g(gr) {}

    int root(int t) const {
        while (g.nodes[t].alias_of >= 0) t = g.nodes[t].alias_of;
        return t;
    }
    int buf_of(int t) {
        const int r = root(t);
        if (g.nodes[r].buffer < 0 && planar.count(r)) planar_to_nhwc(r); // a reader outside the LDS-ring conv: the map as NHWC, once
        const int b = g.nodes[r].buffer;
        if (b < 0) throw except("graph: tensor %d (%s) is read before it is computed", t, graph_op_name(g.nodes[t].op));
        return b;
    }
    int new_buffer(size_t bytes, bool persistent = false) {
        graph_buffer b;
        b.bytes = round_up<size_t>(bytes, 256) + 256;
        b.persistent = persistent;
        b.first = b.last = (int)g.launches.size();
        g.buffers.push_back(b);
        return (int)g.buffers.size() - 1;
    }
    void materialise(int t) {
        graph_node& n = g.nodes[t];
        n.buffer = new_buffer(n.n_bytes(), n.is_output || n.op == gop_input);
    }
    void emit(std::string desc, std::vector<int> reads, std::vector<int> writes, std::function<void(void*)> run) {
        graph_launch l;
        l.desc = std::move(desc);
        l.run = std::move(run);
        g.launches.push_back(std::move(l));
        io.emplace_back(std::move(reads), std::move(writes));
    }
    void tag(const char* group, double flops = 0, double bytes = 0) { // timing group and algorithmic work of the launch emitted last
        graph_launch& l = g.launches.back();
        l.group = group;
        l.flops = flops;
        l.bytes = bytes;
    }
    // device pointer of a buffer at run time (offsets are assigned after the launch list is complete)
    std::function<char*()> ptr(int buf) {
        graph* gp = &g;
        return [gp, buf]() {
            graph_buffer const& b = gp->buffers[buf];
            return b.external ? static_cast<char*>(b.external) : static_cast<char*>(gp->arena.ptr) + b.offset;
        };
    }

    // device image of constant t in `role`: model weights are cached in the weight store (shared by every graph over the model),
    // constants folded inside this graph belong to the graph
    // `with`: names of the other weights a derived image was made from (a folded LayerScale vector, the k and v parts of a fused
    // q|k|v operand) -- part of its identity
    void* cached(int t, int role, std::function<void*(bool)> make, std::string const& with = {}) {
        auto key = std::make_pair(std::make_pair(t, role), with);
        auto it = const_cache.find(key);
        if (it != const_cache.end()) return it->second;
        graph_node const& n = g.nodes[t];
        void* d = nullptr;
        // only images made from NAMED model weights alone are shared: a derived image whose other operand is a constant folded inside this
        // graph ("#<node>": a node index means nothing to another graph) stays with the graph
        const bool shareable = n.op == gop_weight && with.find('#') == std::string::npos;
        if (shareable && !g.dev) d = make(true); // planning: upload() only counts what the store would take
        else if (shareable) {
            auto skey = std::make_pair(with.empty() ? n.name : n.name + "|" + with, role);
            std::lock_guard<std::mutex> lock(g.store->mutex); // graphs over one store may be allocated from several threads
            auto sit = g.store->packs.find(skey);
            if (sit != g.store->packs.end()) d = sit->second; // uploaded by an earlier graph over the same weights
            else { d = make(true); g.store->packs[skey] = d; }
        } else d = make(false);
        const_cache[key] = d;
        return d;
    }
    void* upload(const void* host, size_t bytes, bool to_store) {
        g.const_bytes += bytes;
        if (!g.dev) {
            if (to_store) g.plan_store_bytes += round_up<size_t>(bytes + 256, 256);
            return nullptr;
        }
        void* d = nullptr;
        weight_store& ws = *g.store;
        if (to_store && ws.arena.ptr && ws.arena_used + round_up<size_t>(bytes + 256, 256) <= ws.arena.bytes) { // the model's one arena, in lowering order
            d = static_cast<char*>(ws.arena.ptr) + ws.arena_used;
            ws.arena_used += round_up<size_t>(bytes + 256, 256);
            ws.device_bytes += bytes;
        } else {
            if (to_store && ws.no_data) throw except("graph: this rank holds no weight data (the arena came by broadcast) and the image it needs is not part of the arena");
            VX(vx_malloc(&d, bytes + 256));
            if (to_store) { ws.allocs.push_back(d); ws.device_bytes += bytes; }
            else g.const_allocs.push_back(d);
        }
        if (to_store && ws.no_data) return d; // laid out, filled by the broadcast
        VX(vx_memcpy_h2d(d, host, bytes, g.dev->stream));
        VX(vx_stream_sync(g.dev->stream)); // the host image is a temporary
        return d;
    }
    float* const_f32(int t) {
        return static_cast<float*>(cached(t, 0, [&](bool st) { return upload(g.nodes[t].values(), (size_t)g.nodes[t].n_elements() * 4, st); }));
    }
    void* const_f16(int t) {
        return cached(t, 1, [&](bool st) {
            std::vector<uint16_t> h((size_t)g.nodes[t].n_elements());
            for (size_t i = 0; i < h.size(); ++i) h[i] = f32_to_f16(g.nodes[t].values()[i]);
            return upload(h.data(), h.size() * 2, st);
        });
    }
    // rows [n][k] -> f16 [N pad][K pad 64] + f32 bias [N pad] (bias of `period` elements repeated)
    packed_operand pack_matrix(int wt, int role, int n, int k, std::function<float(int, int)> at, int bias_t, int bias_period, int scale_t = -1) {
        packed_operand p;
        // LayerScale folded into the operand: W' = f16(lambda[n] * W[n, :]), b' = lambda[n] * b[n] (what csrc/depthany.cpp does for the
        // block kernel: the product's epilogue can then take the residual)
        const float* lam = scale_t >= 0 ? g.nodes[scale_t].values() : nullptr;
        const std::string with = scale_t >= 0 ? "*" + (g.nodes[scale_t].op == gop_weight && !g.nodes[scale_t].name.empty() ? g.nodes[scale_t].name : "#" + std::to_string(scale_t)) : std::string();
        p.n_real = n; p.k_real = k;
        p.N = round_up(n, n > 64 ? 64 : 32);
        p.K = round_up(k, 64);
        p.w = cached(wt, role, [&](bool st) {
            std::vector<uint16_t> h((size_t)p.N * p.K, 0);
            for (int r = 0; r < n; ++r)
                for (int c = 0; c < k; ++c) h[(size_t)r * p.K + c] = f32_to_f16(lam ? lam[r] * at(r, c) : at(r, c));
            return upload(h.data(), h.size() * 2, st);
        }, with);
        if (bias_t >= 0)
            p.bias = static_cast<float*>(cached(bias_t, 16 + role, [&](bool st) {
                std::vector<float> b((size_t)p.N, 0.0f);
                for (int r = 0; r < n; ++r) b[r] = g.nodes[bias_t].values()[r % bias_period] * (lam ? lam[r] : 1.0f);
                return upload(b.data(), b.size() * 4, st);
            }, with));
        return p;
    }
    packed_operand pack_rows(int wt, int bias_t, int n, int scale_t = -1) { // linear [K, N] and conv [Cin, kw, kh, Cout]: the n rows of the host image as they are
        graph_node const& w = g.nodes[wt];
        const int k = (int)(w.n_elements() / n);
        const float* h = w.values();
        return pack_matrix(wt, 2, n, k, [h, k](int r, int c) { return h[(size_t)r * k + c]; }, bias_t, n, scale_t);
    }
    // q | k | v rows of three linears over the same input as one operand (N = 3 * heads * 64), for the head-major epilogue
    packed_operand pack_qkv(const int lin[3]) {
        graph_node const* w[3];
        int bias[3];
        std::string with;
        for (int i = 0; i < 3; ++i) {
            w[i] = &g.nodes[g.nodes[lin[i]].src[1]];
            bias[i] = g.nodes[lin[i]].n_src == 3 ? g.nodes[lin[i]].src[2] : -1;
            if (i) with += (i > 1 ? "|" : "") + w[i]->name;
        }
        const int k = (int)w[0]->ne[0], n1 = (int)w[0]->ne[1];
        packed_operand p;
        p.n_real = 3 * n1; p.k_real = k; p.N = 3 * n1; p.K = round_up(k, 64);
        p.w = cached(g.nodes[lin[0]].src[1], 4, [&](bool st) {
            std::vector<uint16_t> h((size_t)p.N * p.K, 0);
            for (int i = 0; i < 3; ++i)
                for (int r = 0; r < n1; ++r)
                    for (int c = 0; c < k; ++c) h[((size_t)i * n1 + r) * p.K + c] = f32_to_f16(w[i]->values()[(size_t)r * k + c]);
            return upload(h.data(), h.size() * 2, st);
        }, with);
        if (bias[0] >= 0 || bias[1] >= 0 || bias[2] >= 0)
            p.bias = static_cast<float*>(cached(g.nodes[lin[0]].src[1], 20, [&](bool st) {
                std::vector<float> b((size_t)p.N, 0.0f);
                for (int i = 0; i < 3; ++i)
                    if (bias[i] >= 0)
                        for (int r = 0; r < n1; ++r) b[(size_t)i * n1 + r] = g.nodes[bias[i]].values()[r];
                return upload(b.data(), b.size() * 4, st);
            }, with));
        return p;
    }
    packed_operand pack_conv_transpose(int wt, int bias_t, int s) { // ne [kw, kh, Cout, Cin]: row (dy * s + dx) * Cout + co, column ci
        graph_node const& w = g.nodes[wt];
        const int kw = (int)w.ne[0], kh = (int)w.ne[1], cout = (int)w.ne[2], cin = (int)w.ne[3];
        const float* h = w.values();
        return pack_matrix(wt, 3, s * s * cout, cin,
                           [=](int r, int c) {
                               const int tap = r / cout, co = r % cout, dy = tap / s, dx = tap % s;
                               return h[(((size_t)c * cout + co) * kh + dy) * kw + dx];
                           },
                           bias_t, cout);
    }

    // the sole reader of t, if it is not an output
    int sole_consumer(int t) const { return uses[t] == 1 && !g.nodes[t].is_output && consumers[t].size() == 1 ? consumers[t][0] : -1; }

    // epilogue fusion behind a matrix product: [gelu | relu] then [+ residual]
    struct epilogue { int act = 0; int res = -1; int res2 = -1; int last = -1; int scale = -1; };
    epilogue fuse_epilogue(int t, bool allow_gelu, bool allow_scale = false, bool allow_two = false) {
        epilogue e;
        e.last = t;
        int c = sole_consumer(t);
        if (allow_scale && c >= 0 && g.nodes[c].op == gop_mul) { // LayerScale behind a linear (dino.cpp:48-50): folded into the weights
            graph_node const& mu = g.nodes[c];
            const int v = root(mu.src[0]) == t ? mu.src[1] : mu.src[0];
            if (g.nodes[v].constant && g.nodes[v].n_elements() == g.nodes[t].ne[0] && root(mu.src[0]) == t) {
                e.scale = v;
                skip[c] = 1;
                g.nodes[c].alias_of = t;
                if (g.nodes[c].is_output) g.nodes[t].is_output = true;
                e.last = c;
                c = sole_consumer(c);
            }
        }
        if (e.scale < 0 && c >= 0 && ((g.nodes[c].op == gop_gelu && allow_gelu) || g.nodes[c].op == gop_relu)) {
            e.act = g.nodes[c].op == gop_gelu ? 1 : 2;
            skip[c] = 1;
            g.nodes[c].alias_of = t;
            if (g.nodes[c].is_output) { g.nodes[t].is_output = true; }
            e.last = c;
            c = sole_consumer(c);
        }
        // up to two residual adds (dpt::residual_conv's skip, then feature_fusion's x0 + ..., depth-anything.cpp:15-33)
        for (int round = 0; round < 2 && c >= 0 && g.nodes[c].op == gop_add && e.act != 1; ++round) {
            graph_node const& a = g.nodes[c];
            const int other = root(a.src[0]) == root(e.last) ? a.src[1] : a.src[0];
            graph_node const& o = g.nodes[other];
            bool same = !o.constant && o.dtype == gdt_f16 && root(other) != root(t) && root(other) < t; // already computed when this launch runs
            for (int i = 0; i < 4; ++i) same = same && o.ne[i] == g.nodes[t].ne[i];
            if (!(same && g.nodes[root(other)].buffer >= 0)) break;
            if (round == 1 && !allow_two) break;
            (round == 0 ? e.res : e.res2) = other;
            skip[c] = 1;
            g.nodes[c].alias_of = t;
            if (g.nodes[c].is_output) g.nodes[t].is_output = true;
            e.last = c;
            c = sole_consumer(c);
        }
        return e;
    }
    static const char* act_name(int a) { return a == 1 ? "gelu" : (a == 2 ? "relu" : ""); }

    // a plain-GEMM A operand whose row length is not a multiple of 64 is copied into zero-padded rows first (the kernel reads whole
    // 64-wide k tiles; the weights' padding columns are zero, but 0 * garbage must not be NaN)
    int padded_rows(int t, int K, int Kp, int64_t M, std::string const& who) {
        const int src = buf_of(t);
        if (K == Kp) return src;
        const int dst = new_buffer((size_t)M * Kp * 2);
        auto sp = ptr(src), dp = ptr(dst);
        emit("pad_rows K=" + std::to_string(K) + "->" + std::to_string(Kp) + " <- " + who, {src}, {dst}, [=](void* st) {
            VX(vx_memset(dp(), 0, (size_t)M * Kp * 2, st));
            const int64_t ne[4] = {K, M, 1, 1}, ss[4] = {1, K, 0, 0}, ds[4] = {1, Kp, 0, 0};
            VX(vx_copy_strided_f16(sp(), dp(), ne, ss, ds, 1.0f, st));
        });
        return dst;
    }

    // interpolate (bilinear, align_corners) -> conv 3x3 32 -> 32 + bias -> relu -> conv 1x1 -> 1 -> relu [-> scale], every link read by the
    // next one only: the DPT head's tail (depth-anything.cpp:84-95) is ONE launch of the kernel made for it (kernels_headconv.hip)
    bool head_tail(int t) {
        graph_node const& up = g.nodes[t];
        graph_node const& x = g.nodes[up.src[0]];
        if ((up.ip[2] & 255) != 1 || !(up.ip[2] & 256) || x.ne[0] != 32) return false;
        const int c2 = sole_consumer(t);
        if (c2 < 0 || g.nodes[c2].op != gop_conv_2d || g.nodes[c2].src[0] != t || g.nodes[c2].n_src != 3) return false;
        graph_node const& w2 = g.nodes[g.nodes[c2].src[1]];
        if (w2.ne[0] != 32 || w2.ne[1] != 3 || w2.ne[2] != 3 || w2.ne[3] != 32 || g.nodes[c2].ip[0] != 1 || g.nodes[c2].ip[1] != 1) return false;
        const int r1 = sole_consumer(c2);
        if (r1 < 0 || g.nodes[r1].op != gop_relu) return false;
        const int c3 = sole_consumer(r1);
        if (c3 < 0 || g.nodes[c3].op != gop_conv_2d || g.nodes[c3].dtype != gdt_f32 || g.nodes[c3].src[0] != r1) return false;
        const int r2 = sole_consumer(c3);
        if (r2 < 0 || g.nodes[r2].op != gop_relu) return false; // the kernel's last ReLU is not optional
        const int H = (int)up.ne[2], W = (int)up.ne[1], hs = (int)x.ne[2], ws = (int)x.ne[1], B = (int)x.ne[3];
        if (!vx_headconv_supported(32, 32, H, W, hs, ws) || (size_t)B * hs * ws * 64 >= ((size_t)1 << 31)) return false;
        int last = r2;
        float scale = 1.0f;
        const int sc = sole_consumer(r2);
        if (sc >= 0 && g.nodes[sc].op == gop_scale) { scale = g.nodes[sc].fp[0]; last = sc; }
        // the 3x3 kernel as the head kernel's register fragments
        const int wt = g.nodes[c2].src[1];
        void* frag = cached(wt, 5, [&](bool st) {
            std::vector<uint16_t> rows((size_t)32 * 320, 0), out(vx_headconv_frag_bytes() / 2);
            for (int r = 0; r < 32; ++r)
                for (int k = 0; k < 288; ++k) rows[(size_t)r * 320 + k] = f32_to_f16(w2.values()[(size_t)r * 288 + k]);
            VX(vx_headconv_pack(rows.data(), 320, out.data()));
            return upload(out.data(), out.size() * 2, st);
        });
        const float* bias2 = const_f32(g.nodes[c2].src[2]);
        const float* w3 = const_f32(g.nodes[c3].src[1]);
        const float b3 = g.nodes[c3].n_src == 3 ? g.nodes[g.nodes[c3].src[2]].values()[0] : 0.0f;
        if (g.nodes[c3].n_src == 3) (void)const_f32(g.nodes[c3].src[2]); // (a kernel argument by value; its image makes the scalar part of the model's arena)
        const int xbuf = buf_of(up.src[0]);
        for (int f : {t, c2, r1, r2}) skip[f] = 1;
        if (last != r2) skip[last] = 1;
        skip[c3] = 1;
        graph_node& n = g.nodes[c3];
        for (int f : {r2, last}) {
            g.nodes[f].alias_of = c3;
            if (g.nodes[f].is_output) n.is_output = true;
        }
        materialise(c3);
        const int obuf = n.buffer;
        auto xp = ptr(xbuf), op = ptr(obuf);
        char d[160];
        snprintf(d, sizeof d, "head_tail[resize %dx%d -> %dx%d, conv3x3 32->32, relu, conv1x1 -> 1, relu%s] B=%d <- %s", ws, hs, W, H, scale != 1.0f ? ", scale" : "", B,
                 g.nodes[wt].name.c_str());
        emit(d, {xbuf}, {obuf}, [=](void* st) { VX(vx_headconv_bil_f16(xp(), frag, bias2, w3, b3, scale, reinterpret_cast<float*>(op()), B, H, W, hs, ws, st)); });
        tag("head_conv2+3", 2.0 * B * H * W * 32.0 * (288 + 1), (double)B * ((double)hs * ws * 64 + (double)H * W * 4));
        return true;
    }

    // conv 1x1 -> one channel [-> relu] [-> scale]: one f32 launch
    void one_channel_head(int t) {
        graph_node& n = g.nodes[t];
        graph_node const& x = g.nodes[n.src[0]];
        const int C = (int)x.ne[0];
        if (C % 8) throw except("conv_2d to one channel: Cin = %d must be a multiple of 8", C);
        int relu = 0;
        float scale = 1.0f;
        int last = t, c = sole_consumer(t);
        if (c >= 0 && g.nodes[c].op == gop_relu) { relu = 1; skip[c] = 1; g.nodes[c].alias_of = t; last = c; c = sole_consumer(c); }
        if (c >= 0 && g.nodes[c].op == gop_scale) { scale = g.nodes[c].fp[0]; skip[c] = 1; g.nodes[c].alias_of = t; last = c; }
        for (int f = t; ; f = consumers[f][0]) { // an output anywhere in the fused chain keeps the buffer
            if (g.nodes[f].is_output) n.is_output = true;
            if (f == last) break;
        }
        const float* w = const_f32(n.src[1]);
        const float bias = n.n_src == 3 ? g.nodes[n.src[2]].values()[0] : 0.0f;
        if (n.n_src == 3) (void)const_f32(n.src[2]);
        const int xbuf = buf_of(n.src[0]);
        materialise(t);
        const int obuf = n.buffer;
        auto xp = ptr(xbuf), op = ptr(obuf);
        const int64_t M = n.n_elements();
        std::string who = n.name.empty() ? g.nodes[n.src[1]].name : n.name;
        emit(std::string("conv1x1_to_1") + (relu ? "[relu]" : "") + (scale != 1.0f ? "[scale]" : "") + " M=" + std::to_string(M) + " C=" + std::to_string(C) + " <- " + who, {xbuf}, {obuf},
             [=](void* st) { VX(vx_conv1x1_to1_f32(xp(), w, bias, relu, scale, reinterpret_cast<float*>(op()), M, C, st)); });
        tag("head_out", 2.0 * M * C, (double)M * (C * 2 + 4));
    }

    // timing group of a matrix product from the weight's name (the groups the model entries and bench.py report)
    static std::string group_of(std::string const& who, graph_node const& n, int stride) {
        auto has = [&](const char* k) { return who.find(k) != std::string::npos; };
        if (has("reassemble") && has("projection")) return "neck_proj";
        if (has("reassemble") && has("resize")) return n.op == gop_conv_transpose_2d ? "neck_convT" : (stride > 1 ? "neck_conv_s2" : "neck_resize");
        if (has("neck.convs")) return "neck_convs";
        if (has("residual_layer")) return "fusion_rcu";
        if (has("fusion_stage") && has("projection")) return "fusion_proj";
        if (has("head.conv1")) return "head_conv1";
        if (has("head.conv")) return "head_conv2+3";
        return n.op == gop_linear ? "gemm" : "conv";
    }

    // ---- inputs that a consumer reads in place instead of through a copy / a resize launch (g.fused_models)
    struct sliced_rows { int src; int64_t begin, rows, stride; };  // rows [begin, begin + rows) of every group of `stride` rows of src
    std::map<int, sliced_rows> sliced_in;                           // slice node -> what its (sole, plain-GEMM) reader addresses directly
    struct resized_map { int buf; int hs, ws; };                    // the bilinear (align_corners) resize of [.., ws, hs, B] in `buf`
    std::map<int, resized_map> resized_in;                          // interpolate / projected node -> what its (sole) conv reader interpolates itself
    std::map<int, int> row_stride;                                  // node -> elements between rows when its buffer is padded for a GEMM reader

    int plain_gemm_reader(int t) const { // the sole reader of t (through single-reader views) if it reads t's rows as a plain GEMM A operand
        int c = sole_consumer(t);
        while (c >= 0 && (g.nodes[c].op == gop_reshape || g.nodes[c].op == gop_cont)) c = sole_consumer(c);
        if (c < 0 || root_view(g.nodes[c].src[0]) != t) return -1;
        graph_node const& n = g.nodes[c];
        if (n.op == gop_linear || n.op == gop_conv_transpose_2d) return c;
        if (n.op == gop_conv_2d && n.dtype == gdt_f16) {
            graph_node const& w = g.nodes[n.src[1]];
            if (w.ne[1] == 1 && w.ne[2] == 1 && n.ip[0] == 1 && n.ip[1] == 0) return c;
        }
        return -1;
    }
    int root_view(int v) const { // the node behind reshape / cont views (before aliases are assigned)
        while (g.nodes[v].op == gop_reshape || g.nodes[v].op == gop_cont) v = g.nodes[v].src[0];
        return v;
    }
    bool dconv_shape(graph_node const& n, graph_node const& x, graph_node const& w) const {
        static const bool off = getenv("VISP_NO_DCONV") != nullptr;
        static const int min_w = getenv("VISP_DCONV_MIN_W") ? atoi(getenv("VISP_DCONV_MIN_W")) : 16; // A/B runs: narrower maps take the GEMM family's conv forms
        return g.fused_models && !off && n.op == gop_conv_2d && n.dtype == gdt_f16 && w.ne[1] == 3 && w.ne[2] == 3 && n.ip[0] == 1 && n.ip[1] == 1 && x.ne[0] % 16 == 0 &&
               (w.ne[3] == 32 || w.ne[3] == 64) && n.ne[1] >= min_w;
    }
    // the LDS-ring conv's operand: slabs [cin pad 32 / 32][9 taps][cout][32] f16, 16-byte groups swizzled (kernels_dconv.hip)
    void* pack_dconv(int wt, int cin, int cout, int* d_cin) {
        const int dc = round_up(cin, 32);
        *d_cin = dc;
        return cached(wt, 6, [&](bool st) {
            std::vector<uint16_t> h((size_t)dc * 9 * cout, 0);
            const float* w = g.nodes[wt].values(); // [cout][ky][kx][cin]
            for (int n = 0; n < cout; ++n)
                for (int tap = 0; tap < 9; ++tap)
                    for (int c = 0; c < cin; ++c) {
                        const size_t row = ((size_t)(c / 32) * 9 + tap) * cout + n;
                        h[row * 32 + (size_t)(((c % 32) / 8) ^ ((n >> 2) & 3)) * 8 + c % 8] = f32_to_f16(w[((size_t)n * 9 + tap) * cin + c]);
                    }
            return upload(h.data(), h.size() * 2, st);
        });
    }

    // ---- maps kept as 32-channel PLANES ([plane][B * H * W][32] f16) -- the layout the LDS-ring conv was built on for the ESRGAN dense blocks
    // (kernels_dconv.hip; profiles/r04_esrgan_layout_experiment.txt: 5 % faster than NHWC). Layout is this lowering's choice: a map is planar when
    // the conv that writes it read a planar map (or the f32 image), every reader that is such a conv takes it as it is, and any other reader gets
    // an NHWC copy made on first use. A channel concat of planar maps is then no launch at all: the dense block [x | x1 | x2 | x3 | x4] of
    // esrgan.cpp:27-41 is six planes of ONE buffer, x written by its producer into planes 0-1, every conv_block into the plane behind its input.
    struct planar_map { int buf; int plane0; int n_planes; int64_t plane_elems; };
    std::map<int, planar_map> planar;                 // ROOT node -> where it lives
    std::map<int, std::pair<int, int>> placed;        // node a concat chain will read -> (buffer, first plane) reserved for it
    std::map<int, int> up2_of;                        // interpolate(NEAREST, x2) node -> its (planar) source, resized by the reading conv's loader
    std::map<int, int> image_plane;                   // f32 image input -> buffer of its value | residue plane

    void planar_to_nhwc(int r) {
        graph_node& n = g.nodes[r];
        const planar_map pm = planar.at(r);
        materialise(r);
        const int obuf = n.buffer;
        auto sp = ptr(pm.buf), op = ptr(obuf);
        const int64_t npix = pm.plane_elems / 32, C = n.ne[0], off = (int64_t)pm.plane0 * pm.plane_elems;
        std::array<int64_t, 4> A{32, npix, pm.n_planes, 1}, S{1, 32, pm.plane_elems, 0}, D{1, C, 32, 0};
        emit("planes_to_nhwc " + shape_str(n.ne), {pm.buf}, {obuf}, [=](void* st) { VX(vx_copy_strided_f16(sp() + off * 2, op(), A.data(), S.data(), D.data(), 1.0f, st)); });
    }
    static bool is_conv3x3(graph_node const& c, graph_node const& w) {
        return c.op == gop_conv_2d && w.constant && w.ne[1] == 3 && w.ne[2] == 3 && c.ip[0] == 1 && c.ip[1] == 1;
    }
    // the LDS-ring conv's operand with the input columns repeated (`dup`: an image plane carries value | residue) and the outputs padded to `cout_pad`
    void* pack_dconv_ex(int wt, int cin, int dup, int cout, int cout_pad, int* d_cin) {
        const int n_in = dup ? 2 * cin : cin, dc = round_up(n_in, 32);
        *d_cin = dc;
        return cached(wt, 8, [&](bool st) {
            std::vector<uint16_t> h((size_t)dc * 9 * cout_pad, 0);
            const float* w = g.nodes[wt].values(); // [cout][ky][kx][cin]
            for (int n = 0; n < cout; ++n)
                for (int tap = 0; tap < 9; ++tap)
                    for (int c = 0; c < n_in; ++c) {
                        const size_t row = ((size_t)(c / 32) * 9 + tap) * cout_pad + n;
                        h[row * 32 + (size_t)(((c % 32) / 8) ^ ((n >> 2) & 3)) * 8 + c % 8] = f32_to_f16(w[((size_t)n * 9 + tap) * cin + c % cin]);
                    }
            return upload(h.data(), h.size() * 2, st);
        }, "dup" + std::to_string(dup) + "pad" + std::to_string(cout_pad));
    }
    float* bias_padded(int bt, int n_pad) {
        return static_cast<float*>(cached(bt, 9, [&](bool st) {
            std::vector<float> b((size_t)n_pad, 0.0f);
            for (int64_t i = 0; i < g.nodes[bt].n_elements() && i < n_pad; ++i) b[(size_t)i] = g.nodes[bt].values()[i];
            return upload(b.data(), b.size() * 4, st);
        }, "pad" + std::to_string(n_pad)));
    }
    // interpolate(NEAREST) to exactly twice the extent of a planar map, read by one 3x3 conv: the conv's loader does it (esrgan.cpp:13-19)
    bool nearest_into_reader(int t) {
        graph_node const& n = g.nodes[t];
        const int xs = root(n.src[0]);
        graph_node const& x = g.nodes[xs];
        if (!g.fused_models || (n.ip[2] & 255) != 0 || n.is_output || !planar.count(xs) || n.ne[1] != 2 * x.ne[1] || n.ne[2] != 2 * x.ne[2]) return false;
        const int c = sole_consumer(t);
        if (c < 0 || g.nodes[c].src[0] != t || !is_conv3x3(g.nodes[c], g.nodes[g.nodes[c].src[1]])) return false;
        up2_of[t] = xs;
        return true;
    }
    // A 3x3 / stride 1 / pad 1 conv on a planar map (or on the f32 image, or on the x2 nearest resize of a planar map) through the LDS-ring conv in
    // its native layout, with esrgan.cpp's epilogues: LeakyReLU 0.2 or ReLU, then up to two [* scale] + residual steps (conv5 * 0.2 + x, and
    // (.) * 0.2 + the rrdb's input, esrgan.cpp:38-40, 49-50), the skip x taken from the halo when it is the conv's own first 64 input channels;
    // the RGB conv writes the f32 image. Returns false (nothing changed) when the conv is not of this kind.
    bool planar_conv(int t, std::string const& who) {
        graph_node& n = g.nodes[t];
        if (!g.fused_models || !is_conv3x3(n, g.nodes[n.src[1]]) || (n.n_src == 3 && !g.nodes[n.src[2]].constant)) return false;
        const int xs = n.src[0], xr = root(xs);
        graph_node const& x = g.nodes[xr];
        const int cin = (int)g.nodes[xs].ne[0], cout = (int)n.ne[0], H = (int)n.ne[2], W = (int)n.ne[1], B = (int)n.ne[3];
        const bool rgb = cout == 3 && n.dtype == gdt_f32;
        if (!rgb && !((cout == 32 || cout == 64) && n.dtype == gdt_f16)) return false;
        const int64_t npix = (int64_t)B * H * W;
        planar_map in{};
        int up2 = 0, dup = 0;
        if (auto it = up2_of.find(xs); it != up2_of.end()) {
            in = planar.at(it->second);
            up2 = 1;
        } else if (planar.count(xr)) {
            in = planar.at(xr);
        } else if (x.op == gop_input && x.dtype == gdt_f32 && cin <= 16) {
            auto it = image_plane.find(xr);
            if (it == image_plane.end()) {
                const int ibuf = buf_of(xr), pbuf = new_buffer((size_t)npix * 32 * 2);
                auto ip = ptr(ibuf), pp = ptr(pbuf);
                emit("image_planes " + shape_str(x.ne), {ibuf}, {pbuf}, [=](void* st) { VX(vx_image_planes_f32(reinterpret_cast<const float*>(ip()), pp(), npix, cin, st)); });
                tag("image_in", 0, (double)npix * (cin * 4 + 64));
                it = image_plane.emplace(xr, pbuf).first;
            }
            in = {it->second, 0, 1, npix * 32};
            dup = cin;
        } else return false;
        if (!dup && cin % 32) return false;

        // ---- the epilogue chain
        int act = 0, last = t;
        float s1 = 1.0f, s2 = 1.0f;
        int res[2] = {-1, -1};
        std::vector<int> absorbed;
        {
            int c = rgb ? -1 : sole_consumer(t);
            if (c >= 0 && (g.nodes[c].op == gop_relu || (g.nodes[c].op == gop_leaky_relu && g.nodes[c].fp[0] == 0.2f))) {
                act = g.nodes[c].op == gop_relu ? 2 : 1;
                absorbed.push_back(c);
                last = c;
                c = sole_consumer(c);
            }
            for (int round = 0; round < 2 && c >= 0; ++round) {
                float sc = 1.0f;
                int sn = -1, a = c;
                if (g.nodes[c].op == gop_scale) { sn = c; sc = g.nodes[c].fp[0]; a = sole_consumer(c); }
                if (a < 0 || g.nodes[a].op != gop_add) break;
                graph_node const& an = g.nodes[a];
                const int prev = sn >= 0 ? sn : last;
                const int other = an.src[0] == prev ? an.src[1] : (an.src[1] == prev ? an.src[0] : -1);
                if (other < 0 || !planar.count(root(other)) || g.nodes[other].n_elements() != n.n_elements() || g.nodes[other].ne[0] != cout) break;
                (round == 0 ? s1 : s2) = sc;
                res[round] = root(other);
                if (sn >= 0) absorbed.push_back(sn);
                absorbed.push_back(a);
                last = a;
                c = sole_consumer(a);
            }
        }
        for (int c : absorbed) {
            skip[c] = 1;
            g.nodes[c].alias_of = t;
            if (g.nodes[c].is_output) n.is_output = true;
        }

        vx_dconv_args d;
        memset(&d, 0, sizeof d);
        int d_cin = 0;
        d.w = (dup || rgb) ? pack_dconv_ex(n.src[1], cin, dup, cout, rgb ? 32 : cout, &d_cin) : pack_dconv(n.src[1], cin, cout, &d_cin);
        d.bias = n.n_src == 3 ? (rgb ? bias_padded(n.src[2], 32) : const_f32(n.src[2])) : nullptr;
        d.cin = d_cin; d.cout = rgb ? 32 : cout;
        d.up2 = up2;
        d.B = B; d.H = H; d.W = W;
        d.epi = rgb ? VX_DC_RGB_F32 : VX_DC_F16;
        d.act = act;
        d.s1 = s1; d.s2 = s2;
        d.x_plane = in.plane_elems;
        const int64_t x_off = (int64_t)in.plane0 * in.plane_elems;
        std::vector<int> reads = {in.buf};
        std::function<char*()> rp[2];
        int64_t r_off[2] = {0, 0};
        for (int i = 0; i < 2; ++i) {
            if (res[i] < 0) continue;
            planar_map const& rm = planar.at(res[i]);
            if (i == 0 && !up2 && cout == 64 && rm.buf == in.buf && rm.plane0 == in.plane0 && in.n_planes >= 2 && s1 != 0.0f) { d.x_residual = 1; continue; } // x = the halo's first two planes
            reads.push_back(rm.buf);
            rp[i] = ptr(rm.buf);
            r_off[i] = (int64_t)rm.plane0 * rm.plane_elems;
            (i == 0 ? d.res1_plane : d.res2_plane) = rm.plane_elems;
        }

        // ---- where the result goes
        int obuf;
        int64_t o_off = 0;
        if (rgb) {
            materialise(t);
            obuf = n.buffer;
        } else {
            planar_map om{-1, 0, cout / 32, npix * 32};
            if (auto it = placed.find(last); it != placed.end()) { om.buf = it->second.first; om.plane0 = it->second.second; placed.erase(it); }
            else {
                // the head of a concat chain along the channels: reserve the whole dense block and the slots of the maps that will join it
                std::vector<std::pair<int, int>> joins; // (concat node, second operand)
                int cur = last, width = cout;
                for (;;) {
                    int cc = -1;
                    for (int c : consumers[cur])
                        if (g.nodes[c].op == gop_concat && g.nodes[c].ip[0] == 0 && g.nodes[c].src[0] == cur && !g.nodes[c].is_output) { cc = c; break; }
                    if (cc < 0) break;
                    const int s2n = g.nodes[cc].src[1];
                    graph_node const& sn = g.nodes[s2n];
                    // the joining map must be one this function will write: a 3x3 conv of the chain so far to 32 or 64 channels (constant bias), alone or behind
                    // an activation its epilogue absorbs
                    auto joins_by_conv = [&](int cn) {
                        graph_node const& c = g.nodes[cn];
                        return !c.constant && c.op == gop_conv_2d && c.dtype == gdt_f16 && is_conv3x3(c, g.nodes[c.src[1]]) && (c.n_src == 2 || g.nodes[c.src[2]].constant) &&
                               (c.ne[0] == 32 || c.ne[0] == 64) && c.src[0] == cur;
                    };
                    const bool act_of_conv = (sn.op == gop_relu || (sn.op == gop_leaky_relu && sn.fp[0] == 0.2f)) && !sn.constant && joins_by_conv(sn.src[0]) && uses[sn.src[0]] == 1 &&
                                             !g.nodes[sn.src[0]].is_output;
                    const bool conv_itself = joins_by_conv(s2n);
                    if (!(act_of_conv || conv_itself) || uses[s2n] != 1 || sn.is_output || s2n < cur) break;
                    joins.emplace_back(cc, s2n);
                    width += (int)sn.ne[0];
                    cur = cc;
                }
                om.buf = new_buffer((size_t)npix * width * 2, n.is_output);
                int at = cout / 32;
                for (auto const& j : joins) {
                    placed[j.second] = {om.buf, at};
                    at += (int)g.nodes[j.second].ne[0] / 32;
                    planar[j.first] = {om.buf, 0, at, npix * 32};
                    skip[j.first] = 1;
                }
            }
            planar[t] = om;
            obuf = om.buf;
            o_off = (int64_t)om.plane0 * om.plane_elems;
            d.out_plane = om.plane_elems;
        }
        char desc[320];
        snprintf(desc, sizeof desc, "dconv3x3(planes)%s%s%s%s%s%s M=%lld N=%d K=%d <- %s", dup ? "[f32 image]" : "", up2 ? "[nearest x2 in the loader]" : "",
                 act == 1 ? "[leaky_relu]" : (act == 2 ? "[relu]" : ""), res[0] >= 0 ? (d.x_residual ? "[*s + x]" : "[*s + res]") : "", res[1] >= 0 ? "[*s + res]" : "", rgb ? "[rgb f32]" : "",
                 (long long)npix, cout, 9 * cin, who.c_str());
        auto xp = ptr(in.buf), op = ptr(obuf);
        emit(desc, reads, {obuf}, [=](void* st) {
            vx_dconv_args r = d;
            r.x = xp() + x_off * 2;
            r.out = op() + o_off * 2;
            if (rp[0]) r.res1 = rp[0]() + r_off[0] * 2;
            if (rp[1]) r.res2 = rp[1]() + r_off[1] * 2;
            VX(vx_dconv3x3_f16(&r, st));
        });
        std::string grp = rgb ? "last" : (dup ? "first" : (up2 ? "upconv" : "conv"));
        if (auto k = who.find(".RDB"); k != std::string::npos && who.find(".conv", k) != std::string::npos) grp = "rdb_conv" + who.substr(who.find(".conv", k) + 5, 1);
        else if (!rgb && !dup && !up2 && who.find(".sub.") != std::string::npos) grp = "trunk";
        else if (!rgb && !dup && !up2 && act == 1) grp = "hrconv";
        tag(grp.c_str(), 2.0 * npix * (double)cout * 9 * cin, (double)npix * ((up2 ? 0.25 : 1.0) * d_cin * 2 + (rgb ? 12 : cout * 2)));
        if (!rgb && n.is_output) planar_to_nhwc(t); // a graph output is read back as NHWC
        return true;
    }

    // the residual-unit kernel's operand: [taps][64 n][64 c] f16, the 16-byte groups of row n at position g ^ ((n >> 1) & 7) (kernels_rcu.hip)
    void* pack_rcu(int wt, int taps) {
        return cached(wt, 7, [&](bool st) {
            std::vector<uint16_t> h((size_t)taps * 64 * 64);
            const float* w = g.nodes[wt].values(); // [n][ky][kx][c]
            for (int n = 0; n < 64; ++n)
                for (int tap = 0; tap < taps; ++tap)
                    for (int c = 0; c < 64; ++c)
                        h[((size_t)tap * 64 + n) * 64 + (size_t)((c >> 3) ^ ((n >> 1) & 7)) * 8 + (c & 7)] = f32_to_f16(w[((size_t)n * taps + tap) * 64 + c]);
            return upload(h.data(), h.size() * 2, st);
        });
    }
    std::map<int, int> preprojected; // interpolate node -> buffer that already holds the 1x1 projection of its SOURCE (written by the residual-unit launch)

    // dpt::residual_conv on a small map (depth-anything.cpp:15-23): relu -> conv3x3 -> relu -> conv3x3 -> + x, every link read by the next one only,
    // [+ feature_fusion's other addend (:28-31)] [-> the 1x1 out_conv that commutes with the resize behind the unit (:36-40)]: ONE launch of
    // kernels_rcu.hip with the intermediate map in LDS. t = the first conv (its ReLU is on its loader already). Returns false and leaves
    // everything untouched when the pattern or the kernel's limits do not hold.
    bool residual_unit(int t, std::string const& who) {
        static const bool off = getenv("VISP_NO_RCU_FUSE") != nullptr;
        graph_node& n = g.nodes[t];
        const int xs = n.src[0];
        graph_node const& x = g.nodes[xs];
        const int H = (int)n.ne[2], W = (int)n.ne[1], B = (int)n.ne[3];
        auto conv64 = [&](graph_node const& c) {
            graph_node const& w = g.nodes[c.src[1]];
            return c.op == gop_conv_2d && c.dtype == gdt_f16 && w.constant && w.ne[0] == 64 && w.ne[1] == 3 && w.ne[2] == 3 && w.ne[3] == 64 && c.ip[0] == 1 && c.ip[1] == 1 &&
                   (c.n_src == 2 || g.nodes[c.src[2]].constant);
        };
        if (off || !g.fused_models || !relu_on_load[t] || !conv64(n) || x.ne[0] != 64 || n.is_output || resized_in.count(xs) || !vx_rcu_supported(H, W)) return false;
        const int r1 = sole_consumer(t);
        if (r1 < 0 || g.nodes[r1].op != gop_relu) return false;
        const int c2 = sole_consumer(r1);
        if (c2 < 0 || !conv64(g.nodes[c2]) || g.nodes[c2].src[0] != r1) return false;
        const int a1 = sole_consumer(c2);
        if (a1 < 0 || g.nodes[a1].op != gop_add) return false;
        {
            graph_node const& a = g.nodes[a1];
            const int other = root(a.src[0]) == c2 ? a.src[1] : a.src[0];
            if (root(other) != root(xs) || (root(a.src[0]) != c2 && root(a.src[1]) != c2)) return false; // the first addend must be the unit's own input
        }
        skip[r1] = 1;
        g.nodes[r1].alias_of = t;
        epilogue e = fuse_epilogue(c2, false, false, true);
        if (e.act != 0 || e.res < 0 || root(e.res) != root(xs)) throw except("graph: internal: residual unit at %s did not fuse as expected", who.c_str());
        skip[c2] = 1;
        // the projection behind the unit's resize
        int up = g.nodes[e.last].is_output ? -1 : sole_consumer(e.last), pj = -1;
        if (up >= 0) {
            graph_node const& u = g.nodes[up];
            const bool bil = u.op == gop_interpolate && (u.ip[2] & 255) == 1 && (u.ip[2] & 256) && !u.is_output && root(u.src[0]) == c2;
            pj = bil ? sole_consumer(up) : -1;
            if (pj >= 0) {
                graph_node const& c = g.nodes[pj];
                graph_node const& w = g.nodes[c.src[1]];
                if (!(c.op == gop_conv_2d && c.dtype == gdt_f16 && c.src[0] == up && w.constant && w.ne[0] == 64 && w.ne[1] == 1 && w.ne[2] == 1 && w.ne[3] == 64 && c.ip[0] == 1 && c.ip[1] == 0 &&
                      (c.n_src == 2 || g.nodes[c.src[2]].constant)))
                    pj = -1;
            }
        }
        vx_rcu_args a;
        memset(&a, 0, sizeof a);
        a.w1 = pack_rcu(n.src[1], 9);
        a.b1 = n.n_src == 3 ? const_f32(n.src[2]) : nullptr;
        a.w2 = pack_rcu(g.nodes[c2].src[1], 9);
        a.b2 = g.nodes[c2].n_src == 3 ? const_f32(g.nodes[c2].src[2]) : nullptr;
        a.B = B; a.H = H; a.W = W;
        const int xbuf = buf_of(xs);
        std::vector<int> reads = {xbuf};
        std::function<char*()> rp2;
        if (e.res2 >= 0) { reads.push_back(buf_of(e.res2)); rp2 = ptr(buf_of(e.res2)); }
        int obuf;
        if (pj >= 0) {
            a.wp = pack_rcu(g.nodes[pj].src[1], 1);
            a.bp = g.nodes[pj].n_src == 3 ? const_f32(g.nodes[pj].src[2]) : nullptr;
            obuf = new_buffer((size_t)B * H * W * 64 * 2);
            preprojected[up] = obuf;
        } else {
            materialise(c2);
            obuf = g.nodes[c2].buffer;
        }
        char d[320];
        snprintf(d, sizeof d, "residual_unit[relu, conv3x3, relu, conv3x3, + x%s%s] %dx%d B=%d <- %s", e.res2 >= 0 ? ", + x0" : "", pj >= 0 ? ", conv1x1 before its resize" : "", W, H, B,
                 who.c_str());
        auto xp = ptr(xbuf), op = ptr(obuf);
        emit(d, reads, {obuf}, [=](void* st) {
            vx_rcu_args r = a;
            r.x = xp();
            r.out = op();
            if (rp2) r.res2 = rp2();
            VX(vx_rcu_fused_f16(&r, st));
        });
        const double px = (double)B * H * W;
        tag("fusion_rcu", 2.0 * px * 64 * (576 * 2 + (pj >= 0 ? 64 : 0)), px * 64 * 2 * (2 + (e.res2 >= 0 ? 1 : 0)));
        return true;
    }

    void gemm_like(int t) {
        graph_node& n = g.nodes[t];
        {
            std::string w0 = g.nodes[n.src[1]].op == gop_weight && !g.nodes[n.src[1]].name.empty() ? g.nodes[n.src[1]].name : n.name;
            if (n.op == gop_conv_2d && planar_conv(t, w0)) return;
        }
        if (n.op == gop_conv_2d && n.dtype == gdt_f32 && n.ne[0] != 1) throw except("conv_2d %s to an f32 image of %lld channels is lowered onto the LDS-ring conv only (3x3 / stride 1 / pad 1 behind planar maps)", n.name.c_str(), (long long)n.ne[0]);
        if (n.op == gop_conv_2d && n.dtype == gdt_f32) return one_channel_head(t);
        const int xs = n.src[0], wt = n.src[1], bt = n.n_src == 3 ? n.src[2] : -1;
        graph_node const& x = g.nodes[xs];
        std::string who = n.name.empty() ? g.nodes[wt].name : n.name;
        if (g.nodes[wt].op == gop_weight && !g.nodes[wt].name.empty()) who = g.nodes[wt].name; // groups are keyed on the weight's name
        vx_gemm_args a;
        memset(&a, 0, sizeof a);
        packed_operand p;
        int xbuf = -1;
        std::string kind;
        bool conv = false;
        epilogue e;
        e.last = t;
        int stride = 1;
        // the A rows of a plain product: in place from a sliced source, from a padded producer, or padded by a copy
        auto plain_rows = [&](int K, int Kp) {
            const int v = root_view(xs);
            if (auto it = sliced_in.find(v); it != sliced_in.end()) {
                a.a_group = (int)it->second.rows; a.a_group_stride = (int)it->second.stride; a.a_row_off = (int)it->second.begin;
                a.lda = K;
                kind += "[rows " + std::to_string(it->second.begin) + ".. of " + std::to_string(it->second.stride) + "]";
                return buf_of(it->second.src);
            }
            if (auto it = row_stride.find(v); it != row_stride.end() && it->second == Kp) { a.lda = Kp; return buf_of(xs); }
            a.lda = Kp;
            return padded_rows(xs, K, Kp, a.M, who);
        };
        if (n.op == gop_linear) {
            e = fuse_epilogue(t, true, true);
            p = pack_rows(wt, bt, (int)n.ne[0], e.scale);
            a.M = (int)(x.n_elements() / x.ne[0]);
            kind = "gemm";
            xbuf = plain_rows(p.k_real, p.K);
        } else if (n.op == gop_conv_2d) {
            graph_node const& w = g.nodes[wt];
            const int kw = (int)w.ne[1], kh = (int)w.ne[2], pad = (int)n.ip[1];
            stride = (int)n.ip[0];
            a.M = (int)(n.ne[1] * n.ne[2] * n.ne[3]);
            if (kw == 1 && kh == 1 && stride == 1 && pad == 0) { // 1x1: a plain product on the pixel rows (nn.cpp:76-81)
                p = pack_rows(wt, bt, (int)n.ne[0]);
                kind = "gemm(conv1x1)";
                xbuf = plain_rows(p.k_real, p.K);
            } else {
                if (x.ne[0] % 8) throw except("conv_2d %s: Cin = %lld must be a multiple of 8", who.c_str(), (long long)x.ne[0]);
                if (dconv_shape(n, x, w)) return dconv(t, who);
                p = pack_rows(wt, bt, (int)n.ne[0]);
                xbuf = buf_of(xs);
                a.conv_kh = kh; a.conv_kw = kw; a.conv_stride = stride; a.conv_pad = pad;
                a.conv_H = (int)x.ne[2]; a.conv_W = (int)x.ne[1]; a.conv_Cin = (int)x.ne[0];
                a.conv_OH = (int)n.ne[2]; a.conv_OW = (int)n.ne[1];
                a.a_relu = relu_on_load[t];
                conv = true;
                kind = "conv" + std::to_string(kw) + "x" + std::to_string(kh) + (stride > 1 ? "s" + std::to_string(stride) : "") + (a.a_relu ? "[relu-in]" : "");
            }
        } else { // conv_transpose_2d with kernel == stride: a product + pixel shuffle
            const int s = (int)n.ip[0];
            p = pack_conv_transpose(wt, bt, s);
            a.M = (int)(x.ne[1] * x.ne[2] * x.ne[3]);
            kind = "gemm+pixel_shuffle(s" + std::to_string(s) + ")";
            xbuf = plain_rows(p.k_real, p.K);
            a.ps_s = s; a.ps_Cout = (int)n.ne[0]; a.ps_H = (int)x.ne[2]; a.ps_W = (int)x.ne[1];
        }
        if (n.ne[0] % 8) throw except("%s %s: %lld output channels; the f16 epilogues store 8 at a time", graph_op_name(n.op), who.c_str(), (long long)n.ne[0]);
        a.W = p.w; a.bias = p.bias; a.N = p.N; a.K = p.K; a.n_valid = n.op == gop_conv_transpose_2d ? p.n_real : (int)n.ne[0];
        a.ldo = n.ne[0];
        if (n.op == gop_conv_transpose_2d) a.epi = VX_EPI_PIXSHUF;
        else {
            if (n.op != gop_linear) e = fuse_epilogue(t, true, false, true);
            a.epi = e.res >= 0 ? VX_EPI_F16_ADD : (e.act == 1 ? VX_EPI_F16_GELU : (e.act == 2 ? VX_EPI_F16_RELU : VX_EPI_F16));
            a.relu = e.res >= 0 && e.act == 2;
        }
        // rows of a width that is no multiple of the GEMM's 64-wide k tile, read by one plain product only: written with the padded row
        // stride that reader wants (the pad columns are exact zeros: zero weights, zero bias), no pad_rows copy in between
        size_t out_bytes = n.n_bytes();
        if (g.fused_models && a.epi == VX_EPI_F16 && n.op != gop_conv_transpose_2d && n.ne[0] % 64 != 0 && !n.is_output && p.N >= round_up((int)n.ne[0], 64) && plain_gemm_reader(t) >= 0) {
            const int Kp = round_up((int)n.ne[0], 64);
            row_stride[t] = Kp;
            a.ldo = Kp;
            a.n_valid = Kp;
            out_bytes = (size_t)(n.n_elements() / n.ne[0]) * Kp * 2;
        }
        n.buffer = new_buffer(out_bytes, n.is_output || n.op == gop_input);
        const int obuf = n.buffer;
        std::vector<int> reads = {xbuf};
        std::function<char*()> rp, rp2;
        if (e.res >= 0) { reads.push_back(buf_of(e.res)); rp = ptr(buf_of(e.res)); }
        if (e.res2 >= 0) { reads.push_back(buf_of(e.res2)); rp2 = ptr(buf_of(e.res2)); }
        static const int halo_min_w = getenv("VISP_HALO_MIN_W") ? atoi(getenv("VISP_HALO_MIN_W")) : 96;
        const bool halo = conv && a.conv_kh == 3 && a.conv_kw == 3 && a.conv_stride == 1 && a.conv_pad == 1 && a.conv_W >= halo_min_w;
        char d[320];
        snprintf(d, sizeof d, "%s%s%s%s%s%s%s M=%d N=%d K=%d <- %s", kind.c_str(), e.scale >= 0 ? "[*scale]" : "", e.act ? "[" : "", act_name(e.act), e.act ? "]" : "",
                 e.res >= 0 ? "[+res]" : "", e.res2 >= 0 ? "[+res]" : "", a.M, (int)n.ne[0], p.k_real, who.c_str());
        auto xp = ptr(xbuf), op = ptr(obuf);
        emit(d, reads, {obuf}, [=](void* st) {
            vx_gemm_args r = a;
            r.A = xp();
            r.out = op();
            if (rp) r.res1 = rp();
            if (rp2) r.res2 = rp2();
            if (halo && vx_conv3x3_supported(&r)) VX(vx_conv3x3_f16(&r, st));
            else VX(vx_gemm_f16(&r, st));
        });
        tag(group_of(who, n, stride).c_str(), 2.0 * a.M * (double)p.n_real * p.k_real, (double)a.M * (p.k_real + n.ne[0]) * 2 + (double)p.N * p.K * 2);
    }

    // 3x3 / stride 1 / pad 1 with Cin % 16 == 0 and 32 or 64 output channels: the persistent LDS-ring conv (kernels_dconv.hip), with the
    // ReLU of a relu -> conv chain applied to the fragments, ReLU or up to two residual maps in the epilogue, and -- where the input is the
    // bilinear (align_corners) resize of a smaller map read by this conv only -- the resize done by the conv's halo loader
    void dconv(int t, std::string const& who) {
        if (residual_unit(t, who)) return;
        graph_node& n = g.nodes[t];
        const int xs = n.src[0], wt = n.src[1], bt = n.n_src == 3 ? n.src[2] : -1;
        graph_node const& x = g.nodes[xs];
        const int cin = (int)x.ne[0], cout = (int)n.ne[0], H = (int)n.ne[2], W = (int)n.ne[1], B = (int)n.ne[3];
        epilogue e = fuse_epilogue(t, false, false, true);
        if (e.act == 2 && e.res >= 0) throw except("graph: internal: relu between a conv and its residual add is not a dconv epilogue");
        vx_dconv_args d;
        memset(&d, 0, sizeof d);
        int d_cin = 0;
        d.w = pack_dconv(wt, cin, cout, &d_cin);
        d.bias = bt >= 0 ? const_f32(bt) : nullptr;
        d.cin = d_cin; d.cout = cout;
        d.x_pix = cin; d.x_plane = 32;
        d.B = B; d.H = H; d.W = W;
        d.epi = VX_DC_F16;
        d.act = e.act == 2 ? 2 : 0;
        d.a_relu = relu_on_load[t];
        d.s1 = d.s2 = 1.0f;
        d.res1_pix = d.res2_pix = d.out_pix = cout;
        d.res1_plane = d.res2_plane = d.out_plane = 32;
        int xbuf;
        std::string kind = "dconv3x3";
        if (auto it = resized_in.find(xs); it != resized_in.end()) {
            xbuf = it->second.buf;
            d.bil_hs = it->second.hs; d.bil_ws = it->second.ws;
            kind += "[resize " + std::to_string(it->second.ws) + "x" + std::to_string(it->second.hs) + " in the loader]";
        } else xbuf = buf_of(xs);
        if (d.a_relu) kind += "[relu-in]";
        materialise(t);
        const int obuf = n.buffer;
        std::vector<int> reads = {xbuf};
        std::function<char*()> rp, rp2;
        if (e.res >= 0) { reads.push_back(buf_of(e.res)); rp = ptr(buf_of(e.res)); }
        if (e.res2 >= 0) { reads.push_back(buf_of(e.res2)); rp2 = ptr(buf_of(e.res2)); }
        char desc[320];
        snprintf(desc, sizeof desc, "%s%s%s%s M=%d N=%d K=%d <- %s", kind.c_str(), e.act == 2 ? "[relu]" : "", e.res >= 0 ? "[+res]" : "", e.res2 >= 0 ? "[+res]" : "", B * H * W, cout,
                 9 * cin, who.c_str());
        auto xp = ptr(xbuf), op = ptr(obuf);
        emit(desc, reads, {obuf}, [=](void* st) {
            vx_dconv_args r = d;
            r.x = xp();
            r.out = op();
            if (rp) r.res1 = rp();
            if (rp2) r.res2 = rp2();
            VX(vx_dconv3x3_f16(&r, st));
        });
        tag(group_of(who, n, 1).c_str(), 2.0 * B * H * W * (double)cout * 9 * cin, (double)B * H * W * (cin + cout) * 2);
    }

    // interpolate (bilinear, align_corners) read by ONE consumer:
    //   * conv 1x1 (+ bias): the two commute (the bilinear weights of a pixel sum to one), so the projection runs on the small map -- a
    //     quarter of the products and no full-resolution intermediate (depth-anything.cpp:36-40) -- and the resize follows;
    //   * conv 3x3 with 32 output channels that the LDS-ring kernel can feed from the small map: no resize launch at all.
    // Returns true when the node needs no launch of its own.
    bool resize_into_reader(int t) {
        if (!g.fused_models) return false;
        graph_node const& up = g.nodes[t];
        if ((up.ip[2] & 255) != 1 || !(up.ip[2] & 256) || up.is_output) return false;
        const int c = sole_consumer(t);
        if (c < 0 || g.nodes[c].op != gop_conv_2d || g.nodes[c].src[0] != t || g.nodes[c].dtype != gdt_f16) return false;
        graph_node const& x = g.nodes[up.src[0]];
        graph_node& cn = g.nodes[c];
        graph_node const& w = g.nodes[cn.src[1]];
        const int hs = (int)x.ne[2], ws = (int)x.ne[1], H = (int)up.ne[2], W = (int)up.ne[1], B = (int)x.ne[3];
        static const bool no_bil = getenv("VISP_NO_BIL_FUSE") != nullptr;
        auto loader_can = [&](graph_node const& conv, graph_node const& cw, int cin) {
            return !no_bil && dconv_shape(conv, g.nodes[conv.src[0]], cw) && cw.ne[3] == 32 && cin % 32 == 0 && !relu_on_load[&conv - g.nodes.data()] &&
                   vx_dconv_bilinear_supported(32, H, W, hs, ws);
        };
        if (w.ne[1] == 3 && w.ne[2] == 3) {
            if (!loader_can(cn, w, (int)x.ne[0])) return false;
            resized_in[t] = {buf_of(up.src[0]), hs, ws};
            return true;
        }
        if (!(w.ne[1] == 1 && w.ne[2] == 1 && cn.ip[0] == 1 && cn.ip[1] == 0) || cn.ne[0] % 8 || x.ne[0] % 64) return false;
        // the projection on the small map (where the residual unit in front of the resize has not applied it already)
        const int bt = cn.n_src == 3 ? cn.src[2] : -1, cout = (int)cn.ne[0];
        const int64_t M = (int64_t)B * hs * ws;
        int sbuf;
        if (auto it = preprojected.find(t); it != preprojected.end()) sbuf = it->second;
        else {
            packed_operand p = pack_rows(cn.src[1], bt, cout);
            sbuf = new_buffer((size_t)M * cout * 2);
            const int xbuf = buf_of(up.src[0]);
            vx_gemm_args a;
            memset(&a, 0, sizeof a);
            a.lda = p.K; a.W = p.w; a.bias = p.bias; a.M = (int)M; a.N = p.N; a.K = p.K; a.n_valid = cout; a.ldo = cout; a.epi = VX_EPI_F16;
            auto xp = ptr(xbuf), sp0 = ptr(sbuf);
            std::string who = g.nodes[cn.src[1]].name;
            emit("gemm(conv1x1 before its resize) M=" + std::to_string(M) + " N=" + std::to_string(cout) + " K=" + std::to_string(p.k_real) + " <- " + who, {xbuf}, {sbuf}, [=](void* st) {
                vx_gemm_args r = a;
                r.A = xp(); r.out = sp0();
                VX(vx_gemm_f16(&r, st));
            });
            tag(group_of(who, cn, 1).c_str(), 2.0 * M * (double)cout * p.k_real, (double)M * (p.k_real + cout) * 2);
        }
        auto sp = ptr(sbuf);
        skip[t] = 1;
        skip[c] = 1;
        // ... and its resize: by the next conv's loader where that is the only reader, else by the resize kernel into the conv node's buffer
        const int c2 = cn.is_output ? -1 : sole_consumer(c);
        if (c2 >= 0 && g.nodes[c2].op == gop_conv_2d && g.nodes[c2].src[0] == c) {
            graph_node const& w2 = g.nodes[g.nodes[c2].src[1]];
            if (w2.ne[1] == 3 && w2.ne[2] == 3 && loader_can(g.nodes[c2], w2, cout)) {
                resized_in[c] = {sbuf, hs, ws};
                return true;
            }
        }
        materialise(c);
        const int obuf = cn.buffer;
        auto op = ptr(obuf);
        char d[128];
        snprintf(d, sizeof d, "bilinear_ac %dx%d -> %dx%d C=%d", ws, hs, W, H, cout);
        emit(d, {sbuf}, {obuf}, [=](void* st) { VX(vx_bilinear_ac_f16(sp(), op(), B, hs, ws, cout, H, W, st)); });
        tag("bilinear", 0, (double)B * ((double)hs * ws + (double)H * W) * cout * 2);
        return true;
    }

    void patch_embed(int t) {
        graph_node& n = g.nodes[t];
        graph_node const& x = g.nodes[n.src[0]];
        const int ps = (int)n.ip[0], C = (int)x.ne[0], W = (int)x.ne[1], H = (int)x.ne[2], B = (int)x.ne[3];
        packed_operand p = pack_rows(n.src[1], n.n_src == 3 ? n.src[2] : -1, (int)n.ne[0]);
        const int64_t M = (int64_t)B * (H / ps) * (W / ps);
        const int xbuf = buf_of(n.src[0]), pbuf = new_buffer((size_t)M * p.K * 2);
        std::string who = n.name.empty() ? g.nodes[n.src[1]].name : n.name;
        auto xp = ptr(xbuf), pp = ptr(pbuf);
        const int Kp = p.K;
        emit("im2col_patches " + std::to_string(ps) + "x" + std::to_string(ps) + " M=" + std::to_string(M) + " <- " + who, {xbuf}, {pbuf},
             [=](void* st) { VX(vx_im2col_patches_f32(reinterpret_cast<const float*>(xp()), pp(), B, H, W, C, ps, Kp, st)); });
        if (n.ne[0] % 8) throw except("patch_embed %s: %lld output channels", who.c_str(), (long long)n.ne[0]);
        materialise(t);
        const int obuf = n.buffer;
        vx_gemm_args a;
        memset(&a, 0, sizeof a);
        a.lda = p.K; a.W = p.w; a.bias = p.bias; a.M = (int)M; a.N = p.N; a.K = p.K; a.n_valid = (int)n.ne[0]; a.ldo = n.ne[0]; a.epi = VX_EPI_F16;
        auto op = ptr(obuf);
        emit("gemm M=" + std::to_string(M) + " N=" + std::to_string(n.ne[0]) + " K=" + std::to_string(p.k_real) + " <- " + who, {pbuf}, {obuf}, [=](void* st) {
            vx_gemm_args r = a;
            r.A = pp();
            r.out = op();
            VX(vx_gemm_f16(&r, st));
        });
    }

    // q, k, v of an attention that are views of three linears over ONE input, read by nothing else: the three products become one GEMM
    // whose epilogue writes head-major q (scaled), k, v (dino.cpp:59-70 as one launch instead of three products + three permutes)
    std::map<int, int> qkv_first;                 // first of the three linears (node order) -> attention node
    std::map<int, std::array<int, 3>> qkv_lin;    // attention node -> its three linears
    std::map<int, std::array<int, 3>> qkv_bufs;   // attention node -> head-major buffers, once the fused launch is emitted
    void find_qkv_groups() {
        for (int t = 0; t < (int)g.nodes.size(); ++t) {
            graph_node const& n = g.nodes[t];
            if (!needed[t] || n.op != gop_attention) continue;
            std::array<int, 3> lin{-1, -1, -1};
            bool ok = g.nodes[n.src[0]].ne[0] == 64 && g.nodes[n.src[0]].ne[2] == g.nodes[n.src[1]].ne[2];
            for (int i = 0; i < 3 && ok; ++i) {
                int v = n.src[i];
                while (ok && (g.nodes[v].op == gop_reshape || g.nodes[v].op == gop_cont)) { // single-reader views only
                    ok = uses[v] == 1 && !g.nodes[v].is_output;
                    v = g.nodes[v].src[0];
                }
                ok = ok && g.nodes[v].op == gop_linear && uses[v] == 1 && !g.nodes[v].is_output;
                lin[i] = v;
            }
            ok = ok && lin[0] != lin[1] && lin[1] != lin[2] && lin[0] != lin[2];
            for (int i = 1; i < 3 && ok; ++i)
                ok = g.nodes[lin[i]].src[0] == g.nodes[lin[0]].src[0] && g.nodes[g.nodes[lin[i]].src[1]].ne[0] == g.nodes[g.nodes[lin[0]].src[1]].ne[0] &&
                     g.nodes[g.nodes[lin[i]].src[1]].ne[1] == g.nodes[g.nodes[lin[0]].src[1]].ne[1];
            ok = ok && g.nodes[g.nodes[lin[0]].src[1]].ne[0] % 64 == 0 && g.nodes[g.nodes[lin[0]].src[1]].ne[1] == g.nodes[n.src[0]].ne[0] * g.nodes[n.src[0]].ne[1];
            if (!ok) continue;
            qkv_lin[t] = lin;
            qkv_first[std::min({lin[0], lin[1], lin[2]})] = t;
        }
    }
    void fused_qkv(int att) {
        graph_node const& n = g.nodes[att];
        graph_node const& q = g.nodes[n.src[0]];
        const int64_t H = q.ne[1], T = q.ne[2], B = q.ne[3];
        auto const& lin = qkv_lin[att];
        packed_operand p = pack_qkv(lin.data());
        const int xs = g.nodes[lin[0]].src[0];
        const int xbuf = buf_of(xs);
        std::array<int, 3> hb;
        for (int i = 0; i < 3; ++i) hb[i] = new_buffer((size_t)(64 * H * T * B) * 2);
        for (int i = 0; i < 3; ++i) { // the linears and their views are computed by this launch
            skip[lin[i]] = 1;
            for (int v = n.src[i]; v != lin[i]; v = g.nodes[v].src[0]) skip[v] = 1;
        }
        vx_gemm_args a;
        memset(&a, 0, sizeof a);
        a.lda = p.K; a.W = p.w; a.bias = p.bias; a.M = (int)(B * T); a.N = p.N; a.K = p.K; a.n_valid = p.N;
        a.epi = VX_EPI_QKV;
        a.qkv_T = (int)T; a.qkv_H = (int)H;
        a.q_scale = n.fp[0] * 1.4426950408889634f; // the attention kernel works in the exp2 domain (VX_ATTN_Q_SCALE)
        auto xp = ptr(xbuf), qp = ptr(hb[0]), kp = ptr(hb[1]), vp = ptr(hb[2]);
        char d[200];
        snprintf(d, sizeof d, "gemm[qkv heads-major] M=%d N=%d K=%d <- %s", a.M, p.N, p.k_real, g.nodes[g.nodes[lin[0]].src[1]].name.c_str());
        emit(d, {xbuf}, {hb[0], hb[1], hb[2]}, [=](void* st) {
            vx_gemm_args r = a;
            r.A = xp(); r.q = qp(); r.k = kp(); r.vt = vp();
            VX(vx_gemm_f16(&r, st));
        });
        tag("gemm_qkv", 2.0 * a.M * (double)p.N * p.k_real, (double)a.M * (p.k_real + p.N) * 2);
        qkv_bufs[att] = hb;
    }

    void attention(int t) {
        graph_node& n = g.nodes[t];
        graph_node const &q = g.nodes[n.src[0]], &k = g.nodes[n.src[1]];
        const int64_t hd = q.ne[0], H = q.ne[1], Tq = q.ne[2], Tk = k.ne[2], B = q.ne[3];
        if (hd != 64 || Tq != Tk) throw except("attention: head_dim %lld, %lld queries on %lld keys; the fused kernel is built for head_dim 64 self-attention", (long long)hd, (long long)Tq, (long long)Tk);
        const float q_scale = n.fp[0] * 1.4426950408889634f; // the kernel works in the exp2 domain (VX_ATTN_Q_SCALE)
        int hb[3];
        if (auto it = qkv_bufs.find(t); it != qkv_bufs.end()) {
            for (int i = 0; i < 3; ++i) hb[i] = it->second[i];
        } else
        for (int i = 0; i < 3; ++i) { // [hd, H, T, B] -> head-major [hd, T, H, B]
            const int sb = buf_of(n.src[i]);
            hb[i] = new_buffer((size_t)(hd * H * Tq * B) * 2);
            auto sp = ptr(sb), dp = ptr(hb[i]);
            const float s = i == 0 ? q_scale : 1.0f;
            emit(std::string("heads_major ") + "qkv"[i] + (i == 0 ? " (scaled)" : ""), {sb}, {hb[i]}, [=](void* st) {
                const int64_t ne[4] = {hd, Tq, H, B}, ss[4] = {1, H * hd, hd, Tq * H * hd}, ds[4] = {1, hd, Tq * hd, H * Tq * hd};
                VX(vx_copy_strided_f16(sp(), dp(), ne, ss, ds, s, st));
            });
        }
        materialise(t);
        const int obuf = n.buffer;
        auto qp = ptr(hb[0]), kp = ptr(hb[1]), vp = ptr(hb[2]), op = ptr(obuf);
        char d[128];
        snprintf(d, sizeof d, "attention B=%lld heads=%lld T=%lld", (long long)B, (long long)H, (long long)Tq);
        emit(d, {hb[0], hb[1], hb[2]}, {obuf}, [=](void* st) { VX(vx_attention_f16(qp(), kp(), vp(), op(), (int)B, (int)H, (int)Tq, st)); });
        tag("attention", 4.0 * B * H * (double)Tq * Tq * 64, (double)B * H * Tq * 64 * 2 * 4);
    }


    // ---- fused model group 1: the DINOv2 encoder (dino.cpp:10-110) -----------------------------------------------------------------------
    // prepare_tokens (patch_embed -> reshape -> concat with the cls constant -> + position constant) followed by a chain of dino::layer
    // groups  x -> LN1 -> q|k|v -> attention -> out-proj * lambda1 + x -> LN2 -> fc1 -> gelu -> fc2 * lambda2 + x1 [-> final LN taps]
    // at the block kernel's widths (384 / 1536 / heads of 64) lowers to what csrc/kernels_block16.hip was written for: the residual stream
    // as ONE f32 buffer, per layer one attention launch and one token-stationary block launch (out-proj .. fc2 of layer i, the taps' final
    // LayerNorm, LN1 + QKV of layer i + 1), the first layer's LN1 + QKV as the kernel's QKV-only instance. Every interior node must be
    // read inside the group only; the token tensor and the layer outputs may be graph outputs (they are copied out of the stream as f32).
    struct enc_layer {
        int ln1 = -1, lin[3] = {-1, -1, -1}, att = -1, o = -1, s1 = -1, x1 = -1, ln2 = -1, f1 = -1, ge = -1, f2 = -1, s2 = -1, x2 = -1;
        int lam1 = -1, lam2 = -1;
        std::vector<int> taps;
    };
    struct enc_chain {
        bool ok = false;
        int img = -1, u8 = -1, pe = -1, resh = -1, cat = -1, tok = -1, cls = -1, pos = -1, first = -1;
        std::vector<int> views;
        std::vector<enc_layer> layers;
    } enc;

    int through_views(int v, std::vector<int>* seen = nullptr) const { // the producer behind single-reader reshape / cont views
        while (g.nodes[v].op == gop_reshape || g.nodes[v].op == gop_cont) {
            if (uses[v] != 1 || g.nodes[v].is_output) return -1;
            if (seen) seen->push_back(v);
            v = g.nodes[v].src[0];
        }
        return v;
    }
    bool match_layer(int att, enc_layer& L) const {
        auto it = qkv_lin.find(att);
        if (it == qkv_lin.end()) return false;
        for (int i = 0; i < 3; ++i) L.lin[i] = it->second[i];
        L.att = att;
        L.ln1 = g.nodes[L.lin[0]].src[0];
        graph_node const& ln1 = g.nodes[L.ln1];
        if (ln1.op != gop_layer_norm || uses[L.ln1] != 3 || ln1.is_output || ln1.ne[0] != 384) return false;
        if (g.nodes[att].ne[0] != 384 || g.nodes[att].is_output) return false;
        auto sole = [&](int t, int op) { const int c = sole_consumer(t); return c >= 0 && g.nodes[c].op == op ? c : -1; };
        auto scaled = [&](int t, int& lam) { // mul by a 384-element constant behind t
            const int c = sole(t, gop_mul);
            if (c < 0 || root(g.nodes[c].src[0]) != t || !g.nodes[g.nodes[c].src[1]].constant || g.nodes[g.nodes[c].src[1]].n_elements() != 384) return -1;
            lam = g.nodes[c].src[1];
            return c;
        };
        auto residual = [&](int t, int x) { // add(x, t), either order
            const int c = sole(t, gop_add);
            if (c < 0) return -1;
            const int a = g.nodes[c].src[0], b = g.nodes[c].src[1];
            return ((a == x && b == t) || (a == t && b == x)) ? c : -1;
        };
        const int x = ln1.src[0];
        if ((L.o = sole(att, gop_linear)) < 0 || g.nodes[L.o].src[0] != att || g.nodes[L.o].ne[0] != 384) return false;
        if ((L.s1 = scaled(L.o, L.lam1)) < 0 || (L.x1 = residual(L.s1, x)) < 0) return false;
        if (uses[L.x1] != 2 || g.nodes[L.x1].is_output) return false;
        for (int c : consumers[L.x1])
            if (g.nodes[c].op == gop_layer_norm && g.nodes[c].src[0] == L.x1) L.ln2 = c;
        if (L.ln2 < 0 || g.nodes[L.ln2].is_output || g.nodes[L.ln2].fp[0] != ln1.fp[0]) return false;
        if ((L.f1 = sole(L.ln2, gop_linear)) < 0 || g.nodes[L.f1].ne[0] != 1536 || (L.ge = sole(L.f1, gop_gelu)) < 0) return false;
        if ((L.f2 = sole(L.ge, gop_linear)) < 0 || g.nodes[L.f2].ne[0] != 384) return false;
        if ((L.s2 = scaled(L.f2, L.lam2)) < 0 || (L.x2 = residual(L.s2, L.x1)) < 0) return false;
        for (int t : {L.o, L.s1, L.f1, L.ge, L.f2, L.s2})
            if (g.nodes[t].is_output) return false;
        return true;
    }
    void find_encoder_chain() {
        if (!g.fused_models || !vx_dino_block_supported(384, 1536, 64)) return;
        std::vector<enc_layer> found;
        for (auto const& kv : qkv_lin) { // std::map: attention nodes in node order = layer order
            enc_layer L;
            if (match_layer(kv.first, L)) found.push_back(L);
        }
        if (found.empty()) return;
        // a chain: layer l + 1 normalises what layer l produced; x2's other readers are the final-LayerNorm taps
        std::vector<enc_layer> chain;
        for (auto& L : found) {
            const int x = g.nodes[L.ln1].src[0];
            if (!chain.empty() && x != chain.back().x2) return; // two encoders / a branch: not this pattern
            chain.push_back(L);
        }
        for (size_t l = 0; l < chain.size(); ++l) {
            enc_layer& L = chain[l];
            int known = 0;
            for (int c : consumers[L.x2]) {
                if (l + 1 < chain.size() && (c == chain[l + 1].ln1 || c == chain[l + 1].x1)) { ++known; continue; }
                if (g.nodes[c].op == gop_layer_norm && g.nodes[c].src[0] == L.x2) { L.taps.push_back(c); ++known; continue; }
                return;
            }
            if (known != uses[L.x2]) return;
        }
        int tap_w = -1, tap_b = -1;
        float eps = g.nodes[chain[0].ln1].fp[0];
        for (auto& L : chain) {
            if (g.nodes[L.ln1].fp[0] != eps) return;
            for (int t : L.taps) {
                if (tap_w < 0) { tap_w = g.nodes[t].src[1]; tap_b = g.nodes[t].src[2]; }
                if (g.nodes[t].src[1] != tap_w || g.nodes[t].src[2] != tap_b || g.nodes[t].fp[0] != eps) return; // one shared final LayerNorm
            }
        }
        // prepare_tokens in front of the first layer
        enc_chain e;
        e.tok = g.nodes[chain[0].ln1].src[0];
        graph_node const& tok = g.nodes[e.tok];
        if (tok.op != gop_add || uses[e.tok] != 2 || !g.nodes[tok.src[1]].constant) return;
        e.pos = tok.src[1];
        e.cat = tok.src[0];
        graph_node const& cat = g.nodes[e.cat];
        const int64_t D = tok.ne[0], T = tok.ne[1], B = tok.ne[2];
        if (cat.op != gop_concat || cat.ip[0] != 1 || sole_consumer(e.cat) != e.tok || g.nodes[e.pos].n_elements() != D * T || tok.ne[3] != 1) return;
        if (!g.nodes[cat.src[0]].constant || g.nodes[cat.src[0]].ne[1] != 1) return;
        e.cls = cat.src[0];
        if (sole_consumer(cat.src[1]) != e.cat) return;
        e.pe = through_views(cat.src[1], &e.views);
        if (e.pe < 0 || g.nodes[e.pe].op != gop_patch_embed || uses[e.pe] != 1 || g.nodes[e.pe].is_output) return;
        graph_node const& pe = g.nodes[e.pe];
        if (pe.ne[0] != D || pe.ne[1] * pe.ne[2] + 1 != T || pe.ne[3] != B) return;
        e.img = pe.src[0];
        if (g.nodes[e.img].op == gop_image_u8_to_f32 && sole_consumer(e.img) == e.pe) { e.u8 = g.nodes[e.img].src[0]; }
        e.layers = std::move(chain);
        e.first = e.u8 >= 0 ? e.img : e.pe;
        e.ok = true;
        enc = std::move(e);
        for (auto const& L : enc.layers) qkv_first.erase(std::min({L.lin[0], L.lin[1], L.lin[2]})); // these attentions are not lowered one by one
    }

    float* vec_cached(int key_node, int role, std::string const& with, std::function<std::vector<float>()> make_host) {
        return static_cast<float*>(cached(key_node, role, [&](bool st) { std::vector<float> v = make_host(); return upload(v.data(), v.size() * 4, st); }, with));
    }
    const float* values_or_null(int t) const { return t >= 0 ? g.nodes[t].values() : nullptr; }
    static void put(std::vector<float>& v, const float* src, size_t n) { // n floats (zeros if the tensor is absent)
        if (src) v.insert(v.end(), src, src + n); else v.insert(v.end(), n, 0.0f);
    }
    std::string wname(int t) const { return t >= 0 ? (g.nodes[t].op == gop_weight && !g.nodes[t].name.empty() ? g.nodes[t].name : "#" + std::to_string(t)) : std::string("-"); }

    void emit_encoder() {
        enc_chain const& e = enc;
        graph_node const& tok = g.nodes[e.tok];
        graph_node const& pe = g.nodes[e.pe];
        const int D = 384, HID = 1536, NH = 6;
        const int T = (int)tok.ne[1], B = (int)tok.ne[2], Pn = T - 1;
        const long M = (long)B * T, MP = (long)B * Pn;
        const int ps = (int)pe.ip[0], Wimg = (int)g.nodes[e.img].ne[1], Himg = (int)g.nodes[e.img].ne[2];
        const float eps = g.nodes[enc.layers[0].ln1].fp[0];
        auto bias_of = [&](int lin) { return g.nodes[lin].n_src == 3 ? g.nodes[lin].src[2] : -1; };
        std::vector<int> all = {e.pe, e.cat, e.tok};
        if (e.u8 >= 0) all.push_back(e.img);
        all.insert(all.end(), e.views.begin(), e.views.end());

        // ---- tokens: patches (u8 -> f16 with the normalisation folded in, or from the f32 image), cls rows, patch GEMM whose epilogue adds
        // bias + position embedding and writes f32 token rows
        packed_operand p = pack_rows(pe.src[1], pe.n_src == 3 ? pe.src[2] : -1, D);
        const int Kp = p.K;
        const int pbuf = new_buffer((size_t)MP * Kp * 2), xbuf = new_buffer((size_t)M * D * 4);
        auto pp = ptr(pbuf), xp = ptr(xbuf);
        if (e.u8 >= 0) {
            const int ibuf = buf_of(e.u8);
            auto ip = ptr(ibuf);
            graph_node const& im = g.nodes[e.img];
            std::array<float, 3> mean{im.fp[0], im.fp[1], im.fp[2]}, istd{im.fp[3], im.fp[4], im.fp[5]};
            emit("preprocess_patches " + std::to_string(ps) + "x" + std::to_string(ps) + " M=" + std::to_string(MP), {ibuf}, {pbuf}, [=](void* st) {
                VX(vx_preprocess_patches(reinterpret_cast<const uint8_t*>(ip()), pp(), B, Himg, Wimg, ps, Kp, mean.data(), istd.data(), st));
            });
            tag("preprocess", 0, (double)B * Himg * Wimg * 3 + (double)MP * Kp * 2);
        } else {
            const int ibuf = buf_of(e.img);
            auto ip = ptr(ibuf);
            const int C = (int)g.nodes[e.img].ne[0];
            emit("im2col_patches " + std::to_string(ps) + "x" + std::to_string(ps) + " M=" + std::to_string(MP), {ibuf}, {pbuf},
                 [=](void* st) { VX(vx_im2col_patches_f32(reinterpret_cast<const float*>(ip()), pp(), B, Himg, Wimg, C, ps, Kp, st)); });
            tag("preprocess", 0, (double)B * Himg * Wimg * C * 4 + (double)MP * Kp * 2);
        }
        const float* cls = const_f32(e.cls);
        const float* pos = const_f32(e.pos);
        emit("cls_rows B=" + std::to_string(B), {}, {xbuf}, [=](void* st) { VX(vx_write_cls_rows(reinterpret_cast<float*>(xp()), cls, pos, B, T, D, st)); });
        tag("preprocess");
        {
            vx_gemm_args a;
            memset(&a, 0, sizeof a);
            a.lda = Kp; a.W = p.w; a.bias = p.bias; a.M = (int)MP; a.N = p.N; a.K = p.K; a.n_valid = D;
            a.epi = VX_EPI_TOKENS; a.ldo = D; a.pos = pos; a.tokens_P = Pn;
            emit("gemm[tokens f32: + bias + pos] M=" + std::to_string(MP) + " N=384 K=" + std::to_string(p.k_real) + " <- " + g.nodes[pe.src[1]].name, {pbuf}, {xbuf}, [=](void* st) {
                vx_gemm_args r = a;
                r.A = pp(); r.out = xp();
                VX(vx_gemm_f16(&r, st));
            });
            tag("patch_embed", 2.0 * MP * D * p.k_real, (double)MP * Kp * 2 + (double)M * D * 4);
        }
        auto copy_out = [&](int node, const char* what) { // a graph output inside the group: the residual stream as it stands, f32
            graph_node& n = g.nodes[node];
            n.dtype = gdt_f32;
            materialise(node);
            const int ob = n.buffer;
            auto op = ptr(ob);
            const size_t bytes = (size_t)M * D * 4;
            emit(std::string("copy_f32 ") + what, {xbuf}, {ob}, [=](void* st) { VX(vx_memcpy_d2d(op(), xp(), bytes, st)); });
            tag("capture");
        };
        if (tok.is_output) copy_out(e.tok, "tokens");

        // ---- per layer: attention + one block launch
        const int qb = new_buffer((size_t)M * D * 2), kb = new_buffer((size_t)M * D * 2), vb = new_buffer((size_t)M * D * 2), ab = new_buffer((size_t)M * D * 2);
        auto qp = ptr(qb), kp = ptr(kb), vp = ptr(vb), ap = ptr(ab);
        const float q_scale = g.nodes[enc.layers[0].att].fp[0] * 1.4426950408889634f; // the attention kernel works in the exp2 domain
        auto pack_mlp = [&](enc_layer const& L) -> void* {
            const int wo = g.nodes[L.o].src[1], w1 = g.nodes[L.f1].src[1], w2 = g.nodes[L.f2].src[1];
            return cached(wo, 40, [&](bool st) {
                // LayerScale folded into the two residual products (W' = f16(lambda[n] * W[n, :])): the residual stream stays in the accumulators
                std::vector<uint16_t> a((size_t)D * D), b((size_t)HID * D), c((size_t)D * HID), out(vx_dino_block_mlp_bytes() / 2);
                const float *ho = g.nodes[wo].values(), *h1 = g.nodes[w1].values(), *h2 = g.nodes[w2].values(), *l1 = g.nodes[L.lam1].values(), *l2 = g.nodes[L.lam2].values();
                for (int n = 0; n < D; ++n) {
                    for (int k = 0; k < D; ++k) a[(size_t)n * D + k] = f32_to_f16(f16_to_f32(f32_to_f16(ho[(size_t)n * D + k])) * l1[n]);
                    for (int k = 0; k < HID; ++k) c[(size_t)n * HID + k] = f32_to_f16(f16_to_f32(f32_to_f16(h2[(size_t)n * HID + k])) * l2[n]);
                }
                for (size_t i = 0; i < b.size(); ++i) b[i] = f32_to_f16(h1[i]);
                VX(vx_dino_block16_pack_mlp(a.data(), b.data(), c.data(), out.data()));
                return upload(out.data(), out.size() * 2, st);
            }, wname(w1) + "|" + wname(w2) + "|*" + wname(L.lam1) + "|*" + wname(L.lam2));
        };
        auto pack_qkv_block = [&](enc_layer const& L) -> void* {
            return cached(g.nodes[L.lin[0]].src[1], 41, [&](bool st) {
                std::vector<uint16_t> rows((size_t)3 * D * D), out(vx_dino_block_qkv_bytes() / 2);
                for (int i = 0; i < 3; ++i) {
                    const float* h = g.nodes[g.nodes[L.lin[i]].src[1]].values();
                    for (size_t j = 0; j < (size_t)D * D; ++j) rows[(size_t)i * D * D + j] = f32_to_f16(h[j]);
                }
                VX(vx_dino_block16_pack_qkv(rows.data(), out.data()));
                return upload(out.data(), out.size() * 2, st);
            }, wname(g.nodes[L.lin[1]].src[1]) + "|" + wname(g.nodes[L.lin[2]].src[1]));
        };
        auto vec_mlp = [&](enc_layer const& L) {
            const int bo = bias_of(L.o), b1 = bias_of(L.f1), b2 = bias_of(L.f2);
            return vec_cached(g.nodes[L.o].src[1], 42, wname(bo) + "|" + wname(b1) + "|" + wname(b2) + "|" + wname(g.nodes[L.ln2].src[1]) + "|*" + wname(L.lam1) + "|*" + wname(L.lam2), [&]() {
                std::vector<float> v;
                const float *l1 = g.nodes[L.lam1].values(), *l2 = g.nodes[L.lam2].values();
                put(v, values_or_null(bo), D);
                for (int n = 0; n < D; ++n) v[n] *= l1[n];
                put(v, l1, D);
                put(v, g.nodes[g.nodes[L.ln2].src[1]].values(), D);
                put(v, g.nodes[g.nodes[L.ln2].src[2]].values(), D);
                put(v, values_or_null(b1), HID);
                const size_t at = v.size();
                put(v, values_or_null(b2), D);
                for (int n = 0; n < D; ++n) v[at + n] *= l2[n];
                put(v, l2, D);
                return v;
            });
        };
        auto vec_qkv = [&](enc_layer const& L) {
            return vec_cached(g.nodes[L.ln1].src[1], 43, wname(g.nodes[L.ln1].src[2]) + "|" + wname(bias_of(L.lin[0])) + "|" + wname(bias_of(L.lin[1])) + "|" + wname(bias_of(L.lin[2])), [&]() {
                std::vector<float> v;
                put(v, g.nodes[g.nodes[L.ln1].src[1]].values(), D);
                put(v, g.nodes[g.nodes[L.ln1].src[2]].values(), D);
                for (int i = 0; i < 3; ++i) put(v, values_or_null(bias_of(L.lin[i])), D);
                return v;
            });
        };
        float* vec_tap = nullptr;
        for (auto const& L : enc.layers)
            if (!L.taps.empty() && !vec_tap) {
                const int t = L.taps[0];
                vec_tap = vec_cached(g.nodes[t].src[1], 44, wname(g.nodes[t].src[2]), [&]() {
                    std::vector<float> v;
                    put(v, g.nodes[g.nodes[t].src[1]].values(), D);
                    put(v, g.nodes[g.nodes[t].src[2]].values(), D);
                    return v;
                });
            }
        auto block = [&](const enc_layer* mlp, const enc_layer* qkv, int feat_buf, const char* group, std::string const& who) {
            vx_dino_block_args a;
            memset(&a, 0, sizeof a);
            a.M = (int)M; a.T = T; a.H = NH; a.q_scale = q_scale; a.eps = eps;
            double flops = 0, bytes = (double)M * D * 4;
            std::vector<int> reads = {xbuf}, writes = {xbuf};
            if (mlp) {
                a.w_mlp = pack_mlp(*mlp); a.vec_mlp = vec_mlp(*mlp);
                flops += 2.0 * M * D * (D + 2.0 * HID);
                bytes += (double)M * D * (2 + 4);
                reads.push_back(ab);
            }
            if (qkv) {
                a.w_qkv = pack_qkv_block(*qkv); a.vec_qkv = vec_qkv(*qkv);
                flops += 2.0 * M * D * 3.0 * D;
                bytes += (double)M * D * 2 * 3;
                writes.insert(writes.end(), {qb, kb, vb});
            }
            std::function<char*()> fp;
            if (feat_buf >= 0) { a.vec_tap = vec_tap; fp = ptr(feat_buf); bytes += (double)M * D * 2; writes.push_back(feat_buf); }
            const bool has_mlp = mlp != nullptr, has_qkv = qkv != nullptr;
            emit(std::string("dino_block[") + (has_mlp ? "out-proj + mlp" : "") + (feat_buf >= 0 ? " + tap" : "") + (has_qkv ? (has_mlp ? " + next ln1 + qkv" : "ln1 + qkv") : "") + "] M=" + std::to_string(M) + " <- " + who,
                 reads, writes, [=](void* st) {
                     vx_dino_block_args r = a;
                     r.x = reinterpret_cast<float*>(xp());
                     if (has_mlp) r.att = ap();
                     if (has_qkv) { r.q = qp(); r.k = kp(); r.v = vp(); }
                     if (fp) r.feat = fp();
                     VX(vx_dino_block16_f16(&r, st));
                 });
            tag(group, flops, bytes);
        };
        block(nullptr, &enc.layers[0], -1, "block_qkv0", g.nodes[g.nodes[enc.layers[0].lin[0]].src[1]].name);
        for (size_t l = 0; l < enc.layers.size(); ++l) {
            enc_layer const& L = enc.layers[l];
            emit("attention B=" + std::to_string(B) + " heads=" + std::to_string(NH) + " T=" + std::to_string(T), {qb, kb, vb}, {ab},
                 [=](void* st) { VX(vx_attention_f16(qp(), kp(), vp(), ap(), B, NH, T, st)); });
            tag("attention", 4.0 * B * NH * (double)T * T * 64, (double)M * D * 2 * 4);
            int feat = -1;
            for (int t : L.taps) {
                materialise(t);
                if (feat < 0) feat = g.nodes[t].buffer;
            }
            block(&L, l + 1 < enc.layers.size() ? &enc.layers[l + 1] : nullptr, feat, "block", g.nodes[g.nodes[L.o].src[1]].name);
            for (size_t i = 1; i < L.taps.size(); ++i) { // further taps that name this layer: the same rows
                const int ob = g.nodes[L.taps[i]].buffer;
                auto fp = ptr(feat), op = ptr(ob);
                const size_t bytes = (size_t)M * D * 2;
                emit("copy tap", {feat}, {ob}, [=](void* st) { VX(vx_memcpy_d2d(op(), fp(), bytes, st)); });
                tag("block");
            }
            if (g.nodes[L.x2].is_output) copy_out(L.x2, "layer output");
            for (int t : {L.ln1, L.lin[0], L.lin[1], L.lin[2], L.att, L.o, L.s1, L.x1, L.ln2, L.f1, L.ge, L.f2, L.s2, L.x2}) all.push_back(t);
            for (int i = 0; i < 3; ++i)
                for (int v = g.nodes[L.att].src[i]; v != L.lin[i]; v = g.nodes[v].src[0]) all.push_back(v);
            for (int t : L.taps) all.push_back(t);
        }
        for (int t : all) skip[t] = 1;
    }

    void copy_op(int t) {
        graph_node& n = g.nodes[t];
        materialise(t);
        const int obuf = n.buffer;
        auto op = ptr(obuf);
        auto strides = [](const int64_t ne[4], int64_t s[4]) { s[0] = 1; s[1] = ne[0]; s[2] = ne[0] * ne[1]; s[3] = ne[0] * ne[1] * ne[2]; };
        if (n.op == gop_slice || n.op == gop_repeat) {
            graph_node const& x = g.nodes[n.src[0]];
            const int xbuf = buf_of(n.src[0]);
            auto xp = ptr(xbuf);
            int64_t xs[4], ds[4], ne[4], ss[4], off = 0;
            strides(x.ne, xs);
            strides(n.ne, ds);
            for (int d = 0; d < 4; ++d) {
                ne[d] = n.ne[d];
                if (n.op == gop_slice) { off += n.ip[3 * d] * xs[d]; ss[d] = xs[d] * n.ip[3 * d + 2]; }
                else ss[d] = x.ne[d] == 1 && n.ne[d] > 1 ? 0 : xs[d];
            }
            std::array<int64_t, 4> A{ne[0], ne[1], ne[2], ne[3]}, S{ss[0], ss[1], ss[2], ss[3]}, D{ds[0], ds[1], ds[2], ds[3]};
            emit(std::string(graph_op_name(n.op)) + " " + shape_str(x.ne) + " -> " + shape_str(n.ne), {xbuf}, {obuf},
                 [=](void* st) { VX(vx_copy_strided_f16(xp() + off * 2, op(), A.data(), S.data(), D.data(), 1.0f, st)); });
            return;
        }
        // concat: each source into its block of the destination
        const int dim = (int)n.ip[0];
        int64_t ds[4];
        strides(n.ne, ds);
        int64_t at = 0;
        for (int i = 0; i < 2; ++i) {
            graph_node const& x = g.nodes[n.src[i]];
            int64_t xs[4];
            strides(x.ne, xs);
            std::array<int64_t, 4> A{x.ne[0], x.ne[1], x.ne[2], x.ne[3]}, S{xs[0], xs[1], xs[2], xs[3]}, D{ds[0], ds[1], ds[2], ds[3]};
            const int64_t off = at * ds[dim];
            at += x.ne[dim];
            std::string d = "concat part " + std::to_string(i) + " " + shape_str(x.ne) + " -> " + shape_str(n.ne);
            if (x.constant) {
                void* cp = const_f16(n.src[i]);
                emit(d + " (constant)", {}, {obuf}, [=](void* st) { VX(vx_copy_strided_f16(cp, op() + off * 2, A.data(), S.data(), D.data(), 1.0f, st)); });
            } else {
                const int xbuf = buf_of(n.src[i]);
                auto xp = ptr(xbuf);
                emit(d, {xbuf}, {obuf}, [=](void* st) { VX(vx_copy_strided_f16(xp(), op() + off * 2, A.data(), S.data(), D.data(), 1.0f, st)); });
            }
        }
    }

    void run() {
        const int N = (int)g.nodes.size();
        needed.assign(N, 0);
        skip.assign(N, 0);
        uses.assign(N, 0);
        relu_on_load.assign(N, 0);
        consumers.assign(N, {});
        bool any_out = false;
        for (int t = N - 1; t >= 0; --t) {
            graph_node const& n = g.nodes[t];
            if (n.is_output) { needed[t] = 1; any_out = true; }
            if (!needed[t]) continue;
            for (int i = 0; i < n.n_src; ++i) needed[n.src[i]] = 1;
        }
        if (!any_out) throw except("graph_allocate: the graph has no output (compute_graph_output)");
        for (int t = 0; t < N; ++t) {
            graph_node const& n = g.nodes[t];
            if (!needed[t] || n.constant) continue;
            for (int i = 0; i < n.n_src; ++i) {
                uses[n.src[i]]++;
                consumers[n.src[i]].push_back(t);
            }
        }
        find_qkv_groups();
        find_encoder_chain();
        for (int t = 0; t < N; ++t) {
            graph_node& n = g.nodes[t];
            if (enc.ok && t == enc.first) emit_encoder();
            if (auto it = qkv_first.find(t); it != qkv_first.end() && !skip[t]) fused_qkv(it->second);
            if (!needed[t] || skip[t]) continue;
            if (n.constant) {
                if (n.is_output) throw except("graph_allocate: output '%s' is a constant", n.name.c_str());
                continue;
            }
            std::string who = n.name;
            switch (n.op) {
                case gop_input: materialise(t); break;
                case gop_reshape:
                case gop_cont:
                    n.alias_of = n.src[0];
                    if (n.is_output) { // an output view keeps its source alive
                        graph_node& r = g.nodes[root(t)];
                        r.is_output = true;
                        if (r.buffer >= 0) g.buffers[r.buffer].persistent = true;
                    }
                    break;
                case gop_linear:
                case gop_conv_2d:
                case gop_conv_transpose_2d: gemm_like(t); break;
                case gop_patch_embed: patch_embed(t); break;
                case gop_attention: attention(t); break;
                case gop_relu: {
                    const int c = sole_consumer(t);
                    if (c >= 0 && g.nodes[c].op == gop_conv_2d && g.nodes[c].src[0] == t && !(g.nodes[g.nodes[c].src[1]].ne[1] == 1 && g.nodes[g.nodes[c].src[1]].ne[2] == 1)) {
                        relu_on_load[c] = 1; // the conv's loader applies it (depth-anything.cpp:17-20)
                        n.alias_of = n.src[0];
                        break;
                    }
                }
                    [[fallthrough]];
                case gop_gelu:
                case gop_leaky_relu:
                case gop_scale: {
                    if (n.dtype != gdt_f16) throw except("%s on the f32 map of a one-channel head that has other readers is not built", graph_op_name(n.op));
                    const int xbuf = buf_of(n.src[0]);
                    materialise(t);
                    const int obuf = n.buffer;
                    auto xp = ptr(xbuf), op = ptr(obuf);
                    const int64_t cnt = n.n_elements();
                    const int uop = n.op == gop_gelu ? 0 : (n.op == gop_relu ? 1 : (n.op == gop_leaky_relu ? 3 : 2));
                    const float s = n.fp[0];
                    emit(std::string(graph_op_name(n.op)) + " n=" + std::to_string(cnt), {xbuf}, {obuf}, [=](void* st) { VX(vx_unary_f16(uop, xp(), op(), cnt, s, st)); });
                } break;
                case gop_add:
                case gop_mul: {
                    graph_node const& b = g.nodes[n.src[1]];
                    if (g.nodes[n.src[0]].constant) throw except("%s: the larger operand is a constant; broadcast of an activation onto it is not built", graph_op_name(n.op));
                    const int abuf = buf_of(n.src[0]);
                    const int64_t period = b.n_elements(), cnt = n.n_elements();
                    const int bop = n.op == gop_add ? 0 : 1;
                    std::vector<int> reads = {abuf};
                    std::function<char*()> bp;
                    float* bc = nullptr;
                    if (b.constant) bc = const_f32(n.src[1]);
                    else { reads.push_back(buf_of(n.src[1])); bp = ptr(buf_of(n.src[1])); }
                    materialise(t);
                    const int obuf = n.buffer;
                    auto ap = ptr(abuf), op = ptr(obuf);
                    emit(std::string(graph_op_name(n.op)) + " n=" + std::to_string(cnt) + " period=" + std::to_string(period) + (bc || !bp ? " (constant)" : ""), reads, {obuf}, [=](void* st) {
                        if (bp) VX(vx_binary_rows(bop, ap(), 0, bp(), 0, period, op(), 0, cnt, st));
                        else VX(vx_binary_rows(bop, ap(), 0, bc, 1, period, op(), 0, cnt, st));
                    });
                } break;
                case gop_layer_norm: {
                    const int xbuf = buf_of(n.src[0]);
                    const int C = (int)n.ne[0];
                    if (C % 8 || C > 512) throw except("layer_norm: %d channels (a multiple of 8, at most 512)", C);
                    float *w = const_f32(n.src[1]), *b = const_f32(n.src[2]);
                    materialise(t);
                    const int obuf = n.buffer;
                    auto xp = ptr(xbuf), op = ptr(obuf);
                    const int64_t rows = n.n_elements() / C;
                    const float eps = n.fp[0];
                    emit("layer_norm rows=" + std::to_string(rows) + " C=" + std::to_string(C) + (who.empty() ? "" : " <- " + who), {xbuf}, {obuf},
                         [=](void* st) { VX(vx_layernorm_f16(xp(), w, b, op(), rows, C, eps, 0, 0, 0, st)); });
                    tag("layernorm", 0, (double)rows * C * 4);
                } break;
                case gop_interpolate: {
                    if (nearest_into_reader(t)) break;
                    if ((n.ip[2] & 255) == 0) { // NEAREST on its own
                        graph_node const& x = g.nodes[n.src[0]];
                        if (x.ne[0] % 8) throw except("interpolate: %lld channels (a multiple of 8)", (long long)x.ne[0]);
                        const int xbuf = buf_of(n.src[0]);
                        materialise(t);
                        const int obuf = n.buffer;
                        auto xp = ptr(xbuf), op = ptr(obuf);
                        const int B = (int)x.ne[3], H = (int)x.ne[2], W = (int)x.ne[1], C = (int)x.ne[0], OH = (int)n.ne[2], OW = (int)n.ne[1];
                        char d[128];
                        snprintf(d, sizeof d, "nearest %dx%d -> %dx%d C=%d", W, H, OW, OH, C);
                        emit(d, {xbuf}, {obuf}, [=](void* st) { VX(vx_nearest_f16(xp(), op(), B, H, W, C, OH, OW, st)); });
                        break;
                    }
                    if (!n.is_output && head_tail(t)) break;
                    if (resize_into_reader(t)) break;
                    graph_node const& x = g.nodes[n.src[0]];
                    if ((n.ip[2] & 255) != 1 || !(n.ip[2] & 256)) throw except("interpolate: mode %lld on activations is not built (bilinear | align_corners is; bicubic on constants)", (long long)n.ip[2]);
                    if (x.ne[0] % 8) throw except("interpolate: %lld channels (a multiple of 8)", (long long)x.ne[0]);
                    const int xbuf = buf_of(n.src[0]);
                    materialise(t);
                    const int obuf = n.buffer;
                    auto xp = ptr(xbuf), op = ptr(obuf);
                    const int B = (int)x.ne[3], H = (int)x.ne[2], W = (int)x.ne[1], C = (int)x.ne[0], OH = (int)n.ne[2], OW = (int)n.ne[1];
                    char d[128];
                    snprintf(d, sizeof d, "bilinear_ac %dx%d -> %dx%d C=%d", W, H, OW, OH, C);
                    emit(d, {xbuf}, {obuf}, [=](void* st) { VX(vx_bilinear_ac_f16(xp(), op(), B, H, W, C, OH, OW, st)); });
                    tag("bilinear", 0, (double)B * ((double)H * W + (double)OH * OW) * C * 2);
                } break;
                case gop_image_u8_to_f32: { // not read by a fused patch embedding: the f32 image itself
                    graph_node const& x = g.nodes[n.src[0]];
                    const int xbuf = buf_of(n.src[0]);
                    materialise(t);
                    const int obuf = n.buffer;
                    auto xp = ptr(xbuf), op = ptr(obuf);
                    const int B = (int)x.ne[3], H = (int)x.ne[2], W = (int)x.ne[1];
                    std::array<float, 3> mean{n.fp[0], n.fp[1], n.fp[2]}, istd{n.fp[3], n.fp[4], n.fp[5]};
                    emit("image_u8_to_f32 " + shape_str(x.ne), {xbuf}, {obuf},
                         [=](void* st) { VX(vx_preprocess_f32(reinterpret_cast<const uint8_t*>(xp()), reinterpret_cast<float*>(op()), B, H, W, mean.data(), istd.data(), st)); });
                    tag("preprocess", 0, (double)B * H * W * 15);
                } break;
                case gop_image_normalize: {
                    graph_node const& x = g.nodes[n.src[0]];
                    const int xbuf = buf_of(n.src[0]);
                    materialise(t);
                    const int obuf = n.buffer, mbuf = new_buffer((size_t)x.ne[3] * 8);
                    auto xp = ptr(xbuf), op = ptr(obuf), mp = ptr(mbuf);
                    const int B = (int)x.ne[3];
                    const int64_t px = x.ne[1] * x.ne[2];
                    emit("image_normalize B=" + std::to_string(B), {xbuf}, {obuf, mbuf}, [=](void* st) {
                        VX(vx_minmax_normalize(reinterpret_cast<const float*>(xp()), reinterpret_cast<float*>(op()), reinterpret_cast<float*>(mp()), B, px, st));
                    });
                    tag("normalize", 0, (double)B * px * 12);
                } break;
                case gop_slice:
                    // rows [b, e) of every image's token block, read by one plain product only (dpt::neck drops the cls token, depth-anything.cpp:50):
                    // the GEMM's A rows are addressed in the source, no copy
                    if (g.fused_models && !n.is_output && plain_gemm_reader(t) >= 0 && n.ip[0] == 0 && n.ne[0] == g.nodes[n.src[0]].ne[0] && n.ne[0] % 64 == 0 && n.ip[5] == 1 &&
                        n.ne[2] == g.nodes[n.src[0]].ne[2] && n.ip[6] == 0 && n.ne[3] == g.nodes[n.src[0]].ne[3] && n.ip[9] == 0 && g.nodes[n.src[0]].ne[3] == 1 &&
                        row_stride.find(root_view(n.src[0])) == row_stride.end()) {
                        sliced_in[t] = {n.src[0], n.ip[3], n.ne[1], g.nodes[n.src[0]].ne[1]};
                        n.alias_of = n.src[0];
                        break;
                    }
                    copy_op(t);
                    break;
                case gop_concat:
                case gop_repeat: copy_op(t); break;
                default: throw except("graph_allocate: %s is not lowered", graph_op_name(n.op));
            }
        }
        if (!placed.empty()) throw except("graph: internal: a plane reserved for tensor %d of a dense block was never written", placed.begin()->first);
        plan_arena();
    }

    // buffer lifetimes from the launch list, then offsets: persistent buffers first, the others first-fit into the holes their
    // predecessors left (ggml_gallocr's job in the reference, ml.cpp:545-552)
    void plan_arena() {
        const int L = (int)g.launches.size();
        for (auto& b : g.buffers) { b.first = L; b.last = -1; }
        for (int i = 0; i < L; ++i) {
            for (int b : io[i].second) { g.buffers[b].first = std::min(g.buffers[b].first, i); g.buffers[b].last = std::max(g.buffers[b].last, i); }
            for (int b : io[i].first) { g.buffers[b].first = std::min(g.buffers[b].first, i); g.buffers[b].last = std::max(g.buffers[b].last, i); }
        }
        for (auto& n : g.nodes)
            if (n.buffer >= 0 && (n.is_output || n.op == gop_input)) g.buffers[n.buffer].persistent = true;
        size_t top = 0;
        g.sum_bytes = 0;
        for (auto& b : g.buffers) {
            g.sum_bytes += b.bytes;
            if (b.persistent) { b.offset = top; top += b.bytes; }
        }
        const size_t base = top;
        struct block { size_t off, bytes; };
        std::vector<block> free_list;
        std::vector<std::vector<int>> starts(L + 1), ends(L + 1);
        for (int i = 0; i < (int)g.buffers.size(); ++i) {
            graph_buffer const& b = g.buffers[i];
            if (b.persistent || b.last < 0) continue;
            starts[b.first].push_back(i);
            ends[b.last].push_back(i);
        }
        size_t high = base;
        for (int i = 0; i < L; ++i) {
            for (int bi : starts[i]) {
                graph_buffer& b = g.buffers[bi];
                int best = -1;
                for (int f = 0; f < (int)free_list.size(); ++f)
                    if (free_list[f].bytes >= b.bytes && (best < 0 || free_list[f].bytes < free_list[best].bytes)) best = f;
                if (best >= 0) {
                    b.offset = free_list[best].off;
                    free_list[best].off += b.bytes;
                    free_list[best].bytes -= b.bytes;
                    if (free_list[best].bytes == 0) free_list.erase(free_list.begin() + best);
                } else if (!free_list.empty() && free_list.back().off + free_list.back().bytes == high) { // grow the block at the top
                    b.offset = free_list.back().off;
                    high = b.offset + b.bytes;
                    free_list.pop_back();
                } else {
                    b.offset = high;
                    high += b.bytes;
                }
            }
            for (int bi : ends[i]) {
                graph_buffer const& b = g.buffers[bi];
                free_list.push_back({b.offset, b.bytes});
                std::sort(free_list.begin(), free_list.end(), [](block const& x, block const& y) { return x.off < y.off; });
                for (size_t f = 0; f + 1 < free_list.size();) {
                    if (free_list[f].off + free_list[f].bytes == free_list[f + 1].off) {
                        free_list[f].bytes += free_list[f + 1].bytes;
                        free_list.erase(free_list.begin() + f + 1);
                    } else ++f;
                }
            }
        }
        g.arena_bytes = high + 65536; // slack behind the last buffer: tile-granular loaders may read past a tail row
    }
};

} // namespace

void graph_allocate(graph& g, backend_device const* dev) {
    if (g.allocated) return;
    g.dev = dev;
    std::optional<device_turn> turn; // the arena, the packed weights and the kernel attributes belong to THIS device; uploads use its stream
    if (dev) {
        turn.emplace(*dev);
        std::lock_guard<std::mutex> lock(g.store->mutex);
        if (g.store->dev && g.store->dev != dev) throw except("compute_graph_allocate: the weights of this graph live on another device");
        g.store->dev = dev;
    }
    lowering low(g);
    low.run();
    if (g.dev) {
        VX(vx_malloc(&g.arena.ptr, g.arena_bytes));
        g.arena.bytes = g.arena_bytes;
        VX(vx_memset(g.arena.ptr, 0, g.arena_bytes, g.dev->stream));
        VX(vx_stream_sync(g.dev->stream));
    }
    g.allocated = true;
}

static void require_device(graph const& g, const char* what) {
    if (!g.allocated) throw except("%s: call compute_graph_allocate first", what);
    if (!g.dev) throw except("%s: this graph was made without a device (planning only)", what);
}

void graph_compute(graph& g) {
    require_device(g, "compute");
    device_turn turn(*g.dev);
    void* st = g.dev->stream;
    if (g.use_hip_graph && g.graph_exec) {
        VX(vx_graph_launch(g.graph_exec, st));
        VX(vx_stream_sync(st));
        return;
    }
    for (auto& l : g.launches) l.run(st);
    VX(vx_stream_sync(st));
    if (g.use_hip_graph) { // the eager pass above set every kernel attribute; now record the same launches once
        VX(vx_graph_begin_capture(st));
        try {
            for (auto& l : g.launches) l.run(st);
        } catch (...) {
            void* dead = nullptr;
            vx_graph_end_capture(st, &dead);
            if (dead) vx_graph_destroy(dead);
            throw;
        }
        VX(vx_graph_end_capture(st, &g.graph_exec));
    }
}

static char* tensor_ptr(graph& g, int t, const char* what) {
    check_tensor(g, t, what);
    int r = t;
    while (g.nodes[r].alias_of >= 0) r = g.nodes[r].alias_of;
    graph_node const& n = g.nodes[r];
    if (n.buffer < 0) throw except("%s: tensor %d (%s) has no storage (a constant, fused away, or not needed by any output)", what, t, graph_op_name(g.nodes[t].op));
    if (!g.buffers[n.buffer].persistent) throw except("%s: tensor %d (%s) is an intermediate whose buffer is recycled; mark it with compute_graph_output", what, t, graph_op_name(g.nodes[t].op));
    return static_cast<char*>(g.arena.ptr) + g.buffers[n.buffer].offset;
}

void graph_tensor_set(graph& g, int t, const void* data, size_t bytes) {
    require_device(g, "transfer_to_backend");
    device_turn turn(*g.dev);
    char* p = tensor_ptr(g, t, "transfer_to_backend");
    if (bytes != g.nodes[t].n_bytes()) throw except("transfer_to_backend: %zu bytes for a tensor of %zu", bytes, g.nodes[t].n_bytes());
    VX(vx_memcpy_h2d(p, data, bytes, g.dev->stream));
    VX(vx_stream_sync(g.dev->stream));
}

void graph_tensor_get(graph& g, int t, void* data, size_t bytes, bool as_f32) {
    require_device(g, "transfer_from_backend");
    device_turn turn(*g.dev);
    char* p = tensor_ptr(g, t, "transfer_from_backend");
    graph_node const& n = g.nodes[t];
    const bool convert = as_f32 && n.dtype == gdt_f16;
    const size_t want = convert ? (size_t)n.n_elements() * 4 : n.n_bytes();
    if (bytes != want) throw except("transfer_from_backend: %zu bytes for a tensor of %zu", bytes, want);
    if (!convert) {
        VX(vx_memcpy_d2h(data, p, bytes, g.dev->stream));
        VX(vx_stream_sync(g.dev->stream));
        return;
    }
    std::vector<uint16_t> h((size_t)n.n_elements());
    VX(vx_memcpy_d2h(h.data(), p, h.size() * 2, g.dev->stream));
    VX(vx_stream_sync(g.dev->stream));
    float* out = static_cast<float*>(data);
    for (size_t i = 0; i < h.size(); ++i) out[i] = f16_to_f32(h[i]);
}

void graph_bind_external(graph& g, int t, void* ptr) {
    if (!g.allocated) throw except("graph_bind_external: call compute_graph_allocate first");
    check_tensor(g, t, "graph_bind_external");
    int r = t;
    while (g.nodes[r].alias_of >= 0) r = g.nodes[r].alias_of;
    graph_node const& n = g.nodes[r];
    if (n.buffer < 0 || !g.buffers[n.buffer].persistent) throw except("graph_bind_external: tensor %d is not an input or output of the graph", t);
    g.buffers[n.buffer].external = ptr;
}
void* graph_tensor_device_ptr(graph& g, int t) {
    require_device(g, "graph_tensor_device_ptr");
    char* p = tensor_ptr(g, t, "graph_tensor_device_ptr");
    int r = t;
    while (g.nodes[r].alias_of >= 0) r = g.nodes[r].alias_of;
    graph_buffer const& b = g.buffers[g.nodes[r].buffer];
    return b.external ? b.external : p;
}

std::string graph_describe(graph const& g) {
    std::string s;
    for (auto const& l : g.launches) { s += l.desc; s += '\n'; }
    char b[256];
    snprintf(b, sizeof b, "launches=%zu nodes=%zu buffers=%zu arena_bytes=%zu unshared_bytes=%zu constant_bytes=%zu\n", g.launches.size(), g.nodes.size(), g.buffers.size(),
             g.arena_bytes, g.sum_bytes, g.const_bytes);
    s += b;
    return s;
}

} // namespace visp
