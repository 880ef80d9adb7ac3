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
                                         "patch_embed", "cont"};

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
        tmp.resize((size_t)n);
        if (t.type == GGML_F32) memcpy(tmp.data(), t.data, (size_t)n * 4);
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
    if (dtype != gdt_f32 && dtype != gdt_f16) throw except("graph_input: dtype %d (0 = f32, 1 = f16)", dtype);
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
    if (n_src < 1 || n_src > 4 || n_ip < 0 || n_ip > 12 || n_fp < 0 || n_fp > 2) throw except("graph_add(%s): bad argument counts", graph_op_name(op));
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
            if (x.dtype != gdt_f32) throw except("patch_embed: the input image tensor is f32 (the tensor the reference uploads)");
            if (ps <= 0 || w.ne[1] != ps || w.ne[2] != ps || w.ne[0] != x.ne[0]) throw except("patch_embed: kernel %s does not match patch size %lld and input %s", shape_str(w.ne).c_str(), (long long)ps, shape_str(x.ne).c_str());
            if (x.ne[1] % ps || x.ne[2] % ps) throw except("patch_embed: extent %lldx%lld is not a multiple of the patch size %lld", (long long)x.ne[1], (long long)x.ne[2], (long long)ps);
            n.ne[0] = w.ne[3];
            n.ne[1] = x.ne[1] / ps;
            n.ne[2] = x.ne[2] / ps;
            n.ne[3] = x.ne[3];
        } break;
        default: throw except("graph_add: op %d", op);
    }
    // f32 tensors exist as inputs only; an f32 operand anywhere else would need kernels this executor does not have
    if (op != gop_patch_embed)
        for (int i = 0; i < n_src; ++i) {
            const bool head_tail = (op == gop_relu || op == gop_scale) && S(i).op != gop_input;
            if (!S(i).constant && S(i).dtype != gdt_f16 && !head_tail) throw except("%s: operand %d is f32; only patch_embed reads the f32 input tensor", nm, i);
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
    int buf_of(int t) const {
        const int b = g.nodes[root(t)].buffer;
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
        g.launches.push_back({std::move(desc), std::move(run)});
        io.emplace_back(std::move(reads), std::move(writes));
    }
    // device pointer of a buffer at run time (offsets are assigned after the launch list is complete)
    std::function<char*()> ptr(int buf) {
        graph* gp = &g;
        return [gp, buf]() { return static_cast<char*>(gp->arena.ptr) + gp->buffers[buf].offset; };
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
        if (n.op == gop_weight && g.dev && with.find('#') == std::string::npos) {
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
        if (!g.dev) return nullptr;
        void* d = nullptr;
        VX(vx_malloc(&d, bytes + 256));
        if (to_store) { g.store->allocs.push_back(d); g.store->device_bytes += bytes; }
        else g.const_allocs.push_back(d);
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
    struct epilogue { int act = 0; int res = -1; int last = -1; int scale = -1; };
    epilogue fuse_epilogue(int t, bool allow_gelu, bool allow_scale = false) {
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
        if (c >= 0 && g.nodes[c].op == gop_add && e.act != 1) {
            graph_node const& a = g.nodes[c];
            const int other = root(a.src[0]) == root(e.last) ? a.src[1] : a.src[0];
            graph_node const& o = g.nodes[other];
            bool same = !o.constant && o.dtype == gdt_f16 && root(other) != root(t) && root(other) < t; // already computed when this launch runs
            for (int i = 0; i < 4; ++i) same = same && o.ne[i] == g.nodes[t].ne[i];
            if (same && g.nodes[root(other)].buffer >= 0) {
                e.res = other;
                skip[c] = 1;
                g.nodes[c].alias_of = t;
                if (g.nodes[c].is_output) g.nodes[t].is_output = true;
                e.last = c;
            }
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
        const int xbuf = buf_of(n.src[0]);
        materialise(t);
        const int obuf = n.buffer;
        auto xp = ptr(xbuf), op = ptr(obuf);
        const int64_t M = n.n_elements();
        std::string who = n.name.empty() ? g.nodes[n.src[1]].name : n.name;
        emit(std::string("conv1x1_to_1") + (relu ? "[relu]" : "") + (scale != 1.0f ? "[scale]" : "") + " M=" + std::to_string(M) + " C=" + std::to_string(C) + " <- " + who, {xbuf}, {obuf},
             [=](void* st) { VX(vx_conv1x1_to1_f32(xp(), w, bias, relu, scale, reinterpret_cast<float*>(op()), M, C, st)); });
    }

    void gemm_like(int t) {
        graph_node& n = g.nodes[t];
        if (n.op == gop_conv_2d && n.dtype == gdt_f32) return one_channel_head(t);
        const int xs = n.src[0], wt = n.src[1], bt = n.n_src == 3 ? n.src[2] : -1;
        graph_node const& x = g.nodes[xs];
        std::string who = n.name.empty() ? g.nodes[wt].name : n.name;
        vx_gemm_args a;
        memset(&a, 0, sizeof a);
        packed_operand p;
        int xbuf = -1;
        std::string kind;
        bool conv = false;
        epilogue e;
        e.last = t;
        if (n.op == gop_linear) {
            e = fuse_epilogue(t, true, true);
            p = pack_rows(wt, bt, (int)n.ne[0], e.scale);
            a.M = (int)(x.n_elements() / x.ne[0]);
            xbuf = padded_rows(xs, p.k_real, p.K, a.M, who);
            a.lda = p.K;
            kind = "gemm";
        } else if (n.op == gop_conv_2d) {
            graph_node const& w = g.nodes[wt];
            p = pack_rows(wt, bt, (int)n.ne[0]);
            const int kw = (int)w.ne[1], kh = (int)w.ne[2], stride = (int)n.ip[0], pad = (int)n.ip[1];
            a.M = (int)(n.ne[1] * n.ne[2] * n.ne[3]);
            if (kw == 1 && kh == 1 && stride == 1 && pad == 0) { // 1x1: a plain product on the pixel rows (nn.cpp:76-81)
                xbuf = padded_rows(xs, p.k_real, p.K, a.M, who);
                a.lda = p.K;
                kind = "gemm(conv1x1)";
            } else {
                if (x.ne[0] % 8) throw except("conv_2d %s: Cin = %lld must be a multiple of 8", who.c_str(), (long long)x.ne[0]);
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
            xbuf = padded_rows(xs, p.k_real, p.K, a.M, who);
            a.lda = p.K;
            a.ps_s = s; a.ps_Cout = (int)n.ne[0]; a.ps_H = (int)x.ne[2]; a.ps_W = (int)x.ne[1];
            kind = "gemm+pixel_shuffle(s" + std::to_string(s) + ")";
        }
        if (n.ne[0] % 8) throw except("%s %s: %lld output channels; the f16 epilogues store 8 at a time", graph_op_name(n.op), who.c_str(), (long long)n.ne[0]);
        a.W = p.w; a.bias = p.bias; a.N = p.N; a.K = p.K; a.n_valid = n.op == gop_conv_transpose_2d ? p.n_real : (int)n.ne[0];
        a.ldo = n.ne[0];
        if (n.op == gop_conv_transpose_2d) a.epi = VX_EPI_PIXSHUF;
        else {
            if (n.op != gop_linear) e = fuse_epilogue(t, true);
            a.epi = e.res >= 0 ? VX_EPI_F16_ADD : (e.act == 1 ? VX_EPI_F16_GELU : (e.act == 2 ? VX_EPI_F16_RELU : VX_EPI_F16));
            a.relu = e.res >= 0 && e.act == 2;
        }
        materialise(t);
        const int obuf = n.buffer;
        std::vector<int> reads = {xbuf};
        std::function<char*()> rp;
        if (e.res >= 0) { reads.push_back(buf_of(e.res)); rp = ptr(buf_of(e.res)); }
        const bool halo = conv && a.conv_kh == 3 && a.conv_kw == 3 && a.conv_stride == 1 && a.conv_pad == 1 && a.conv_W >= 96;
        char d[256];
        snprintf(d, sizeof d, "%s%s%s%s%s%s M=%d N=%d K=%d <- %s", kind.c_str(), e.scale >= 0 ? "[*scale]" : "", e.act ? "[" : "", act_name(e.act), e.act ? "]" : "",
                 e.res >= 0 ? "[+res]" : "", a.M, (int)n.ne[0], p.k_real, who.c_str());
        auto xp = ptr(xbuf), op = ptr(obuf);
        emit(d, reads, {obuf}, [=](void* st) {
            vx_gemm_args r = a;
            r.A = xp();
            r.out = op();
            if (rp) r.res1 = rp();
            if (halo && vx_conv3x3_supported(&r)) VX(vx_conv3x3_f16(&r, st));
            else VX(vx_gemm_f16(&r, st));
        });
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
        for (int t = 0; t < N; ++t) {
            graph_node& n = g.nodes[t];
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
                case gop_scale: {
                    if (n.dtype != gdt_f16) throw except("%s on the f32 map of a one-channel head that has other readers is not built", graph_op_name(n.op));
                    const int xbuf = buf_of(n.src[0]);
                    materialise(t);
                    const int obuf = n.buffer;
                    auto xp = ptr(xbuf), op = ptr(obuf);
                    const int64_t cnt = n.n_elements();
                    const int uop = n.op == gop_gelu ? 0 : (n.op == gop_relu ? 1 : 2);
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
                } break;
                case gop_interpolate: {
                    if (!n.is_output && head_tail(t)) break;
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
                } break;
                case gop_concat:
                case gop_slice:
                case gop_repeat: copy_op(t); break;
                default: throw except("graph_allocate: %s is not lowered", graph_op_name(n.op));
            }
        }
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
